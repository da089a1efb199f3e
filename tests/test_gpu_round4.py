"""Round-4 GPU tests: cinv_t and cinv_p of a simulation at the same time on two streams (filt_cinv.apply_ivf_tp / run_tp,
shts.plan_context), filter_sims on filters that cannot take block vectors, and the reference's own filt_cinv.cinv_t / cinv_p
classes at the smallest size they accept (tests/golden/cinv_golden.npz, made by tests/golden/make_golden.py cinv)."""
import os

import numpy as np
import pytest

from helpers import relrms

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _survey_setup(nb, seed=4):
    from plancklens_amd import hp, shts
    rng = np.random.default_rng(seed)
    nside, lmax = 512, 1024
    npix = 12 * nside ** 2
    ell = np.arange(lmax + 1.)
    cl = {'tt': np.where(ell >= 2, 1e4 / np.maximum(ell, 1) ** 2.5, 0.), 'ee': np.where(ell >= 2, 50. / np.maximum(ell, 1) ** 2, 0.),
          'bb': np.where(ell >= 2, 1. / np.maximum(ell, 1) ** 2, 0.)}
    transf = hp.gauss_beam(10. / 60 / 180 * np.pi, lmax=lmax)
    x, y, z = hp.pix2vec(nside, np.arange(npix))
    mask = (np.abs(z) > 0.25).astype(float)
    vamin = np.sqrt(hp.nside2pixarea(nside, degrees=True)) * 60
    tmaps, pmaps = [], []
    for i in range(nb):
        tmaps.append(shts.alm2map(hp.almxfl(hp.synalm(cl['tt'], lmax, rng), transf), nside) + 30. / vamin * rng.standard_normal(npix))
        q, u = shts.alm2map_spin([hp.almxfl(hp.synalm(cl['ee'], lmax, rng), transf), hp.almxfl(hp.synalm(cl['bb'], lmax, rng), transf)], nside, 2, lmax)
        pmaps.append([q + 40. / vamin * rng.standard_normal(npix), u + 40. / vamin * rng.standard_normal(npix)])
    return nside, lmax, cl, transf, mask, vamin, tmaps, pmaps


def test_cinv_t_and_cinv_p_at_the_same_time(tmp_path):
    """apply_ivf_tp: the temperature solve on the calling thread / current stream and the polarization solve on a helper thread / side
    stream inside its own plan context give exactly what the two solves give one after the other (same kernels, same order within
    each solve; only workspaces, scratch buffers and captured graphs are per context) -- single solves, block solves, and through
    library_cinv_sepTP.filter_sims."""
    import torch
    from plancklens_amd import dev, shts
    from plancklens_amd.filt import filt_cinv
    nside, lmax, cl, transf, mask, vamin, tmaps, pmaps = _survey_setup(4)
    cinv_t = filt_cinv.cinv_t(str(tmp_path / 'cinv_t'), lmax, nside, cl, transf, [mask * (vamin / 30.) ** 2])
    cinv_p = filt_cinv.cinv_p(str(tmp_path / 'cinv_p'), lmax, nside, cl, transf, [[mask * (vamin / 40.) ** 2]])
    ones_t = [cinv_t.apply_ivf(m) for m in tmaps]
    ones_p = [cinv_p.apply_ivf(m) for m in pmaps]
    assert shts.context() == 0
    for rep in range(3):  # first call: one after the other in the two contexts (set-up, graph capture); then overlapped
        i = rep % 2
        t, (e, b) = filt_cinv.apply_ivf_tp(cinv_t, tmaps[i], cinv_p, pmaps[i])
        assert np.array_equal(t, ones_t[i]), (rep, relrms(t, ones_t[i]))
        assert np.array_equal(e, ones_p[i][0]) and np.array_equal(b, ones_p[i][1]), (rep, relrms(e, ones_p[i][0]))
    assert shts.context() == 0
    # device maps in, device alms out, usable on the caller's stream
    td, pd = dev.to_dev(tmaps[2]), [dev.to_dev(m) for m in pmaps[2]]
    t, (e, b) = filt_cinv.apply_ivf_tp(cinv_t, td, cinv_p, pd)
    assert isinstance(e, torch.Tensor) and np.array_equal(dev.to_host(t), ones_t[2]) and np.array_equal(dev.to_host(b), ones_p[2][1])
    # block solves of two simulations each
    for rep in range(2):
        ts, ps = filt_cinv.apply_ivf_tp(cinv_t, tmaps[:2], cinv_p, pmaps[:2])
        for i in range(2):
            assert relrms(ts[i], ones_t[i]) < 1e-12 and relrms(ps[i][0], ones_p[i][0]) < 1e-12 and relrms(ps[i][1], ones_p[i][1]) < 1e-12

    class sims(object):
        def hashdict(self):
            return {'tp': 4}

        def get_sim_tmap(self, idx):
            return tmaps[idx]

        def get_sim_pmap(self, idx):
            return pmaps[idx]
    lib = filt_cinv.library_cinv_sepTP(str(tmp_path / 'ivfs'), sims(), cinv_t, cinv_p, cl)
    lib.filter_sims([0, 1, 2, 3], fields='tp', batch=2)  # the second block runs overlapped
    for i in range(4):
        assert relrms(lib.get_sim_tlm(i), ones_t[i]) < 1e-12
        assert relrms(lib.get_sim_elm(i), ones_p[i][0]) < 1e-12 and relrms(lib.get_sim_blm(i), ones_p[i][1]) < 1e-12
    # one by one through the library: the polarization solve in its own context as well
    lib2 = filt_cinv.library_cinv_sepTP(str(tmp_path / 'ivfs2'), sims(), cinv_t, cinv_p, cl, )
    assert relrms(lib2.get_sim_elm(3), ones_p[3][0]) < 1e-12 and relrms(lib2.get_sim_tlm(3), ones_t[3]) < 1e-12


def test_filter_sims_falls_back_when_a_filter_takes_no_block_vectors(tmp_path):
    """A cinv_p with marginalised Q / U templates has no block operator (opfilt_pp.one_call_ok needs single vectors for the
    harmonic-space projection): filter_sims must serve it one simulation at a time instead of failing in fwd_op, with the
    temperature filter still going through block solves; cache=False keeps every result of the call resident."""
    from plancklens_amd import hp
    from plancklens_amd.filt import filt_cinv
    nside, lmax, cl, transf, mask, vamin, tmaps, pmaps = _survey_setup(3, seed=5)
    npix = 12 * nside ** 2
    th, ph = hp.pix2ang(nside, np.arange(npix))
    tq = [np.cos(th) * mask, np.sin(th) * np.cos(ph) * mask]
    tu = [np.sin(2 * th) * np.sin(ph) * mask]
    cinv_t = filt_cinv.cinv_t(str(tmp_path / 'cinv_t'), lmax, nside, cl, transf, [mask * (vamin / 30.) ** 2])
    cinv_p = filt_cinv.cinv_p(str(tmp_path / 'cinv_p'), lmax, nside, cl, transf, [[mask * (vamin / 40.) ** 2]], marge_qmaps=tq, marge_umaps=tu)

    class sims(object):
        def hashdict(self):
            return {'tmpl': 3}

        def get_sim_tmap(self, idx):
            return tmaps[idx]

        def get_sim_pmap(self, idx):
            return pmaps[idx]
    lib = filt_cinv.library_cinv_sepTP(str(tmp_path / 'ivfs'), sims(), cinv_t, cinv_p, cl)
    lib.cache = False
    assert lib.supports_block('t') and not lib.supports_block('p')
    lib.filter_sims([0, 1, 2], fields='tp', batch=3)
    assert all(set(lib._dev_cache.get(i, {})) == {'t', 'e', 'b'} for i in range(3)), 'every simulation of the call stays resident'
    for i in range(3):
        e, b = cinv_p.apply_ivf(pmaps[i])
        assert relrms(lib.get_sim_elm(i), e) < 1e-12 and relrms(lib.get_sim_blm(i), b) < 1e-12
        assert relrms(lib.get_sim_tlm(i), cinv_t.apply_ivf(tmaps[i])) < 1e-12


def test_cinv_t_and_cinv_p_vs_the_reference_classes(tmp_path, oracle):
    """filt_cinv.cinv_t / cinv_p against the reference's own classes (filt_cinv.py:56-338) at the smallest size their constructors
    accept (nside 512, lmax 1024): default 4-stage / 3-stage chains with the dense(64) / dense(32) levels, D_l rescaling, eps = 1e-5
    stopping rule, galactic-cut + point-source mask, inhomogeneous noise, monopole + dipole marginalised.  The reference ran over
    the oracle's transforms (tests/golden/make_golden.py cinv -> cinv_golden.npz); inputs are re-made here from the shared recipe
    (tests/helpers.py::cinv_golden_inputs, checksums stored).  Compared: the side files, the number of top-level iterations, the
    residual trace, C_l of the solutions, every entry with l <= 64 and a seeded 10 000-entry subset."""
    from helpers import cinv_golden_inputs
    from plancklens_amd import hp
    from plancklens_amd.filt import filt_cinv
    g = np.load(os.path.join(HERE, 'golden', 'cinv_golden.npz'))
    d = cinv_golden_inputs(oracle.alm2map, oracle.alm2map_spin)
    nside, lmax, cl, transf = d['nside'], d['lmax'], d['cl'], d['transf']
    assert nside == int(g['nside']) and lmax == int(g['lmax']) and np.array_equal(transf, g['transf'])
    for k in ['ninv_t', 'ninv_p', 'tmap', 'qmap', 'umap']:  # the same inputs as the generator's (their transforms may round differently)
        chk = np.array([d[k].sum(), (d[k] ** 2).sum(), d[k][::9973].sum()])
        assert np.allclose(chk, g['chk_' + k], rtol=1e-9, atol=1e-9 * np.sqrt(chk[1])), (k, chk, g['chk_' + k])
    sub, low = g['subset'], g['low']
    report = []
    for kind in ('t', 'p'):
        trace = []
        if kind == 't':
            filt = filt_cinv.cinv_t(str(tmp_path / 'cinv_t'), lmax, nside, cl, transf, [d['ninv_t']])
        else:
            filt = filt_cinv.cinv_p(str(tmp_path / 'cinv_p'), lmax, nside, cl, transf, [[d['ninv_p']]])
        log0 = filt.chain.log
        filt.chain.log = lambda stage, it, eps, **kw: (trace.append((stage.depth, it, eps)), log0(stage, it, eps, **kw))
        if kind == 't':
            sols = {'tlm': filt.apply_ivf(d['tmap'])}
            assert np.allclose(filt.get_ftl(), g['ftl'], rtol=1e-12, atol=0) and np.allclose(filt.get_tal('t'), g['tal_t'], rtol=1e-12)
            assert filt.get_fmask().sum() == float(g['fmask_t_sum'])
        else:
            e, b = filt.apply_ivf([d['qmap'], d['umap']])
            sols = {'elm': e, 'blm': b}
            assert np.allclose(filt.get_fel(), g['fel'], rtol=1e-12, atol=0) and np.allclose(filt.get_fbl(), g['fbl'], rtol=1e-12, atol=0)
        tr = np.array([t[2] for t in trace if t[0] == 0])
        ref = g['trace_' + kind]
        assert len(tr) == len(ref), (kind, len(tr), len(ref), tr[-3:], ref[-3:])  # same number of top-level iterations
        assert np.allclose(tr, ref, rtol=1e-6), (kind, np.max(np.abs(tr / ref - 1)))
        for nm, a in sols.items():
            e_sub, e_low = relrms(a[sub], g[nm + '_sub']), relrms(a[low], g[nm + '_low'])
            e_cl = float(np.max(np.abs(hp.alm2cl(a)[2:] / g[nm + '_cl'][2:] - 1)))
            report.append('%s: %d iterations, subset %.1e, l <= 64 %.1e, C_l %.1e, trace %.1e' % (nm, len(tr) - 1, e_sub, e_low, e_cl,
                                                                                              np.max(np.abs(tr / ref - 1))))
            assert e_sub < 1e-8 and e_low < 1e-8 and e_cl < 1e-8, report[-1]
    with open(os.path.join(os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(HERE)), 'gpurun_out', 'cinv_reference_parity.txt'), 'w') as f:
        f.write('\n'.join(report) + '\n')
    print('\n'.join(report))


def test_paired_sky_synthesis_of_device_simulations(tmp_path):
    """sims.maps.cmb_maps.hint_pair: the polarization maps of two announced simulations come out of one batched synthesis
    (pl_alm2map_batch2) and equal the maps made one by one bit for bit; the hint is forwarded through sim_lib_shuffle, and a request
    in any other order still gives the right maps."""
    import torch
    from plancklens_amd import hp, utils
    from plancklens_amd.sims import cmbs, maps, phas, utils as sutils
    nside, lmax = 64, 128
    cl = utils.camb_clfile(os.path.join(os.path.dirname(HERE), 'plancklens_amd', 'data', 'cls', 'FFP10_wdipole_lensedCls.dat'), lmax=lmax)
    transf = hp.gauss_beam(20. / 60. / 180. * np.pi, lmax=lmax)

    def lib(tag):
        pix = phas.pix_lib_phas_dev(str(tmp_path / ('pix' + tag)), 3, (hp.nside2npix(nside),), seed=3)
        sky = phas.lib_phas_dev(str(tmp_path / ('sky' + tag)), 3, lmax, seed=4)
        skies = cmbs.sims_cmb_unl({k: cl[k] for k in ['tt', 'ee', 'bb', 'te']}, sky)
        return maps.cmb_maps_nlev(skies, transf, 30., 40., nside, pix_lib_phas=pix, device_maps=True)
    ref, a = lib('a'), lib('b')
    sh = sutils.sim_lib_shuffle(a, {0: 0, 1: 1, 2: 2, 3: 3})
    sh.hint_pair(0, 1)
    q0, u0 = sh.get_sim_pmap(0)
    assert a.__dict__.get('_pair_held') is not None and a._pair_held[0] == 1
    q1, u1 = sh.get_sim_pmap(1)
    assert a._pair_held is None
    for i, (q, u) in enumerate(((q0, u0), (q1, u1))):
        rq, ru = ref.get_sim_pmap(i)
        assert isinstance(q, torch.Tensor) and bool((q == rq).all()) and bool((u == ru).all()), i
    sh.hint_pair(2, 3)      # the hint is not followed: 3 first
    q3, u3 = sh.get_sim_pmap(3)
    q2, u2 = sh.get_sim_pmap(2)
    assert bool((q3 == ref.get_sim_pmap(3)[0]).all()) and bool((u2 == ref.get_sim_pmap(2)[1]).all())


@pytest.mark.parametrize('lsplit,lmax', [(6, 16), (32, 64), (64, 256)])
def test_split_preconditioner_in_one_launch(lsplit, lmax):
    """pl_gemv_split: pre_op_split with a dense low block and a diagonal high part (multigrid.py:163-182, dense.py:118-119) in one launch
    equals truncating alm_copy + pl_gemv + pl_alm_splice_fl bit for bit, for one field (temperature) and two (E, B), and pre_op_split
    takes that route for device vectors."""
    import torch
    from plancklens_amd import dev, hp
    from plancklens_amd.qcinv import dense, multigrid, opfilt_pp, util_alm
    rng = np.random.default_rng(lsplit)

    def ralm():
        x = rng.standard_normal(hp.Alm.getsize(lmax)) + 1j * rng.standard_normal(hp.Alm.getsize(lmax))
        x[:lmax + 1] = x[:lmax + 1].real
        return dev.to_dev(x)
    for nf in (1, 2):
        nr = nf * (lsplit + 1) ** 2
        a = rng.standard_normal((nr, nr))
        base = dense.pre_op_dense_tt if nf == 1 else dense.pre_op_dense_pp

        class low(base):  # a dense block with a given (symmetric) matrix
            def __init__(self):
                self.lmax = lsplit
                self.minv = dev.to_dev(a + a.T, torch.float64)
        if nf == 1:
            class diag(object):
                filt = rng.uniform(0.5, 2., lmax + 1)

                def __call__(self, talm):
                    return dev.almxfl(talm, self.filt)

                def splice_above(self, alm_low, talm, ls):
                    return dev.alm_splice_fl(alm_low, talm, self.filt, ls)
            ph, x = diag(), ralm()
        else:
            ph = opfilt_pp.pre_op_diag.__new__(opfilt_pp.pre_op_diag)
            ph.flmat = np.zeros((lmax + 1, 2, 2))
            ph.flmat[:, 0, 0], ph.flmat[:, 1, 1] = rng.uniform(0.5, 2., lmax + 1), rng.uniform(0.5, 2., lmax + 1)
            x = util_alm.eblm([ralm(), ralm()])
        pl = low()
        ref = ph.splice_above(pl(util_alm.alm_copy(x, lmax=lsplit)), x, lsplit)   # the step-by-step launches
        one = pl.split_apply(x, lsplit, ph)
        assert one is not None
        for r, o in zip(multigrid._parts(ref), multigrid._parts(one)):
            assert torch.equal(o, r)
        sp = multigrid.pre_op_split(lsplit, lmax, pl, ph)
        for r, o in zip(multigrid._parts(ref), multigrid._parts(sp(x))):
            assert torch.equal(o, r)
        # block vectors keep the step-by-step route
        blk = torch.stack([multigrid._parts(x)[0]] * 2)
        assert pl.split_apply(blk if nf == 1 else util_alm.eblm([blk, blk]), lsplit, ph) is None


@pytest.mark.parametrize('nb', [1, 3])
def test_step_scalar_products_from_the_operator_kernels(nb):
    """fwd_op.with_dots: the post-processing kernel of the operator's analysis (k_post0 / k_posts) leaves <d, q> and <d, r> as partial
    sums (pl_plan_arm_post_dots); they equal the scalar products of pl_alm_dot to rounding, the result q is untouched (bit-identical),
    and dot_op.step(pre=...) makes the updates of the step from them (pl_cg_axpy_pre_b) -- temperature with monopole + dipole
    marginalised (the low-rank update folded into k_post0, block vectors too) and polarization."""
    import torch
    from plancklens_amd import dev, hp
    from plancklens_amd.qcinv import opfilt_pp, opfilt_tt
    from plancklens_amd.qcinv.util_alm import eblm
    rng = np.random.default_rng(11)
    nside, lmax = 64, 128
    npix, nalm = 12 * nside ** 2, hp.Alm.getsize(lmax)
    ell = np.arange(lmax + 1.)
    cl = {'tt': np.where(ell >= 1, 1e3 / np.maximum(ell, 1) ** 2, 1.), 'ee': np.where(ell >= 2, 30. / np.maximum(ell, 1) ** 2, 0.),
          'bb': np.where(ell >= 2, 3. / np.maximum(ell, 1) ** 2, 0.)}
    transf = hp.gauss_beam(30. / 60 / 180 * np.pi, lmax=lmax)
    z = hp.pix2vec(nside, np.arange(npix))[2]
    ninv = (np.abs(z) > 0.3) * (1. + 0.5 * rng.random(npix))

    def vec(shape_nb):
        v = rng.standard_normal((shape_nb, nalm)) + 1j * rng.standard_normal((shape_nb, nalm))
        v[:, :lmax + 1].imag = 0.
        return dev.to_dev(v[0] if nb == 1 else v, torch.complex128).contiguous()

    def tot(parts):
        return dev.to_host(parts.sum(-1))

    # temperature
    ft = opfilt_tt.alm_filter_ninv(ninv, transf, marge_monopole=True, marge_dipole=True)
    op, dot = opfilt_tt.fwd_op(cl, ft), opfilt_tt.dot_op()
    d, r, x = vec(nb), vec(nb), vec(nb)
    q_ref = op(d)
    q, pre = op.with_dots(d, r)
    assert pre is not None and torch.equal(q, q_ref)
    np.testing.assert_allclose(tot(pre[0]), tot(dot.parts(d, q)), rtol=1e-12)
    np.testing.assert_allclose(tot(pre[1]), tot(dot.parts(d, r)), rtol=1e-12, atol=1e-9 * abs(tot(dot.parts(d, d))).max())
    x1, r1, x2, r2 = x.clone(), r.clone(), x.clone(), r.clone()
    dtad1, delta1 = dot.step(x1, d, r1, q)
    dtad2, delta2 = dot.step(x2, d, r2, q, pre=pre)
    np.testing.assert_allclose(tot(dtad2), tot(dtad1), rtol=1e-12)
    assert relrms(dev.to_host(x2), dev.to_host(x1)) < 1e-12 and relrms(dev.to_host(r2), dev.to_host(r1)) < 1e-12
    # polarization
    fp = opfilt_pp.alm_filter_ninv([ninv], transf)
    opp, dotp = opfilt_pp.fwd_op(cl, fp), opfilt_pp.dot_op()
    mk = lambda: eblm([vec(nb), vec(nb)])
    d, r, x = mk(), mk(), mk()
    q_ref = opp(d)
    q, pre = opp.with_dots(d, r)
    assert pre is not None and torch.equal(q.elm, q_ref.elm) and torch.equal(q.blm, q_ref.blm)
    np.testing.assert_allclose(tot(pre[0]), tot(dotp.parts(d, q)), rtol=1e-12)
    np.testing.assert_allclose(tot(pre[1]), tot(dotp.parts(d, r)), rtol=1e-12, atol=1e-9 * abs(tot(dotp.parts(d, d))).max())
    cp = lambda v: eblm([v.elm.clone(), v.blm.clone()])
    x1, r1, x2, r2 = cp(x), cp(r), cp(x), cp(r)
    dotp.step(x1, d, r1, q)
    dotp.step(x2, d, r2, q, pre=pre)
    assert relrms(dev.to_host(x2.elm), dev.to_host(x1.elm)) < 1e-12 and relrms(dev.to_host(r2.blm), dev.to_host(r1.blm)) < 1e-12


def test_ortho_scalar_product_from_the_preconditioner_kernels():
    """pre_op_split.with_dot: the kernel that writes the new search direction (pl_gemv_split_dot around the dense block, pl_alm_splice_dot_b
    around a nested stage) leaves <s, q'> as partial sums; s is bit-identical to calc's, the sum equals pl_alm_dot's to rounding and
    dot_op.ortho(pre=...) makes the update of cd_solve.py:96-103 from it -- temperature and polarization."""
    import torch
    from plancklens_amd import dev, hp
    from plancklens_amd.qcinv import multigrid, opfilt_pp, opfilt_tt
    from plancklens_amd.qcinv.util_alm import eblm
    rng = np.random.default_rng(12)
    nside, lmax, lsplit = 32, 64, 12
    npix, nalm = 12 * nside ** 2, hp.Alm.getsize(lmax)
    ell = np.arange(lmax + 1.)
    cl = {'tt': np.where(ell >= 1, 1e3 / np.maximum(ell, 1) ** 2, 1.), 'ee': np.where(ell >= 2, 30. / np.maximum(ell, 1) ** 2, 0.),
          'bb': np.where(ell >= 2, 3. / np.maximum(ell, 1) ** 2, 0.)}
    transf = hp.gauss_beam(60. / 60 / 180 * np.pi, lmax=lmax)
    z = hp.pix2vec(nside, np.arange(npix))[2]
    ninv = (np.abs(z) > 0.3) * (1. + 0.5 * rng.random(npix))

    def vec():
        v = rng.standard_normal(nalm) + 1j * rng.standard_normal(nalm)
        v[:lmax + 1].imag = 0.
        return dev.to_dev(v, torch.complex128).contiguous()

    tot = lambda parts: float(parts.sum())
    for opfilt, mk, parts_of in ((opfilt_tt, vec, lambda v: [v]), (opfilt_pp, lambda: eblm([vec(), vec()]), lambda v: [v.elm, v.blm])):
        filt = opfilt.alm_filter_ninv(ninv, transf) if opfilt is opfilt_tt else opfilt.alm_filter_ninv([ninv], transf)
        dot = opfilt.dot_op()
        diag = opfilt.pre_op_diag(cl, filt)
        dense_op = opfilt.pre_op_dense(lsplit, opfilt.fwd_op(cl, filt.degrade(nside)))
        # (a) the one-launch split around the dense block; (b) the splice around any other low-l preconditioner (here: the same dense
        # block without its split form -- stands in for a nested multigrid stage)
        class plain(object):
            def __init__(self, op): self.op = op
            def __call__(self, v): return self.op(v)
        for low in (dense_op, plain(dense_op)):
            op = multigrid.pre_op_split(lsplit, lmax, low, diag)
            r, q, pd = mk(), mk(), mk()
            s_ref = op(r)
            s, pre = op.with_dot(r, q, dot.lmin)
            assert pre is not None
            for a, b in zip(parts_of(s), parts_of(s_ref)):
                assert torch.equal(a, b)
            ref = tot(dot.parts(s, q))
            assert abs(float(pre.sum()) - ref) <= 1e-12 * abs(tot(dot.parts(s, s))), (float(pre.sum()), ref)
            dtad = dot.parts(pd, pd)
            s1 = s_ref if opfilt is opfilt_tt else eblm([s_ref.elm.clone(), s_ref.blm.clone()])
            s1 = s1.clone() if opfilt is opfilt_tt else s1
            dot.ortho(s1, q, pd, dtad)
            dot.ortho(s, q, pd, dtad, pre=pre)
            for a, b in zip(parts_of(s), parts_of(s1)):
                assert relrms(dev.to_host(a), dev.to_host(b)) < 1e-12


@pytest.mark.parametrize('nside,lmax,spin', [(32, 64, 1), (128, 200, 1), (64, 95, 3), (512, 512, 1), (2048, 2048, 1)])
def test_two_gradient_only_syntheses_on_one_recursion(nside, lmax, spin):
    """pl_alm2map_grad_pair (k_leg_synths<R, true, 1>: the gradient legs of the temperature estimator of two simulations, 12 instead of
    2 x 8 FMAs per step) gives the maps of two pl_alm2map_grad calls bit for bit, with one filter or two."""
    import torch
    from plancklens_amd import dev, hp, shts
    rng = np.random.default_rng(nside + lmax)
    nalm = hp.Alm.getsize(lmax)

    def galm():
        a = rng.standard_normal(nalm) + 1j * rng.standard_normal(nalm)
        a[:lmax + 1].imag = 0.
        return dev.to_dev(a, torch.complex128)
    g1, g2 = galm(), galm()
    fl, fl2 = 1. / (1. + np.arange(lmax + 1.)), np.sqrt(np.arange(lmax + 1.))
    for fa, fb in ((fl, None), (fl, fl2), (None, None)):
        r1 = shts.alm2map_spin([g1, None], nside, spin, lmax, fl=fa)
        r2 = shts.alm2map_spin([g2, None], nside, spin, lmax, fl=fa if fb is None else fb)
        p1, p2 = shts.alm2map_spin_grad_pair(g1, g2, nside, spin, lmax, fl=fa, fl2=fb)
        for a, b in zip(p1 + p2, r1 + r2):
            assert torch.equal(a, b)
