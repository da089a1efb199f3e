"""Block solves of the qcinv conjugate-gradient filter: several right-hand sides (simulations) through every launch
(`_b` entry points of include/plshts.h, multigrid_chain.solve on [nb, nalm] vectors, filt_cinv.cinv_*.apply_ivf_batch).

The reference filters one simulation at a time (examples/run_qlms.py:57-62 -> filt_cinv.py:196-203,275-289); a block solve
must give what nb separate solves give.  Every batched kernel forms each entry's result with the arithmetic of the un-batched
launch, so the comparisons below are bit-for-bit wherever both sides go through the same route."""
import os

import numpy as np
import pytest

from helpers import relrms

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'cg_golden.npz')


def _ralm(rng, lmax, nb):
    import torch
    from plancklens_amd import hp
    n = hp.Alm.getsize(lmax)
    a = rng.standard_normal((nb, n)) + 1j * rng.standard_normal((nb, n))
    a[:, :lmax + 1] = a[:, :lmax + 1].real
    return torch.from_numpy(a).cuda()


def test_block_vector_helpers_equal_their_loops():
    """almxfl, alm_copy, alm_splice(_fl), almxfl_add, alm_dot, axpy_dev, cg_dot_axpy (with and without `active`), gemv and
    template_project on blocks against the same call entry by entry: bit-identical."""
    import torch
    from plancklens_amd import dev
    rng = np.random.default_rng(0)
    nb, lmax, llo = 3, 37, 12
    a, b, c, d = (_ralm(rng, lmax, nb) for _ in range(4))
    lo = _ralm(rng, llo, nb)
    fl = rng.standard_normal(lmax + 1)
    eq = lambda x, y: bool((x == y).all())
    assert eq(dev.almxfl(a, fl), torch.stack([dev.almxfl(a[i], fl) for i in range(nb)]))
    assert eq(dev.alm_copy(a, llo), torch.stack([dev.alm_copy(a[i], llo) for i in range(nb)]))
    assert eq(dev.alm_copy(lo, lmax), torch.stack([dev.alm_copy(lo[i], lmax) for i in range(nb)]))
    assert eq(dev.alm_splice(lo, a, 9), torch.stack([dev.alm_splice(lo[i], a[i], 9) for i in range(nb)]))
    assert eq(dev.alm_splice_fl(lo, a, fl, 9), torch.stack([dev.alm_splice_fl(lo[i], a[i], fl, 9) for i in range(nb)]))
    assert eq(dev.almxfl_add(a, b, fl), torch.stack([dev.almxfl_add(a[i], b[i], fl) for i in range(nb)]))
    for lmin in (0, 2):
        assert eq(dev.alm_dot([(a, b), (c, d)], lmin=lmin), torch.stack([dev.alm_dot([(a[i], b[i]), (c[i], d[i])], lmin=lmin) for i in range(nb)]))
    num, den = dev.alm_dot([(a, b)]), dev.alm_dot([(c, c)])
    y1, y2 = a.clone(), a.clone()
    dev.axpy_dev(y1, d, num, den, -1.0)
    for i in range(nb):
        dev.axpy_dev(y2[i], d[i], num[i], den[i], -1.0)
    assert eq(y1, y2)
    # one conjugate-directions update over two fields, blocks against loops; then with one entry frozen
    for active in (None, torch.tensor([1., 0., 1.], dtype=torch.float64, device='cuda')):
        xs = [[t.clone() for t in (a, b)] for _ in range(2)]   # x (two fields)
        rs = [[t.clone() for t in (c, d)] for _ in range(2)]   # r
        dd, qq = [c + a, d - b], [a * 2. + d, b - c * .5]
        p1, p2 = dev.cg_dot_axpy(dd, qq, xs[0], dd, 1.0, b2=rs[0], y2=rs[0], x2=qq, sign2=-1.0, lmin=2, active=active)
        for i in range(nb):
            if active is not None and float(active[i]) == 0.:
                continue
            q1, q2 = dev.cg_dot_axpy([t[i] for t in dd], [t[i] for t in qq], [t[i] for t in xs[1]], [t[i] for t in dd], 1.0, b2=[t[i] for t in rs[1]],
                                     y2=[t[i] for t in rs[1]], x2=[t[i] for t in qq], sign2=-1.0, lmin=2)
            assert eq(p1[i], q1) and eq(p2[i], q2)
        for k in range(2):
            assert eq(xs[0][k], xs[1][k]) and eq(rs[0][k], rs[1][k])
        if active is not None:
            assert eq(xs[0][0][1], a[1]) and eq(rs[0][1][1], d[1])  # the frozen entry did not move
    # dense preconditioner product
    for nrhs in (2, 3, 5, 8, 11):
        n = 2 * 46  # even, as the (re, im) views of alm arrays are
        A = torch.from_numpy(rng.standard_normal((n + 5, n))).cuda()
        X = torch.from_numpy(rng.standard_normal((nrhs, n))).cuda()
        assert eq(dev.gemv(A, X), torch.stack([dev.gemv(A, X[i]) for i in range(nrhs)]))
    A = torch.from_numpy(rng.standard_normal((3000, 4290))).cuda()  # the row length of the temperature block at lmax 64
    X = torch.from_numpy(rng.standard_normal((4, 4290))).cuda()
    assert eq(dev.gemv(A, X), torch.stack([dev.gemv(A, X[i]) for i in range(4)]))
    assert relrms(dev.to_host(dev.gemv(A, X)), dev.to_host(X) @ dev.to_host(A).T) < 1e-13
    # template projection, coarse- and fine-grid variants of the kernels
    for npix in (12 * 16 ** 2, 12 * 512 ** 2):
        t = torch.from_numpy(rng.standard_normal((nb, npix))).cuda()
        ninv = torch.from_numpy(rng.random(npix)).cuda()
        pm = torch.from_numpy(rng.standard_normal((4, npix))).cuda()
        rm = torch.from_numpy(rng.standard_normal((4, npix)) * 1e-3).cuda()
        t1, t2 = t.clone(), t.clone()
        dev.template_project(t1, ninv, pm, rm)
        for i in range(nb):
            dev.template_project(t2[i], ninv, pm, rm)
        assert eq(t1, t2)


@pytest.mark.parametrize('nside,lmax,marge,nb', [(8, 16, True, 3), (32, 64, True, 3), (512, 600, True, 3), (512, 600, False, 2), (1024, 1100, True, 3),
                                                 (1024, 1100, False, 4), (2048, 2048, True, 2)])
def test_block_operators_equal_their_loops(nside, lmax, marge, nb):
    """pl_cg_fwd_tt_b / pl_cg_fwd_pp_b against entry-by-entry pl_cg_fwd_tt / pl_cg_fwd_pp (fwd_op.calc of opfilt_tt / opfilt_pp) on
    grids of every route: all rings in the generic FFT kernel with the projection folded in (8, 32), register FFT classes with
    the separate projection kernels (512, 1024); even blocks at nside >= 1024 send the polarization entries through the synthesis
    two at a time on one recursion (k_leg_synths<R, false, 2>), odd ones singly.  Bit-identical."""
    import torch
    from plancklens_amd import dev, hp
    from plancklens_amd.qcinv import opfilt_pp, opfilt_tt
    from plancklens_amd.qcinv.util_alm import eblm
    rng = np.random.default_rng(nside)
    npix = 12 * nside ** 2
    ell = np.arange(lmax + 1.)
    bl = np.exp(-ell * (ell + 1.) * 1e-6)
    x, y, z = hp.pix2vec(nside, np.arange(npix))
    ninv = (1. + 0.3 * x) * (np.abs(z) > 0.3)
    cl = {'tt': 1. / (ell + 3.) ** 2, 'ee': .1 / (ell + 3.) ** 2, 'bb': .01 / (ell + 3.) ** 2}
    nf = opfilt_tt.alm_filter_ninv(ninv, bl, marge_monopole=marge, marge_dipole=marge)
    op = opfilt_tt.fwd_op(cl, nf)
    xt = _ralm(rng, lmax, nb)
    x0 = xt.clone()
    got = op(xt)
    assert got.shape == xt.shape and bool((xt == x0).all())
    for i in range(nb):
        assert bool((got[i] == op(xt[i].contiguous())).all()), (nside, i)
    assert bool((opfilt_tt.pre_op_diag(cl, nf)(xt) == torch.stack([opfilt_tt.pre_op_diag(cl, nf)(xt[i]) for i in range(nb)])).all())
    nfp = opfilt_pp.alm_filter_ninv([ninv], bl)
    opp = opfilt_pp.fwd_op(cl, nfp)
    xe, xb = _ralm(rng, lmax, nb), _ralm(rng, lmax, nb)
    gp = opp(eblm([xe, xb]))
    for i in range(nb):
        one = opp(eblm([xe[i].contiguous(), xb[i].contiguous()]))
        assert bool((gp.elm[i] == one.elm).all()) and bool((gp.blm[i] == one.blm).all()), (nside, i)
    d = opfilt_pp.dot_op()
    vals = d(eblm([xe, xb]), gp)
    assert vals.shape == (nb,)
    for i in range(nb):
        one = d(eblm([xe[i].contiguous(), xb[i].contiguous()]), eblm([gp.elm[i].contiguous(), gp.blm[i].contiguous()]))
        assert abs(vals[i] - one) <= 1e-14 * abs(one)  # (the 64 partial sums are equal bit for bit; torch adds them in its own order)


def _chain_descr(lmax, nside, niter, dense_lmax, eps=0.0):
    from plancklens_amd.qcinv import cd_solve
    return [[1, ["split(dense(), %d, diag_cl)" % dense_lmax], 16, 8, 3, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()],
            [0, ["split(stage(1), 16, diag_cl)"], lmax, nside, niter, eps, cd_solve.tr_cg, cd_solve.cache_mem()]]


def test_reference_chains_reproduce_inside_a_block():
    """The reference's stored multigrid solutions (tests/golden/cg_golden.npz: multigrid_chain + opfilt_tt / opfilt_pp + dense +
    template_removal run by the reference itself) come out of a block solve of two right-hand sides -- the golden one and a
    different one -- and the block equals two separate solves."""
    import torch
    from plancklens_amd import dev
    from plancklens_amd.qcinv import multigrid, opfilt_pp, opfilt_tt
    from plancklens_amd.qcinv.util_alm import eblm
    g = np.load(GOLD)
    rng = np.random.default_rng(5)
    lmax, nside = int(g['lmax']), int(g['nside'])
    cl = {'tt': g['cl_tt'], 'ee': g['cl_ee'], 'bb': g['cl_bb']}
    n = g['cg_tlm'].size
    other = g['tmap'][::-1].copy() * 0.7 + 0.1 * rng.standard_normal(g['tmap'].size)
    nf = opfilt_tt.alm_filter_ninv(g['ninv_t'], g['transf'], marge_monopole=True, marge_dipole=True)
    mk = lambda: multigrid.multigrid_chain(opfilt_tt, _chain_descr(lmax, nside, 6, 6), cl, nf)
    blk = torch.zeros((2, n), dtype=torch.complex128, device='cuda')
    mk().solve(blk, [g['tmap'], other])
    assert relrms(dev.to_host(blk[0]), g['cg_tlm']) < 1e-8
    for i, m in enumerate((g['tmap'], other)):
        one = torch.zeros(n, dtype=torch.complex128, device='cuda')
        mk().solve(one, m)
        assert relrms(dev.to_host(blk[i]), dev.to_host(one)) < 1e-12, i
    # polarization
    nfp = opfilt_pp.alm_filter_ninv([g['ninv_p']], g['transf'])
    mkp = lambda: multigrid.multigrid_chain(opfilt_pp, _chain_descr(lmax, nside, 5, 5), cl, nfp)
    q2, u2 = g['umap'] * 0.5 + 0.2 * g['qmap'], g['qmap'][::-1].copy()
    pblk = eblm([torch.zeros((2, n), dtype=torch.complex128, device='cuda'), torch.zeros((2, n), dtype=torch.complex128, device='cuda')])
    mkp().solve(pblk, [[g['qmap'], g['umap']], [q2, u2]])
    assert relrms(dev.to_host(pblk.elm[0]), g['cg_elm']) < 1e-8 and relrms(dev.to_host(pblk.blm[0]), g['cg_blm']) < 1e-7
    for i, m in enumerate(([g['qmap'], g['umap']], [q2, u2])):
        one = eblm([torch.zeros(n, dtype=torch.complex128, device='cuda'), torch.zeros(n, dtype=torch.complex128, device='cuda')])
        mkp().solve(one, m)
        assert relrms(dev.to_host(pblk.elm[i]), dev.to_host(one.elm)) < 1e-12 and relrms(dev.to_host(pblk.blm[i]), dev.to_host(one.blm)) < 1e-12, i


def test_joint_filter_chain_reproduces_inside_a_block():
    """opfilt_tp (teblm block vectors, the T and P blocks through pl_cg_fwd_tt_b / pl_cg_fwd_pp_b, dense.pre_op_dense_tp on blocks):
    the reference's stored joint solution comes out of a block of two right-hand sides, equal to separate solves."""
    import torch
    from plancklens_amd import dev
    from plancklens_amd.qcinv import multigrid, opfilt_tp
    from plancklens_amd.qcinv.util_alm import teblm
    g = np.load(GOLD)
    lmax, nside = int(g['lmax']), int(g['nside'])
    cl = {'tt': g['cl_tt'], 'ee': g['cl_ee'], 'bb': g['cl_bb'], 'te': g['cl_te']}
    nf = opfilt_tp.alm_filter_ninv([g['ninv_t'], g['ninv_p']], g['transf'], b_transf_e=g['transf_e'], b_transf_b=g['transf_e'],
                                   marge_monopole=True, marge_dipole=True)
    mk = lambda: multigrid.multigrid_chain(opfilt_tp, _chain_descr(lmax, nside, 5, 4), cl, nf)
    n = g['cg_tp_tlm'].size
    other = [g['tmap'][::-1].copy(), g['umap'] * 0.5, g['qmap'][::-1].copy()]
    blk = teblm([torch.zeros((2, n), dtype=torch.complex128, device='cuda') for _ in range(3)])
    mk().solve(blk, [[g['tmap'], g['qmap'], g['umap']], other])
    assert relrms(dev.to_host(blk.tlm[0]), g['cg_tp_tlm']) < 1e-8
    assert relrms(dev.to_host(blk.elm[0]), g['cg_tp_elm']) < 1e-8 and relrms(dev.to_host(blk.blm[0]), g['cg_tp_blm']) < 1e-7
    for i, m in enumerate(([g['tmap'], g['qmap'], g['umap']], other)):
        one = teblm([torch.zeros(n, dtype=torch.complex128, device='cuda') for _ in range(3)])
        mk().solve(one, m)
        for a in ('tlm', 'elm', 'blm'):
            assert relrms(dev.to_host(getattr(blk, a)[i]), dev.to_host(getattr(one, a))) < 1e-12, (i, a)


def test_block_entries_stop_where_their_own_solves_stop():
    """eps_min > 0: the entries of a block converge after different numbers of iterations; each is frozen at the iterate its own
    solve returns (cd_monitors stopping rule per entry), while the others go on."""
    import torch
    from plancklens_amd import dev
    from plancklens_amd.qcinv import multigrid, opfilt_tt
    g = np.load(GOLD)
    rng = np.random.default_rng(9)
    lmax, nside = int(g['lmax']), int(g['nside'])
    cl = {'tt': g['cl_tt']}
    n = g['cg_tlm'].size
    nf = opfilt_tt.alm_filter_ninv(g['ninv_t'], g['transf'], marge_monopole=True, marge_dipole=True)
    # (the multigrid preconditioner makes the iteration count nearly independent of the right-hand side; the all-zero map -- masked
    # data, say -- is the entry that certainly stops elsewhere: at once, with 0 / 0 step lengths that must not leak into it)
    maps = [g['tmap'], rng.standard_normal(g['tmap'].size) * g['tmap'].std(), np.zeros(g['tmap'].size), g['tmap'] * np.linspace(0., 2., g['tmap'].size)]
    differed = False
    for eps in (1e-4, 1e-6, 1e-8, 1e-10):
        mk = lambda: multigrid.multigrid_chain(opfilt_tt, _chain_descr(lmax, nside, np.inf, 6, eps=eps), cl, nf)
        iters = []
        for m in maps:
            c = mk()
            one = torch.zeros(n, dtype=torch.complex128, device='cuda')
            c.solve(one, m)
            iters.append((c.last_iters, dev.to_host(one)))
        cb = mk()
        blk = torch.zeros((len(maps), n), dtype=torch.complex128, device='cuda')
        cb.solve(blk, maps)
        assert cb.last_iters == max(i for i, _ in iters), (eps, cb.last_iters, [i for i, _ in iters])
        differed = differed or len(set(i for i, _ in iters)) > 1
        for i, (_, ref) in enumerate(iters):
            assert np.all(np.isfinite(dev.to_host(blk[i]).real)) and relrms(dev.to_host(blk[i]), ref) < 1e-12, (eps, i, [k for k, _ in iters])
        assert not dev.to_host(blk[2]).any() and iters[2][0] == 0
    assert differed, 'the right-hand sides were meant to converge at different iterations for at least one tolerance'


def test_cinv_block_filtering_at_survey_size(tmp_path):
    """filt_cinv.cinv_t / cinv_p.apply_ivf_batch with the reference's default chains (4 and 3 stages, eps 1e-5, dense coarse level)
    on a masked sky at nside 512, lmax 1024: three simulations in one block solve equal three separate apply_ivf calls (which
    replay captured HIP graphs of the nested stages; the block solve captures its own)."""
    from plancklens_amd import hp, shts
    from plancklens_amd.filt import filt_cinv
    rng = np.random.default_rng(4)
    nside, lmax = 512, 1024
    npix = 12 * nside ** 2
    ell = np.arange(lmax + 1.)
    cl = {'tt': np.where(ell >= 2, 1e4 / np.maximum(ell, 1) ** 2.5, 0.), 'ee': np.where(ell >= 2, 50. / np.maximum(ell, 1) ** 2, 0.),
          'bb': np.where(ell >= 2, 1. / np.maximum(ell, 1) ** 2, 0.)}
    transf = hp.gauss_beam(10. / 60 / 180 * np.pi, lmax=lmax)
    x, y, z = hp.pix2vec(nside, np.arange(npix))
    mask = (np.abs(z) > 0.25).astype(float)
    vamin = np.sqrt(hp.nside2pixarea(nside, degrees=True)) * 60
    nb = 3
    tmaps, pmaps = [], []
    for i in range(nb):
        tmaps.append(shts.alm2map(hp.almxfl(hp.synalm(cl['tt'], lmax, rng), transf), nside) + 30. / vamin * rng.standard_normal(npix))
        q, u = shts.alm2map_spin([hp.almxfl(hp.synalm(cl['ee'], lmax, rng), transf), hp.almxfl(hp.synalm(cl['bb'], lmax, rng), transf)], nside, 2, lmax)
        pmaps.append([q + 40. / vamin * rng.standard_normal(npix), u + 40. / vamin * rng.standard_normal(npix)])
    cinv_t = filt_cinv.cinv_t(str(tmp_path / 'cinv_t'), lmax, nside, cl, transf, [mask * (vamin / 30.) ** 2])
    ones = [cinv_t.apply_ivf(m) for m in tmaps]
    blk = cinv_t.apply_ivf_batch(tmaps)
    for i in range(nb):
        assert relrms(blk[i], ones[i]) < 1e-12, i
    cinv_p = filt_cinv.cinv_p(str(tmp_path / 'cinv_p'), lmax, nside, cl, transf, [[mask * (vamin / 40.) ** 2]])
    onesp = [cinv_p.apply_ivf(m) for m in pmaps]
    blkp = cinv_p.apply_ivf_batch(pmaps)
    for i in range(nb):
        assert relrms(blkp[i][0], onesp[i][0]) < 1e-12 and relrms(blkp[i][1], onesp[i][1]) < 1e-12, i
    # the library level: the driver's filtering phase in block solves, same cache files as one-by-one filtering

    class sims(object):
        def hashdict(self):
            return {'blk': 4}

        def get_sim_tmap(self, idx):
            return tmaps[idx]

        def get_sim_pmap(self, idx):
            return pmaps[idx]
    lib = filt_cinv.library_cinv_sepTP(str(tmp_path / 'ivfs'), sims(), cinv_t, cinv_p, cl)
    lib.filter_sims([0, 1, 2], fields='tp', batch=3)
    for i in range(nb):
        assert os.path.exists(str(tmp_path / 'ivfs' / ('sim_%04d_tlm.fits' % i))) and os.path.exists(str(tmp_path / 'ivfs' / ('sim_%04d_blm.fits' % i)))
        assert relrms(lib.get_sim_tlm(i), ones[i]) < 1e-12
        assert relrms(lib.get_sim_elm(i), onesp[i][0]) < 1e-12 and relrms(lib.get_sim_blm(i), onesp[i][1]) < 1e-12



def test_dense_preconditioner_built_through_block_vectors(monkeypatch):
    """dense.pre_op_dense_tt / _pp / _tp.compute_minv (dense.py:57-285: fwd_op applied to every unit vector) with the unit vectors sent
    through the operator 32 at a time as block vectors: the matrix equals the one-vector-at-a-time build bit for bit."""
    from plancklens_amd import hp
    from plancklens_amd.qcinv import dense, opfilt_pp, opfilt_tp, opfilt_tt
    g = np.load(GOLD)
    nside = 8
    cl = {'tt': g['cl_tt'], 'ee': g['cl_ee'], 'bb': g['cl_bb'], 'te': g['cl_te']}
    nt = hp.ud_grade(g['ninv_t'], nside, power=-2)
    npol = hp.ud_grade(g['ninv_p'], nside, power=-2)
    ops = [(dense.pre_op_dense_tt, opfilt_tt.fwd_op(cl, opfilt_tt.alm_filter_ninv(nt, g['transf'][:17], marge_monopole=True, marge_dipole=True)), 6),
           (dense.pre_op_dense_pp, opfilt_pp.fwd_op(cl, opfilt_pp.alm_filter_ninv([npol], g['transf'][:17])), 5),
           (dense.pre_op_dense_tp, opfilt_tp.fwd_op(cl, opfilt_tp.alm_filter_ninv([nt, npol], g['transf'][:17], marge_monopole=True, marge_dipole=True)), 4)]
    for cls_, op, lmax in ops:
        from plancklens_amd import options
        monkeypatch.setattr(options.opts, 'dense_block', 1)
        one = cls_(lmax, op).minv
        monkeypatch.setattr(options.opts, 'dense_block', 32)
        blk = cls_(lmax, op).minv
        assert bool((one == blk).all()), cls_.__name__


@pytest.mark.parametrize('nside,lmax,nb', [(16, 32, 1), (32, 64, 3), (256, 512, 1), (512, 600, 2), (1024, 1100, 1)])
def test_template_projection_in_harmonic_space_equals_the_pixel_space_one(nside, lmax, nb, monkeypatch):
    """B^t Y^t [N^-1 - N^-1 T (T^t N^-1 T)^-1 T^t N^-1] Y B x (opfilt_tt.py:196-205: the bracket applied to the map) against
    B^t Y^t N^-1 Y B x - V (T^t N^-1 T)^-1 V^t x with V = B^t Y^t N^-1 T (pl_lowrank_update_b, the default): the same operator up to
    rounding, for monopole + dipole and for stored template maps, single vectors and blocks; the projected modes are annihilated."""
    import torch
    from plancklens_amd import dev, hp, shts
    from plancklens_amd.qcinv import opfilt_tt
    rng = np.random.default_rng(3 * nside + nb)
    npix = 12 * nside ** 2
    ell = np.arange(lmax + 1.)
    bl = np.exp(-ell * (ell + 1.) * 1e-6)
    x, y, z = hp.pix2vec(nside, np.arange(npix))
    ninv = (1. + 0.3 * x) * (np.abs(z) > 0.3) * (1. + 0.2 * rng.random(npix))
    cl = {'tt': 1. / (ell + 3.) ** 2}
    marge_maps = [np.cos(3. * np.arctan2(y, x)) * (1. - z ** 2), rng.standard_normal(npix)] if nside <= 256 else []
    nf = opfilt_tt.alm_filter_ninv(ninv, bl, marge_monopole=True, marge_dipole=True, marge_maps=marge_maps)
    op = opfilt_tt.fwd_op(cl, nf)
    xt = _ralm(rng, lmax, nb) if nb > 1 else _ralm(rng, lmax, 2)[0].contiguous()
    from plancklens_amd import options
    monkeypatch.setattr(options.opts, 'tproj_harm', False)
    ref = op(xt)
    monkeypatch.setattr(options.opts, 'tproj_harm', True)
    got = op(xt)
    assert got.shape == ref.shape
    assert relrms(dev.to_host(got), dev.to_host(ref)) < 1e-12
    if nb > 1:  # block entries equal single calls bit for bit
        for i in range(nb):
            assert bool((got[i] == op(xt[i].contiguous())).all())
    # a vector made of the templates alone, Y^t-side: N^-1-weighted operator part vanishes (S^-1 x remains)
    tmap = dev.to_dev(1. + 2. * x - y + .5 * z + (3. * marge_maps[0] if marge_maps else 0.), torch.float64)
    nf2 = opfilt_tt.alm_filter_ninv(ninv, bl, marge_monopole=True, marge_dipole=True, marge_maps=marge_maps)
    t1, t2 = tmap.clone(), tmap.clone()
    nf2.apply_map(t1)
    assert float(t1.abs().max()) < 1e-9 * float((dev.to_dev(ninv, torch.float64) * t2).abs().max())


def test_ring_roundtrip_kernel_is_bit_identical_to_the_two_launches():
    """k_ring_roundtrip (synthesis FFT, N^-1 weighting and analysis FFT of an all-generic grid in one launch, the pixel values kept in
    registers) against k_phase2map + k_map2phase: the switch is read once per process, so two child processes print a checksum of the
    temperature and polarization operators (with and without templates, single vectors and blocks, nside 8 ... 256) and must agree."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sums = []
    for v in ('0', '1'):
        env = dict(os.environ, PLSHTS_DEBUG='1', PLSHTS_CG_ROUNDTRIP=v)  # (development knobs are read under PLSHTS_DEBUG only)
        out = subprocess.run([sys.executable, os.path.join(root, 'tests', 'workers', 'roundtrip_check.py')], cwd=root, env=env,
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        sums.append(out.stdout.strip().splitlines()[-1])
    assert len(sums[0]) == 64 and sums[0] == sums[1]


def test_lowrank_coefficient_pass_in_the_prologue_launch_is_bit_identical_to_its_own_launch():
    """pl_cg_fwd_tt_lr_b on coarse grids: the coefficient pass c = hpm x of the low-rank template update rides in extra workgroups of the
    synthesis prologue (k_prep0_lr) instead of a forked side-stream launch (k_tproj_coeffs).  Same workgroup bodies (tproj_device.h), so the
    operators (monopole + dipole marginalised; single vectors and blocks of 2 ... 4) must agree bit for bit: two child processes (the switch
    is read once per process, under PLSHTS_DEBUG) print a checksum of their results."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sums = []
    for v in ('0', '1'):
        env = dict(os.environ, PLSHTS_DEBUG='1', PLSHTS_LR_PROLOGUE=v)
        out = subprocess.run([sys.executable, os.path.join(root, 'tests', 'workers', 'roundtrip_check.py')], cwd=root, env=env,
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        sums.append(out.stdout.strip().splitlines()[-1])
    assert len(sums[0]) == 64 and sums[0] == sums[1]
