"""GPU parity of the qcinv conjugate-gradient filter (plancklens_amd.qcinv + filt.filt_cinv) against outputs of the
reference's own multigrid CG stored in tests/golden/cg_golden.npz (made by tests/golden/make_golden.py), plus the
known answer that on a full sky with white noise the CG filter equals the isotropic filter."""
import os

import numpy as np
import pytest

from helpers import relrms

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'cg_golden.npz')


@pytest.fixture(scope='module')
def g():
    import torch
    assert torch.cuda.is_available()
    return np.load(GOLD)


def _chain_descr(lmax, nside, niter, dense_lmax):
    from plancklens_amd.qcinv import cd_solve
    return [[1, ["split(dense(), %d, diag_cl)" % dense_lmax], 16, 8, 3, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()],
            [0, ["split(stage(1), 16, diag_cl)"], lmax, nside, niter, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()]]


def test_tt_operators_vs_reference(g):
    from plancklens_amd import dev
    from plancklens_amd.qcinv import opfilt_tt
    cl = {'tt': g['cl_tt']}
    nf = opfilt_tt.alm_filter_ninv(g['ninv_t'], g['transf'], marge_monopole=True, marge_dipole=True)
    assert relrms(dev.to_host(opfilt_tt.calc_prep(g['tmap'], cl, nf)), g['cg_t_prep']) < 1e-11
    x = dev.to_dev(g['cg_t_x'])
    x0 = x.clone()
    assert relrms(dev.to_host(opfilt_tt.fwd_op(cl, nf)(x)), g['cg_t_fwd']) < 1e-11
    assert relrms(dev.to_host(opfilt_tt.pre_op_diag(cl, nf)(x)), g['cg_t_diag']) < 1e-12
    assert bool((x == x0).all())  # operators must not modify their argument (cd_solve.py:50-51)
    d = opfilt_tt.dot_op()
    from plancklens_amd import hp
    lmax = int(g['lmax'])
    assert abs(d(x, x) - np.sum(hp.alm2cl(g['cg_t_x']) * (2 * np.arange(lmax + 1) + 1))) < 1e-9 * d(x, x)


def test_tt_chain_vs_reference(g):
    import torch
    from plancklens_amd import dev
    from plancklens_amd.qcinv import multigrid, opfilt_tt
    lmax, nside = int(g['lmax']), int(g['nside'])
    cl = {'tt': g['cl_tt']}
    nf = opfilt_tt.alm_filter_ninv(g['ninv_t'], g['transf'], marge_monopole=True, marge_dipole=True)
    chain = multigrid.multigrid_chain(opfilt_tt, _chain_descr(lmax, nside, 6, 6), cl, nf)
    trace = []
    log0 = chain.log
    chain.log = lambda stage, it, eps, **kw: (trace.append((stage.depth, it, eps)), log0(stage, it, eps, **kw))
    talm = torch.zeros(g['cg_tlm'].size, dtype=torch.complex128, device='cuda')
    chain.solve(talm, g['tmap'])
    assert relrms(dev.to_host(talm), g['cg_tlm']) < 1e-8
    tr = np.array([t[2] for t in trace if t[0] == 0])
    assert len(tr) == len(g['cg_t_trace'])
    assert np.allclose(tr[:4], g['cg_t_trace'][:4], rtol=1e-5)  # later entries are at the rounding floor


def test_pp_operators_and_chain_vs_reference(g):
    import torch
    from plancklens_amd import dev
    from plancklens_amd.qcinv import multigrid, opfilt_pp
    from plancklens_amd.qcinv.util_alm import eblm
    lmax, nside = int(g['lmax']), int(g['nside'])
    cl = {'ee': g['cl_ee'], 'bb': g['cl_bb'], 'tt': g['cl_tt']}
    nf = opfilt_pp.alm_filter_ninv([g['ninv_p']], g['transf'])
    x = eblm([dev.to_dev(g['cg_p_xe']), dev.to_dev(g['cg_p_xb'])])
    f = opfilt_pp.fwd_op(cl, nf)(x)
    assert relrms(dev.to_host(f.elm), g['cg_p_fwd_e']) < 1e-11 and relrms(dev.to_host(f.blm), g['cg_p_fwd_b']) < 1e-11
    chain = multigrid.multigrid_chain(opfilt_pp, _chain_descr(lmax, nside, 5, 5), cl, nf)
    n = g['cg_elm'].size
    palm = eblm([torch.zeros(n, dtype=torch.complex128, device='cuda'), torch.zeros(n, dtype=torch.complex128, device='cuda')])
    chain.solve(palm, [g['qmap'], g['umap']])
    assert relrms(dev.to_host(palm.elm), g['cg_elm']) < 1e-8 and relrms(dev.to_host(palm.blm), g['cg_blm']) < 1e-7


def test_cinv_t_p_fullsky_white_noise_known_answer(tmp_path):
    """No mask, homogeneous noise: N^-1 is a multiple of the identity (the marginalised monopole / dipole only touch l < 2), so for l >= 2 the CG solution must equal
    the isotropic filter F_l / b_l map2alm(map) up to the quadrature error of uniform-weight HEALPix sums (tested at
    lmax = 1024 on nside 512 with the reference's default 4-stage / 3-stage chains and its eps = 1e-5 stopping rule)."""
    from plancklens_amd import hp, shts, utils
    from plancklens_amd.filt import filt_cinv
    rng = np.random.default_rng(3)
    nside, lmax = 512, 1024
    npix = 12 * nside ** 2
    ell = np.arange(lmax + 1.)
    cl = {'tt': np.where(ell >= 2, 1e4 / np.maximum(ell, 1) ** 2.5, 0.), 'ee': np.where(ell >= 2, 50. / np.maximum(ell, 1) ** 2, 0.),
          'bb': np.where(ell >= 2, 1. / np.maximum(ell, 1) ** 2, 0.)}
    transf = hp.gauss_beam(10. / 60 / 180 * np.pi, lmax=lmax)
    nlev_t, nlev_p = 30., 40.
    vamin = np.sqrt(hp.nside2pixarea(nside, degrees=True)) * 60
    tmap = shts.alm2map(hp.almxfl(hp.synalm(cl['tt'], lmax, rng), transf), nside) + nlev_t / vamin * rng.standard_normal(npix)
    ninv_t = [np.ones(npix) * (vamin / nlev_t) ** 2]
    cinv_t = filt_cinv.cinv_t(str(tmp_path / 'cinv_t'), lmax, nside, cl, transf, ninv_t)  # monopole + dipole marginalised (default)
    tlm = cinv_t.apply_ivf(tmap)
    ref = hp.almxfl(shts.map2alm(tmap, lmax=lmax, iter=0), cinv_t.get_ftl() * utils.cli(transf))
    ls = np.concatenate([np.arange(m, lmax + 1) for m in range(lmax + 1)])
    sel = (ls >= 2) & (ls <= 800)
    assert relrms(tlm[sel], ref[sel]) < 2e-3
    assert os.path.exists(str(tmp_path / 'cinv_t' / 'ftl.dat')) and os.path.exists(str(tmp_path / 'cinv_t' / 'fmask.fits.gz'))
    assert os.path.exists(str(tmp_path / 'cinv_t' / 'dense.pk'))
    # polarization
    e, b = hp.synalm(cl['ee'], lmax, rng), hp.synalm(cl['bb'], lmax, rng)
    q, u = shts.alm2map_spin([hp.almxfl(e, transf), hp.almxfl(b, transf)], nside, 2, lmax)
    q = q + nlev_p / vamin * rng.standard_normal(npix)
    u = u + nlev_p / vamin * rng.standard_normal(npix)
    cinv_p = filt_cinv.cinv_p(str(tmp_path / 'cinv_p'), lmax, nside, cl, transf, [[np.ones(npix) * (vamin / nlev_p) ** 2]])
    elm, blm = cinv_p.apply_ivf([q, u])
    er, br = shts.map2alm_spin([q, u], 2, lmax)
    er, br = hp.almxfl(er, cinv_p.get_fel() * utils.cli(transf)), hp.almxfl(br, cinv_p.get_fbl() * utils.cli(transf))
    assert relrms(elm[sel], er[sel]) < 2e-3 and relrms(blm[sel], br[sel]) < 2e-2
    # The nested preconditioners are replayed as captured HIP graphs from the third top-level iteration on.  The temperature
    # graphs were recorded before the polarization solve made the shared coarse plans grow their workspaces (spin 0 -> 2):
    # replaying them afterwards must still give the first answer (outgrown workspaces are retired, not freed).
    assert relrms(cinv_t.apply_ivf(tmap), tlm) < 1e-12


def test_tp_operators_and_chain_vs_reference(g):
    """Joint T + P filter (opfilt_tp, dense.pre_op_dense_tp, teblm): operators and the multigrid solve against the
    reference's own run on the same TE-correlated inputs (T monopole + dipole marginalised, separate P beam)."""
    import torch
    from plancklens_amd import dev
    from plancklens_amd.qcinv import multigrid, opfilt_tp
    from plancklens_amd.qcinv.util_alm import teblm
    lmax, nside = int(g['lmax']), int(g['nside'])
    cl = {'tt': g['cl_tt'], 'ee': g['cl_ee'], 'bb': g['cl_bb'], 'te': g['cl_te']}
    nf = opfilt_tp.alm_filter_ninv([g['ninv_t'], g['ninv_p']], g['transf'], b_transf_e=g['transf_e'], b_transf_b=g['transf_e'],
                                   marge_monopole=True, marge_dipole=True)
    x = teblm([dev.to_dev(g['cg_tp_xt']), dev.to_dev(g['cg_tp_xe']), dev.to_dev(g['cg_tp_xb'])])
    x0 = [a.clone() for a in (x.tlm, x.elm, x.blm)]
    f = opfilt_tp.fwd_op(cl, nf)(x)
    for a, k in ((f.tlm, 'cg_tp_fwd_t'), (f.elm, 'cg_tp_fwd_e'), (f.blm, 'cg_tp_fwd_b')):
        assert relrms(dev.to_host(a), g[k]) < 1e-11
    dg = opfilt_tp.pre_op_diag(cl, nf)(x)
    for a, k in ((dg.tlm, 'cg_tp_diag_t'), (dg.elm, 'cg_tp_diag_e'), (dg.blm, 'cg_tp_diag_b')):
        assert relrms(dev.to_host(a), g[k]) < 1e-11
    assert all(bool((a == b).all()) for a, b in zip((x.tlm, x.elm, x.blm), x0))  # operators leave their argument alone
    assert abs(opfilt_tp.dot_op()(x, f) - float(g['cg_tp_dot'])) < 1e-10 * abs(float(g['cg_tp_dot']))
    pr = opfilt_tp.calc_prep([g['tmap'], g['qmap'], g['umap']], cl, nf)
    for a, k in ((pr.tlm, 'cg_tp_prep_t'), (pr.elm, 'cg_tp_prep_e'), (pr.blm, 'cg_tp_prep_b')):
        assert relrms(dev.to_host(a), g[k]) < 1e-11
    chain = multigrid.multigrid_chain(opfilt_tp, _chain_descr(lmax, nside, 5, 4), cl, nf)
    n = g['cg_tp_tlm'].size
    z = lambda: torch.zeros(n, dtype=torch.complex128, device='cuda')
    sol = teblm([z(), z(), z()])
    chain.solve(sol, [g['tmap'], g['qmap'], g['umap']])
    assert relrms(dev.to_host(sol.tlm), g['cg_tp_tlm']) < 1e-8
    assert relrms(dev.to_host(sol.elm), g['cg_tp_elm']) < 1e-8 and relrms(dev.to_host(sol.blm), g['cg_tp_blm']) < 1e-7


def test_cinv_tp_fullsky_white_noise_known_answer(tmp_path):
    """Joint filter on a full sky with white noise: for l >= 2 the solution is the per-l 3x3 isotropic filter
    (C + N / b^2)^-1 applied to the beam-deconvolved alms (cinv_tp.get_fal), TE coupling included."""
    from plancklens_amd import hp, shts, utils
    from plancklens_amd.filt import filt_cinv
    rng = np.random.default_rng(5)
    nside, lmax = 512, 1024
    npix = 12 * nside ** 2
    ell = np.arange(lmax + 1.)
    cl = {'tt': np.where(ell >= 2, 1e4 / np.maximum(ell, 1) ** 2.5, 0.), 'ee': np.where(ell >= 2, 50. / np.maximum(ell, 1) ** 2, 0.),
          'bb': np.where(ell >= 2, 1. / np.maximum(ell, 1) ** 2, 0.)}
    cl['te'] = 0.5 * np.sqrt(cl['tt'] * cl['ee'])
    transf = hp.gauss_beam(10. / 60 / 180 * np.pi, lmax=lmax)
    nlev_t, nlev_p = 30., 40.
    vamin = np.sqrt(hp.nside2pixarea(nside, degrees=True)) * 60
    u1, u2 = hp.synalm(np.ones(lmax + 1), lmax, rng), hp.synalm(np.ones(lmax + 1), lmax, rng)
    r = cl['te'] * utils.cli(np.sqrt(cl['tt']))
    tlm = hp.almxfl(u1, np.sqrt(cl['tt']))
    elm = hp.almxfl(u1, r) + hp.almxfl(u2, np.sqrt(np.maximum(cl['ee'] - r ** 2, 0.)))
    blm = hp.synalm(cl['bb'], lmax, rng)
    tmap = shts.alm2map(hp.almxfl(tlm, transf), nside) + nlev_t / vamin * rng.standard_normal(npix)
    q, u = shts.alm2map_spin([hp.almxfl(elm, transf), hp.almxfl(blm, transf)], nside, 2, lmax)
    q = q + nlev_p / vamin * rng.standard_normal(npix)
    u = u + nlev_p / vamin * rng.standard_normal(npix)
    ninv = [[np.ones(npix) * (vamin / nlev_t) ** 2], [np.ones(npix) * (vamin / nlev_p) ** 2]]
    # the reference's default 4-stage chain, with the dense block at lmax 32 instead of 64: its 12 675 column build
    # (one coarse fwd_op each) would take two minutes of the GPU test budget, 3 267 columns take half a minute
    from plancklens_amd.qcinv import cd_solve
    pcf = str(tmp_path / 'cinv_tp' / 'dense_tp.pk')
    chain_descr = [[3, ["split(dense(%s), 32, diag_cl)" % pcf], 256, 128, 3, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()],
                   [2, ["split(stage(3),  256, diag_cl)"], 512, 256, 3, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()],
                   [1, ["split(stage(2),  512, diag_cl)"], 1024, 512, 3, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()],
                   [0, ["split(stage(1), 1024, diag_cl)"], lmax, nside, np.inf, 1.0e-5, cd_solve.tr_cg, cd_solve.cache_mem()]]
    cinv = filt_cinv.cinv_tp(str(tmp_path / 'cinv_tp'), lmax, nside, cl, transf, ninv, marge_monopole=True, marge_dipole=True,
                             chain_descr=chain_descr)
    st, se, sb = cinv.apply_ivf([tmap, q, u])
    fal = cinv.get_fal()
    dt = hp.almxfl(shts.map2alm(tmap, lmax=lmax, iter=0), utils.cli(transf))
    de, db = shts.map2alm_spin([q, u], 2, lmax)
    de, db = hp.almxfl(de, utils.cli(transf)), hp.almxfl(db, utils.cli(transf))
    rt = hp.almxfl(dt, fal['tt']) + hp.almxfl(de, fal['te'])
    re = hp.almxfl(dt, fal['te']) + hp.almxfl(de, fal['ee'])
    rb = hp.almxfl(db, fal['bb'])
    ls = np.concatenate([np.arange(m, lmax + 1) for m in range(lmax + 1)])
    sel = (ls >= 2) & (ls <= 800)
    assert relrms(st[sel], rt[sel]) < 2e-3 and relrms(se[sel], re[sel]) < 2e-3 and relrms(sb[sel], rb[sel]) < 2e-2
    assert os.path.exists(str(tmp_path / 'cinv_tp' / 'fal.pk')) and os.path.exists(str(tmp_path / 'cinv_tp' / 'dense_tp.pk'))


def test_masked_cg_at_baseline_size_graph_replay_equals_eager(tmp_path):
    """BASELINE config 4 at full size (nside = lmax = 2048, masked sky, default 4-stage chain): the captured HIP graph of the
    nested preconditioner must reproduce the eager solve -- same iterates to rounding -- and the residual must fall.  Few
    top-level iterations; the dense block (4225 coarse operator applications) is built once and shared through its cache."""
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tools'))
import cg_bench
from plancklens_amd import dev, hp, shts, utils
from plancklens_amd.filt import filt_cinv
nside = lmax = 2048
rng = np.random.default_rng(7)
cl = utils.camb_clfile(os.path.join(%r, 'plancklens_amd', 'data', 'cls', 'FFP10_wdipole_lensedCls.dat'), lmax=lmax)
transf = hp.gauss_beam(5. / 60. / 180. * np.pi, lmax=lmax)
mask = cg_bench.make_mask(nside, rng)
vamin = np.sqrt(hp.nside2pixarea(nside, degrees=True)) * 60
tmap = shts.alm2map(hp.almxfl(hp.synalm(cl['tt'], lmax, rng), transf), nside) + 35. / vamin * rng.standard_normal(12 * nside ** 2)
tmp = sys.argv[1]
f = filt_cinv.cinv_t(os.path.join(tmp, 'cinv_t'), lmax, nside, cl, transf, [np.array([3. / 35. ** 2]) * mask],
                     chain_descr=cg_bench.chain('t', 6, lmax, nside, os.path.join(tmp, 'dense_t.pk')))
trace = []
log0 = f.chain.log
f.chain.log = lambda stage, it, eps, **kw: (trace.append(float(eps)) if stage.depth == 0 else None, log0(stage, it, eps, **kw))
out = dev.to_host(dev.to_dev(f.apply_ivf(dev.to_dev(tmap))))
from plancklens_amd import options
tag = '1' if options.opts.cg_graph else '0'
assert options.stats['cg_graph_fallbacks'] == 0 and (options.stats['cg_graph_captures'] > 0) == options.opts.cg_graph, options.stats
np.save(os.path.join(tmp, 'sol_%%s.npy' %% tag), out)
np.save(os.path.join(tmp, 'eps_%%s.npy' %% tag), np.array(trace))
''' % (ROOT, ROOT, ROOT)
    script = tmp_path / 'cg_full.py'
    script.write_text(code)
    for graph in ('0', '1'):  # the second run reads the dense block cached by the first
        out = subprocess.run([sys.executable, str(script), str(tmp_path)], env=dict(os.environ, PLENS_OPTIONS='cg_graph=' + graph), stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT, timeout=1500)
        assert out.returncode == 0, out.stdout.decode()[-3000:]
    a, b = np.load(tmp_path / 'sol_0.npy'), np.load(tmp_path / 'sol_1.npy')
    assert np.all(np.isfinite(a)) and np.abs(a).max() > 0
    assert np.sqrt(np.sum(np.abs(a - b) ** 2) / np.sum(np.abs(a) ** 2)) < 1e-10
    ea, eb = np.load(tmp_path / 'eps_0.npy'), np.load(tmp_path / 'eps_1.npy')
    if ea.size and eb.size:  # residual norms are only evaluated when the top level logs them
        assert ea[-1] < 1e-2 * ea[0] and np.allclose(ea, eb, rtol=1e-6)


@pytest.mark.parametrize('nside,lmax,marge', [(32, 64, 'md'), (128, 256, 'md'), (256, 400, 'm'), (256, 512, ''), (512, 700, 'md'), (64, 100, 'maps')])
def test_tt_one_call_operator(nside, lmax, marge):
    """pl_cg_fwd_tt (weighting and projection inside the ring-FFT launches on the all-generic grids, nside <= 256 here; the separate
    projection launches above) against the operator assembled from alm2map / apply_map / map2alm / almxfl_add."""
    import torch
    from plancklens_amd import dev, hp, shts
    from plancklens_amd.qcinv import opfilt_tt
    rng = np.random.default_rng(nside + lmax)
    npix = 12 * nside ** 2
    ninv = rng.uniform(0.5, 1.5, npix) * (rng.uniform(size=npix) > 0.2)
    maps = [rng.standard_normal(npix) for _ in range(6)] if marge == 'maps' else []
    bl = hp.gauss_beam(np.radians(0.3), lmax=lmax)
    nf = opfilt_tt.alm_filter_ninv(ninv, bl, marge_monopole='m' in marge and marge != 'maps', marge_dipole='d' in marge, marge_maps=maps)
    cl = {'tt': 1e3 / (np.arange(lmax + 1) + 10.) ** 2}
    cl['tt'][:2] = 0.
    x = rng.standard_normal(hp.Alm.getsize(lmax)) + 1j * rng.standard_normal(hp.Alm.getsize(lmax))
    x[:lmax + 1] = x[:lmax + 1].real
    x = dev.to_dev(x)
    x0 = x.clone()
    op = opfilt_tt.fwd_op(cl, nf)
    assert nf.one_call_ok(x)
    got = op(x)
    tmap = shts.alm2map(x, nside, lmax=lmax, fl=bl)
    nf.apply_map(tmap)
    ref = shts.map2alm(tmap, lmax=lmax, iter=0, fl=bl * (npix / (4. * np.pi)))
    ref = dev.almxfl_add(ref, x, op.cltt_inv)
    assert bool((x == x0).all())
    assert relrms(dev.to_host(got), dev.to_host(ref)) < 1e-13
    assert bool((op(x) == got).all())  # bit-reproducible
    got2 = nf.apply_alm_new(x)  # without the S^-1 term
    assert relrms(dev.to_host(got2), dev.to_host(shts.map2alm(tmap, lmax=lmax, iter=0, fl=bl * (npix / (4. * np.pi))))) < 1e-13


@pytest.mark.parametrize('nside,lmax', [(32, 64), (128, 256), (256, 400), (512, 700), (1024, 1500)])
def test_pp_one_call_operator(nside, lmax):
    """pl_cg_fwd_pp (inverse-noise weighting inside the synthesis-side ring-FFT launches of every kernel class, S^-1 term inside the
    analysis post-processing) against the operator assembled from alm2map_spin / apply_map / map2alm_spin / almxfl_add."""
    from plancklens_amd import dev, hp, shts
    from plancklens_amd.qcinv import opfilt_pp
    from plancklens_amd.qcinv.util_alm import eblm
    rng = np.random.default_rng(nside + lmax)
    npix = 12 * nside ** 2
    ninv = rng.uniform(0.5, 1.5, npix) * (rng.uniform(size=npix) > 0.2)
    bl = hp.gauss_beam(np.radians(0.3), lmax=lmax)
    nf = opfilt_pp.alm_filter_ninv([ninv], bl)
    cl = {'ee': 1e2 / (np.arange(lmax + 1) + 10.) ** 2, 'bb': 1e1 / (np.arange(lmax + 1) + 10.) ** 2}
    cl['ee'][:2] = 0.
    cl['bb'][:2] = 0.

    def ralm():
        x = rng.standard_normal(hp.Alm.getsize(lmax)) + 1j * rng.standard_normal(hp.Alm.getsize(lmax))
        x[:lmax + 1] = x[:lmax + 1].real
        return dev.to_dev(x)
    x = eblm([ralm(), ralm()])
    e0, b0 = x.elm.clone(), x.blm.clone()
    op = opfilt_pp.fwd_op(cl, nf)
    assert nf.one_call_ok(x)
    got = op(x)
    ref = nf._apply_alm_steps(x)
    sl = op.s_inv_filt.slinv
    ref_e = dev.almxfl_add(ref.elm, x.elm, sl[:, 0, 0])
    ref_b = dev.almxfl_add(ref.blm, x.blm, sl[:, 1, 1])
    assert bool((x.elm == e0).all()) and bool((x.blm == b0).all())
    assert relrms(dev.to_host(got.elm), dev.to_host(ref_e)) < 1e-13
    assert relrms(dev.to_host(got.blm), dev.to_host(ref_b)) < 1e-13
    again = op(x)
    assert bool((again.elm == got.elm).all()) and bool((again.blm == got.blm).all())
    got2 = nf.apply_alm_new(x)  # without the S^-1 term
    assert relrms(dev.to_host(got2.elm), dev.to_host(ref.elm)) < 1e-13 and relrms(dev.to_host(got2.blm), dev.to_host(ref.blm)) < 1e-13
