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
