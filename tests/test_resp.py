"""Analytic responses and N0 (plancklens_amd.qresp.get_response, nhl.get_nhl, utils_spin.wignerc, wigners) on the CPU.

* the numpy Wigner series against vectors produced by the reference's Fortran module (oracle/_ref/libwigners_ref.so,
  built from /root/reference by oracle/Makefile) -- committed in tests/golden/resp_golden.npz;
* responses / N0 against the reference's own Python (qresp.get_response, nhl.get_nhl run on that Fortran build), same file;
* the reference's own known answer (tests/test_w.py of the reference): for optimally filtered Gaussian fields the
  unnormalised N0 equals the response, for sources 'p' and 'f', separately and jointly filtered."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, 'golden', 'resp_golden.npz')


@pytest.fixture(scope='module')
def g():
    return np.load(GOLD)


def test_gauss_legendre_and_wigner_series_vs_reference_fortran(g):
    from plancklens_amd import wigners as w
    for n in (7, 64, 301):
        x, wg = w.get_xgwg(-1., 1., n)
        assert np.allclose(x, g['xg_%d' % n], rtol=0, atol=2e-15) and np.allclose(wg, g['wg_%d' % n], rtol=0, atol=5e-15)
    x, cl, xi = g['w_x'], g['w_cl'], g['w_xi']
    lmax = cl.size - 1
    k = 0
    for s1 in range(-3, 4):
        for s2 in range(-3, 4):
            a, b = w.wignerpos(cl, x, s1, s2), g['w_pos'][k]
            assert np.abs(a - b).max() <= 1e-12 * max(np.abs(b).max(), 1e-30), (s1, s2)
            c, d = w.wignercoeff(xi, x, s1, s2, lmax), g['w_coeff'][k]
            assert np.abs(c - d).max() <= 1e-12 * max(np.abs(d).max(), 1e-30), (s1, s2)
            k += 1


def _inputs(g):
    cls = {k: g['cl_' + k] for k in ['tt', 'te', 'ee', 'bb']}
    fal = {k: g['fal_' + k] for k in ['tt', 'ee', 'bb']}
    return cls, fal, int(g['lmax_ivf']), int(g['lmax_qlm'])


@pytest.mark.parametrize('key,source', [('ptt', 'p'), ('p_p', 'p'), ('p', 'p'), ('x', 'x'), ('p', 'f'), ('ftt', 'f'), ('ptt_bh_s', 'p'), ('a_p', 'a')])
def test_get_response_vs_reference(g, key, source):
    from plancklens_amd import qresp
    cls, fal, lmax_ivf, lmax_qlm = _inputs(g)
    R = qresp.get_response(key, lmax_ivf, source, cls, cls, fal, lmax_qlm=lmax_qlm)
    for r, tag in zip(R, ['GG', 'CC', 'GC', 'CG']):
        ref = g['R_%s_%s_%s' % (key, source, tag)]
        assert np.abs(r - ref).max() <= 1e-10 * max(np.abs(ref).max(), 1e-300), (key, source, tag)


@pytest.mark.parametrize('k1,k2', [('ptt', 'ptt'), ('p_p', 'p_p'), ('p', 'p'), ('p_p', 'ptt'), ('x', 'x'), ('p', 'x')])
def test_get_nhl_vs_reference(g, k1, k2):
    from plancklens_amd import nhl
    cls, fal, lmax_ivf, lmax_qlm = _inputs(g)
    cls_ivfs = {k: g['ivf_' + k] for k in ['tt', 'ee', 'bb', 'te']}
    N = nhl.get_nhl(k1, k2, cls, cls_ivfs, lmax_ivf, lmax_ivf, lmax_out=lmax_qlm)
    for n, tag in zip(N, ['GG', 'CC', 'GC', 'CG']):
        ref = g['N_%s_%s_%s' % (k1, k2, tag)]
        assert np.abs(n - ref).max() <= 1e-10 * max(np.abs(ref).max(), 1e-300), (k1, k2, tag)


def test_n0_equals_response_for_optimal_filtering():
    """tests/test_w.py of the reference: lmax_ivf = 500, lmin_ivf = 100, 35 / 35 sqrt(2) uK-amin, 6' beam."""
    from plancklens_amd import hp, nhl, qresp, utils
    cls_path = os.path.join(os.path.dirname(HERE), 'plancklens_amd', 'data', 'cls', 'FFP10_wdipole_lensedCls.dat')
    lmax_ivf, lmin_ivf, nlev_t, nlev_p = 500, 100, 35., 35. * np.sqrt(2.)
    lmax_qlm = lmax_ivf
    transf = hp.gauss_beam(6. / 60. / 180. * np.pi, lmax=lmax_ivf)
    cls_len = utils.camb_clfile(cls_path)
    cls_weight = utils.camb_clfile(cls_path)
    nt, npol = (nlev_t / 60. / 180. * np.pi) ** 2 / transf ** 2, (nlev_p / 60. / 180. * np.pi) ** 2 / transf ** 2
    cls_dat = {'tt': cls_len['tt'][:lmax_ivf + 1] + nt, 'ee': cls_len['ee'][:lmax_ivf + 1] + npol,
               'bb': cls_len['bb'][:lmax_ivf + 1] + npol, 'te': np.copy(cls_len['te'][:lmax_ivf + 1])}
    fal_sep = {k: utils.cli(cls_dat[k]) for k in ['tt', 'ee', 'bb']}
    ivf_sep = {k: fal_sep[k].copy() for k in ['tt', 'ee', 'bb']}
    ivf_sep['te'] = cls_len['te'][:lmax_ivf + 1] * fal_sep['tt'] * fal_sep['ee']
    fal_jt, ivf_jt = utils.cl_inverse(cls_dat), utils.cl_inverse(cls_dat)
    for d in (fal_sep, fal_jt, ivf_sep, ivf_jt):
        for cl in d.values():
            cl[:max(1, lmin_ivf)] *= 0.
    for src in ['p', 'f']:
        for key in [src + 'tt', src + '_p', src]:
            NG, NC, NGC, NCG = nhl.get_nhl(key, key, cls_weight, ivf_sep, lmax_ivf, lmax_ivf, lmax_out=lmax_qlm)
            RG, RC, RGC, RCG = qresp.get_response(key, lmax_ivf, src, cls_weight, cls_len, fal_sep, lmax_qlm=lmax_qlm)
            if key[1:] in ['tt', '_p']:
                assert np.allclose(NG[1:], RG[1:], rtol=1e-6), key
                assert np.allclose(NC[2:], RC[2:], rtol=1e-6), key
            assert np.all(NCG == 0.) and np.all(NGC == 0.) and np.all(RCG == 0.) and np.all(RGC == 0.)
        NG, NC, NGC, NCG = nhl.get_nhl(src, src, cls_weight, ivf_jt, lmax_ivf, lmax_ivf, lmax_out=lmax_qlm)
        RG, RC, RGC, RCG = qresp.get_response(src, lmax_ivf, src, cls_weight, cls_len, fal_jt, lmax_qlm=lmax_qlm)
        assert np.allclose(NG[1:], RG[1:], rtol=1e-6) and np.allclose(NC[2:], RC[2:], rtol=1e-6), src
        assert np.all(NCG == 0.) and np.all(NGC == 0.) and np.all(RCG == 0.) and np.all(RGC == 0.)


def test_mean_field_response_vs_reference():
    """qresp.get_mf_resp (qresp.py:421-500) against the reference's own Python on its Fortran Wigner module
    (tests/golden/mfresp_golden.npz, made by tests/golden/make_golden.py mfresp): 'ptt' and 'p_p', gradient and curl parts and the
    three pieces of the gradient response.  The responses are differences of large terms (C_L=1 ~ 8e4 against G - C ~ 2e2): the
    tolerance is relative to the largest piece."""
    from plancklens_amd import qresp
    g = np.load(os.path.join(HERE, 'golden', 'mfresp_golden.npz'))
    cls = {k: g['cl_' + k] for k in ['tt', 'te', 'ee', 'bb']}
    ivf = {k: g['ivf_' + k] for k in ['tt', 'ee', 'bb']}
    lmax_qe, lmax_out = int(g['lmax_qe']), int(g['lmax_out'])
    for key in ['ptt', 'p_p']:
        GL, CL, terms = qresp.get_mf_resp(key, cls, ivf, lmax_qe, lmax_out, retterms=True)
        scale = np.abs(g['GK_' + key]).max()
        assert np.abs(GL - g['G_' + key]).max() < 1e-11 * scale and np.abs(CL - g['C_' + key]).max() < 1e-11 * scale, key
        assert set(terms) == {'GK', 'GxiK', 'Gcons'}
        for t in terms:
            assert np.abs(terms[t] - g['%s_%s' % (t, key)]).max() < 1e-11 * scale, (key, t)
        G2, C2 = qresp.get_mf_resp(key, cls, ivf, lmax_qe, lmax_out)
        assert np.array_equal(G2, GL) and np.array_equal(C2, CL) and CL[1] == 0.
