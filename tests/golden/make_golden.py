"""Generates tests/golden/*.npz by running the REFERENCE's own Python (imported from /root/reference, in this
container only) on seeded inputs, with its `healpy` dependency (absent here) stood in by the CPU oracle's SHTs
(oracle/sht_oracle.py) and the oracle's own restatement of the healpy host helpers (oracle/hp_oracle.py: Alm, almxfl, alm2cl,
gauss_beam, ud_grade, ... -- independent of the product's plancklens_amd/hp.py, which only lends the FITS readers / writers (the
file formats are pinned by tests/test_host.py) and the pixel angles used to draw the input masks).

What the fixtures pin: everything the reference does ABOVE the SHT seam -- isotropic filtering
(filt_simple.library_fullsky_sepTP), the specialised QE route (qest.library_sepTP.get_sim_qlm), the generic
route (qest.eval_qe), index shuffling / leg symmetrisation (filt_util.library_shuffle), spectra (qecl) --
evaluated with SHTs that are themselves pinned by tests/test_oracle.py.  Only data (inputs and the
reference's outputs) is stored; no reference source is copied.

Run:  python tests/golden/make_golden.py      (needs /root/reference; not needed on the GPU box; `... cinv` separately: 20 minutes; `... cinv2048`
      separately: BASELINE config 4 at nside = lmax = 2048, 3 top-level iterations of the reference's cinv_t / cinv_p)
The generator is reproducible to rounding only: the oracle's threaded stages and the reference's OpenMP Fortran sum in an order that
depends on the thread count and, for the Fortran reductions, on the run (two runs of `resp` differ in 26 of 80 arrays at <= 2e-15).
"""
import os
import shutil
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference'

from oracle import sht_oracle as so  # noqa: E402
from oracle import hp_oracle as oh  # noqa: E402
from plancklens_amd import hp as myhp  # noqa: E402  (FITS IO and the angles of the input masks only)


def install_healpy_standin():
    """A module object named `healpy` built from the oracle SHTs + host helpers (never written to disk)."""
    m = types.ModuleType('healpy')
    for name in ['Alm', 'almxfl', 'alm2cl', 'gauss_beam', 'nside2npix', 'npix2nside', 'nside2pixarea', 'ud_grade', 'pix2ang', 'pix2vec', 'UNSEEN']:
        setattr(m, name, getattr(oh, name))     # arithmetic the reference's results depend on: the oracle's own
    for name in ['read_alm', 'write_alm', 'read_map', 'write_map']:
        setattr(m, name, getattr(myhp, name))   # file formats (tests/test_host.py)
    def alm2map(alm, nside, lmax=None, mmax=None, pol=False, **kw):
        if pol and not isinstance(alm, np.ndarray) and len(alm) == 3:  # (T, E, B) -> (T, Q, U), as healpy's pol=True
            q, u = so.alm2map_spin([np.asarray(alm[1]), np.asarray(alm[2])], nside, 2, so.alm_lmax(len(alm[1])))
            return np.array([so.alm2map(np.asarray(alm[0]), nside, lmax=lmax), q, u])
        return so.alm2map(np.asarray(alm), nside, lmax=lmax)

    def map2alm(mp, lmax=None, mmax=None, iter=0, pol=False, **kw):
        if pol and np.ndim(mp[0]) == 1 and len(mp) == 3:
            e, b = so.map2alm_spin([np.asarray(mp[1]), np.asarray(mp[2])], 2, lmax)
            return np.array([so.map2alm(np.asarray(mp[0]), lmax=lmax, iter=iter), e, b])
        return so.map2alm(np.asarray(mp), lmax=lmax, iter=iter)
    m.alm2map, m.map2alm = alm2map, map2alm
    m.alm2map_spin = lambda gclm, nside, spin, lmax, mmax=None: so.alm2map_spin(gclm, nside, spin, lmax)
    m.map2alm_spin = lambda maps, spin, lmax=None, mmax=None: so.map2alm_spin(maps, spin, lmax)
    proj = types.ModuleType('healpy.projector')
    proj.CartesianProj = object
    m.projector = proj
    sys.modules['healpy'] = m
    sys.modules['healpy.projector'] = proj


class tiny_sims(object):
    """Seeded T, Q, U maps: correlated Gaussian sky x beam + white noise (same recipe as tests/helpers)."""

    def __init__(self, nside, lmax, cls, transf, nlev_t, nlev_p):
        self.nside, self.lmax, self.cls, self.transf, self.nlev_t, self.nlev_p = nside, lmax, cls, transf, nlev_t, nlev_p

    def hashdict(self):
        return {'nside': self.nside, 'lmax': self.lmax}

    def _alms(self, idx):
        rng = np.random.default_rng(1000 + idx)
        n = so.alm_size(self.lmax)

        def unit():
            a = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) / np.sqrt(2.)
            a[:self.lmax + 1] = np.sqrt(2.) * a[:self.lmax + 1].real
            return a
        u1, u2, u3 = unit(), unit(), unit()
        tt, ee, bb, te = (self.cls[k][:self.lmax + 1] for k in ['tt', 'ee', 'bb', 'te'])
        tlm = oh.almxfl(u1, np.sqrt(tt))
        r = te * np.where(tt > 0, 1. / np.sqrt(np.where(tt > 0, tt, 1.)), 0.)
        elm = oh.almxfl(u1, r) + oh.almxfl(u2, np.sqrt(np.maximum(ee - r ** 2, 0.)))
        blm = oh.almxfl(u3, np.sqrt(bb))
        return tlm, elm, blm

    def _noise(self, idx, f):
        rng = np.random.default_rng(2000 + 3 * idx + f)
        vamin = np.sqrt(oh.nside2pixarea(self.nside, degrees=True)) * 60
        return (self.nlev_t if f == 0 else self.nlev_p) / vamin * rng.standard_normal(12 * self.nside ** 2)

    def get_sim_tmap(self, idx):
        tlm, _, _ = self._alms(idx)
        return so.alm2map(oh.almxfl(tlm, self.transf), self.nside, lmax=self.lmax) + self._noise(idx, 0)

    def get_sim_pmap(self, idx):
        _, elm, blm = self._alms(idx)
        q, u = so.alm2map_spin([oh.almxfl(elm, self.transf), oh.almxfl(blm, self.transf)], self.nside, 2, self.lmax)
        return q + self._noise(idx, 1), u + self._noise(idx, 2)


def main():
    assert os.path.isdir(REF), 'the reference is only present in the build container'
    install_healpy_standin()
    sys.path.insert(0, REF)
    from plancklens import qest, qecl, utils  # the REFERENCE modules
    from plancklens.filt import filt_simple, filt_util

    nside, lmax_ivf, lmax_qlm, lmin_ivf = 16, 40, 47, 4
    cls_path = os.path.join(ROOT, 'plancklens_amd', 'data', 'cls', 'FFP10_wdipole_lensedCls.dat')
    cl_len = utils.camb_clfile(cls_path, lmax=lmax_ivf)
    # bring the spectra to O(1) signal-to-noise at these tiny multipoles
    nlev_t, nlev_p = 1200., 35.
    transf = oh.gauss_beam(4. / 180. * np.pi, lmax=lmax_ivf)
    sims = tiny_sims(nside, lmax_ivf, cl_len, transf, nlev_t, nlev_p)
    arcmin = np.pi / 180. / 60.
    ftl = utils.cli(cl_len['tt'][:lmax_ivf + 1] + (nlev_t * arcmin) ** 2 * utils.cli(transf ** 2))
    fel = utils.cli(cl_len['ee'][:lmax_ivf + 1] + (nlev_p * arcmin) ** 2 * utils.cli(transf ** 2))
    fbl = utils.cli(cl_len['bb'][:lmax_ivf + 1] + (nlev_p * arcmin) ** 2 * utils.cli(transf ** 2))
    ftl[:lmin_ivf] = 0; fel[:lmin_ivf] = 0; fbl[:lmin_ivf] = 0

    tmp = tempfile.mkdtemp(prefix='plgolden_')
    out = {'nside': nside, 'lmax_ivf': lmax_ivf, 'lmax_qlm': lmax_qlm, 'nlev_t': nlev_t, 'nlev_p': nlev_p,
           'transf': transf, 'ftl': ftl, 'fel': fel, 'fbl': fbl}
    for k in ['tt', 'ee', 'bb', 'te']:
        out['cl_' + k] = cl_len[k]
    try:
        ivfs = filt_simple.library_fullsky_sepTP(os.path.join(tmp, 'ivfs'), sims, nside, transf, cl_len, ftl, fel, fbl, cache=False)
        for idx in (0, 1):
            out['tmap_%d' % idx] = sims.get_sim_tmap(idx)
            q, u = sims.get_sim_pmap(idx)
            out['qmap_%d' % idx], out['umap_%d' % idx] = q, u
            out['tlm_%d' % idx], out['elm_%d' % idx], out['blm_%d' % idx] = ivfs.get_sim_tlm(idx), ivfs.get_sim_elm(idx), ivfs.get_sim_blm(idx)
        qlms_dd = qest.library_sepTP(os.path.join(tmp, 'qlms_dd'), ivfs, ivfs, cl_len['te'], nside, lmax_qlm=lmax_qlm)
        for k in ['ptt', 'xtt', 'p_p', 'x_p', 'p', 'x', 'stt', 'ftt', 'f_p', 'a_p', 'pte', 'peb', 'p_tp', 'p_eb']:
            out['dd_%s_0' % k] = qlms_dd.get_sim_qlm(k, 0)
        out['dd_p_1'] = qlms_dd.get_sim_qlm('p', 1)
        out['dd_mf_p'] = qlms_dd.get_sim_qlm_mf('p', np.array([0, 1]))
        # different legs -> symmetrised estimators (qest.py:327-332)
        ivfs_s = filt_util.library_shuffle(ivfs, {0: 1, 1: 0})
        qlms_ds = qest.library_sepTP(os.path.join(tmp, 'qlms_ds'), ivfs, ivfs_s, cl_len['te'], nside, lmax_qlm=lmax_qlm)
        for k in ['ptt', 'p_p', 'p', 'x', 'ftt', 'f_p']:
            out['ds_%s_0' % k] = qlms_ds.get_sim_qlm(k, 0)
        # generic spin-weight route (qest.py:19-39)
        get_alm = lambda a: {'t': ivfs.get_sim_tlm, 'e': ivfs.get_sim_elm, 'b': ivfs.get_sim_blm}[a](0)
        for k in ['ptt', 'p_p', 'p']:
            G, C = qest.eval_qe(k, lmax_ivf, cl_len, get_alm, nside, lmax_qlm, verbose=False)
            out['gen_%s_G' % k], out['gen_%s_C' % k] = G, C
        # spectra (qecl.py:85-124)
        qcls = qecl.library(os.path.join(tmp, 'qcls'), qlms_dd, qlms_dd, np.array([]))
        out['qcl_p_0'] = qcls.get_sim_qcl('p', 0)
        out['qcl_ptt_p_p_0'] = qcls.get_sim_qcl('ptt', 0, k2='p_p')
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    np.savez_compressed(os.path.join(HERE, 'qe_golden.npz'), **out)
    print('wrote qe_golden.npz with %d arrays' % len(out))
    for k in ['ptt', 'p_p', 'p']:
        d = np.abs(out['gen_%s_G' % k] - out['dd_%s_0' % k]).max() / np.abs(out['dd_%s_0' % k]).max()
        print('reference route agreement (library vs eval_qe) %s: %.2e' % (k, d))


def make_cg_golden():
    """Reference multigrid CG (qcinv.multigrid_chain + opfilt_tt / opfilt_pp + dense + template_removal) on a masked,
    inhomogeneous-noise sky at nside 16: solutions after a fixed number of iterations and the residual trace."""
    from plancklens.qcinv import multigrid, opfilt_tt, opfilt_pp, cd_solve, util_alm
    nside, lmax = 16, 32
    npix = 12 * nside ** 2
    rng = np.random.default_rng(77)
    ell = np.arange(lmax + 1.)
    cl = {'tt': np.where(ell >= 2, 3e3 / np.maximum(ell, 1.) ** 2.2, 0.), 'ee': np.where(ell >= 2, 60. / np.maximum(ell, 1.) ** 1.8, 0.),
          'bb': np.where(ell >= 2, 2. / np.maximum(ell, 1.) ** 1.5, 0.)}
    transf = oh.gauss_beam(6. / 180. * np.pi, lmax=lmax)
    th, ph = myhp.pix2ang(nside)
    mask = (np.abs(np.cos(th)) > 0.25).astype(float)                         # galactic-like band removed
    ninv_t = mask * (0.5 + 0.4 * np.sin(3 * ph) * np.sin(th)) / 40. ** 2 * (npix / (4 * np.pi)) * 1e-4
    ninv_p = mask * (0.6 + 0.3 * np.cos(2 * ph)) / 10. ** 2 * (npix / (4 * np.pi)) * 1e-4
    tmap = so.alm2map(oh.almxfl(oh.synalm(cl['tt'], lmax, rng), transf), nside) + rng.standard_normal(npix) * 40.
    e, b = oh.synalm(cl['ee'], lmax, rng), oh.synalm(cl['bb'], lmax, rng)
    q, u = so.alm2map_spin([oh.almxfl(e, transf), oh.almxfl(b, transf)], nside, 2, lmax)
    qmap, umap = q + rng.standard_normal(npix) * 10., u + rng.standard_normal(npix) * 10.
    out = {'nside': nside, 'lmax': lmax, 'transf': transf, 'ninv_t': ninv_t, 'ninv_p': ninv_p, 'tmap': tmap, 'qmap': qmap,
           'umap': umap, 'cl_tt': cl['tt'], 'cl_ee': cl['ee'], 'cl_bb': cl['bb']}
    trace = []

    def chain_descr(niter, dense_lmax):
        return [[1, ["split(dense(), %d, diag_cl)" % dense_lmax], 16, 8, 3, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()],
                [0, ["split(stage(1), 16, diag_cl)"], lmax, nside, niter, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()]]

    # temperature, monopole + dipole marginalised
    n_inv_filt = opfilt_tt.alm_filter_ninv(ninv_t, transf, marge_monopole=True, marge_dipole=True)
    chain = multigrid.multigrid_chain(opfilt_tt, chain_descr(6, 6), cl, n_inv_filt)
    orig_log = chain.log
    chain.log = lambda stage, it, eps, **kw: (trace.append((stage.depth, it, eps)), orig_log(stage, it, eps, **kw))
    talm = np.zeros(so.alm_size(lmax), dtype=complex)
    chain.solve(talm, tmap)
    out['cg_tlm'] = talm
    out['cg_t_trace'] = np.array([t[2] for t in trace if t[0] == 0])
    out['cg_t_prep'] = opfilt_tt.calc_prep(tmap, cl, n_inv_filt)
    x = oh.synalm(cl['tt'], lmax, rng)
    out['cg_t_x'] = x
    out['cg_t_fwd'] = opfilt_tt.fwd_op(cl, n_inv_filt)(x)
    out['cg_t_diag'] = opfilt_tt.pre_op_diag(cl, n_inv_filt)(x)
    # polarization
    del trace[:]
    n_inv_filt_p = opfilt_pp.alm_filter_ninv([ninv_p], transf)
    chain_p = multigrid.multigrid_chain(opfilt_pp, chain_descr(5, 5), cl, n_inv_filt_p)
    orig_log_p = chain_p.log
    chain_p.log = lambda stage, it, eps, **kw: (trace.append((stage.depth, it, eps)), orig_log_p(stage, it, eps, **kw))
    palm = util_alm.eblm([np.zeros(so.alm_size(lmax), dtype=complex), np.zeros(so.alm_size(lmax), dtype=complex)])
    chain_p.solve(palm, [qmap, umap])
    out['cg_elm'], out['cg_blm'] = palm.elm, palm.blm
    out['cg_p_trace'] = np.array([t[2] for t in trace if t[0] == 0])
    xe = util_alm.eblm([oh.synalm(cl['ee'], lmax, rng), oh.synalm(cl['bb'], lmax, rng)])
    out['cg_p_xe'], out['cg_p_xb'] = xe.elm, xe.blm
    f = opfilt_pp.fwd_op(cl, n_inv_filt_p)(xe)
    out['cg_p_fwd_e'], out['cg_p_fwd_b'] = f.elm, f.blm
    # joint temperature + polarization (opfilt_tp): TE-correlated spectra, T monopole + dipole marginalised
    from plancklens.qcinv import opfilt_tp
    del trace[:]
    cl_tp = dict(cl)
    cl_tp['te'] = 0.6 * np.sqrt(cl['tt'] * cl['ee'])
    out['cl_te'] = cl_tp['te']
    transf_e = oh.gauss_beam(7. / 180. * np.pi, lmax=lmax)   # different beam for polarization
    out['transf_e'] = transf_e
    n_inv_filt_tp = opfilt_tp.alm_filter_ninv([ninv_t, ninv_p], transf, b_transf_e=transf_e, b_transf_b=transf_e,
                                              marge_monopole=True, marge_dipole=True)
    chain_tp = multigrid.multigrid_chain(opfilt_tp, chain_descr(5, 4), cl_tp, n_inv_filt_tp)
    orig_log_tp = chain_tp.log
    chain_tp.log = lambda stage, it, eps, **kw: (trace.append((stage.depth, it, eps)), orig_log_tp(stage, it, eps, **kw))
    z = lambda: np.zeros(so.alm_size(lmax), dtype=complex)
    tpalm = util_alm.teblm([z(), z(), z()])
    chain_tp.solve(tpalm, [tmap, qmap, umap])
    out['cg_tp_tlm'], out['cg_tp_elm'], out['cg_tp_blm'] = tpalm.tlm, tpalm.elm, tpalm.blm
    out['cg_tp_trace'] = np.array([t[2] for t in trace if t[0] == 0])
    xt = util_alm.teblm([oh.synalm(cl['tt'], lmax, rng), oh.synalm(cl['ee'], lmax, rng), oh.synalm(cl['bb'], lmax, rng)])
    out['cg_tp_xt'], out['cg_tp_xe'], out['cg_tp_xb'] = xt.tlm, xt.elm, xt.blm
    f = opfilt_tp.fwd_op(cl_tp, n_inv_filt_tp)(xt)
    out['cg_tp_fwd_t'], out['cg_tp_fwd_e'], out['cg_tp_fwd_b'] = f.tlm, f.elm, f.blm
    dg = opfilt_tp.pre_op_diag(cl_tp, n_inv_filt_tp)(xt)
    out['cg_tp_diag_t'], out['cg_tp_diag_e'], out['cg_tp_diag_b'] = dg.tlm, dg.elm, dg.blm
    pr = opfilt_tp.calc_prep([tmap, qmap, umap], cl_tp, n_inv_filt_tp)
    out['cg_tp_prep_t'], out['cg_tp_prep_e'], out['cg_tp_prep_b'] = pr.tlm, pr.elm, pr.blm
    out['cg_tp_dot'] = opfilt_tp.dot_op()(xt, f)
    np.savez_compressed(os.path.join(HERE, 'cg_golden.npz'), **out)
    print('wrote cg_golden.npz; T residual trace', out['cg_t_trace'], 'P', out['cg_p_trace'], 'TP', out['cg_tp_trace'])


def install_fortran_wigners():
    """`plancklens.wigners.wigners` (an f2py extension in the reference) stood in by ctypes calls into
    oracle/_ref/libwigners_ref.so, which oracle/Makefile builds from the reference's own wigners.f90."""
    import ctypes
    lib = ctypes.CDLL(os.path.join(ROOT, 'oracle', '_ref', 'libwigners_ref.so'))
    dp = ctypes.POINTER(ctypes.c_double)
    ci, cd, ref = ctypes.c_int, ctypes.c_double, ctypes.byref
    m = types.ModuleType('plancklens.wigners.wigners')

    def get_xgwg(x1, x2, n):
        x, w = np.zeros(n), np.zeros(n)
        lib.get_xgwg_(ref(cd(x1)), ref(cd(x2)), x.ctypes.data_as(dp), w.ctypes.data_as(dp), ref(ci(n)))
        return x, w

    def wignerpos(cl, x, s1, s2):
        cl, x = np.ascontiguousarray(cl, dtype=float), np.ascontiguousarray(x, dtype=float)
        xi = np.zeros_like(x)
        if cl.size - 1 <= max(abs(s1), abs(s2)):  # the Fortran writes out of bounds when lmax == lmin: pad
            cl = np.concatenate([cl, np.zeros(max(abs(s1), abs(s2)) + 2 - cl.size)])
        lib.wignerpos_(xi.ctypes.data_as(dp), ref(ci(x.size)), ref(ci(cl.size - 1)), cl.ctypes.data_as(dp), x.ctypes.data_as(dp),
                       ref(ci(s1)), ref(ci(s2)))
        return xi

    def wignercoeff(xi, x, s1, s2, lmax):
        xi, x = np.ascontiguousarray(xi, dtype=float), np.ascontiguousarray(x, dtype=float)
        cl = np.zeros(lmax + 1)
        lib.wignercoeff_(cl.ctypes.data_as(dp), xi.ctypes.data_as(dp), x.ctypes.data_as(dp), ref(ci(s1)), ref(ci(s2)),
                         ref(ci(lmax)), ref(ci(x.size)))
        return cl
    m.get_xgwg, m.wignerpos, m.wignercoeff = get_xgwg, wignerpos, wignercoeff
    import plancklens.wigners as pw
    pw.wigners = m
    sys.modules['plancklens.wigners.wigners'] = m
    return m


def make_resp_golden():
    """Reference qresp.get_response / nhl.get_nhl (its Python + its Fortran Wigner module) on the tiny configuration of
    the QE fixtures, plus raw Gauss-Legendre / Wigner-series vectors of the Fortran module."""
    wig = install_fortran_wigners()
    from plancklens import qresp, nhl, utils, utils_spin
    assert utils_spin.HASWIGNER
    lmax_ivf, lmax_qlm, lmin_ivf = 40, 47, 4
    cls_path = os.path.join(ROOT, 'plancklens_amd', 'data', 'cls', 'FFP10_wdipole_lensedCls.dat')
    cl_len = utils.camb_clfile(cls_path, lmax=lmax_ivf)
    transf = oh.gauss_beam(4. / 180. * np.pi, lmax=lmax_ivf)
    arcmin = np.pi / 180. / 60.
    fal = {'tt': utils.cli(cl_len['tt'][:lmax_ivf + 1] + (1200. * arcmin) ** 2 * utils.cli(transf ** 2)),
           'ee': utils.cli(cl_len['ee'][:lmax_ivf + 1] + (35. * arcmin) ** 2 * utils.cli(transf ** 2)),
           'bb': utils.cli(cl_len['bb'][:lmax_ivf + 1] + (35. * arcmin) ** 2 * utils.cli(transf ** 2))}
    for f in fal.values():
        f[:lmin_ivf] = 0
    out = {'lmax_ivf': lmax_ivf, 'lmax_qlm': lmax_qlm}
    for k in ['tt', 'te', 'ee', 'bb']:
        out['cl_' + k] = cl_len[k]
    for k in fal:
        out['fal_' + k] = fal[k]
    for key, source in [('ptt', 'p'), ('p_p', 'p'), ('p', 'p'), ('x', 'x'), ('p', 'f'), ('ftt', 'f'), ('ptt_bh_s', 'p'), ('a_p', 'a')]:
        R = qresp.get_response(key, lmax_ivf, source, cl_len, cl_len, fal, lmax_qlm=lmax_qlm)
        for r, tag in zip(R, ['GG', 'CC', 'GC', 'CG']):
            out['R_%s_%s_%s' % (key, source, tag)] = r
    ivf = {k: fal[k].copy() for k in fal}
    ivf['te'] = cl_len['te'][:lmax_ivf + 1] * fal['tt'] * fal['ee']
    for k in ivf:
        out['ivf_' + k] = ivf[k]
    for k1, k2 in [('ptt', 'ptt'), ('p_p', 'p_p'), ('p', 'p'), ('p_p', 'ptt'), ('x', 'x'), ('p', 'x')]:
        N = nhl.get_nhl(k1, k2, cl_len, ivf, lmax_ivf, lmax_ivf, lmax_out=lmax_qlm)
        for n, tag in zip(N, ['GG', 'CC', 'GC', 'CG']):
            out['N_%s_%s_%s' % (k1, k2, tag)] = n
    for n in (7, 64, 301):
        out['xg_%d' % n], out['wg_%d' % n] = wig.get_xgwg(-1., 1., n)
    rng = np.random.default_rng(11)
    x = np.sort(rng.uniform(-1, 1, 40))
    cl, xi = rng.standard_normal(31), rng.standard_normal(40)
    pos, coeff = [], []
    for s1 in range(-3, 4):
        for s2 in range(-3, 4):
            pos.append(wig.wignerpos(cl, x, s1, s2))
            coeff.append(wig.wignercoeff(xi, x, s1, s2, 30))
    out.update({'w_x': x, 'w_cl': cl, 'w_xi': xi, 'w_pos': np.array(pos), 'w_coeff': np.array(coeff)})
    np.savez_compressed(os.path.join(HERE, 'resp_golden.npz'), **out)
    print('wrote resp_golden.npz with %d arrays' % len(out))


def make_lib_golden():
    """Library classes the first fixture set did not reach, on the same tiny configuration (nside 16, lmax_ivf 40, lmax_qlm 47):
    qest.library_jtTP over a jointly filtered library (lib_filt2map, qest.py:441-530), the 'ntt' estimator, one bias-hardened
    key (qest.py:155-201, with the reference's resp_lib_simple on its own Fortran Wigner module), filt_simple.library_apo_sepTP
    (filt_simple.py:473-535) and filt_util.library_ftl (filt_util.py:39-103)."""
    install_fortran_wigners()
    from plancklens import qest, qresp, utils
    from plancklens.filt import filt_simple, filt_util
    import healpy as hp  # the stand-in installed by install_healpy_standin()

    nside, lmax_ivf, lmax_qlm, lmin_ivf = 16, 40, 47, 4
    cls_path = os.path.join(ROOT, 'plancklens_amd', 'data', 'cls', 'FFP10_wdipole_lensedCls.dat')
    cl_len = utils.camb_clfile(cls_path, lmax=lmax_ivf)
    nlev_t, nlev_p = 1200., 35.
    transf = oh.gauss_beam(4. / 180. * np.pi, lmax=lmax_ivf)
    sims = tiny_sims(nside, lmax_ivf, cl_len, transf, nlev_t, nlev_p)
    arcmin = np.pi / 180. / 60.
    ftl = utils.cli(cl_len['tt'][:lmax_ivf + 1] + (nlev_t * arcmin) ** 2 * utils.cli(transf ** 2))
    fel = utils.cli(cl_len['ee'][:lmax_ivf + 1] + (nlev_p * arcmin) ** 2 * utils.cli(transf ** 2))
    fbl = utils.cli(cl_len['bb'][:lmax_ivf + 1] + (nlev_p * arcmin) ** 2 * utils.cli(transf ** 2))
    ftl[:lmin_ivf] = 0; fel[:lmin_ivf] = 0; fbl[:lmin_ivf] = 0
    out = {'nside': nside, 'lmax_ivf': lmax_ivf, 'lmax_qlm': lmax_qlm, 'nlev_t': nlev_t, 'nlev_p': nlev_p, 'transf': transf,
           'ftl': ftl, 'fel': fel, 'fbl': fbl}
    for k in ['tt', 'ee', 'bb', 'te']:
        out['cl_' + k] = cl_len[k]
    tmp = tempfile.mkdtemp(prefix='plgolden_lib_')
    try:
        # ---- joint T-P filtering: an isotropic 3 x 3 filter with a TE block, as a subclass of the reference's template
        fal = {'tt': ftl.copy(), 'ee': fel.copy(), 'bb': fbl.copy(), 'te': -0.3 * np.sqrt(ftl * fel)}

        class iso_jTP(filt_simple.library_jTP):
            def hashdict(self):
                return {'sims': self.sim_lib.hashdict(), 'fal': {k: utils.clhash(v) for k, v in fal.items()}}

            def get_fmask(self):
                return np.ones(12 * nside ** 2)

            def get_fal(self):
                return {k: v.copy() for k, v in fal.items()}

            def _apply_ivf(self, tqumap, soltn=None):
                bi = utils.cli(transf)
                t = hp.almxfl(hp.map2alm(tqumap[0], lmax=lmax_ivf, iter=0), bi)
                e, b = hp.map2alm_spin([tqumap[1], tqumap[2]], 2, lmax=lmax_ivf)
                e, b = hp.almxfl(e, bi), hp.almxfl(b, bi)
                return (hp.almxfl(t, fal['tt']) + hp.almxfl(e, fal['te']), hp.almxfl(t, fal['te']) + hp.almxfl(e, fal['ee']),
                        hp.almxfl(b, fal['bb']))
        for k in fal:
            out['jt_fal_' + k] = fal[k]
        ivfs_j = iso_jTP(os.path.join(tmp, 'ivfs_j'), sims, {k: cl_len[k] for k in ['tt', 'ee', 'bb', 'te']}, cache=True)
        out['jt_tlm_0'], out['jt_elm_0'], out['jt_blm_0'] = ivfs_j.get_sim_tlm(0), ivfs_j.get_sim_elm(0), ivfs_j.get_sim_blm(0)
        out['jt_tmliklm_0'], out['jt_emliklm_0'] = ivfs_j.get_sim_tmliklm(0), ivfs_j.get_sim_emliklm(0)
        qlms_j = qest.library_jtTP(os.path.join(tmp, 'qlms_j'), ivfs_j, ivfs_j, nside, lmax_qlm=lmax_qlm)
        for k in ['p', 'x', 'ptt', 'p_p', 'stt']:  # ('f' raises inside the reference for jointly filtered libraries)
            out['jt_%s_0' % k] = qlms_j.get_sim_qlm(k, 0)
        # ---- sepTP: noise-inhomogeneity estimator and a bias-hardened key
        ivfs = filt_simple.library_fullsky_sepTP(os.path.join(tmp, 'ivfs'), sims, nside, transf, cl_len, ftl, fel, fbl, cache=False)
        resp = qresp.resp_lib_simple(os.path.join(tmp, 'resp'), lmax_ivf, cl_len, cl_len, {'t': ftl, 'e': fel, 'b': fbl}, lmax_qlm)
        qlms = qest.library_sepTP(os.path.join(tmp, 'qlms'), ivfs, ivfs, cl_len['te'], nside, lmax_qlm=lmax_qlm, resplib=resp)
        out['dd_ntt_0'] = qlms.get_sim_qlm('ntt', 0)
        out['dd_ptt_bh_s_0'] = qlms.get_sim_qlm('ptt_bh_s', 0)
        out['dd_mf_ptt_bh_s'] = qlms.get_sim_qlm_mf('ptt_bh_s', np.array([0, 1]))
        # ---- apodised-mask isotropic filtering
        th, ph = myhp.pix2ang(nside)
        apo = np.clip((np.abs(np.cos(th)) - 0.15) / 0.3, 0., 1.) ** 2 * (1. + 0.1 * np.cos(ph))
        apo_path = os.path.join(tmp, 'apomask.fits')
        hp.write_map(apo_path, apo)
        out['apomask'] = apo
        ivfs_a = filt_simple.library_apo_sepTP(os.path.join(tmp, 'ivfs_apo'), sims, apo_path, cl_len, transf, ftl, fel, fbl, cache=False)
        out['apo_tlm_1'], out['apo_elm_1'], out['apo_blm_1'] = ivfs_a.get_sim_tlm(1), ivfs_a.get_sim_elm(1), ivfs_a.get_sim_blm(1)
        out['apo_tmliklm_1'] = ivfs_a.get_sim_tmliklm(1)
        # ---- a-posteriori rescaling of a filtering library, to a smaller band-limit
        lmax_f = 33
        lt, le, lb = 1. / (1. + np.arange(lmax_f + 3.)), np.cos(0.1 * np.arange(lmax_f + 3.)), np.ones(lmax_f + 3) * 0.7
        out['ftl_lmax'], out['ftl_lt'], out['ftl_le'], out['ftl_lb'] = lmax_f, lt, le, lb
        ivfs_f = filt_util.library_ftl(ivfs, lmax_f, lt, le, lb)
        out['ftl_tlm_0'], out['ftl_elm_0'], out['ftl_blm_0'] = ivfs_f.get_sim_tlm(0), ivfs_f.get_sim_elm(0), ivfs_f.get_sim_blm(0)
        out['ftl_emliklm_0'] = ivfs_f.get_sim_emliklm(0)
        out['ftl_get_fel'] = ivfs_f.get_fel()
        qlms_f = qest.library_sepTP(os.path.join(tmp, 'qlms_f'), ivfs_f, ivfs_f, cl_len['te'], nside, lmax_qlm=lmax_qlm)
        out['ftl_p_0'] = qlms_f.get_sim_qlm('p', 0)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    np.savez_compressed(os.path.join(HERE, 'lib_golden.npz'), **out)
    print('wrote lib_golden.npz with %d arrays' % len(out))


def make_sims_golden():
    """Simulation inputs and the small wrapper classes, from the reference's own classes:
    phas.lib_phas / pix_lib_phas (numpy generator states in sqlite: the rngdb.db files themselves are stored, so that the test can
    hand an *existing* library to the new code), cmbs.sims_cmb_unl / sims_cmb_unl_fixed_phi, maps.cmb_maps_nlev /
    cmb_maps_noisefree / cmb_maps_harmonicspace, filt_util.library_fml, and the cross-filtered estimator keys of qest
    (_build_sim_xfiltMVgclm, qest.py:372-402)."""
    from plancklens import qest, utils
    from plancklens.sims import phas, cmbs, maps
    from plancklens.filt import filt_simple, filt_util
    nside, lmax = 8, 20
    npix = 12 * nside ** 2
    out = {'nside': nside, 'lmax': lmax}
    tmp = tempfile.mkdtemp(prefix='plgolden_sims_')
    try:
        np.random.seed(4242)
        lp = phas.lib_phas(os.path.join(tmp, 'pha'), 4, lmax)
        pp = phas.pix_lib_phas(os.path.join(tmp, 'pix'), 3, (npix,))
        for idx in (0, 1):   # request order matters: every first request records the global generator's state
            out['pha_%d' % idx] = lp.get_sim(idx)
            out['pix_%d' % idx] = pp.get_sim(idx)
        assert np.array_equal(lp.get_sim(0, idf=2), out['pha_0'][2])
        ell = np.arange(lmax + 1.)
        cls = {'tt': 1e3 / (ell + 5.) ** 2, 'ee': 30. / (ell + 5.) ** 2, 'bb': 2. / (ell + 5.) ** 2, 'pp': 1e-2 / (ell + 5.) ** 4}
        cls['te'] = 0.5 * np.sqrt(cls['tt'] * cls['ee'])
        cls['tp'] = 0.2 * np.sqrt(cls['tt'] * cls['pp'])
        for k, v in cls.items():
            out['cls_' + k] = v
        sky = cmbs.sims_cmb_unl(cls, lp)
        out['sky_fields'] = np.array(sky.fields)
        for f in 'pteb':
            out['sky_%slm_1' % f] = sky.get_sim_alm(1, f)
        fixed = cmbs.sims_cmb_unl_fixed_phi(cls, lp)
        out['fixed_plm_1'], out['fixed_tlm_1'] = fixed.get_sim_plm(1), fixed.get_sim_tlm(1)
        transf = oh.gauss_beam(8. / 180. * np.pi, lmax=lmax)
        out['transf'] = transf
        nl = maps.cmb_maps_nlev(sky, transf, 50., 70., nside, pix_lib_phas=pp)
        out['nlev_tmap_0'] = nl.get_sim_tmap(0)
        out['nlev_qmap_0'], out['nlev_umap_0'] = nl.get_sim_pmap(0)
        nf = maps.cmb_maps_noisefree(sky, transf, nside=nside, cl_transf_P=transf ** 2)
        out['nf_tmap_1'] = nf.get_sim_tmap(1)
        out['nf_qmap_1'], out['nf_umap_1'] = nf.get_sim_pmap(1)
        cls_noise = {'t': 1. / (1. + ell), 'e': 0.5 * np.ones(lmax + 1), 'b': 0.25 * np.ones(lmax + 1)}
        cls_transf = {'t': transf, 'e': transf ** 2, 'b': transf ** 3}
        for k in 'teb':
            out['hs_noise_' + k], out['hs_transf_' + k] = cls_noise[k], cls_transf[k]
        hs = maps.cmb_maps_harmonicspace(sky, cls_transf, cls_noise, lp)
        out['hs_tlm_0'] = hs.get_sim_tmap(0)
        out['hs_elm_0'], out['hs_blm_0'] = hs.get_sim_pmap(0)
        hsm = maps.cmb_maps_harmonicspace(sky, cls_transf, cls_noise, lp, nside=nside)
        out['hs_tmap_0'] = hsm.get_sim_tmap(0)
        out['hs_qmap_0'], out['hs_umap_0'] = hsm.get_sim_pmap(0)
        # the library files as data: an existing $PLENS tree the new code must read
        for name, lib, nf_ in (('pha', 'pha', 4), ('pix', 'pix_pha', 3)):
            for i in range(nf_):
                d = os.path.join(tmp, name, '%s_%04d' % (lib, i))
                lp_, pp_ = None, None
                out['file_%s_%d_rngdb' % (name, i)] = np.frombuffer(open(os.path.join(d, 'rngdb.db'), 'rb').read(), dtype=np.uint8)
                out['file_%s_%d_hash' % (name, i)] = np.frombuffer(open(os.path.join(d, 'sim_hash.pk'), 'rb').read(), dtype=np.uint8)

        # ---- library_fml over a stub filtering library of seeded random alms
        lmax_i, lmax_f = 24, 17

        class stub_ivfs(object):
            lib_dir = tmp

            def hashdict(self):
                return {'stub': 1}

            def _a(self, idx, k):
                rng = np.random.default_rng(100 * idx + k)
                a = rng.standard_normal(so.alm_size(lmax_i)) + 1j * rng.standard_normal(so.alm_size(lmax_i))
                a[:lmax_i + 1] = a[:lmax_i + 1].real
                return a

            def get_fmask(self):
                return np.ones(48)

            def get_tal(self, a):
                return np.ones(lmax_i + 1)

            def get_ftl(self):
                return 1. / (1. + np.arange(lmax_i + 1.))

            def get_fel(self):
                return 2. / (2. + np.arange(lmax_i + 1.))

            def get_fbl(self):
                return 3. / (3. + np.arange(lmax_i + 1.))
        for k, name in enumerate(['tlm', 'elm', 'blm', 'tmliklm', 'emliklm', 'bmliklm']):
            setattr(stub_ivfs, 'get_sim_' + name, (lambda self, idx, k=k: self._a(idx, k)))
        mt, me, mb = 1. / (1. + 0.1 * np.arange(lmax_i + 1.)), np.cos(0.2 * np.arange(lmax_i + 1.)) ** 2, 0.5 + 0.5 * (np.arange(lmax_i + 1) % 2)
        out['fml_lmax_in'], out['fml_lmax'], out['fml_mt'], out['fml_me'], out['fml_mb'] = lmax_i, lmax_f, mt, me, mb
        fml = filt_util.library_fml(stub_ivfs(), lmax_f, mt, me, mb)
        for name in ['tlm', 'elm', 'blm', 'tmliklm', 'emliklm', 'bmliklm']:
            out['fml_%s_3' % name] = getattr(fml, 'get_sim_' + name)(3)
        out['fml_ftl'], out['fml_fel'], out['fml_fbl'] = fml.get_ftl(), fml.get_fel(), fml.get_fbl()

        # ---- cross-filtered estimator keys on the tiny estimator configuration of main()
        nside_q, lmax_ivf, lmax_qlm, lmin_ivf = 16, 40, 47, 4
        cls_path = os.path.join(ROOT, 'plancklens_amd', 'data', 'cls', 'FFP10_wdipole_lensedCls.dat')
        cl_len = utils.camb_clfile(cls_path, lmax=lmax_ivf)
        nlev_t, nlev_p = 1200., 35.
        transf_q = oh.gauss_beam(4. / 180. * np.pi, lmax=lmax_ivf)
        sims = tiny_sims(nside_q, lmax_ivf, cl_len, transf_q, nlev_t, nlev_p)
        arcmin = np.pi / 180. / 60.
        ftl = utils.cli(cl_len['tt'][:lmax_ivf + 1] + (nlev_t * arcmin) ** 2 * utils.cli(transf_q ** 2))
        fel = utils.cli(cl_len['ee'][:lmax_ivf + 1] + (nlev_p * arcmin) ** 2 * utils.cli(transf_q ** 2))
        fbl = utils.cli(cl_len['bb'][:lmax_ivf + 1] + (nlev_p * arcmin) ** 2 * utils.cli(transf_q ** 2))
        ftl[:lmin_ivf] = 0; fel[:lmin_ivf] = 0; fbl[:lmin_ivf] = 0
        out.update({'q_nside': nside_q, 'q_lmax_ivf': lmax_ivf, 'q_lmax_qlm': lmax_qlm, 'q_nlev_t': nlev_t, 'q_nlev_p': nlev_p,
                    'q_transf': transf_q, 'q_ftl': ftl, 'q_fel': fel, 'q_fbl': fbl})
        for k in ['tt', 'ee', 'bb', 'te']:
            out['q_cl_' + k] = cl_len[k]
        ivfs = filt_simple.library_fullsky_sepTP(os.path.join(tmp, 'ivfs'), sims, nside_q, transf_q, cl_len, ftl, fel, fbl, cache=False)
        ivfs_s = filt_util.library_shuffle(ivfs, {0: 1, 1: 0})
        qdd = qest.library_sepTP(os.path.join(tmp, 'qdd'), ivfs, ivfs, cl_len['te'], nside_q, lmax_qlm=lmax_qlm)
        qds = qest.library_sepTP(os.path.join(tmp, 'qds'), ivfs, ivfs_s, cl_len['te'], nside_q, lmax_qlm=lmax_qlm)
        for k in ['pte', 'pet', 'pee', 'peb', 'pbe', 'ptb', 'xeb', 'xte']:
            out['xf_dd_%s_0' % k] = qdd.get_sim_qlm(k, 0)
        for k in ['pte', 'peb', 'xbe']:
            out['xf_ds_%s_0' % k] = qds.get_sim_qlm(k, 0)
        out['xf_dd_p_eb_0'] = qdd.get_sim_qlm('p_eb', 0)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    np.savez_compressed(os.path.join(HERE, 'sims_golden.npz'), **out)
    print('wrote sims_golden.npz with %d arrays' % len(out))


def make_cg2_golden():
    """Operator fixtures of the noise models the first CG set did not reach: opfilt_pp.alm_filter_ninv with three maps (QQ, QU, UU)
    and with marginalised Q / U template maps (apply_map, opfilt_pp.py:272-303), opfilt_tp.alm_filter_ninv with four maps
    (TT, QQ, QU, UU; opfilt_tp.py:306-327), each through fwd_op and calc_prep."""
    from plancklens.qcinv import opfilt_pp, opfilt_tp, util_alm
    nside, lmax = 16, 32
    npix = 12 * nside ** 2
    rng = np.random.default_rng(78)
    ell = np.arange(lmax + 1.)
    cl = {'tt': np.where(ell >= 2, 3e3 / np.maximum(ell, 1.) ** 2.2, 0.), 'ee': np.where(ell >= 2, 60. / np.maximum(ell, 1.) ** 1.8, 0.),
          'bb': np.where(ell >= 2, 2. / np.maximum(ell, 1.) ** 1.5, 0.)}
    cl['te'] = 0.6 * np.sqrt(cl['tt'] * cl['ee'])
    transf = oh.gauss_beam(6. / 180. * np.pi, lmax=lmax)
    th, ph = myhp.pix2ang(nside)
    mask = (np.abs(np.cos(th)) > 0.25).astype(float)
    sc = (npix / (4 * np.pi)) * 1e-4
    nqq = mask * (0.6 + 0.3 * np.cos(2 * ph)) / 10. ** 2 * sc
    nuu = mask * (0.7 + 0.2 * np.sin(ph)) / 10. ** 2 * sc
    nqu = mask * 0.15 * np.sin(2 * ph) * np.sin(th) / 10. ** 2 * sc
    ntt = mask * (0.5 + 0.4 * np.sin(3 * ph) * np.sin(th)) / 40. ** 2 * sc
    qmap, umap, tmap = rng.standard_normal(npix) * 10., rng.standard_normal(npix) * 10., rng.standard_normal(npix) * 40.
    tq = [np.cos(th) * mask, np.sin(th) * np.cos(ph) * mask]   # two Q templates
    tu = [np.sin(2 * th) * np.sin(ph) * mask]                   # one U template
    out = {'nside': nside, 'lmax': lmax, 'transf': transf, 'nqq': nqq, 'nqu': nqu, 'nuu': nuu, 'ntt': ntt, 'qmap': qmap, 'umap': umap,
           'tmap': tmap, 'tq0': tq[0], 'tq1': tq[1], 'tu0': tu[0]}
    for k in cl:
        out['cl_' + k] = cl[k]
    x = util_alm.eblm([oh.synalm(cl['ee'], lmax, rng), oh.synalm(cl['bb'], lmax, rng)])
    out['xe'], out['xb'] = x.elm, x.blm
    # (QQ, QU, UU)
    f3 = opfilt_pp.alm_filter_ninv([nqq, nqu, nuu], transf)
    r = opfilt_pp.fwd_op(cl, f3)(x)
    out['pp3_fwd_e'], out['pp3_fwd_b'] = r.elm, r.blm
    pr = opfilt_pp.calc_prep([qmap, umap], cl, f3)
    out['pp3_prep_e'], out['pp3_prep_b'] = pr.elm, pr.blm
    # one map + marginalised templates
    fm = opfilt_pp.alm_filter_ninv([nqq], transf, marge_qmaps=tq, marge_umaps=tu)
    r = opfilt_pp.fwd_op(cl, fm)(x)
    out['ppm_fwd_e'], out['ppm_fwd_b'] = r.elm, r.blm
    pr = opfilt_pp.calc_prep([qmap, umap], cl, fm)
    out['ppm_prep_e'], out['ppm_prep_b'] = pr.elm, pr.blm
    # joint filter, four maps
    f4 = opfilt_tp.alm_filter_ninv([ntt, nqq, nqu, nuu], transf, marge_monopole=True, marge_dipole=True)
    xt = util_alm.teblm([oh.synalm(cl['tt'], lmax, rng), oh.synalm(cl['ee'], lmax, rng), oh.synalm(cl['bb'], lmax, rng)])
    out['tp4_xt'], out['tp4_xe'], out['tp4_xb'] = xt.tlm, xt.elm, xt.blm
    r = opfilt_tp.fwd_op(cl, f4)(xt)
    out['tp4_fwd_t'], out['tp4_fwd_e'], out['tp4_fwd_b'] = r.tlm, r.elm, r.blm
    pr = opfilt_tp.calc_prep([tmap, qmap, umap], cl, f4)
    out['tp4_prep_t'], out['tp4_prep_e'], out['tp4_prep_b'] = pr.tlm, pr.elm, pr.blm
    np.savez_compressed(os.path.join(HERE, 'cg2_golden.npz'), **out)
    print('wrote cg2_golden.npz with %d arrays' % len(out))


def make_cinv_golden():
    """The reference's own filt_cinv.cinv_t and cinv_p (filt_cinv.py:56-338) at the smallest size their constructors accept
    (nside 512, lmax 1024; :77, :225), default 4-stage / 3-stage chains with the dense(64) / dense(32) levels and the D_l rescaling,
    eps = 1e-5 stopping rule, on a masked sky with inhomogeneous noise, monopole + dipole marginalised, over the oracle SHTs.
    Inputs are the recipe tests/helpers.py::cinv_golden_inputs (checksums stored); outputs: side files, the top-level residual
    trace, C_l of the solutions, every entry with l <= 64 and a seeded 10 000-entry subset of the rest."""
    import time
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import cinv_golden_inputs
    from plancklens.filt import filt_cinv
    d = cinv_golden_inputs(so.alm2map, so.alm2map_spin)
    nside, lmax, cl, transf = d['nside'], d['lmax'], d['cl'], d['transf']
    out = {'nside': nside, 'lmax': lmax, 'transf': transf}
    for k in ['ninv_t', 'ninv_p', 'tmap', 'qmap', 'umap']:
        out['chk_' + k] = np.array([d[k].sum(), (d[k] ** 2).sum(), d[k][::9973].sum()])
    for k in cl:
        out['cl_' + k] = cl[k]
    l_of = np.concatenate([np.arange(m, lmax + 1) for m in range(lmax + 1)])
    sub = np.sort(np.random.default_rng(99).choice(l_of.size, 10000, replace=False))
    out['subset'] = sub
    out['low'] = np.nonzero(l_of <= 64)[0]
    tmp = tempfile.mkdtemp(prefix='plgolden_cinv_')
    try:
        for kind in ('t', 'p'):
            trace = []
            t0 = time.time()
            if kind == 't':
                filt = filt_cinv.cinv_t(os.path.join(tmp, 'cinv_t'), lmax, nside, cl, transf, [d['ninv_t']])
            else:
                filt = filt_cinv.cinv_p(os.path.join(tmp, 'cinv_p'), lmax, nside, cl, transf, [[d['ninv_p']]])
            log0 = filt.chain.log
            filt.chain.log = lambda stage, it, eps, **kw: (trace.append((stage.depth, it, eps)), log0(stage, it, eps, **kw))
            if kind == 't':
                sol = [filt.apply_ivf(d['tmap'])]
                out['ftl'] = np.loadtxt(os.path.join(tmp, 'cinv_t', 'ftl.dat'))
                out['tal_t'] = np.loadtxt(os.path.join(tmp, 'cinv_t', 'tal.dat'))
                out['fmask_t_sum'] = myhp.read_map(os.path.join(tmp, 'cinv_t', 'fmask.fits.gz')).sum()
                names = ['tlm']
            else:
                sol = list(filt.apply_ivf([d['qmap'], d['umap']]))
                out['fel'] = np.loadtxt(os.path.join(tmp, 'cinv_p', 'fel.dat'))
                out['fbl'] = np.loadtxt(os.path.join(tmp, 'cinv_p', 'fbl.dat'))
                names = ['elm', 'blm']
            out['trace_' + kind] = np.array([t[2] for t in trace if t[0] == 0])
            for nm, a in zip(names, sol):
                out[nm + '_cl'] = oh.alm2cl(a)
                out[nm + '_sub'] = a[sub]
                out[nm + '_low'] = a[out['low']]
            print('cinv_%s: %d top-level iterations in %.0f s, last eps %.3e' % (kind, len(out['trace_' + kind]), time.time() - t0,
                                                                               out['trace_' + kind][-1]), flush=True)
            np.savez_compressed(os.path.join(HERE, 'cinv_golden.npz'), **out)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print('wrote cinv_golden.npz with %d arrays' % len(out))


def make_cinv2048_golden():
    """BASELINE config 4 at its own size (nside = lmax = 2048): the reference's own filt_cinv.cinv_t and cinv_p (filt_cinv.py:56-338) with the
    default chains (:112-116, :236-239), the benchmark's mask / noise model / data maps (tools/cg_bench.py::inputs, here on the oracle's
    transforms), the top level cut to 3 iterations (iter_max = 3, eps_min = 0: two to three fine operators and the full nested coarse solves
    per iteration), solver cd_solve.py:35-107.  Stored: checksums of the inputs, the top-level residual trace, C_l and <x, x> of each
    solution, 4096 seeded entries + every entry with l <= 8."""
    import time
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import cg_bench
    from plancklens.filt import filt_cinv
    from plancklens.qcinv import cd_solve as ref_cd_solve
    nside = lmax = 2048
    niter = 3
    t0 = time.time()
    d = cg_bench.inputs(nside, lmax, lambda a, ns: so.alm2map(a, ns, lmax=lmax), so.alm2map_spin)
    print('inputs: %.0f s' % (time.time() - t0), flush=True)
    out = {'nside': nside, 'lmax': lmax, 'niter': niter}
    for k in ['mask', 'tmap', 'qmap', 'umap']:
        out['chk_' + k] = np.array([d[k].sum(), (d[k] ** 2).sum(), d[k][::9973].sum()])
    l_of = np.concatenate([np.arange(m, lmax + 1) for m in range(lmax + 1)])
    sub = np.sort(np.random.default_rng(2048).choice(l_of.size, 4096, replace=False))
    out['subset'] = sub
    out['low'] = np.nonzero(l_of <= 8)[0]
    w = np.full(l_of.size, 2.)
    w[:lmax + 1] = 1.
    tmp = tempfile.mkdtemp(prefix='plgolden_cinv2048_')
    try:
        for kind in ('t', 'p'):
            trace = []
            t0 = time.time()
            pcf = os.path.join(tmp, 'dense_%s.pk' % kind)
            descr = cg_bench.chain(kind, niter, lmax, nside, pcf, cd_solve=ref_cd_solve)
            if kind == 't':
                filt = filt_cinv.cinv_t(os.path.join(tmp, 'cinv_t'), lmax, nside, d['cl'], d['transf'], d['ninv_t'], chain_descr=descr)
            else:
                filt = filt_cinv.cinv_p(os.path.join(tmp, 'cinv_p'), lmax, nside, d['cl'], d['transf'], d['ninv_p'], chain_descr=descr)
            log0 = filt.chain.log
            filt.chain.log = lambda stage, it, eps, **kw: (trace.append((stage.depth, it, eps)), log0(stage, it, eps, **kw))
            if kind == 't':
                sol, names = [filt.apply_ivf(d['tmap'])], ['tlm']
            else:
                sol, names = list(filt.apply_ivf([d['qmap'], d['umap']])), ['elm', 'blm']
            out['trace_' + kind] = np.array([t[2] for t in trace if t[0] == 0])
            for nm, a in zip(names, sol):
                out[nm + '_cl'] = oh.alm2cl(a)
                out[nm + '_sub'] = a[sub]
                out[nm + '_low'] = a[out['low']]
                out[nm + '_xx'] = np.array(float(np.sum(w * (a.real ** 2 + a.imag ** 2))))
            print('cinv_%s at 2048: %d top-level iterations in %.0f s, eps %s' % (kind, len(out['trace_' + kind]), time.time() - t0,
                                                                                 out['trace_' + kind]), flush=True)
            np.savez_compressed(os.path.join(HERE, 'cinv2048_golden.npz'), **out)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print('wrote cinv2048_golden.npz with %d arrays' % len(out))


def make_mfresp_golden():
    """Reference qresp.get_mf_resp (qresp.py:421-500; its Python on its own Fortran Wigner module) for 'ptt' and 'p_p' on the tiny
    configuration of the response fixtures, with the three pieces of the gradient response (retterms)."""
    install_fortran_wigners()
    from plancklens import qresp, utils
    lmax_qe, lmax_out, lmin = 40, 47, 4
    cls_path = os.path.join(ROOT, 'plancklens_amd', 'data', 'cls', 'FFP10_wdipole_lensedCls.dat')
    cl_len = utils.camb_clfile(cls_path, lmax=lmax_qe + 20)   # the CMB spectra reach beyond the filtered range, as in an analysis
    transf = oh.gauss_beam(4. / 180. * np.pi, lmax=lmax_qe)
    arcmin = np.pi / 180. / 60.
    ivf = {'tt': utils.cli(cl_len['tt'][:lmax_qe + 1] + (1200. * arcmin) ** 2 * utils.cli(transf ** 2)),
           'ee': utils.cli(cl_len['ee'][:lmax_qe + 1] + (35. * arcmin) ** 2 * utils.cli(transf ** 2)),
           'bb': utils.cli(cl_len['bb'][:lmax_qe + 1] + (35. * arcmin) ** 2 * utils.cli(transf ** 2))}
    for f in ivf.values():
        f[:lmin] = 0
    out = {'lmax_qe': lmax_qe, 'lmax_out': lmax_out}
    for k in ['tt', 'ee', 'bb', 'te']:
        out['cl_' + k] = cl_len[k]
    for k in ivf:
        out['ivf_' + k] = ivf[k]
    for key in ['ptt', 'p_p']:
        GL, CL, terms = qresp.get_mf_resp(key, cl_len, ivf, lmax_qe, lmax_out, retterms=True)
        out['G_' + key], out['C_' + key] = GL, CL
        for t in terms:
            out['%s_%s' % (t, key)] = terms[t]
    np.savez_compressed(os.path.join(HERE, 'mfresp_golden.npz'), **out)
    print('wrote mfresp_golden.npz with %d arrays' % len(out))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'resp':   # only the response / N0 fixtures
        assert os.path.isdir(REF), 'the reference is only present in the build container'
        install_healpy_standin()
        sys.path.insert(0, REF)
        make_resp_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == 'lib':   # only the fixtures of the further library classes
        assert os.path.isdir(REF), 'the reference is only present in the build container'
        install_healpy_standin()
        sys.path.insert(0, REF)
        make_lib_golden()
    elif len(sys.argv) > 1 and sys.argv[1] in ('sims', 'cg2', 'cinv', 'cinv2048', 'mfresp'):   # simulation inputs / small wrapper classes; further noise models of the CG
        assert os.path.isdir(REF), 'the reference is only present in the build container'
        install_healpy_standin()
        sys.path.insert(0, REF)
        {'sims': make_sims_golden, 'cg2': make_cg2_golden, 'cinv': make_cinv_golden, 'cinv2048': make_cinv2048_golden, 'mfresp': make_mfresp_golden}[sys.argv[1]]()
    elif len(sys.argv) > 1 and sys.argv[1] == 'cg':   # only the CG fixtures
        assert os.path.isdir(REF), 'the reference is only present in the build container'
        install_healpy_standin()
        sys.path.insert(0, REF)
        make_cg_golden()
    else:
        main()
        # every other fixture set in a process of its own (the response fixtures install the Fortran Wigner stand-in BEFORE the reference's
        # modules are imported; 'cinv' -- the reference's own cinv_t / cinv_p solves at nside 512 -- takes ~20 minutes of CPU and is run by name)
        import subprocess
        for part in ('cg', 'resp', 'lib', 'sims', 'cg2', 'mfresp'):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), part])
