"""Temperature Wiener / inverse-variance filtering operators on the device, API of plancklens/qcinv/opfilt_tt.py
(`calc_prep` :30-36, `apply_fini` :39-41, `dot_op` :43-51, `fwd_op` :54-73, `pre_op_diag` :76-93, `alm_filter_ninv`
:99-205).  The CG solves  [S^-1 + B^t Y^t N^-1 Y B] x = B^t Y^t N^-1 d  and returns S^-1 x.

Y = alm2map and Y^t = (npix / 4 pi) map2alm run on the GPU (plancklens_amd.shts); the beam factors are fused into the
transforms; the inverse-noise map, the template projector and every CG vector live in HBM."""
from __future__ import print_function

import hashlib

import numpy as np
import torch

from .. import dev, hp, options, shts
from ..utils import clhash, enumerate_progress
from . import dense, template_removal, util

alm2map, map2alm = shts.alm2map, shts.map2alm  # exported so that they can be customised, as in the reference


def _cli(cl):
    ret = np.zeros_like(cl)
    ret[np.where(cl != 0.)] = 1. / cl[np.where(cl != 0.)]
    return ret


def calc_prep(m, s_cls, n_inv_filt):
    """b = B^t Y^t N^-1 d."""
    tmap = dev.to_dev(m, torch.float64).clone()
    n_inv_filt.apply_map(tmap)
    lmax = len(n_inv_filt.b_transf) - 1
    return map2alm(tmap, lmax=lmax, iter=0, fl=n_inv_filt.b_transf * (tmap.numel() / (4. * np.pi)))


def calc_prep_batch(maps, s_cls, n_inv_filt):
    """calc_prep of every map of the list as one block vector [nb, nalm] (several right-hand sides solved together)"""
    return torch.stack([calc_prep(m, s_cls, n_inv_filt) for m in maps]).contiguous()


def apply_fini(alm, s_cls, n_inv_filt):
    """Wiener-filtered solution -> inverse-variance filtered: x <- S^-1 x (in place)."""
    alm.copy_(dev.almxfl(alm, _cli(s_cls['tt'])))


class dot_op(object):
    """sum_l (2l + 1) C_l^{ab}: the scalar product of the CG (opfilt_tt.py:43-51)."""
    lmin = 0  # first multipole of the sum

    def parts(self, alm1, alm2):
        """the scalar product as dev.DOT_PARTS partial sums in device memory (one launch per field, no host synchronisation);
        its value is their sum, formed by whoever consumes it (axpy below, dev(), __call__)"""
        assert alm1.shape == alm2.shape
        return dev.alm_dot([(alm1, alm2)])

    def dev(self, alm1, alm2):
        """the scalar product as a 0-dim device tensor (block vectors [nb, nalm]: nb values)"""
        return self.parts(alm1, alm2).sum(-1)

    @staticmethod
    def axpy(y, x, num, den, sign):
        """y += sign num / den x in place (num, den: scalar products as returned by parts): the vector updates of cd_solve in one launch"""
        dev.axpy_dev(y, x, num, den, sign)

    @staticmethod
    def step(x, d, r, q, update_r=True, active=None, pre=None, x_init=False):
        """one conjugate-directions update, all fields in two launches (or one with a grid barrier): dTAd = <d, q>, delta = <d, r>,
        x += (delta / dTAd) d and, if update_r, r -= (delta / dTAd) q; returns (dTAd, delta) as `parts` does.
        active (block vectors): 0 / 1 per entry, multiplies the step lengths.
        pre: the two scalar products as left by the operator that made q (fwd_op.with_dots): the updates alone, one launch;
        x_init (with pre): x = (delta / dTAd) d -- x is written, not read (first step of a solve from zero)"""
        f = (lambda v: [v])
        if pre is not None:
            return dev.cg_axpy_pre(pre, f(x), f(d), 1.0, y2=f(r) if update_r else None, x2=f(q) if update_r else None, sign2=-1.0, active=active,
                                   assign_y1=x_init)
        assert not x_init
        return dev.cg_dot_axpy(f(d), f(q), f(x), f(d), 1.0, b2=f(r), y2=f(r) if update_r else None, x2=f(q) if update_r else None,
                               sign2=-1.0, lmin=0, active=active)

    @staticmethod
    def ortho(s, pq, pd, prev_dtad, pre=None):
        """s -= (<s, pq> / prev_dtad) pd, all fields in two launches (or one with a grid barrier).
        pre: <s, pq> as partial sums left by the preconditioner kernel that wrote s (pre_op.with_dot): the update alone"""
        f = (lambda v: [v])
        if pre is not None:
            dev.cg_axpy_pre((pre, None), f(s), f(pd), -1.0, den=prev_dtad)
            return
        dev.cg_dot_axpy(f(s), f(pq), f(s), f(pd), -1.0, den=prev_dtad, lmin=0)

    def __call__(self, alm1, alm2):
        p = self.parts(alm1, alm2)
        return float(p.sum()) if p.dim() == 1 else dev.to_host(p.sum(-1))  # block vectors: one value per entry


class fwd_op(object):
    """x -> S^-1 x + B^t Y^t N^-1 Y B x."""

    def __init__(self, s_cls, n_inv_filt):
        self.cltt_inv = _cli(s_cls['tt'])
        self.n_inv_filt = n_inv_filt

    def hashdict(self):
        return {'cltt_inv': clhash(self.cltt_inv), 'n_inv_filt': self.n_inv_filt.hashdict()}

    def __call__(self, talm):
        return self.calc(talm)

    def with_dots(self, talm, r):
        """(fwd_op(talm), pre): pre = the scalar products <talm, result> and <talm, r> as partial sums left by the kernel that writes the
        result (dot_op.step takes them), or None where the operator is not the one-call form"""
        f = util.unjit(self.n_inv_filt)
        if (isinstance(f, alm_filter_ninv) and f.one_call_ok(talm) and f.one_call_final(talm) and isinstance(r, torch.Tensor) and r.is_cuda
                and r.dtype == torch.complex128 and talm.dtype == torch.complex128 and r.shape == talm.shape and r.is_contiguous() and talm.is_contiguous()):
            req = shts.post_dots([r], lmin=0)
            q = f.apply_alm_new(talm, alm_add=talm, fl_add=self.cltt_inv, dots=req)
            return q, req.pre
        return self.calc(talm), None

    def calc(self, talm):
        f = util.unjit(self.n_inv_filt)  # (the filter libraries hand over a lazily built filter)
        if isinstance(f, alm_filter_ninv) and f.one_call_ok(talm):  # the whole operator in pl_cg_fwd_tt
            return f.apply_alm_new(talm, alm_add=talm, fl_add=self.cltt_inv)
        assert not (isinstance(talm, torch.Tensor) and talm.dim() == 2), 'block vectors take the one-call operator (pl_cg_fwd_tt_b)'
        alm = f.apply_alm_new(talm)
        return dev.almxfl_add(alm, talm, self.cltt_inv, out=alm)


class pre_op_diag(object):
    """Harmonic-space diagonal preconditioner 1 / (1 / C_l + b_l^2 sum(N^-1) / 4 pi)."""

    def __init__(self, s_cls, n_inv_filt):
        cltt = s_cls['tt']
        assert len(cltt) >= len(n_inv_filt.b_transf)
        n_inv_cl = float(n_inv_filt.n_inv.sum()) / (4.0 * np.pi)
        lmax = len(n_inv_filt.b_transf) - 1
        filt = _cli(cltt[:lmax + 1])
        filt += n_inv_cl * n_inv_filt.b_transf[:lmax + 1] ** 2
        self.filt = _cli(filt)

    def __call__(self, talm):
        return self.calc(talm)

    def calc(self, talm):
        return dev.almxfl(talm, self.filt)

    def splice_above(self, alm_low, talm, lsplit, dot=None):
        """alm_low for l <= lsplit, this preconditioner applied to talm above: pre_op_split's result in one launch (None: not here).
        dot = (q, lmin): returns (result, pre), pre the partial sums of <result, q> formed by the same launch"""
        if not (isinstance(talm, torch.Tensor) and talm.is_cuda and talm.dtype == torch.complex128 and alm_low.dtype == torch.complex128):
            return None
        if dot is not None:
            q, lmin = dot
            lmax_hi = hp.Alm.getlmax(talm.shape[-1])
            pre = torch.empty(tuple(talm.shape[:-1]) + (dev.alm_splice_dot_count(lmax_hi),), dtype=torch.float64, device=talm.device)
            return dev.alm_splice_fl(alm_low, talm, self.filt, lsplit, dot=(q, lmin, pre)), pre
        return dev.alm_splice_fl(alm_low, talm, self.filt, lsplit)


def pre_op_dense(lmax, fwd_op, cache_fname=None):
    return dense.pre_op_dense_tt(lmax, fwd_op, cache_fname=cache_fname)


class alm_filter_ninv(object):
    """N^-1 in pixel space (product of the listed maps), with optional marginalisation of monopole, dipole and
    arbitrary template maps: N^-1 <- N^-1 - N^-1 P (P^t N^-1 P)^-1 P^t N^-1."""

    def __init__(self, n_inv, b_transf, marge_monopole=False, marge_dipole=False, marge_uptolmin=-1, marge_maps=(), nlev_ftl=None):
        n_inv = util.load_map(n_inv)
        self.n_inv = dev.to_dev(n_inv, torch.float64)
        nz = self.n_inv[self.n_inv != 0.0]
        print("opfilt_tt: inverse noise map std dev / av = %.3e" % (float(nz.std(unbiased=False)) / float(nz.mean())))
        templates, templates_hash = [], []
        for tmap in [util.load_map(m) for m in marge_maps]:
            assert len(n_inv) == len(tmap)
            templates.append(template_removal.template_map(tmap))
            templates_hash.append(hashlib.sha1(np.ascontiguousarray(tmap).view(np.uint8)).hexdigest())
        assert marge_uptolmin < 0, 'marge_uptolmin not implemented'
        if marge_monopole:
            templates.append(template_removal.template_monopole())
        if marge_dipole:
            templates.append(template_removal.template_dipole())
        if len(templates) != 0:
            nmodes = int(np.sum([t.nmodes for t in templates]))
            modes_idx_t = np.concatenate([t.nmodes * [int(im)] for im, t in enumerate(templates)])
            modes_idx_i = np.concatenate([range(0, t.nmodes) for t in templates])
            Pt_Nn1_P = np.zeros((nmodes, nmodes))
            for ir in range(nmodes):
                tmap = self.n_inv.clone()
                templates[modes_idx_t[ir]].apply_mode(tmap, int(modes_idx_i[ir]))
                ic = 0
                for tc in templates[0:modes_idx_t[ir] + 1]:
                    Pt_Nn1_P[ir, ic:(ic + tc.nmodes)] = tc.dot(tmap)
                    Pt_Nn1_P[ic:(ic + tc.nmodes), ir] = Pt_Nn1_P[ir, ic:(ic + tc.nmodes)]
                    ic += tc.nmodes
            eigv, eigw = np.linalg.eigh(Pt_Nn1_P)
            self.Pt_Nn1_P_inv = np.dot(np.dot(eigw, np.diag(1.0 / eigv)), np.transpose(eigw))
        self.b_transf = b_transf
        self.npix = self.n_inv.numel()
        self.nside = hp.npix2nside(self.npix)
        self.marge_monopole = marge_monopole
        self.marge_dipole = marge_dipole
        self.marge_uptolmin = marge_uptolmin
        self.templates = templates
        self.templates_hash = templates_hash
        if nlev_ftl is None:
            nlev_ftl = 10800. / np.sqrt(float(self.n_inv.sum()) / (4.0 * np.pi)) / np.pi
        self.nlev_ftl = nlev_ftl
        print("ninv_ftl: using %.2f uK-amin noise Cl" % self.nlev_ftl)

    def hashdict(self):
        return {'n_inv': clhash(dev.to_host(self.n_inv)), 'b_transf': clhash(self.b_transf),
                'marge_monopole': self.marge_monopole, 'marge_dipole': self.marge_dipole,
                'templates_hash': self.templates_hash, 'marge_uptolmin': self.marge_uptolmin}

    def get_ftl(self):
        return self.b_transf ** 2 / (self.nlev_ftl / 60. / 180. * np.pi) ** 2

    def degrade(self, nside):
        """Coarser copy: hp.ud_grade(power=-2) sums the inverse variances of the children; template maps are dropped."""
        if nside == self.nside:
            return self
        print("DEGRADING WITH NO MARGE MAPS")
        return alm_filter_ninv(hp.ud_grade(dev.to_host(self.n_inv), nside, power=-2), self.b_transf,
                               marge_monopole=self.marge_monopole, marge_dipole=self.marge_dipole,
                               marge_uptolmin=self.marge_uptolmin, marge_maps=[])

    def apply_alm(self, alm):
        """alm <- B^t Y^t N^-1 Y B alm (in place)."""
        alm.copy_(self.apply_alm_new(alm))

    def one_call_ok(self, alm):
        """pl_cg_fwd_tt applies: device input, the module's transforms not replaced, few enough template modes."""
        nmodes = sum(t.nmodes for t in self.templates)
        return (isinstance(alm, torch.Tensor) and alm.is_cuda and alm2map is shts.alm2map and map2alm is shts.map2alm and not shts.lane_active()
                and nmodes <= dev.TEMPLATE_MAX_MODES and self.n_inv.is_contiguous() and self.n_inv.dtype == torch.float64)

    def supports_block(self):
        """True when block vectors [nb, nalm] can go through this filter (they take the one-call operator only: fwd_op.calc)"""
        return self.one_call_ok(torch.empty((1, 1), dtype=torch.complex128, device=self.n_inv.device))

    def one_call_final(self, alm):
        """True when pl_cg_fwd_tt's result is the operator's (no update applied to it afterwards): scalar products may ride in it"""
        if len(self.templates) != 0 and options.opts.tproj_harm:
            return sum(t.nmodes for t in self.templates) <= dev.TEMPLATE_MAX_MODES
        return True

    def apply_alm_new(self, alm, alm_add=None, fl_add=None, dots=None):
        """B^t Y^t N^-1 Y B alm (+ fl_add alm_add) as a new array (the input is left alone).
        dots (shts.post_dots, one-call forms only): scalar products of the result formed by the kernel that writes it"""
        lmax = hp.Alm.getlmax(alm.shape[-1] if isinstance(alm, torch.Tensor) else alm.size)
        fl_out = self.b_transf * (self.npix / (4. * np.pi))
        if self.one_call_ok(alm):
            if len(self.templates) != 0 and options.opts.tproj_harm:
                # the projection in harmonic space: the transforms carry the plain N^-1 weighting (inside the ring-FFT launches),
                # the templates are a rank-nmodes update of the result (pl_lowrank_update_b)
                hpm, hrm = self._harm_matrices(lmax)
                if hpm.shape[0] <= dev.TEMPLATE_MAX_MODES:  # one call: the coefficient pass of the update runs beside the transforms
                    return shts.cg_fwd_tt(alm, self.nside, lmax, self.n_inv, fl_in=self.b_transf, fl_out=fl_out, alm_add=alm_add, fl_add=fl_add,
                                          lowrank=(hpm, hrm), dots=dots)
                assert dots is None
                ret = shts.cg_fwd_tt(alm, self.nside, lmax, self.n_inv, fl_in=self.b_transf, fl_out=fl_out, alm_add=alm_add, fl_add=fl_add)
                return dev.lowrank_update(ret, alm.to(torch.complex128).contiguous(), hpm, hrm)
            if self._md_only() and not shts.plan_all_generic(self.nside, lmax):
                # monopole + dipole on a grid with register FFT classes: the templates come from the ring geometry, no stored maps
                return shts.cg_fwd_tt(alm, self.nside, lmax, self.n_inv, fl_in=self.b_transf, fl_out=fl_out, pinv_md=self._pinv_md(),
                                      alm_add=alm_add, fl_add=fl_add, dots=dots)
            pmat, rmat = self._proj_matrices()
            nb = alm.shape[0] if alm.dim() == 2 else 1
            return shts.cg_fwd_tt(alm, self.nside, lmax, self.n_inv, fl_in=self.b_transf, fl_out=fl_out, pmat=pmat, rmat=rmat,
                                  scratch=dev.tproj_scratch(nb) if pmat is not None else None, alm_add=alm_add, fl_add=fl_add, dots=dots)
        assert dots is None
        tmap = alm2map(alm, self.nside, lmax=lmax, fl=self.b_transf)
        self.apply_map(tmap)
        ret = map2alm(tmap, lmax=lmax, iter=0, fl=fl_out)
        return ret if alm_add is None else dev.almxfl_add(ret, alm_add, fl_add, out=ret)

    def _harm_matrices(self, lmax):
        """the template modes in harmonic space (template_removal.harmonic_matrices), once per band-limit"""
        cache = self.__dict__.setdefault('_harm', {})
        if lmax not in cache:
            cache[lmax] = template_removal.harmonic_matrices(self.templates, self.n_inv, map2alm, lmax, self.b_transf * (self.npix / (4. * np.pi)),
                                                             self.Pt_Nn1_P_inv)
        return cache[lmax]

    def _md_only(self):
        """the templates are exactly (monopole, dipole): (1, x, y, z) of the pixel centres, which the kernels can evaluate themselves"""
        if not options.opts.tproj_md or len(self.templates) != 2:
            return False
        return isinstance(self.templates[0], template_removal.template_monopole) and isinstance(self.templates[1], template_removal.template_dipole)

    def _pinv_md(self):
        if getattr(self, '_pinv_md_dev', None) is None:
            self._pinv_md_dev = dev.to_dev(np.ascontiguousarray(self.Pt_Nn1_P_inv), torch.float64).contiguous()
        return self._pinv_md_dev

    def _proj_matrices(self):
        """All template modes as one device matrix P (nmodes x npix) and R = (P^t N^-1 P)^-1 P^t N^-1: the projection is two
        mat-vecs, c = P^t (N^-1 t) and t -= R^t c; coefficients, the small solve and the projected map are device operations,
        nothing comes back to the host inside a CG iteration."""
        if len(self.templates) == 0:
            return None, None
        if getattr(self, '_pmat', None) is None:
            rows = []
            for t in self.templates:
                for i in range(t.nmodes):
                    row = torch.ones_like(self.n_inv)
                    t.apply_mode(row, i)
                    rows.append(row)
            self._pmat = torch.stack(rows).contiguous()
            pinv = dev.to_dev(np.ascontiguousarray(self.Pt_Nn1_P_inv), torch.float64)
            self._rmat = torch.mm(pinv, self._pmat * self.n_inv.unsqueeze(0)).contiguous()
        return self._pmat, self._rmat

    def apply_map(self, tmap):
        """tmap <- N^-1 tmap with the templates projected out (in place)."""
        if self._md_only() and tmap.is_contiguous() and tmap.is_cuda and tmap.dtype == torch.float64:
            dev.template_project_md(tmap, self.n_inv, self.nside, len(self.b_transf) - 1, self._pinv_md())
            return
        pmat, rmat = self._proj_matrices()
        if pmat is None:
            tmap *= self.n_inv
        elif pmat.shape[0] <= dev.TEMPLATE_MAX_MODES and tmap.is_contiguous():
            dev.template_project(tmap, self.n_inv, pmat, rmat)  # N^-1 weighting + projection, two launches
        else:
            tmap *= self.n_inv
            tmap.addmv_(rmat.t(), dev.gemv(pmat, tmap), alpha=-1.0)  # more modes than pl_template_project takes
