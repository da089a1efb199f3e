"""Joint temperature + polarization Wiener / inverse-variance filtering operators on the device, API of
plancklens/qcinv/opfilt_tp.py (`calc_prep` :14-31, `apply_fini` :34-40, `dot_op` :46-59, `fwd_op` :62-83, `pre_op_diag`
:88-118, `alm_filter_sinv` :126-160, `alm_filter_ninv` :163-327).  Vectors are util_alm.teblm triplets of device tensors.

Y = (alm2map, alm2map_spin 2) and Y^t = npix / 4 pi (map2alm, map2alm_spin 2) run on the GPU with the beams fused into
the transforms; the inverse-noise maps, the T template projector and every CG vector live in HBM."""
from __future__ import print_function


import numpy as np
import torch

from .. import dev, hp, options, shts
from ..utils import clhash
from . import dense, template_removal, util
from .util_alm import teblm

alm2map, map2alm, alm2map_spin, map2alm_spin = shts.alm2map, shts.map2alm, shts.alm2map_spin, shts.map2alm_spin


def calc_prep(maps, s_cls, n_inv_filt):
    """b = B^t Y^t N^-1 d for d = (T, Q, U)."""
    tmap, qmap, umap = (dev.to_dev(m, torch.float64).clone() for m in maps)
    assert tmap.numel() == qmap.numel() == umap.numel()
    npix = tmap.numel()
    n_inv_filt.apply_map([tmap, qmap, umap])
    lmax = len(n_inv_filt.b_transf) - 1
    tlm = map2alm(tmap, lmax=lmax, iter=0, fl=n_inv_filt.b_transf_t * (npix / (4. * np.pi)))
    elm, blm = map2alm_spin([qmap, umap], 2, lmax=lmax)
    elm = dev.almxfl(elm, n_inv_filt.b_transf_e * (npix / (4. * np.pi)))
    blm = dev.almxfl(blm, n_inv_filt.b_transf_b * (npix / (4. * np.pi)))
    return teblm([tlm, elm, blm])


def calc_prep_batch(maps, s_cls, n_inv_filt):
    """calc_prep of every (T, Q, U) triplet of the list as one block vector: teblm of [nb, nalm] tensors"""
    preps = [calc_prep(m, s_cls, n_inv_filt) for m in maps]
    return teblm([torch.stack([getattr(p, a) for p in preps]).contiguous() for a in ('tlm', 'elm', 'blm')])


def apply_fini(alm, s_cls, n_inv_filt):
    """Wiener-filtered solution -> inverse-variance filtered: x <- S^-1 x (in place)."""
    lmax = len(n_inv_filt.b_transf) - 1
    ret = alm_filter_sinv(s_cls, lmax).calc(alm)
    alm.tlm.copy_(ret.tlm)
    alm.elm.copy_(ret.elm)
    alm.blm.copy_(ret.blm)


def apply_finiMLIK(alm, s_cls, n_inv_filt):
    pass


class dot_op(object):
    """sum_l (2l + 1) (C_l^{TT'} + C_l^{EE'} + C_l^{BB'})."""

    def parts(self, alm1, alm2):
        """the scalar product as dev.DOT_PARTS partial sums in device memory (one launch per field, no host synchronisation);
        its value is their sum, formed by whoever consumes it (axpy below, dev(), __call__)"""
        assert alm1.lmaxt == alm2.lmaxt, (alm1.lmaxt, alm2.lmaxt)
        assert alm1.lmaxe == alm2.lmaxe, (alm1.lmaxe, alm2.lmaxe)
        assert alm1.lmaxb == alm2.lmaxb, (alm1.lmaxb, alm2.lmaxb)
        return dev.alm_dot([(alm1.tlm, alm2.tlm), (alm1.elm, alm2.elm), (alm1.blm, alm2.blm)])

    def dev(self, alm1, alm2):
        """the scalar product as a 0-dim device tensor (block vectors: one value per entry)"""
        return self.parts(alm1, alm2).sum(-1)

    @staticmethod
    def axpy(y, x, num, den, sign):
        """y += sign num / den x in place (num, den: scalar products as returned by parts)"""
        dev.axpy_dev(y.tlm, x.tlm, num, den, sign)
        dev.axpy_dev(y.elm, x.elm, num, den, sign)
        dev.axpy_dev(y.blm, x.blm, num, den, sign)

    @staticmethod
    def step(x, d, r, q, update_r=True, active=None):
        """one conjugate-directions update, all fields in two launches (or one with a grid barrier): dTAd = <d, q>, delta = <d, r>,
        x += (delta / dTAd) d and, if update_r, r -= (delta / dTAd) q; returns (dTAd, delta) as `parts` does.
        active (block vectors): 0 / 1 per entry, an entry with 0 stands still"""
        f = (lambda v: [v.tlm, v.elm, v.blm])
        return dev.cg_dot_axpy(f(d), f(q), f(x), f(d), 1.0, b2=f(r), y2=f(r) if update_r else None, x2=f(q) if update_r else None,
                               sign2=-1.0, lmin=0, active=active)

    @staticmethod
    def ortho(s, pq, pd, prev_dtad):
        """s -= (<s, pq> / prev_dtad) pd, all fields in two launches (or one with a grid barrier)"""
        f = (lambda v: [v.tlm, v.elm, v.blm])
        dev.cg_dot_axpy(f(s), f(pq), f(s), f(pd), -1.0, den=prev_dtad, lmin=0)

    def __call__(self, alm1, alm2):
        p = self.parts(alm1, alm2)
        return float(p.sum()) if p.dim() == 1 else dev.to_host(p.sum(-1))  # block vectors: one value per entry


class fwd_op(object):
    """x -> S^-1 x + B^t Y^t N^-1 Y B x."""

    def __init__(self, s_cls, n_inv_filt):
        lmax = len(n_inv_filt.b_transf) - 1
        self.s_inv_filt = alm_filter_sinv(s_cls, lmax)
        self.n_inv_filt = n_inv_filt

    def hashdict(self):
        return {'s_inv_filt': self.s_inv_filt.hashdict(), 'n_inv_filt': self.n_inv_filt.hashdict()}

    def __call__(self, alm):
        return self.calc(alm)

    def calc(self, alm):
        nlm = self.n_inv_filt.apply_alm_new(alm)
        return _apply_3x3(self.s_inv_filt.slinv, alm, self.s_inv_filt.te_only, add_to=nlm)


def _apply_3x3(tmat, alm, te_only, add_to=None):
    """(T, E, B) <- per-l 3x3 matrix applied to (T, E, B); B decouples when there is no TB / EB power (te_only).
    add_to: a teblm the result is accumulated into in place (and which is returned)."""
    if add_to is None:
        rtlm, relm, rblm = dev.almxfl(alm.tlm, tmat[:, 0, 0]), dev.almxfl(alm.tlm, tmat[:, 1, 0]), dev.almxfl(alm.blm, tmat[:, 2, 2])
    else:
        rtlm = dev.almxfl_add(add_to.tlm, alm.tlm, tmat[:, 0, 0], out=add_to.tlm)
        relm = dev.almxfl_add(add_to.elm, alm.tlm, tmat[:, 1, 0], out=add_to.elm)
        rblm = dev.almxfl_add(add_to.blm, alm.blm, tmat[:, 2, 2], out=add_to.blm)
    dev.almxfl_add(rtlm, alm.elm, tmat[:, 0, 1], out=rtlm)
    dev.almxfl_add(relm, alm.elm, tmat[:, 1, 1], out=relm)
    if not te_only:
        dev.almxfl_add(rtlm, alm.blm, tmat[:, 0, 2], out=rtlm)
        dev.almxfl_add(relm, alm.blm, tmat[:, 1, 2], out=relm)
        dev.almxfl_add(rblm, alm.tlm, tmat[:, 2, 0], out=rblm)
        dev.almxfl_add(rblm, alm.elm, tmat[:, 2, 1], out=rblm)
    return add_to if add_to is not None else teblm([rtlm, relm, rblm])


class pre_op_diag(object):
    """Harmonic-space block-diagonal preconditioner: per-l pseudo-inverse of S^-1 + diag(N^-1 b^2) (3 x 3 in T, E, B)."""

    def __init__(self, s_cls, n_inv_filt):
        lmax = len(n_inv_filt.b_transf) - 1
        s_inv_filt = alm_filter_sinv(s_cls, lmax)
        assert (s_inv_filt.lmax + 1) >= len(n_inv_filt.b_transf)
        ninv_ftl, ninv_fel, ninv_fbl = n_inv_filt.get_ftebl()
        flmat = s_inv_filt.slinv[0:lmax + 1, :, :].copy()
        flmat[:, 0, 0] += ninv_ftl
        flmat[:, 1, 1] += ninv_fel
        flmat[:, 2, 2] += ninv_fbl
        self.flmat = np.linalg.pinv(flmat)
        self.te_only = s_inv_filt.te_only

    def __call__(self, talm):
        return self.calc(talm)

    def calc(self, alm):
        return _apply_3x3(self.flmat, alm, self.te_only)


def pre_op_dense(lmax, fwd_op, cache_fname=None):
    return dense.pre_op_dense_tp(lmax, fwd_op, cache_fname=cache_fname)


class alm_filter_sinv(object):
    """S^-1: per-l pseudo-inverse of the 3 x 3 matrix of TEB spectra."""

    def __init__(self, s_cls, lmax):
        slmat = np.zeros((lmax + 1, 3, 3))
        z = np.zeros(lmax + 1)
        slmat[:, 0, 0] = s_cls.get('tt', z)[:lmax + 1]
        slmat[:, 0, 1] = s_cls.get('te', z)[:lmax + 1]
        slmat[:, 1, 0] = slmat[:, 0, 1]
        slmat[:, 0, 2] = s_cls.get('tb', z)[:lmax + 1]
        slmat[:, 2, 0] = slmat[:, 0, 2]
        slmat[:, 1, 1] = s_cls.get('ee', z)[:lmax + 1]
        slmat[:, 1, 2] = s_cls.get('eb', z)[:lmax + 1]
        slmat[:, 2, 1] = slmat[:, 1, 2]
        slmat[:, 2, 2] = s_cls.get('bb', z)[:lmax + 1]
        self.lmax = lmax
        self.slinv = np.linalg.pinv(slmat)
        self.te_only = not (np.any(slmat[:, 0, 2]) or np.any(slmat[:, 1, 2]))

    def calc(self, alm):
        return _apply_3x3(self.slinv, alm, self.te_only)

    def hashdict(self):
        return {'slinv': clhash(self.slinv.flatten())}


class alm_filter_ninv(object):
    """Pixel-space inverse noise for (T, Q, U): maps (TT, (QQ + UU) / 2) or (TT, QQ, QU, UU); each entry may be a list of
    maps / paths / scalars to be multiplied together.  Optional T templates (monopole, dipole, maps) are projected out."""

    def __init__(self, n_inv, b_transf, b_transf_e=None, b_transf_b=None, marge_monopole=False, marge_dipole=False,
                 marge_maps_t=(), marge_maps_p=()):
        self.n_inv = []
        for tn in n_inv:
            if isinstance(tn, list):
                prod = np.array(util.read_map(tn[0]), dtype=float)
                for n in tn[1:]:
                    prod = prod * util.read_map(n)
                self.n_inv.append(dev.to_dev(prod, torch.float64))
            else:
                self.n_inv.append(dev.to_dev(util.read_map(tn), torch.float64))
        assert len(self.n_inv) in (2, 4), len(self.n_inv)
        npix = self.n_inv[0].numel()
        for n in self.n_inv[1:]:
            assert n.numel() == npix
        templates_t, templates_t_hash = [], []
        for tmap in [util.read_map(m) for m in marge_maps_t]:
            assert npix == len(tmap)
            templates_t.append(template_removal.template_map(tmap))
            templates_t_hash.append(clhash(tmap))
        if marge_monopole:
            templates_t.append(template_removal.template_monopole())
        if marge_dipole:
            templates_t.append(template_removal.template_dipole())
        if len(templates_t) != 0:
            nmodes = int(np.sum([t.nmodes for t in templates_t]))
            modes_idx_t = np.concatenate([t.nmodes * [int(im)] for im, t in enumerate(templates_t)])
            modes_idx_i = np.concatenate([range(0, t.nmodes) for t in templates_t])
            Pt_Nn1_P = np.zeros((nmodes, nmodes))
            for ir in range(nmodes):
                tmap = self.n_inv[0].clone()
                templates_t[modes_idx_t[ir]].apply_mode(tmap, int(modes_idx_i[ir]))
                ic = 0
                for tc in templates_t[0:modes_idx_t[ir] + 1]:
                    Pt_Nn1_P[ir, ic:(ic + tc.nmodes)] = tc.dot(tmap)
                    Pt_Nn1_P[ic:(ic + tc.nmodes), ir] = Pt_Nn1_P[ir, ic:(ic + tc.nmodes)]
                    ic += tc.nmodes
            self.Pt_Nn1_P_inv = np.linalg.inv(Pt_Nn1_P)
        self.b_transf_t = b_transf
        self.b_transf_e = b_transf_e if b_transf_e is not None else b_transf
        self.b_transf_b = b_transf_b if b_transf_b is not None else b_transf
        assert len(self.b_transf_t) == len(self.b_transf_e) and len(self.b_transf_t) == len(self.b_transf_b)
        self.b_transf = (self.b_transf_t + self.b_transf_e + self.b_transf_t) / 3.  # as in the reference (opfilt_tp.py:221)
        self.marge_monopole = marge_monopole
        self.marge_dipole = marge_dipole
        self.templates_t = templates_t
        self.templates_t_hash = templates_t_hash
        assert len(marge_maps_p) == 0
        self.templates_p = []
        self.npix = npix
        self.nside = hp.npix2nside(npix)

    def get_ftebl(self):
        if len(self.n_inv) == 2:  # TT, 1/2 (QQ + UU)
            nt, npol = float(self.n_inv[0].sum()), float(self.n_inv[1].sum())
        else:  # TT, QQ, QU, UU
            nt, npol = float(self.n_inv[0].sum()), float((0.5 * (self.n_inv[1] + self.n_inv[3])).sum())
        return (nt / (4.0 * np.pi) * self.b_transf_t ** 2, npol / (4.0 * np.pi) * self.b_transf_e ** 2,
                npol / (4.0 * np.pi) * self.b_transf_b ** 2)

    def hashdict(self):
        return {'n_inv': [clhash(dev.to_host(n)) for n in self.n_inv], 'b_transf': clhash(self.b_transf),
                'marge_monopole': self.marge_monopole, 'marge_dipole': self.marge_dipole,
                'templates_t_hash': self.templates_t_hash}

    def degrade(self, nside):
        """Coarser copy: hp.ud_grade(power=-2) sums the inverse variances of the children; template maps are dropped."""
        if nside == self.nside:
            return self
        print("DEGRADING WITH NO MARGE MAPS")
        return alm_filter_ninv([hp.ud_grade(dev.to_host(n), nside, power=-2) for n in self.n_inv], self.b_transf_t,
                               b_transf_e=self.b_transf_e, b_transf_b=self.b_transf_b, marge_monopole=self.marge_monopole,
                               marge_dipole=self.marge_dipole, marge_maps_t=(), marge_maps_p=())

    def apply_alm(self, alm):
        """alm <- B^t Y^t N^-1 Y B alm (in place)."""
        ret = self.apply_alm_new(alm)
        alm.tlm.copy_(ret.tlm)
        alm.elm.copy_(ret.elm)
        alm.blm.copy_(ret.blm)

    def _proj_matrices(self):
        """All T template modes as one device matrix P (nmodes x npix) and R = (P^t N^-1 P)^-1 P^t N^-1: the projection is
        c = P^t (N^-1 t), t -= R^t c on the device (nothing comes back to the host inside a CG iteration)."""
        if len(self.templates_t) == 0:
            return None, None
        if getattr(self, '_pmat', None) is None:
            rows = []
            for t in self.templates_t:
                for i in range(t.nmodes):
                    row = torch.ones_like(self.n_inv[0])
                    t.apply_mode(row, i)
                    rows.append(row)
            self._pmat = torch.stack(rows).contiguous()
            pinv = dev.to_dev(np.ascontiguousarray(self.Pt_Nn1_P_inv), torch.float64)
            self._rmat = torch.mm(pinv, self._pmat * self.n_inv[0].unsqueeze(0)).contiguous()
        return self._pmat, self._rmat

    def _harm_matrices(self, lmax):
        """the temperature templates in harmonic space (template_removal.harmonic_matrices), once per band-limit"""
        cache = self.__dict__.setdefault('_harm', {})
        if lmax not in cache:
            cache[lmax] = template_removal.harmonic_matrices(self.templates_t, self.n_inv[0], map2alm, lmax,
                                                             self.b_transf_t * (self.npix / (4. * np.pi)), self.Pt_Nn1_P_inv)
        return cache[lmax]

    def one_call_ok(self, alm):
        """pl_cg_fwd_tt + pl_cg_fwd_pp apply: device vectors, (TT, QQ = UU) noise, one polarization beam, the module's transforms"""
        same_b = self.b_transf_b is self.b_transf_e or np.array_equal(self.b_transf_e, self.b_transf_b)
        nmodes = sum(t.nmodes for t in self.templates_t)
        return (isinstance(alm.tlm, torch.Tensor) and alm.tlm.is_cuda and len(self.n_inv) == 2 and same_b and nmodes <= dev.TEMPLATE_MAX_MODES
                and alm2map is shts.alm2map and map2alm is shts.map2alm and alm2map_spin is shts.alm2map_spin
                and map2alm_spin is shts.map2alm_spin and not shts.lane_active()
                and all(isinstance(n, torch.Tensor) and n.is_contiguous() and n.dtype == torch.float64 for n in self.n_inv))

    def apply_alm_new(self, alm):
        """B^t Y^t N^-1 Y B alm as a new teblm (the input is left alone); the beams are fused into the transforms."""
        lmax = alm.lmax
        assert alm.lmaxt == alm.lmaxe == alm.lmaxb == lmax
        if self.one_call_ok(alm):  # the temperature and polarization blocks as the one-call operators of opfilt_tt / opfilt_pp
            fac = self.npix / (4. * np.pi)
            nb = alm.tlm.shape[0] if alm.tlm.dim() == 2 else 1
            md = (len(self.templates_t) == 2 and isinstance(self.templates_t[0], template_removal.template_monopole)
                  and isinstance(self.templates_t[1], template_removal.template_dipole) and options.opts.tproj_md
                  and not shts.plan_all_generic(self.nside, lmax))
            if len(self.templates_t) != 0 and options.opts.tproj_harm:
                # the temperature templates as a rank-nmodes update in harmonic space, as in opfilt_tt (pl_lowrank_update_b)
                hpm, hrm = self._harm_matrices(lmax)
                ttlm = shts.cg_fwd_tt(alm.tlm, self.nside, lmax, self.n_inv[0], fl_in=self.b_transf_t, fl_out=self.b_transf_t * fac, lowrank=(hpm, hrm))
            elif md:  # monopole + dipole evaluated from the ring geometry (pl_cg_fwd_tt_md_b), as in opfilt_tt
                if getattr(self, '_pinv_md_dev', None) is None:
                    self._pinv_md_dev = dev.to_dev(np.ascontiguousarray(self.Pt_Nn1_P_inv), torch.float64).contiguous()
                ttlm = shts.cg_fwd_tt(alm.tlm, self.nside, lmax, self.n_inv[0], fl_in=self.b_transf_t, fl_out=self.b_transf_t * fac,
                                      pinv_md=self._pinv_md_dev)
            else:
                pmat, rmat = self._proj_matrices()
                ttlm = shts.cg_fwd_tt(alm.tlm, self.nside, lmax, self.n_inv[0], fl_in=self.b_transf_t, fl_out=self.b_transf_t * fac,
                                      pmat=pmat, rmat=rmat, scratch=dev.tproj_scratch(nb) if pmat is not None else None)
            telm, tblm = shts.cg_fwd_pp(alm.elm, alm.blm, self.nside, lmax, self.n_inv[1], fl_in=self.b_transf_e,
                                        fl_out=self.b_transf_e * fac)
            return teblm([ttlm, telm, tblm])
        assert not (isinstance(alm.tlm, torch.Tensor) and alm.tlm.dim() == 2), 'block vectors take the one-call operators (pl_cg_fwd_tt_b / _pp_b)'
        same_b = self.b_transf_b is self.b_transf_e or np.array_equal(self.b_transf_e, self.b_transf_b)
        tmap = alm2map(alm.tlm, self.nside, lmax=lmax, fl=self.b_transf_t)
        if same_b:
            qmap, umap = alm2map_spin([alm.elm, alm.blm], self.nside, 2, lmax, fl=self.b_transf_e)
        else:
            qmap, umap = alm2map_spin([dev.almxfl(alm.elm, self.b_transf_e), dev.almxfl(alm.blm, self.b_transf_b)], self.nside, 2, lmax)
        maps = [tmap, qmap, umap]
        self.apply_map(maps)
        fac = self.npix / (4. * np.pi)
        ttlm = map2alm(maps[0], lmax=lmax, iter=0, fl=self.b_transf_t * fac)
        if same_b:
            telm, tblm = map2alm_spin([maps[1], maps[2]], 2, lmax=lmax, fl=self.b_transf_e * fac)
        else:
            telm, tblm = map2alm_spin([maps[1], maps[2]], 2, lmax=lmax)
            telm, tblm = dev.almxfl(telm, self.b_transf_e * fac), dev.almxfl(tblm, self.b_transf_b * fac)
        return teblm([ttlm, telm, tblm])

    def apply_map(self, amap):
        """(T, Q, U) <- N^-1 (T, Q, U) with the T templates projected out (in place)."""
        tmap, qmap, umap = amap
        self._proj_matrices()
        # temperature: N^-1 weighting and template projection in two launches (pl_template_project), as in opfilt_tt
        fused_t = len(self.templates_t) != 0 and self._pmat.shape[0] <= dev.TEMPLATE_MAX_MODES and tmap.is_contiguous()
        if fused_t:
            dev.template_project(tmap, self.n_inv[0], self._pmat, self._rmat)
        else:
            tmap *= self.n_inv[0]
        if len(self.n_inv) == 2:  # TT, QQ = UU
            qmap *= self.n_inv[1]
            umap *= self.n_inv[1]
        elif qmap.is_contiguous() and umap.is_contiguous() and all(n.is_contiguous() for n in self.n_inv[1:]):  # TT, QQ, QU, UU
            dev.map_qu_weight(qmap, umap, self.n_inv[1], self.n_inv[2], self.n_inv[3])  # one pass (pl_map_qu_weight)
        else:
            qmap_copy = qmap.clone()
            qmap *= self.n_inv[1]
            qmap += self.n_inv[2] * umap
            umap *= self.n_inv[3]
            umap += self.n_inv[2] * qmap_copy
        if len(self.templates_t) != 0 and not fused_t:  # more modes than pl_template_project takes: two mat-vecs (pl_gemv) + rank update
            coeffs = dev.gemv(self._pmat, tmap)
            tmap.addmv_(self._rmat.t(), coeffs, alpha=-1.0)
