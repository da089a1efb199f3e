"""Polarization Wiener / inverse-variance filtering operators on the device, API of plancklens/qcinv/opfilt_pp.py
(`dot_op` :27-34, `fwd_op` :37-55, `pre_op_diag` :57-80, `alm_filter_sinv` :87-110, `alm_filter_ninv` :113-303,
`calc_prep` :306-317, `apply_fini` :320-324).  Vectors are util_alm.eblm pairs of device tensors."""
from __future__ import print_function


import numpy as np
import torch

from .. import dev, hp, options, shts
from ..utils import clhash
from . import dense, template_removal, util
from .util_alm import eblm

alm2map_spin, map2alm_spin = shts.alm2map_spin, shts.map2alm_spin


class dot_op(object):
    """sum_{l >= 2} (2l + 1) (C_l^{EE'} + C_l^{BB'})."""
    lmin = 2  # first multipole of the sum

    def parts(self, alm1, alm2):
        """the scalar product as dev.DOT_PARTS partial sums in device memory (one launch per field, no host synchronisation);
        its value is their sum, formed by whoever consumes it (axpy below, dev(), __call__)"""
        assert alm1.lmax == alm2.lmax
        return dev.alm_dot([(alm1.elm, alm2.elm), (alm1.blm, alm2.blm)], lmin=2)

    def dev(self, alm1, alm2):
        """the scalar product as a 0-dim device tensor (block vectors: one value per entry)"""
        return self.parts(alm1, alm2).sum(-1)

    @staticmethod
    def axpy(y, x, num, den, sign):
        """y += sign num / den x in place (num, den: scalar products as returned by parts)"""
        dev.axpy_dev(y.elm, x.elm, num, den, sign)
        dev.axpy_dev(y.blm, x.blm, num, den, sign)

    @staticmethod
    def step(x, d, r, q, update_r=True, active=None, pre=None, x_init=False):
        """one conjugate-directions update, all fields in two launches (or one with a grid barrier): dTAd = <d, q>, delta = <d, r>,
        x += (delta / dTAd) d and, if update_r, r -= (delta / dTAd) q; returns (dTAd, delta) as `parts` does.
        active (block vectors): 0 / 1 per entry, multiplies the step lengths.
        pre: the two scalar products as left by the operator that made q (fwd_op.with_dots): the updates alone, one launch;
        x_init (with pre): x = (delta / dTAd) d -- x is written, not read (first step of a solve from zero)"""
        f = (lambda v: [v.elm, v.blm])
        if pre is not None:
            return dev.cg_axpy_pre(pre, f(x), f(d), 1.0, y2=f(r) if update_r else None, x2=f(q) if update_r else None, sign2=-1.0, active=active,
                                   assign_y1=x_init)
        assert not x_init
        return dev.cg_dot_axpy(f(d), f(q), f(x), f(d), 1.0, b2=f(r), y2=f(r) if update_r else None, x2=f(q) if update_r else None,
                               sign2=-1.0, lmin=2, active=active)

    @staticmethod
    def ortho(s, pq, pd, prev_dtad, pre=None):
        """s -= (<s, pq> / prev_dtad) pd, all fields in two launches (or one with a grid barrier).
        pre: <s, pq> as partial sums left by the preconditioner kernel(s) that wrote s (pre_op.with_dot): the update alone"""
        f = (lambda v: [v.elm, v.blm])
        if pre is not None:
            dev.cg_axpy_pre((pre, None), f(s), f(pd), -1.0, den=prev_dtad)
            return
        dev.cg_dot_axpy(f(s), f(pq), f(s), f(pd), -1.0, den=prev_dtad, lmin=2)

    def __call__(self, alm1, alm2):
        p = self.parts(alm1, alm2)
        return float(p.sum()) if p.dim() == 1 else dev.to_host(p.sum(-1))  # block vectors: one value per entry


class fwd_op(object):
    def __init__(self, s_cls, n_inv_filt):
        lmax = len(n_inv_filt.b_transf) - 1
        self.s_inv_filt = alm_filter_sinv(s_cls, lmax)
        self.n_inv_filt = n_inv_filt

    def hashdict(self):
        return {'s_inv_filt': self.s_inv_filt.hashdict(), 'n_inv_filt': self.n_inv_filt.hashdict()}

    def __call__(self, alm):
        return self.calc(alm)

    def with_dots(self, alm, r):
        """(fwd_op(alm), pre): pre = the scalar products <alm, result> and <alm, r> as partial sums left by the kernel that writes the
        result (dot_op.step takes them), or None where the operator is not the one-call form"""
        f, sl = util.unjit(self.n_inv_filt), self.s_inv_filt.slinv
        ok = (isinstance(f, alm_filter_ninv) and f.one_call_ok(alm) and not f.wmarg and not np.any(sl[:, 0, 1]) and not np.any(sl[:, 1, 0])
              and isinstance(r, eblm) and all(isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.complex128 and t.is_contiguous()
                                              and t.shape == alm.elm.shape for t in (alm.elm, alm.blm, r.elm, r.blm)))
        if ok:
            req = shts.post_dots([r.elm, r.blm], lmin=2)
            q = f.apply_alm_new(alm, add=alm, fl_add_e=sl[:, 0, 0], fl_add_b=sl[:, 1, 1], dots=req)
            return q, req.pre
        return self.calc(alm), None

    def calc(self, alm):
        f, sl = util.unjit(self.n_inv_filt), self.s_inv_filt.slinv  # (the filter libraries hand over a lazily built filter)
        if isinstance(f, alm_filter_ninv) and f.one_call_ok(alm) and not np.any(sl[:, 0, 1]) and not np.any(sl[:, 1, 0]):
            return f.apply_alm_new(alm, add=alm, fl_add_e=sl[:, 0, 0], fl_add_b=sl[:, 1, 1])  # the whole operator in pl_cg_fwd_pp
        assert not (isinstance(alm.elm, torch.Tensor) and alm.elm.dim() == 2), 'block vectors take the one-call operator (pl_cg_fwd_pp_b)'
        nlm = f.apply_alm_new(alm)
        return _apply_2x2(sl, alm, add_to=nlm)


def _apply_2x2(tmat, alm, add_to=None):
    """(E, B) <- per-l 2x2 matrix applied to (E, B); off-diagonal terms skipped when they vanish identically.
    add_to: an eblm the result is accumulated into in place (and which is returned)."""
    if add_to is None:
        relm, rblm = dev.almxfl(alm.elm, tmat[:, 0, 0]), dev.almxfl(alm.blm, tmat[:, 1, 1])
    else:
        relm = dev.almxfl_add(add_to.elm, alm.elm, tmat[:, 0, 0], out=add_to.elm)
        rblm = dev.almxfl_add(add_to.blm, alm.blm, tmat[:, 1, 1], out=add_to.blm)
    if np.any(tmat[:, 0, 1]):
        dev.almxfl_add(relm, alm.blm, tmat[:, 0, 1], out=relm)
    if np.any(tmat[:, 1, 0]):
        dev.almxfl_add(rblm, alm.elm, tmat[:, 1, 0], out=rblm)
    return add_to if add_to is not None else eblm([relm, rblm])


class pre_op_diag(object):
    def __init__(self, s_cls, n_inv_filt):
        lmax = len(n_inv_filt.b_transf) - 1
        s_inv_filt = alm_filter_sinv(s_cls, lmax)
        assert (s_inv_filt.lmax + 1) >= len(n_inv_filt.b_transf)
        ninv_fel, ninv_fbl = n_inv_filt.get_febl()
        flmat = s_inv_filt.slinv.copy()
        flmat[:, 0, 0] += ninv_fel[:lmax + 1]
        flmat[:, 1, 1] += ninv_fbl[:lmax + 1]
        self.flmat = np.linalg.pinv(flmat)

    def __call__(self, talm):
        return self.calc(talm)

    def calc(self, alm):
        return _apply_2x2(self.flmat, alm)

    def splice_above(self, alm_low, alm, lsplit, dot=None):
        """alm_low for l <= lsplit, this preconditioner applied to alm above: pre_op_split's result in one launch per field
        (None: not here -- host vectors, or E-B coupling in the spectra).
        dot = (q, lmin): returns (result, pre), pre the partial sums of <result, q> formed by the same launches (E's, then B's)"""
        fm = self.flmat
        if np.any(fm[:, 0, 1]) or np.any(fm[:, 1, 0]) or not (isinstance(alm.elm, torch.Tensor) and alm.elm.is_cuda):
            return None
        if dot is not None:
            q, lmin = dot
            n = dev.alm_splice_dot_count(alm.lmax)
            blk = alm.elm.dim() == 2
            pre = torch.empty((alm.elm.shape[0], 2, n) if blk else (2, n), dtype=torch.float64, device=alm.elm.device)
            if blk:  # per entry [E partial sums | B partial sums]: each launch fills its half through a contiguous scratch
                pe, pb = torch.empty((alm.elm.shape[0], n), dtype=torch.float64, device=pre.device), torch.empty((alm.elm.shape[0], n), dtype=torch.float64, device=pre.device)
            else:
                pe, pb = pre[0], pre[1]
            ret = eblm([dev.alm_splice_fl(alm_low.elm, alm.elm, fm[:, 0, 0], lsplit, dot=(q.elm, lmin, pe)),
                        dev.alm_splice_fl(alm_low.blm, alm.blm, fm[:, 1, 1], lsplit, dot=(q.blm, lmin, pb))])
            if blk:
                pre[:, 0].copy_(pe)
                pre[:, 1].copy_(pb)
            return ret, pre.reshape(alm.elm.shape[0], 2 * n) if blk else pre.reshape(2 * n)
        return eblm([dev.alm_splice_fl(alm_low.elm, alm.elm, fm[:, 0, 0], lsplit), dev.alm_splice_fl(alm_low.blm, alm.blm, fm[:, 1, 1], lsplit)])


def pre_op_dense(lmax, fwd_op, cache_fname=None):
    return dense.pre_op_dense_pp(lmax, fwd_op, cache_fname=cache_fname)


class alm_filter_sinv(object):
    """S^-1: per-l pseudo-inverse of the (EE, EB; EB, BB) spectral matrix."""

    def __init__(self, s_cls, lmax):
        slmat = np.zeros((lmax + 1, 2, 2))
        slmat[:, 0, 0] = s_cls.get('ee', np.zeros(lmax + 1))[:lmax + 1]
        slmat[:, 0, 1] = s_cls.get('eb', np.zeros(lmax + 1))[:lmax + 1]
        slmat[:, 1, 0] = s_cls.get('eb', np.zeros(lmax + 1))[:lmax + 1]
        slmat[:, 1, 1] = s_cls.get('bb', np.zeros(lmax + 1))[:lmax + 1]
        self.lmax = lmax
        self.slinv = np.linalg.pinv(slmat)

    def calc(self, alm):
        return _apply_2x2(self.slinv, alm)

    def hashdict(self):
        return {'slinv': clhash(self.slinv.flatten())}


class alm_filter_ninv(object):
    """Pixel-space inverse noise for (Q, U): one map (QQ = UU, QU = 0) or three (QQ, QU, UU); optional Q / U templates."""

    def __init__(self, n_inv, b_transf, nlev_febl=None, b_transf_b=None, marge_qmaps=(), marge_umaps=()):
        self.b_transf_e = b_transf
        self.b_transf_b = b_transf_b if b_transf_b is not None else b_transf
        self.b_transf = 0.5 * (self.b_transf_e + self.b_transf_b)
        self.nside = None
        self.n_inv = None
        self.nlev_febl = nlev_febl
        self._n_inv = n_inv
        self.marge_qmaps = marge_qmaps
        self.marge_umaps = marge_umaps
        self.wmarg = max(len(self.marge_qmaps), len(self.marge_umaps)) > 0
        self.tniti = None
        self.templates_p = []

    def _load_ninv(self):
        if self.n_inv is None:
            self.n_inv = [dev.to_dev(util.read_map(tn), torch.float64) for tn in self._n_inv]
            assert len(self.n_inv) in [1, 3], len(self.n_inv)
            self.nside = hp.npix2nside(self.n_inv[0].numel())

    def _build_tniti(self):
        if not self.wmarg or self.tniti is not None:
            return
        blocks = []
        for im, marge_m in enumerate((self.marge_qmaps, self.marge_umaps)):
            if len(marge_m) == 0:
                continue
            this_n_inv = self.get_ninv()
            assert len(this_n_inv) == 1, 'QQ QU UU not implemented'
            templates = [_template_pmap(m, im) for m in marge_m]
            nmodes = len(templates)
            mat = np.zeros((nmodes, nmodes))
            for ir in range(nmodes):
                w = this_n_inv[0] * templates[ir].map
                for ic in range(ir + 1):
                    mat[ir, ic] = mat[ic, ir] = float(torch.dot(templates[ic].map, w))
            eigv, eigw = np.linalg.eigh(mat)
            blocks.append(np.dot(np.dot(eigw, np.diag(1.0 / eigv)), eigw.T))
            self.templates_p = self.templates_p + templates
        if blocks:
            n = sum(b.shape[0] for b in blocks)
            self.tniti = np.zeros((n, n))
            i = 0
            for b in blocks:
                self.tniti[i:i + b.shape[0], i:i + b.shape[0]] = b
                i += b.shape[0]

    def _calc_febl(self):
        self._load_ninv()
        if len(self.n_inv) == 1:
            s = float(self.n_inv[0].sum())
        else:
            s = float((0.5 * (self.n_inv[0] + self.n_inv[2])).sum())
        nlev_febl = 10800. / np.sqrt(s / (4.0 * np.pi)) / np.pi
        print("ninv_febl: using %.2f uK-amin noise Cl" % nlev_febl)
        return nlev_febl

    def get_ninv(self):
        self._load_ninv()
        return self.n_inv

    def get_mask(self):
        ninv = self.get_ninv()
        mask = (ninv[0] > 0).to(torch.float64)
        for ni in ninv[1:]:
            mask *= (ni > 0)
        return dev.to_host(mask)

    def get_febl(self):
        if self.nlev_febl is None:
            self.nlev_febl = self._calc_febl()
        f = 1. / (self.nlev_febl / 180. / 60. * np.pi) ** 2
        return self.b_transf_e ** 2 * f, self.b_transf_b ** 2 * f

    def hashdict(self):
        t_hash = []
        if self.wmarg:
            t_hash = [util.mask_hash(m, dtype=np.float32) for m in self.marge_qmaps] + \
                     [util.mask_hash(m, dtype=np.float32) for m in self.marge_umaps]
        return {'n_inv': [util.mask_hash(n, dtype=np.float16) for n in self._n_inv], 'b_transf': clhash(self.b_transf),
                'templates_p': t_hash}

    def degrade(self, nside):
        self._load_ninv()
        if nside == self.nside:
            return self
        return alm_filter_ninv([hp.ud_grade(dev.to_host(n), nside, power=-2) for n in self.n_inv], self.b_transf_e,
                               b_transf_b=self.b_transf_b)

    def apply_alm(self, alm):
        """(E, B) <- B^t Y^t N^-1 Y B (E, B), in place."""
        ret = self.apply_alm_new(alm)
        alm.elm.copy_(ret.elm)
        alm.blm.copy_(ret.blm)

    def one_call_ok(self, alm):
        """pl_cg_fwd_pp applies: device vectors, one inverse-noise map or three (QQ, QU, UU), no templates, one beam, the module's
        transforms not replaced."""
        self._load_ninv()
        same_b = self.b_transf_b is self.b_transf_e or np.array_equal(self.b_transf_e, self.b_transf_b)
        # (templates: as a rank-nmodes update in harmonic space, single vectors only -- the (E, B) pair of a block entry is not contiguous)
        harm = (self.wmarg and options.opts.tproj_harm and isinstance(alm.elm, torch.Tensor) and alm.elm.dim() == 1
                and len(self.n_inv) == 1 and len(self.marge_qmaps) + len(self.marge_umaps) <= dev.TEMPLATE_MAX_MODES)
        return (isinstance(alm.elm, torch.Tensor) and alm.elm.is_cuda and len(self.n_inv) in (1, 3) and (not self.wmarg or harm) and same_b
                and alm2map_spin is shts.alm2map_spin and map2alm_spin is shts.map2alm_spin and not shts.lane_active()
                and all(n.is_contiguous() and n.dtype == torch.float64 for n in self.n_inv))

    def supports_block(self):
        """True when block vectors (eblm of [nb, nalm] tensors) can go through this filter: they take the one-call operator only
        (fwd_op.calc), which marginalised Q / U templates, two different beams or replaced transforms rule out"""
        self._load_ninv()
        z = torch.empty((1, 1), dtype=torch.complex128, device=self.n_inv[0].device)
        return self.one_call_ok(eblm([z, z]))

    def apply_alm_new(self, alm, add=None, fl_add_e=None, fl_add_b=None, dots=None):
        """B^t Y^t N^-1 Y B (E, B) (+ (fl_add_e E', fl_add_b B') for add = (E', B')) as a new eblm (the input is left alone)."""
        self._load_ninv()
        lmax = alm.lmax
        if self.one_call_ok(alm):
            npix = self.n_inv[0].numel()
            elm, blm = shts.cg_fwd_pp(alm.elm, alm.blm, self.nside, lmax, self.n_inv[0], fl_in=self.b_transf_e,
                                      fl_out=self.b_transf_e * (npix / (4. * np.pi)),
                                      add=None if add is None else (add.elm, add.blm), fl_add_e=fl_add_e, fl_add_b=fl_add_b,
                                      n_qu=self.n_inv[1] if len(self.n_inv) == 3 else None, n_uu=self.n_inv[2] if len(self.n_inv) == 3 else None,
                                      dots=dots)
            if self.wmarg:  # the Q / U templates: y -= V (T^t N^-1 T)^-1 V^t x on the stacked (E, B) vectors (pl_lowrank_update_b)
                hpm, hrm = self._harm_matrices(lmax)
                y = torch.stack([elm, blm])
                dev.lowrank_update(y.view(-1), torch.stack([alm.elm, alm.blm]).to(torch.complex128).view(-1), hpm, hrm)
                elm, blm = y[0], y[1]
            return eblm([elm, blm])
        assert dots is None
        ret = self._apply_alm_steps(alm)
        if add is not None:
            dev.almxfl_add(ret.elm, add.elm, fl_add_e, out=ret.elm)
            dev.almxfl_add(ret.blm, add.blm, fl_add_b, out=ret.blm)
        return ret

    def _apply_alm_steps(self, alm):
        lmax = alm.lmax
        same_b = self.b_transf_b is self.b_transf_e or np.array_equal(self.b_transf_e, self.b_transf_b)
        if same_b:
            qmap, umap = alm2map_spin([alm.elm, alm.blm], self.nside, 2, lmax, fl=self.b_transf_e)
        else:
            qmap, umap = alm2map_spin([dev.almxfl(alm.elm, self.b_transf_e), dev.almxfl(alm.blm, self.b_transf_b)], self.nside, 2, lmax)
        self.apply_map([qmap, umap])
        npix = qmap.numel()
        if same_b:
            telm, tblm = map2alm_spin([qmap, umap], 2, lmax=lmax, fl=self.b_transf_e * (npix / (4. * np.pi)))
        else:
            telm, tblm = map2alm_spin([qmap, umap], 2, lmax=lmax)
            telm = dev.almxfl(telm, self.b_transf_e * (npix / (4. * np.pi)))
            tblm = dev.almxfl(tblm, self.b_transf_b * (npix / (4. * np.pi)))
        return eblm([telm, tblm])

    def apply_map(self, amap):
        """(Q, U) <- N^-1 (Q, U) in place."""
        self._load_ninv()
        qmap, umap = amap
        if len(self.n_inv) == 1 and self.wmarg:
            # Templates marginalised: N^-1 - N^-1 P (P^t N^-1 P)^-1 P^t N^-1 on the (Q, U) pair taken as one vector of 2 npix pixels.
            # The modes (Q-only or U-only maps) and R = (P^t N^-1 P)^-1 (P . N^-1) are device matrices; weighting, coefficients and
            # projection are the two launches of pl_template_project -- nothing comes back to the host inside a CG iteration
            # (the reference forms the coefficients with np.dot on the host, opfilt_pp.py:284-294).
            pmat, rmat, n2 = self._proj_matrices_p()
            if pmat.shape[0] <= dev.TEMPLATE_MAX_MODES:
                qu = shts._stack([qmap, umap])  # the halves of one (2, npix) tensor as the transforms return them: no copy
                shared = qu.data_ptr() == qmap.data_ptr()
                dev.template_project(qu.view(-1), n2, pmat, rmat)
                if not shared:
                    qmap.copy_(qu[0])
                    umap.copy_(qu[1])
            else:  # more modes than pl_template_project takes: mat-vecs
                qu = torch.cat([qmap, umap])
                qu *= n2
                qu.addmv_(rmat.t(), dev.gemv(pmat, qu), alpha=-1.0)
                qmap.copy_(qu[:qmap.numel()])
                umap.copy_(qu[qmap.numel():])
        elif len(self.n_inv) == 1:
            qmap *= self.n_inv[0]
            umap *= self.n_inv[0]
        elif len(self.n_inv) == 3:
            if qmap.is_contiguous() and umap.is_contiguous() and all(n.is_contiguous() and n.dtype == torch.float64 for n in self.n_inv):
                dev.map_qu_weight(qmap, umap, self.n_inv[0], self.n_inv[1], self.n_inv[2])  # one pass (pl_map_qu_weight)
            else:
                qcopy = qmap.clone()
                qmap *= self.n_inv[0]
                qmap += self.n_inv[1] * umap
                umap *= self.n_inv[2]
                umap += self.n_inv[1] * qcopy
        else:
            assert 0

    def _harm_matrices(self, lmax):
        """The Q / U templates in harmonic space, V_k = B^t Y2^t N^-1 T_k as an (E, B) pair per mode (see
        template_removal.harmonic_matrices for the scalar case): real (nmodes, 4 nalm) device matrices for dev.lowrank_update on
        stacked (E, B) vectors -- V with the weights of the real scalar product folded in, and (T^t N^-1 T)^-1 V."""
        cache = self.__dict__.setdefault('_harm', {})
        if lmax not in cache:
            self._build_tniti()
            npix = self.n_inv[0].numel()
            fl_out = self.b_transf_e * (npix / (4. * np.pi))
            rows = []
            for t in self.templates_p:
                qu = torch.zeros((2, npix), dtype=torch.float64, device=self.n_inv[0].device)
                qu[t.comp] = self.n_inv[0] * t.map
                e, b = map2alm_spin([qu[0], qu[1]], 2, lmax=lmax, fl=fl_out)
                vlm = torch.stack([dev.to_dev(e, torch.complex128), dev.to_dev(b, torch.complex128)]).contiguous()
                rows.append(torch.view_as_real(vlm).reshape(-1))
            v = torch.stack(rows).contiguous()
            nalm = v.shape[1] // 4
            w = torch.full((nalm,), 2., dtype=torch.float64, device=v.device)
            w[:lmax + 1] = 1.
            w4 = w.repeat_interleave(2).repeat(2)
            pinv = dev.to_dev(np.ascontiguousarray(self.tniti), torch.float64)
            cache[lmax] = ((v * w4.unsqueeze(0)).contiguous(), torch.mm(pinv, v).contiguous())
        return cache[lmax]

    def _proj_matrices_p(self):
        """(pmat, rmat, n_inv2): the template modes as rows of a (nmodes, 2 npix) device matrix (a Q template is zero on the U half
        and vice versa), rmat = (P^t N^-1 P)^-1 (pmat . n_inv2) and n_inv2 = the inverse-noise map repeated for Q and U"""
        if getattr(self, '_pmat_p', None) is None:
            self._build_tniti()
            npix = self.n_inv[0].numel()
            pm = torch.zeros((len(self.templates_p), 2 * npix), dtype=torch.float64, device=self.n_inv[0].device)
            for i, t in enumerate(self.templates_p):
                pm[i, t.comp * npix:(t.comp + 1) * npix] = t.map
            self._ninv2 = torch.cat([self.n_inv[0], self.n_inv[0]]).contiguous()
            self._pmat_p = pm.contiguous()
            self._rmat_p = torch.mm(dev.to_dev(np.ascontiguousarray(self.tniti), torch.float64), pm * self._ninv2.unsqueeze(0)).contiguous()
        return self._pmat_p, self._rmat_p, self._ninv2


class _template_pmap(object):
    """A Q-only (comp 0) or U-only (comp 1) template map (template_removal.py template_qmap / template_umap)."""

    def __init__(self, tmap, comp):
        self.nmodes = 1
        self.comp = comp
        self.map = dev.to_dev(util.read_map(tmap), torch.float64)

    def accum(self, pmap, coeffs):
        pmap[self.comp] += self.map * float(coeffs[0])

    def dot(self, pmap):
        return [float(torch.dot(self.map, pmap[self.comp]))]


def calc_prep(maps, s_cls, n_inv_filt):
    qmap = dev.to_dev(util.read_map(maps[0]), torch.float64).clone()
    umap = dev.to_dev(util.read_map(maps[1]), torch.float64).clone()
    assert qmap.numel() == umap.numel()
    lmax = len(n_inv_filt.b_transf) - 1
    npix = qmap.numel()
    n_inv_filt.apply_map([qmap, umap])
    elm, blm = map2alm_spin([qmap, umap], 2, lmax=lmax)
    return eblm([dev.almxfl(elm, n_inv_filt.b_transf_e * npix / (4. * np.pi)), dev.almxfl(blm, n_inv_filt.b_transf_b * npix / (4. * np.pi))])


def calc_prep_batch(maps, s_cls, n_inv_filt):
    """calc_prep of every (Q, U) pair of the list as one block vector: eblm of [nb, nalm] tensors"""
    preps = [calc_prep(m, s_cls, n_inv_filt) for m in maps]
    return eblm([torch.stack([p.elm for p in preps]).contiguous(), torch.stack([p.blm for p in preps]).contiguous()])


def apply_fini(alm, s_cls, n_inv_filt):
    ret = alm_filter_sinv(s_cls, alm.lmax).calc(alm)
    alm.elm.copy_(ret.elm)
    alm.blm.copy_(ret.blm)
