"""Dense low-l preconditioners, API of plancklens/qcinv/dense.py (`alm2rlm` / `rlm2alm` :16-54, `pre_op_dense_tt`
:57-119, `pre_op_dense_pp` :123-202, `pre_op_dense_tp` :204-285).  The (lmax+1)^2 k square matrix is filled by applying the coarse fwd_op to unit
vectors (device SHTs), pseudo-inverted on the host with eigh exactly as the reference does, and applied as one device
mat-vec (pl_gemv, csrc/elementwise.hip)."""
from __future__ import print_function

import os
import pickle as pk

import numpy as np
import torch

from .. import dev, options
from ..hp import Alm
from .util_alm import eblm, teblm

_RLM_IDX = {}


def _rlm_maps(lmax):
    """Index maps between the complex alm layout and the 'real harmonic' layout rlm[l^2 + 2m - 1 | 2m]."""
    if lmax not in _RLM_IDX:
        ls = np.arange(lmax + 1)
        re_alm, re_rlm, im_alm, im_rlm = [ls.copy()], [ls ** 2], [], []
        for m in range(1, lmax + 1):
            a = m * (2 * lmax + 1 - m) // 2 + ls[m:]
            re_alm.append(a); re_rlm.append(ls[m:] ** 2 + 2 * m - 1)
            im_alm.append(a); im_rlm.append(ls[m:] ** 2 + 2 * m)
        cat = lambda x: np.concatenate(x) if len(x) else np.zeros(0, dtype=int)
        _RLM_IDX[lmax] = (cat(re_alm), cat(re_rlm), cat(im_alm), cat(im_rlm))
    return _RLM_IDX[lmax]


_RLM_DEV = {}


def _rlm_maps_dev(lmax, d):
    """device copies of the index maps and m-weights (uploaded once: the dense preconditioner is applied dozens of times
    per CG iteration and every pageable upload is a blocking copy -- and illegal inside a graph capture)"""
    key = (lmax, str(d))
    if key not in _RLM_DEV:
        ra, rr, ia, ir = _rlm_maps(lmax)
        w2 = torch.full((ra.size,), np.sqrt(2.), dtype=torch.float64, device=d)
        w2[:lmax + 1] = 1.
        wi = torch.full((ra.size,), 1.0 / np.sqrt(2.), dtype=torch.float64, device=d)
        wi[:lmax + 1] = 1.
        _RLM_DEV[key] = tuple(torch.from_numpy(x).to(d) for x in (ra, rr, ia, ir)) + (w2, wi)
    return _RLM_DEV[key]


def alm2rlm(alm):
    """Complex alm -> real harmonic coefficients (m = 0 real part; sqrt(2) Re, sqrt(2) Im for m > 0)."""
    is_dev = isinstance(alm, torch.Tensor)
    n = alm.shape[-1] if is_dev else alm.size
    lmax = Alm.getlmax(n)
    ra, rr, ia, ir = _rlm_maps(lmax)
    rt2 = np.sqrt(2.)
    if is_dev:  # (a leading block dimension [nb, nalm] is carried along)
        d = alm.device
        dra, drr, dia, dir_, w2, _ = _rlm_maps_dev(lmax, d)
        rlm = torch.zeros(tuple(alm.shape[:-1]) + ((lmax + 1) ** 2,), dtype=torch.float64, device=d)
        rlm[..., drr] = alm.real[..., dra] * w2
        rlm[..., dir_] = alm.imag[..., dia] * rt2
        return rlm
    rlm = np.zeros((lmax + 1) ** 2)
    w = np.full(ra.size, rt2)
    w[:lmax + 1] = 1.
    rlm[rr] = alm.real[ra] * w
    rlm[ir] = alm.imag[ia] * rt2
    return rlm


def rlm2alm(rlm):
    """Inverse of alm2rlm."""
    is_dev = isinstance(rlm, torch.Tensor)
    n = rlm.shape[-1] if is_dev else len(rlm)
    lmax = int(np.sqrt(n) - 1)
    assert (lmax + 1) ** 2 == n
    ra, rr, ia, ir = _rlm_maps(lmax)
    ir2 = 1.0 / np.sqrt(2.)
    if is_dev:  # (a leading block dimension is carried along)
        d = rlm.device
        dra, drr, dia, dir_, _, wi = _rlm_maps_dev(lmax, d)
        re = torch.zeros(tuple(rlm.shape[:-1]) + (Alm.getsize(lmax),), dtype=torch.float64, device=d)
        im = torch.zeros_like(re)
        re[..., dra] = rlm[..., drr] * wi
        im[..., dia] = rlm[..., dir_] * ir2
        return torch.complex(re, im).contiguous()
    alm = np.zeros(Alm.getsize(lmax), dtype=complex)
    w = np.full(ra.size, ir2)
    w[:lmax + 1] = 1.
    alm.real[ra] = rlm[rr] * w
    alm.imag[ia] = rlm[ir] * ir2
    return alm


def _parts(v):
    if hasattr(v, 'tlm'):
        return [v.tlm, v.elm, v.blm]
    if hasattr(v, 'elm'):
        return [v.elm, v.blm]
    return [v]


def _flat_matrix(minv, lmax, nfields):
    """P_out minv P_in: minv (acting on `nfields` stacked rlm vectors of band-limit lmax) re-indexed and re-scaled so
    that it acts on the stacked interleaved (re, im) views of the alm arrays; rows and columns of Im a_l0 are zero."""
    ra, rr, ia, ir = _rlm_maps(lmax)
    nr, nf = (lmax + 1) ** 2, (lmax + 1) * (lmax + 2)
    assert minv.shape[0] == nfields * nr, (minv.shape, nfields, nr)
    f = np.zeros(nr, dtype=np.int64)
    sin, sout = np.zeros(nr), np.zeros(nr)
    w2 = np.full(ra.size, np.sqrt(2.)); w2[:lmax + 1] = 1.
    f[rr], sin[rr], sout[rr] = 2 * ra, w2, 1. / w2
    f[ir], sin[ir], sout[ir] = 2 * ia + 1, np.sqrt(2.), 1. / np.sqrt(2.)
    d = minv.device
    fa = torch.from_numpy(np.concatenate([f + k * nf for k in range(nfields)])).to(d)
    si = torch.from_numpy(np.tile(sin, nfields)).to(d)
    so = torch.from_numpy(np.tile(sout, nfields)).to(d)
    amat = torch.zeros((nfields * nf, nfields * nf), dtype=torch.float64, device=d)
    amat[fa.unsqueeze(1), fa.unsqueeze(0)] = so.unsqueeze(1) * minv * si.unsqueeze(0)
    return amat


class _pre_op_dense(object):
    """Shared machinery: brute-force matrix, eigen pseudo-inverse with `ntmpl` lowest modes left untouched, cache."""

    def __init__(self, lmax, fwd_op, cache_fname=None):
        self.lmax = lmax
        minv = None
        if cache_fname is not None and os.path.exists(cache_fname):
            cache_lmax, cache_hashdict, cache_minv = pk.load(open(cache_fname, 'rb'))
            if lmax == cache_lmax and self.hashdict(lmax, fwd_op) == cache_hashdict:
                minv = cache_minv
            else:
                print("WARNING: PRE_OP_DENSE CACHE: hashcheck failed. recomputing.")
                os.remove(cache_fname)
        if minv is None:
            minv = self.compute_minv(lmax, fwd_op, cache_fname=cache_fname)
        self.minv = dev.to_dev(minv, torch.float64)

    def _ntmpl(self, fwd_op):
        assert 0, 'override this'

    def _nrlm(self, lmax):
        assert 0, 'override this'

    def _to_rlm(self, alm):
        assert 0, 'override this'

    def _to_alm(self, rlm):
        assert 0, 'override this'

    def compute_minv(self, lmax, fwd_op, cache_fname=None):
        if cache_fname is not None:
            assert not os.path.exists(cache_fname)
        nrlm = self._nrlm(lmax)
        ntmpl = self._ntmpl(fwd_op)
        print("computing dense preconditioner: lmax = %d, ntmpl = %d, size %d" % (lmax, ntmpl, nrlm))
        tmat = torch.zeros((nrlm, nrlm), dtype=torch.float64, device=dev.device())
        # The columns are fwd_op of the unit vectors (dense.py:77-84 applies it to one vector at a time: thousands of launch-bound
        # coarse operators).  Here nbk unit vectors go through the operator as one block vector -- every launch carries all of them,
        # each column bit-identical to the one-at-a-time result; an operator that does not take blocks gets them one by one.
        nbk = max(1, min(int(options.opts.dense_block), 64))
        i0 = 0
        while i0 < nrlm:
            n_ = min(nbk, nrlm - i0)
            blk = torch.zeros((n_, nrlm), dtype=torch.float64, device=dev.device())
            blk[torch.arange(n_), i0 + torch.arange(n_)] = 1.0
            try:
                cols = self._to_rlm(fwd_op(self._to_alm(blk))) if n_ > 1 else None
            except AssertionError:
                cols, nbk = None, 1
            if cols is None:
                n_ = 1
                cols = self._to_rlm(fwd_op(self._to_alm(blk[0].contiguous()))).unsqueeze(0)
            tmat[:, i0:i0 + n_] = cols.t()
            i0 += n_
        eigv, eigw = np.linalg.eigh(dev.to_host(tmat))
        assert np.all(eigv[ntmpl:] > 0.)
        eigv_inv = np.zeros_like(eigv)
        eigv_inv[ntmpl:] = 1.0 / eigv[ntmpl:]
        if ntmpl > 0:  # the ntmpl lowest eigenmodes (marginalised templates) are left untouched
            eigv_inv[0:ntmpl] = 1.0
        minv = np.dot(np.dot(eigw, np.diag(eigv_inv)), np.transpose(eigw))
        if cache_fname is not None:
            pk.dump([lmax, self.hashdict(lmax, fwd_op), minv], open(cache_fname, 'wb'))
        return minv

    @staticmethod
    def hashdict(lmax, fwd_op):
        return {'lmax': lmax, 'fwd_op': fwd_op.hashdict()}

    def __call__(self, talm):
        return self.calc(talm)

    def split_apply(self, talm, lsplit, pre_op_hgh, dot=None):
        """pre_op_split's whole result for a device vector of a higher band-limit -- truncation to lsplit, this dense block, the diagonal
        preconditioner of pre_op_hgh above lsplit and the splice in one launch (pl_gemv_split) -- or None when that form does not apply
        (block vectors, host vectors, three fields, a high-l preconditioner that couples fields).
        dot = (q, lmin): returns (result, pre) with the partial sums of <result, q> formed by the same launch"""
        parts = _parts(talm)
        if len(parts) == 1 and hasattr(pre_op_hgh, 'filt'):
            fls = [pre_op_hgh.filt]
        elif len(parts) == 2 and hasattr(pre_op_hgh, 'flmat') and pre_op_hgh.flmat.shape[1:] == (2, 2) \
                and not (np.any(pre_op_hgh.flmat[:, 0, 1]) or np.any(pre_op_hgh.flmat[:, 1, 0])):
            fls = [pre_op_hgh.flmat[:, 0, 0], pre_op_hgh.flmat[:, 1, 1]]
        else:
            return None
        if lsplit != self.lmax or not all(isinstance(p, torch.Tensor) and p.is_cuda and p.dim() == 1 and p.dtype == torch.complex128 and p.is_contiguous()
                                          and p.shape == parts[0].shape for p in parts):
            return None
        if Alm.getlmax(parts[0].shape[0]) <= lsplit or any(len(f) <= lsplit for f in fls):
            return None
        if getattr(self, '_amat', None) is None:
            self._amat = _flat_matrix(self.minv, self.lmax, len(parts))
        if self._amat.shape[0] != len(parts) * (self.lmax + 1) * (self.lmax + 2):
            return None
        if dot is not None:
            res, pre = dev.gemv_split(self._amat, parts, lsplit, fls, dot=(_parts(dot[0]), dot[1]))
            return (res[0] if len(res) == 1 else eblm(res)), pre
        res = dev.gemv_split(self._amat, parts, lsplit, fls)
        return res[0] if len(res) == 1 else eblm(res)

    def calc(self, talm):
        if isinstance(talm, np.ndarray):
            return dev.to_host(self.calc(dev.to_dev(talm, torch.complex128)))  # host array in, host array out (temperature block)
        # device vectors: alm -> rlm, the pseudo-inverse and rlm -> alm as ONE matrix acting on the interleaved (re, im)
        # view of the alm arrays -- the preconditioner is applied dozens of times per top-level iteration at a
        # resolution where a kernel costs less to run than to launch
        parts = _parts(talm)
        if getattr(self, '_amat', None) is None:
            self._amat = _flat_matrix(self.minv, self.lmax, len(parts))
        if parts[0].dim() == 2:  # block vectors [nb, nalm]: one mat-mat product, the matrix read once (pl_gemv_b)
            nb = parts[0].shape[0]
            flat = [torch.view_as_real(p).reshape(nb, -1) for p in parts]
            out = dev.gemv(self._amat, flat[0] if len(flat) == 1 else torch.cat(flat, dim=1))
            n = flat[0].shape[1]
            res = [torch.view_as_complex(out[:, k * n:(k + 1) * n].contiguous().view(nb, -1, 2)) for k in range(len(parts))]
        else:
            flat = [torch.view_as_real(p).reshape(-1) for p in parts]
            out = dev.gemv(self._amat, flat[0] if len(flat) == 1 else torch.cat(flat))  # pl_gemv: one launch
            n = flat[0].numel()
            res = [torch.view_as_complex(out[k * n:(k + 1) * n].view(-1, 2)) for k in range(len(parts))]
        return res[0] if len(res) == 1 else (eblm(res) if len(res) == 2 else teblm(res))


class pre_op_dense_tt(_pre_op_dense):
    def _ntmpl(self, fwd_op):
        return int(sum(t.nmodes for t in fwd_op.n_inv_filt.templates))

    def _nrlm(self, lmax):
        return (lmax + 1) ** 2

    def _to_rlm(self, alm):
        return alm2rlm(alm)

    def _to_alm(self, rlm):
        return rlm2alm(rlm)


pre_op_dense_kk = pre_op_dense_tt


class pre_op_dense_pp(_pre_op_dense):
    def _ntmpl(self, fwd_op):
        ntmpl = 0
        for t in (getattr(fwd_op.n_inv_filt, 'templates_p', None) or []):
            ntmpl += t.nmodes
        return ntmpl + 8  # (1 mono + 3 dip) * (e + b): the l < 2 modes that do not exist in polarization

    def _nrlm(self, lmax):
        return 2 * (lmax + 1) ** 2

    def _to_rlm(self, alm):
        return torch.cat([alm2rlm(alm.elm), alm2rlm(alm.blm)], dim=-1)

    def _to_alm(self, rlm):
        n = rlm.shape[-1] // 2
        return eblm([rlm2alm(rlm[..., :n]), rlm2alm(rlm[..., n:])])


class pre_op_dense_tp(_pre_op_dense):
    """Joint (T, E, B) dense block (dense.py:204-285): rlm = [T | E | B], 3 (lmax + 1)^2 real coefficients."""

    def _ntmpl(self, fwd_op):
        ntmpl = 0
        for t in fwd_op.n_inv_filt.templates_t:
            ntmpl += t.nmodes  # includes monopole and dipole when marginalised
        for t in fwd_op.n_inv_filt.templates_p:
            ntmpl += t.nmodes
        return ntmpl + 8  # (1 mono + 3 dip) * (e + b)

    def _nrlm(self, lmax):
        return 3 * (lmax + 1) ** 2

    def _to_rlm(self, alm):
        return torch.cat([alm2rlm(alm.tlm), alm2rlm(alm.elm), alm2rlm(alm.blm)], dim=-1)

    def _to_alm(self, rlm):
        n = rlm.shape[-1] // 3
        return teblm([rlm2alm(rlm[..., :n]), rlm2alm(rlm[..., n:2 * n]), rlm2alm(rlm[..., 2 * n:])])
