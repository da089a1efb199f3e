"""alm containers of the CG solver, API of plancklens/qcinv/util_alm.py (`alm_splice` :8-24, `alm_copy` :27-44,
`eblm` :47-86, `teblm` :88-142).  The arrays may live on the host (numpy) or in HBM (torch CUDA tensors): the
per-m slice copies of the reference become one gather with a cached index map."""
import numpy as np

from ..hp import Alm

try:
    import torch
except ImportError:  # pragma: no cover
    torch = None

_IDX = {}


def _is_dev(a):
    return torch is not None and isinstance(a, torch.Tensor)


def _size(a):
    """entries of one alm array; a 2-D device tensor [nb, nalm] is a block of nb arrays (several right-hand sides of one solve)"""
    return a.shape[-1] if _is_dev(a) else len(a)


def _low_index_maps(lmax_a, lmax_b, lcut):
    """Positions, inside alm arrays of band-limits lmax_a and lmax_b, of all (l <= lcut, m <= l) entries."""
    key = (lmax_a, lmax_b, lcut)
    if key not in _IDX:
        ia = np.concatenate([m * (2 * lmax_a + 1 - m) // 2 + np.arange(m, lcut + 1) for m in range(lcut + 1)])
        ib = np.concatenate([m * (2 * lmax_b + 1 - m) // 2 + np.arange(m, lcut + 1) for m in range(lcut + 1)])
        _IDX[key] = [ia, ib, {}]
    return _IDX[key]


def _maps_for(a, lmax_a, lmax_b, lcut):
    ia, ib, devs = _low_index_maps(lmax_a, lmax_b, lcut)
    if not _is_dev(a):
        return ia, ib
    d = a.device
    if d not in devs:
        devs[d] = (torch.from_numpy(ia).to(d), torch.from_numpy(ib).to(d))
    return devs[d]


def alm_splice(alm_lo, alm_hi, lsplit):
    """alm with the band-limit of alm_hi: alm_lo for l <= lsplit, alm_hi above."""
    if hasattr(alm_lo, 'alm_splice'):
        return alm_lo.alm_splice(alm_hi, lsplit)
    lmax_lo, lmax_hi = Alm.getlmax(_size(alm_lo)), Alm.getlmax(_size(alm_hi))
    assert lmax_lo >= lsplit and lmax_hi >= lsplit
    if _is_dev(alm_hi) and _is_dev(alm_lo) and alm_hi.is_cuda and alm_hi.dtype == torch.complex128 and alm_lo.dtype == torch.complex128:
        from .. import dev
        return dev.alm_splice(alm_lo, alm_hi, lsplit)  # one kernel (pl_alm_splice)
    ilo, ihi = _maps_for(alm_hi, lmax_lo, lmax_hi, lsplit)
    ret = alm_hi.clone() if _is_dev(alm_hi) else np.copy(alm_hi)
    ret[ihi] = alm_lo[ilo]
    return ret


def alm_copy(alm, lmax=None):
    """Copy of alm, optionally truncated to a smaller band-limit."""
    if hasattr(alm, 'alm_copy'):
        return alm.alm_copy(lmax=lmax)
    lmox = Alm.getlmax(_size(alm))
    assert lmax is None or lmax <= lmox
    if lmax is None or lmax == lmox:
        return alm.clone() if _is_dev(alm) else np.copy(alm)
    if _is_dev(alm) and alm.is_cuda and alm.dtype == torch.complex128:
        from .. import dev
        return dev.alm_copy(alm, lmax)  # one kernel (pl_alm_copy)
    iin, iout = _maps_for(alm, lmox, lmax, lmax)
    if _is_dev(alm):
        assert alm.dim() == 1, 'blocks of alm arrays live on the GPU as complex128'
        ret = torch.zeros(Alm.getsize(lmax), dtype=alm.dtype, device=alm.device)
    else:
        ret = np.zeros(Alm.getsize(lmax), dtype=complex)
    ret[iout] = alm[iin]
    return ret


class eblm(object):
    """(E, B) pair with the vector-space operations cd_solve needs."""

    def __init__(self, alm):
        elm, blm = alm
        assert _size(elm) == _size(blm), (_size(elm), _size(blm))
        self.lmax = Alm.getlmax(_size(elm))
        self.elm = elm
        self.blm = blm

    def alm_copy(self, lmax=None):
        return eblm([alm_copy(self.elm, lmax=lmax), alm_copy(self.blm, lmax=lmax)])

    def alm_splice(self, alm_hi, lsplit):
        return eblm([alm_splice(self.elm, alm_hi.elm, lsplit), alm_splice(self.blm, alm_hi.blm, lsplit)])

    def __add__(self, other):
        assert self.lmax == other.lmax
        return eblm([self.elm + other.elm, self.blm + other.blm])

    def __sub__(self, other):
        assert self.lmax == other.lmax
        return eblm([self.elm - other.elm, self.blm - other.blm])

    def __iadd__(self, other):
        assert self.lmax == other.lmax
        self.elm += other.elm
        self.blm += other.blm
        return self

    def __isub__(self, other):
        assert self.lmax == other.lmax
        self.elm -= other.elm
        self.blm -= other.blm
        return self

    def __mul__(self, other):
        return eblm([self.elm * other, self.blm * other])


class teblm(object):
    """(T, E, B) triplet, possibly with different band-limits per field."""

    def __init__(self, alm):
        tlm, elm, blm = alm
        self.lmaxt, self.lmaxe, self.lmaxb = (Alm.getlmax(_size(a)) for a in (tlm, elm, blm))
        self.lmax = max(self.lmaxt, self.lmaxe, self.lmaxb)
        self.tlm, self.elm, self.blm = tlm, elm, blm

    def _same(self, other):
        assert (self.lmaxt, self.lmaxe, self.lmaxb) == (other.lmaxt, other.lmaxe, other.lmaxb)

    def alm_copy(self, lmax=None):
        return teblm([alm_copy(a, lmax=lmax) for a in (self.tlm, self.elm, self.blm)])

    def alm_splice(self, alm_hi, lsplit):
        return teblm([alm_splice(self.tlm, alm_hi.tlm, lsplit), alm_splice(self.elm, alm_hi.elm, lsplit),
                      alm_splice(self.blm, alm_hi.blm, lsplit)])

    def __add__(self, other):
        self._same(other)
        return teblm([self.tlm + other.tlm, self.elm + other.elm, self.blm + other.blm])

    def __sub__(self, other):
        self._same(other)
        return teblm([self.tlm - other.tlm, self.elm - other.elm, self.blm - other.blm])

    def __iadd__(self, other):
        self._same(other)
        self.tlm += other.tlm
        self.elm += other.elm
        self.blm += other.blm
        return self

    def __isub__(self, other):
        self._same(other)
        self.tlm -= other.tlm
        self.elm -= other.elm
        self.blm -= other.blm
        return self

    def __mul__(self, other):
        return teblm([self.tlm * other, self.elm * other, self.blm * other])
