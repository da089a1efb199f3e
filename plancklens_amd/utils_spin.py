r"""Spin-weight conventions and helpers, hot-path part of plancklens/utils_spin.py (:1-34, :96-156).

Conventions: :math:`_{\pm |s|} X_{lm} = - (\pm)^{|s|} (G_{lm} \pm i C_{lm})`; for CMB maps
:math:`_0X_{lm} = T_{lm}`, :math:`_{\pm 2}X_{lm} = -\tfrac 12 (E_{lm} \pm i B_{lm})`, hence :math:`G^0 = -T`,
:math:`G^2 = E`, :math:`C^2 = B`.  `wignerc` (utils_spin.py:52-93) runs on the numpy Wigner series of plancklens_amd.wigners
instead of the reference's Fortran extension.
"""
import numpy as np

from . import shts, wigners

HASWIGNER = True
_GL_cache = {}


def alm2map_spin(gclm, nside, spin, lmax, mmax=None):
    """alm2map_spin including spin 0 with its sign flip G^0 = -T (utils_spin.py:21-27)."""
    assert spin >= 0, spin
    assert len(gclm) == 2, len(gclm)
    if spin > 0:
        return shts.alm2map_spin(gclm, nside, spin, lmax, mmax=mmax)
    return shts.alm2map(-gclm[0], nside, lmax=lmax, mmax=mmax), 0.


def map2alm_spin(maps, spin, lmax=None, mmax=None):
    assert spin >= 0, spin
    if spin > 0:
        return shts.map2alm_spin(maps, spin, lmax=lmax, mmax=mmax)
    return -shts.map2alm(maps[0], lmax=lmax, mmax=mmax, iter=0), 0.


def get_spin_raise(s, lmax):
    r"""sqrt((l - s)(l + s + 1)) for |s| <= l <= lmax: response of :math:`_sY_{lm}` to the spin-raising operator."""
    ret = np.zeros(lmax + 1, dtype=float)
    ret[abs(s):] = np.sqrt(np.arange(abs(s) - s, lmax - s + 1) * np.arange(abs(s) + s + 1, lmax + s + 2))
    return ret


def get_spin_lower(s, lmax):
    r"""-sqrt((l + s)(l - s + 1)) for |s| <= l <= lmax: response to the spin-lowering operator."""
    ret = np.zeros(lmax + 1, dtype=float)
    ret[abs(s):] = -np.sqrt(np.arange(s + abs(s), lmax + s + 1) * np.arange(abs(s) - s + 1, lmax - s + 2))
    return ret


def _dict_transpose(cls):
    ret = {}
    for k in cls.keys():
        if len(k) == 1:
            ret[k + k] = np.copy(cls[k])
        else:
            assert len(k) == 2
            ret[k[1] + k[0]] = np.copy(cls[k])
    return ret


def spin_cls(s1, s2, cls):
    r"""Spin-weighted spectrum :math:`\langle _{s1}X_{lm}\, _{s2}X^*_{lm}\rangle` from the T, E, B spectra."""
    if s1 < 0:
        return (-1) ** (s1 + s2) * np.conjugate(spin_cls(-s1, -s2, _dict_transpose(cls)))
    assert s1 in [0, -2, 2] and s2 in [0, -2, 2], (s1, s2, 'not implemented')
    if s1 == 0:
        if s2 == 0:
            return cls['tt']
        tb = cls.get('tb', None)
        te = cls.get('te', cls.get('et'))
        assert te is not None
        return -te if tb is None else -te + 1j * np.sign(s2) * tb
    if s2 == 0:
        tb = cls.get('bt', cls.get('tb', None))
        et = cls.get('et', cls.get('te'))
        assert et is not None
        return -et if tb is None else -et - 1j * tb
    if s2 == 2:
        return cls['ee'] + cls['bb']
    eb = cls.get('be', cls.get('eb', None))
    return cls['ee'] - cls['bb'] if eb is None else cls['ee'] - cls['bb'] + 2j * eb


def wignerc(cl1, cl2, sp1, s1, sp2, s2, lmax_out=None):
    r"""Legendre coefficients of :math:`\xi_{sp1,s1}(\cos\theta)\, \xi_{sp2,s2}(\cos\theta)` given the harmonic series of the
    two factors: an exact Gauss-Legendre quadrature with (lmax1 + lmax2 + lmax_out) / 2 + 1 nodes."""
    lmax1, lmax2 = len(cl1) - 1, len(cl2) - 1
    lmax_out = lmax1 + lmax2 if lmax_out is None else lmax_out
    if not (np.any(cl1) and np.any(cl2)):
        return np.zeros(lmax_out + 1, dtype=float)
    lmaxtot = lmax1 + lmax2 + lmax_out
    npts = (lmaxtot + 2 - lmaxtot % 2) // 2
    if npts not in _GL_cache:
        if len(_GL_cache) > 8:
            _GL_cache.clear()
        _GL_cache[npts] = wigners.get_xgwg(-1., 1., npts)
    xg, wg = _GL_cache[npts]

    def pos(cl, a, b):
        if np.iscomplexobj(cl):
            return wigners.wignerpos(np.real(cl), xg, a, b) + 1j * wigners.wignerpos(np.imag(cl), xg, a, b)
        return wigners.wignerpos(cl, xg, a, b)

    prod = pos(cl1, sp1, s1) * pos(cl2, sp2, s2) * wg
    spo, so = sp1 + sp2, s1 + s2
    if np.iscomplexobj(prod):
        return wigners.wignercoeff(np.real(prod), xg, spo, so, lmax_out) + 1j * wigners.wignercoeff(np.imag(prod), xg, spo, so, lmax_out)
    return wigners.wignercoeff(prod, xg, spo, so, lmax_out)


def get_spin_matrix(sout, sin, cls):
    r"""Spin-space matrix element :math:`(R^{-1}\, {\rm cls}[T, E, B]\, R)_{sout, sin}`, R mapping the spin 0, +-2 fields to
    T, E, B (utils_spin.py:160-198).  `cls` has keys 'tt', 'te', 'ee', 'bb' (+ 'tb', 'eb'); missing spectra are zero."""
    assert sin in [0, 2, -2] and sout in [0, 2, -2], (sin, sout)
    tt = cls.get('tt', cls.get('t', 0.))
    ee, bb = cls.get('ee', cls.get('e', 0.)), cls.get('bb', cls.get('b', 0.))
    te, tb, eb = cls.get('te', 0.), cls.get('tb', None), cls.get('eb', None)
    if sin == 0:
        if sout == 0:
            return tt
        return -te if tb is None else -te - 1j * np.sign(sout) * tb
    if sout == 0:
        return -0.5 * te if tb is None else -0.5 * (te - 1j * np.sign(sin) * tb)
    if sout == sin:
        return 0.5 * (ee + bb)
    ret = 0.5 * (ee - bb)
    return ret if eb is None else ret + 1j * np.sign(sout) * eb  # (sout, sin) = (2, -2): + i EB; (-2, 2): - i EB
