r"""Spin-weight conventions and helpers, hot-path part of plancklens/utils_spin.py (:1-34, :96-156).

Conventions: :math:`_{\pm |s|} X_{lm} = - (\pm)^{|s|} (G_{lm} \pm i C_{lm})`; for CMB maps
:math:`_0X_{lm} = T_{lm}`, :math:`_{\pm 2}X_{lm} = -\tfrac 12 (E_{lm} \pm i B_{lm})`, hence :math:`G^0 = -T`,
:math:`G^2 = E`, :math:`C^2 = B`.  The Wigner-series products (`wignerc`, responses) are out of scope.
"""
import numpy as np

from . import shts


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
