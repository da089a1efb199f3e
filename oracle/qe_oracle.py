"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's filter -> quadratic-estimator chain on top
of the CPU oracle SHTs (oracle/sht_oracle.py).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.

Follows, line by line:
  isotropic filter          plancklens/filt/filt_simple.py:397-407 (Xb = F_l / b_l map2alm(map))
  Wiener-filtered legs      plancklens/filt/filt_simple.py:149-183, qest.py:582-589, 613-618
  T-only estimator          plancklens/qest.py:248-263, 566-595
  polarization estimator    plancklens/qest.py:265-285, 521-530, 597-638
  MV estimator              plancklens/qest.py:318-322
  stt / ftt / f_p / a_p     plancklens/qest.py:287-316

Pinned against the reference itself by tests/golden/make_golden.py (the reference's Python run over this
oracle's SHTs through a `healpy` stand-in) -> tests/golden/qe_golden.npz.
"""
import numpy as np

from . import sht_oracle as so


def almxfl(alm, fl):
    lmax = so.alm_lmax(alm.size)
    f = np.zeros(lmax + 1)
    n = min(lmax + 1, len(fl))
    f[:n] = np.asarray(fl)[:n]
    ls = np.concatenate([np.arange(m, lmax + 1) for m in range(lmax + 1)])
    return alm * f[ls]


def cli(cl):
    ret = np.zeros_like(cl)
    ret[cl > 0] = 1. / cl[cl > 0]
    return ret


def filter_maps(tmap, qmap, umap, lmax, ftl, fel, fbl, transf, mode=1):
    """filt_simple.py:397-407"""
    tlm = almxfl(so.map2alm(tmap, lmax=lmax, mode=mode), ftl * cli(transf[:len(ftl)]))
    elm, blm = so.map2alm_spin([qmap, umap], 2, lmax, mode=mode)
    return tlm, almxfl(elm, fel * cli(transf[:len(fel)])), almxfl(blm, fbl * cli(transf[:len(fbl)]))


def _lensw(lmax):
    ell = np.arange(lmax + 1, dtype=float)
    return -np.sqrt(ell * (ell + 1.))


def _spinw(spin, lmax):
    if spin == 1:
        fl = np.arange(2, lmax + 3, dtype=float) * np.arange(-1, lmax)
    else:
        fl = np.arange(-2, lmax - 1, dtype=float) * np.arange(3, lmax + 4)
    fl[:spin] = 0.
    return np.sqrt(fl)


def qe_T(tlm1, twf2, nside, lmax_qlm, mode=1):
    """qest.py:248-263: legs Tb (leg 1) and T^WF (leg 2)."""
    lmax = so.alm_lmax(tlm1.size)
    tmap = so.alm2map(tlm1, nside, lmax=lmax, mode=mode)
    glm = almxfl(twf2, _lensw(lmax))
    G, C = so.alm2map_spin([glm, np.zeros_like(glm)], nside, 1, lmax, mode=mode)
    G, C = so.map2alm_spin([G * tmap, C * tmap], 1, lmax_qlm, mode=mode)
    return almxfl(G, _lensw(lmax_qlm)), almxfl(C, _lensw(lmax_qlm))


def qe_P(elm1, blm1, ewf2, bwf2, nside, lmax_qlm, mode=1):
    """qest.py:265-285: legs (Eb, Bb) (leg 1) and (E^WF, B^WF) (leg 2)."""
    lmax = so.alm_lmax(elm1.size)
    re, im = so.alm2map_spin([0.5 * elm1, 0.5 * blm1], nside, 2, lmax, mode=mode)
    Gs, Cs = so.alm2map_spin([almxfl(ewf2, _spinw(3, lmax)), almxfl(bwf2, _spinw(3, lmax))], nside, 3, lmax, mode=mode)
    GC = (re - 1j * im) * (Gs + 1j * Cs)
    Gs, Cs = so.alm2map_spin([almxfl(ewf2, _spinw(1, lmax)), almxfl(bwf2, _spinw(1, lmax))], nside, 1, lmax, mode=mode)
    GC -= (re + 1j * im) * (Gs - 1j * Cs)
    G, C = so.map2alm_spin([GC.real, GC.imag], 1, lmax_qlm, mode=mode)
    return almxfl(G, _lensw(lmax_qlm)), almxfl(C, _lensw(lmax_qlm))


def qe_sepTP(key, alms1, alms2, cls, nside, lmax_qlm, mode=1):
    """Gradient and curl of 'ptt', 'p_p' or 'p' for separately filtered T and P (qest.library_sepTP).
    alms = (tlm, elm, blm) inverse-variance filtered; cls has 'tt', 'ee', 'bb', 'te'."""
    t1, e1, b1 = alms1
    t2, e2, b2 = alms2
    if key == 'ptt':
        return qe_T(t1, almxfl(t2, cls['tt']), nside, lmax_qlm, mode)
    if key == 'p_p':
        return qe_P(e1, b1, almxfl(e2, cls['ee']), almxfl(b2, cls['bb']), nside, lmax_qlm, mode)
    if key == 'p':
        GP, CP = qe_P(e1, b1, almxfl(e2, cls['ee']) + almxfl(t2, cls['te']), almxfl(b2, cls['bb']), nside, lmax_qlm, mode)
        GT, CT = qe_T(t1, almxfl(t2, cls['tt']) + almxfl(e2, cls['te']), nside, lmax_qlm, mode)
        return GP + GT, CP + CT
    raise ValueError(key)


def qe_scalar(key, alms1, alms2, cls, nside, lmax_qlm, mode=1):
    """'stt', 'ftt', 'f_p', 'a_p' (qest.py:287-316), same-leg (non swapped) form."""
    t1, e1, b1 = alms1
    t2, e2, b2 = alms2
    lmax = so.alm_lmax(t1.size)
    if key == 'stt':
        m = so.alm2map(t1, nside, lmax=lmax, mode=mode) * so.alm2map(t2, nside, lmax=lmax, mode=mode)
        return -0.5 * so.map2alm(m, lmax=lmax_qlm, mode=mode)
    if key == 'ftt':
        m = so.alm2map(t1, nside, lmax=lmax, mode=mode) * so.alm2map(almxfl(t2, cls['tt']), nside, lmax=lmax, mode=mode)
        return -so.map2alm(m, lmax=lmax_qlm, mode=mode)
    Q1, U1 = so.alm2map_spin([0.5 * e1, 0.5 * b1], nside, 2, lmax, mode=mode)
    Q2, U2 = so.alm2map_spin([almxfl(e2, cls['ee']), almxfl(b2, cls['bb'])], nside, 2, lmax, mode=mode)
    if key == 'f_p':
        return -2 * so.map2alm(Q1 * Q2 + U1 * U2, lmax=lmax_qlm, mode=mode)
    if key == 'a_p':
        return -4. * so.map2alm(Q1 * U2 - U1 * Q2, lmax=lmax_qlm, mode=mode)
    raise ValueError(key)
