"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the four HEALPix SHTs behind plancklens/shts.py:12-35.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The
product path (plancklens_amd.shts -> HIP kernels through the C-ABI) never does.

Parity status: *unpinned by the reference's own tests* (the reference has no SHT test);
the oracle is pinned instead by (i) closed-form known answers, (ii) the reference's own Fortran Wigner
module built in oracle/_ref (spin-weighted Legendre functions), (iii) adjointness, (iv) the two
independent QE routes of the reference (tests/golden/make_golden.py), see tests/test_oracle.py.

Legendre stage: oracle/sht_oracle.c (C; mode 0 long double, mode 1 scaled double + OpenMP).
Fourier stage : numpy (pocketfft), one (i)FFT per iso-latitude ring, aliasing handled explicitly.
healpy semantics restated from SURVEY.md Appendix A.1-A.4:
  alm2map(alm, nside, lmax)                 T(p)   = sum_lm a_lm Y_lm(p)
  map2alm(m, lmax, iter=0)                  a_lm   = 4pi/npix sum_p T_p Y*_lm(p)   (uniform weights)
  alm2map_spin([G, C], nside, s, lmax)      Q + iU = sum_lm -(G + iC) _sY_lm
  map2alm_spin([Q, U], s, lmax)             the 4pi/npix-weighted adjoint
"""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _cpu_tag():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('flags'):
                    import hashlib
                    return hashlib.sha1(line.encode()).hexdigest()[:12]
    except OSError:
        pass
    return 'unknown'


def build(force=False):
    """Compile oracle/libshtoracle.so (and oracle/_ref when /root/reference is present).  The library is
    built with -march=native, so it is rebuilt when the host CPU differs from the one it was built on."""
    so = os.path.join(_HERE, 'libshtoracle.so')
    src = os.path.join(_HERE, 'sht_oracle.c')
    tagf = so + '.cpu'
    tag = _cpu_tag()
    old = open(tagf).read().strip() if os.path.exists(tagf) else ''
    if force or (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src) or old != tag:
        subprocess.check_call(['make', '-B', '-C', _HERE, 'libshtoracle.so'], stdout=subprocess.DEVNULL)
        with open(tagf, 'w') as f:
            f.write(tag)
    if os.path.exists('/root/reference/plancklens/wigners/wigners.f90') and \
            not os.path.exists(os.path.join(_HERE, '_ref', 'libwigners_ref.so')):
        subprocess.call(['make', '-C', _HERE, 'ref'], stdout=subprocess.DEVNULL)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        so = build()
        _LIB = ctypes.CDLL(so)
        dp = ctypes.POINTER(ctypes.c_double)
        ip = ctypes.POINTER(ctypes.c_int)
        _LIB.orc_legendre.argtypes = [ctypes.c_int] * 6 + [dp, dp, ip, dp, dp, ctypes.c_int]
        _LIB.orc_legendre.restype = ctypes.c_int
        _LIB.orc_lambda.argtypes = [ctypes.c_int] * 3 + [ctypes.c_double] * 2 + [dp, dp]
        _LIB.orc_lambda.restype = None
        lp = ctypes.POINTER(ctypes.c_int64)
        _LIB.orc_ring_fft.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, lp, lp, dp, lp, dp, dp, ctypes.c_int]
        _LIB.orc_ring_fft.restype = ctypes.c_int
    return _LIB


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def alm_size(lmax):
    return (lmax + 1) * (lmax + 2) // 2


def alm_lmax(size):
    lmax = int(np.floor(np.sqrt(2 * size) - 1))
    assert alm_size(lmax) == size, size
    return lmax


def ring_geometry(nside):
    """HEALPix RING geometry, rings i = 1 .. 4 nside - 1."""
    i = np.arange(1, 4 * nside, dtype=np.int64)
    north = np.minimum(i, 4 * nside - i)
    cap = north < nside
    nphi = np.where(cap, 4 * north, 4 * nside)
    omz = np.where(cap, north.astype(float) ** 2 / (3. * nside ** 2), 1. - (4. / 3. - 2. * north / (3. * nside)))
    z = 1. - omz
    sth = np.sqrt(omz * (1. + z))
    cth = np.where(i > 2 * nside, -z, z)
    shifted = np.where(cap, True, ((north - nside) % 2) == 0)
    phi0 = np.where(shifted, np.pi / nphi, 0.)
    ofs = np.concatenate([[0], np.cumsum(nphi)[:-1]])
    return cth, sth, nphi, phi0, ofs


def lambda_lm(spin, m, lmax, cth, sth):
    """Fp_l, Fm_l (spin > 0) or (lambda_lm, 0) for spin 0, long-double route."""
    fp = np.zeros(lmax + 1)
    fm = np.zeros(lmax + 1)
    _lib().orc_lambda(spin, m, lmax, float(cth), float(sth), _dp(fp), _dp(fm))
    return fp, fm


def legendre(direction, mode, spin, lmax, mmax, cth, sth, pair, alm=None, phase=None, nthreads=0, out=None):
    """out: a preallocated result array (direction 0: complex128 (ncomp, 2 nring, mmax + 1); direction 1: complex128 (ncomp, nalm)),
    overwritten -- the C stage zeroes it first.  Saves the page faults of a fresh 100 MB+ array per call (bench.py's cpu_baseline)."""
    ncomp = 1 if spin == 0 else 2
    nring = len(cth)
    nalm = alm_size(lmax) if mmax == lmax else mmax * (2 * lmax + 1 - mmax) // 2 + lmax + 1
    cth = np.ascontiguousarray(cth, dtype=np.float64)
    sth = np.ascontiguousarray(sth, dtype=np.float64)
    pair = np.ascontiguousarray(pair, dtype=np.int32)
    if direction == 0:
        alm = np.ascontiguousarray(alm, dtype=np.complex128).reshape(ncomp, nalm)
        phase = np.zeros((ncomp, 2 * nring, mmax + 1), dtype=np.complex128) if out is None else out
        assert phase.shape == (ncomp, 2 * nring, mmax + 1) and phase.dtype == np.complex128 and phase.flags.c_contiguous
    else:
        phase = np.ascontiguousarray(phase, dtype=np.complex128).reshape(ncomp, 2 * nring, mmax + 1)
        alm = np.zeros((ncomp, nalm), dtype=np.complex128) if out is None else out
        assert alm.shape == (ncomp, nalm) and alm.dtype == np.complex128 and alm.flags.c_contiguous
    _lib().orc_legendre(direction, mode, spin, lmax, mmax, nring, _dp(cth), _dp(sth),
                        pair.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                        _dp(alm.view(np.float64)), _dp(phase.view(np.float64)), nthreads)
    return phase if direction == 0 else alm


def _pair_geometry(nside, use_pairs=True):
    """Ring list handed to the C stage and the map from (slot) to ring index."""
    cth, sth, nphi, phi0, ofs = ring_geometry(nside)
    nr = 4 * nside - 1
    if use_pairs:
        north = np.arange(2 * nside)  # ring index 0 .. 2 nside - 1 (equator last)
        pair = np.ones(2 * nside, dtype=np.int32)
        pair[-1] = 0
        slots = np.full(2 * north.size, -1, dtype=np.int64)
        slots[0::2] = north
        slots[1::2] = np.where(pair == 1, nr - 1 - north, -1)
        return cth[north], sth[north], pair, slots
    idx = np.arange(nr)
    slots = np.full(2 * nr, -1, dtype=np.int64)
    slots[0::2] = idx
    return cth, sth, np.zeros(nr, dtype=np.int32), slots


def _phase2map(phase, nside, mmax, slots, out=None):
    """phase[slot, m] -> RING map (one component).  Only the rings listed in `slots` are written (into `out` when given:
    callers that transform ring subsets share one map)."""
    cth, sth, nphi, phi0, ofs = ring_geometry(nside)
    npix = 12 * nside ** 2
    if out is None:
        out = np.empty(npix)
    ms = np.arange(mmax + 1)
    ring_of_slot = slots
    ok = ring_of_slot >= 0
    # group rings by nphi for batched FFTs
    sl = np.nonzero(ok)[0]
    rings = ring_of_slot[sl]
    order = np.argsort(nphi[rings], kind='stable')
    sl, rings = sl[order], rings[order]
    n_sorted = nphi[rings]
    bounds = np.concatenate([[0], np.nonzero(np.diff(n_sorted))[0] + 1, [len(rings)]])
    for b0, b1 in zip(bounds[:-1], bounds[1:]):
        n = int(n_sorted[b0])
        rr = rings[b0:b1]
        f = phase[sl[b0:b1], :] * np.exp(1j * np.outer(phi0[rr], ms))
        Z = np.zeros((b1 - b0, n), dtype=np.complex128)
        k = ms % n
        kneg = (-ms) % n
        for m0 in range(0, mmax + 1, n):  # every chunk of n consecutive m hits distinct bins
            m1 = min(m0 + n, mmax + 1)
            Z[:, k[m0:m1]] += f[:, m0:m1]
            mm0 = max(m0, 1)
            if mm0 < m1:
                Z[:, kneg[mm0:m1]] += np.conj(f[:, mm0:m1])
        x = np.fft.ifft(Z, axis=1).real * n
        for q, r in enumerate(rr):
            out[ofs[r]:ofs[r] + n] = x[q]
    return out


def _map2phase(m, nside, mmax, slots):
    cth, sth, nphi, phi0, ofs = ring_geometry(nside)
    npix = 12 * nside ** 2
    assert m.size == npix
    phase = np.zeros((slots.size, mmax + 1), dtype=np.complex128)
    ms = np.arange(mmax + 1)
    sl = np.nonzero(slots >= 0)[0]
    rings = slots[sl]
    order = np.argsort(nphi[rings], kind='stable')
    sl, rings = sl[order], rings[order]
    n_sorted = nphi[rings]
    bounds = np.concatenate([[0], np.nonzero(np.diff(n_sorted))[0] + 1, [len(rings)]])
    w = 4. * np.pi / npix
    for b0, b1 in zip(bounds[:-1], bounds[1:]):
        n = int(n_sorted[b0])
        rr = rings[b0:b1]
        x = np.stack([m[ofs[r]:ofs[r] + n] for r in rr])
        Z = np.fft.fft(x, axis=1)
        phase[sl[b0:b1], :] = Z[:, ms % n] * np.exp(-1j * np.outer(phi0[rr], ms)) * w
    return phase


def ring_fft_c(direction, nside, mmax, slots, phase=None, m=None, out=None, nthreads=0):
    """The Fourier stage in C, threaded over rings (oracle/sht_oracle.c orc_ring_fft: radix-2 / Bluestein FFTs written out
    in full): same contract as _phase2map (direction 0; returns / fills the map `out`) and _map2phase (direction 1; returns
    phase[slot, m], written into `out` when given).  bench.py's cpu_baseline uses it; tests/test_oracle.py checks it against the numpy route."""
    cth, sth, nphi, phi0, ofs = ring_geometry(nside)
    npix = 12 * nside ** 2
    slots = np.ascontiguousarray(slots, dtype=np.int64)
    nphi = np.ascontiguousarray(nphi, dtype=np.int64)
    ofs = np.ascontiguousarray(ofs, dtype=np.int64)
    phi0 = np.ascontiguousarray(phi0, dtype=np.float64)
    lp = ctypes.POINTER(ctypes.c_int64)
    if direction == 0:
        ph = np.ascontiguousarray(phase, dtype=np.complex128)
        assert ph.shape == (slots.size, mmax + 1)
        if out is None:
            out = np.empty(npix)
        assert out.dtype == np.float64 and out.flags.c_contiguous and out.size == npix
        mp = out
    else:
        mp = np.ascontiguousarray(m, dtype=np.float64)
        assert mp.size == npix
        # out (direction 1): a preallocated phase array; the rows of unused slots (ring -1) are not written -- the caller zeroed them once
        ph = np.zeros((slots.size, mmax + 1), dtype=np.complex128) if out is None else out
        assert ph.shape == (slots.size, mmax + 1) and ph.dtype == np.complex128 and ph.flags.c_contiguous
    _lib().orc_ring_fft(direction, npix, mmax, slots.size, slots.ctypes.data_as(lp), nphi.ctypes.data_as(lp), _dp(phi0),
                        ofs.ctypes.data_as(lp), _dp(ph.view(np.float64)), _dp(mp), nthreads)
    return mp if direction == 0 else ph


def alm2map(alm, nside, lmax=None, mmax=None, mode=1, use_pairs=True, nthreads=0, **kwargs):
    alm = np.asarray(alm, dtype=np.complex128)
    if lmax is None:
        lmax = alm_lmax(alm.size)
    assert mmax is None or mmax == lmax
    c, s, pair, slots = _pair_geometry(nside, use_pairs)
    ph = legendre(0, mode, 0, lmax, lmax, c, s, pair, alm=alm, nthreads=nthreads)
    return _phase2map(ph[0], nside, lmax, slots)


def map2alm(m, lmax=None, mmax=None, iter=0, mode=1, use_pairs=True, nthreads=0, **kwargs):
    m = np.asarray(m, dtype=np.float64)
    nside = int(np.round(np.sqrt(m.size / 12)))
    if lmax is None:
        lmax = 3 * nside - 1
    assert iter == 0, 'every reference call passes iter=0 (SURVEY.md Appendix A.2)'
    assert mmax is None or mmax == lmax
    c, s, pair, slots = _pair_geometry(nside, use_pairs)
    ph = _map2phase(m, nside, lmax, slots)
    return legendre(1, mode, 0, lmax, lmax, c, s, pair, phase=ph[None], nthreads=nthreads)[0]


def alm2map_spin(gclm, nside, spin, lmax, mmax=None, mode=1, use_pairs=True, nthreads=0):
    assert spin > 0 and len(gclm) == 2
    assert mmax is None or mmax == lmax
    alm = np.stack([np.asarray(gclm[0], dtype=np.complex128), np.asarray(gclm[1], dtype=np.complex128)])
    assert alm.shape[1] == alm_size(lmax)
    c, s, pair, slots = _pair_geometry(nside, use_pairs)
    ph = legendre(0, mode, spin, lmax, lmax, c, s, pair, alm=alm, nthreads=nthreads)
    return [_phase2map(ph[0], nside, lmax, slots), _phase2map(ph[1], nside, lmax, slots)]


def map2alm_spin(maps, spin, lmax=None, mmax=None, mode=1, use_pairs=True, nthreads=0):
    assert spin > 0 and len(maps) == 2
    q = np.asarray(maps[0], dtype=np.float64)
    u = np.asarray(maps[1], dtype=np.float64)
    nside = int(np.round(np.sqrt(q.size / 12)))
    if lmax is None:
        lmax = 3 * nside - 1
    assert mmax is None or mmax == lmax
    c, s, pair, slots = _pair_geometry(nside, use_pairs)
    ph = np.stack([_map2phase(q, nside, lmax, slots), _map2phase(u, nside, lmax, slots)])
    alm = legendre(1, mode, spin, lmax, lmax, c, s, pair, phase=ph, nthreads=nthreads)
    return [alm[0], alm[1]]


def brute_alm2map_spin(gclm, nside, spin, lmax):
    """O(npix lmax^2) pixel-by-pixel evaluation straight from the definition (tiny cases only)."""
    cth, sth, nphi, phi0, ofs = ring_geometry(nside)
    npix = 12 * nside ** 2
    out = np.zeros(npix, dtype=np.complex128)
    g = np.asarray(gclm[0], dtype=np.complex128)
    c = np.asarray(gclm[1], dtype=np.complex128) if spin > 0 else None
    sg = (-1.) ** spin
    for r in range(4 * nside - 1):
        n = int(nphi[r])
        ph = phi0[r] + 2 * np.pi * np.arange(n) / n
        acc = np.zeros(n, dtype=np.complex128)
        for m in range(lmax + 1):
            fp, fm = lambda_lm(spin, m, lmax, cth[r], sth[r])
            i0 = m * (2 * lmax + 1 - m) // 2
            ls = np.arange(m, lmax + 1)
            if spin == 0:
                fm_ = np.sum(g[i0 + ls] * fp[ls])
                acc += (fm_ * np.exp(1j * m * ph)).real * (1. if m == 0 else 2.)
            else:
                # Q_m = sum G Fp + i C Fm ; U_m = sum C Fp - i G Fm ; real fields: X(phi) = sum_m w_m Re(X_m e^{i m phi})
                qm = np.sum(g[i0 + ls] * fp[ls] + 1j * c[i0 + ls] * fm[ls])
                um = np.sum(c[i0 + ls] * fp[ls] - 1j * g[i0 + ls] * fm[ls])
                w = 1. if m == 0 else 2.
                acc += w * ((qm * np.exp(1j * m * ph)).real + 1j * (um * np.exp(1j * m * ph)).real)
        out[ofs[r]:ofs[r] + n] = acc
    return (out.real, out.imag) if spin > 0 else out.real
