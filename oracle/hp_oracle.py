"""TEST INFRASTRUCTURE ONLY -- the healpy host helpers the golden generator needs, restated independently of the product.

tests/golden/make_golden.py runs the REFERENCE's Python over a module named `healpy`.  Its transforms come from oracle/sht_oracle.py;
the index / l-filter / spectrum / degrade helpers used to be borrowed from the product (plancklens_amd/hp.py), so that a slip there
would have been in the fixtures and in the product alike.  They are restated here from the public healpy / HEALPix definitions
 with different algorithms where one exists (explicit per-m loops instead of an index table, the degrade by
pixel centres instead of NESTED bit arithmetic); tests/test_oracle.py requires the two sets to agree on random inputs (bit for bit;
ud_grade and the pixel angles, whose sums / inverse functions differ, to a few ulp).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Reference call sites these stand in for: hp.almxfl (147 sites, e.g. plancklens/qest.py:261-262), hp.alm2cl (qecl.py:147-148),
hp.Alm.getlmax / getsize / getidx (utils.py:26-34), hp.gauss_beam (params/idealized_example.py:49), hp.ud_grade(power=-2)
(qcinv/opfilt_tt.py:113-118, opfilt_pp.py:227-236), hp.nside2npix / npix2nside / nside2pixarea (sims/maps.py:146-147).
"""
import numpy as np

UNSEEN = -1.6375e30


class Alm(object):
    """healpy.Alm for mmax = lmax: entry (l, m) sits at m (2 lmax + 1 - m) / 2 + l"""

    @staticmethod
    def getsize(lmax, mmax=None):
        assert mmax is None or mmax < 0 or mmax == lmax, 'mmax = lmax only'
        return (lmax + 1) * (lmax + 2) // 2

    @staticmethod
    def getlmax(s, mmax=None):
        assert mmax is None or mmax < 0, 'mmax = lmax only'
        lmax = int(np.floor(np.sqrt(2. * s))) - 1  # (lmax + 1)(lmax + 2) / 2 = s: search the two candidates around sqrt(2 s)
        for cand in (lmax - 1, lmax, lmax + 1):
            if cand >= 0 and (cand + 1) * (cand + 2) // 2 == s:
                return cand
        return -1

    @staticmethod
    def getidx(lmax, l, m):
        return m * (2 * lmax + 1 - m) // 2 + l

    @staticmethod
    def getlm(lmax, i=None):
        ls = np.concatenate([np.arange(m, lmax + 1) for m in range(lmax + 1)])
        ms = np.concatenate([np.full(lmax + 1 - m, m) for m in range(lmax + 1)])
        if i is None:
            return ls, ms
        return ls[np.asarray(i)], ms[np.asarray(i)]


def nside2npix(nside):
    return 12 * int(nside) * int(nside)


def npix2nside(npix):
    nside = int(round((int(npix) / 12.) ** 0.5))
    if 12 * nside * nside != int(npix):
        raise ValueError("Wrong pixel number (it is not 12*nside**2)")
    return nside


def nside2pixarea(nside, degrees=False):
    area = 4. * np.pi / nside2npix(nside)
    return area * (180. / np.pi) ** 2 if degrees else area


def almxfl(alm, fl, mmax=None, inplace=False):
    """a_lm <- f_l a_lm, f zero-extended to lmax + 1 entries: one slice per m"""
    out = alm if inplace else np.array(alm, dtype=complex)
    lmax = Alm.getlmax(out.size)
    assert lmax >= 0, 'wrong alm size'
    f = np.zeros(lmax + 1, dtype=complex if np.iscomplexobj(fl) else float)
    n = min(lmax + 1, np.size(fl))
    f[:n] = np.asarray(fl)[:n]
    for m in range(lmax + 1):
        i0 = Alm.getidx(lmax, m, m)
        out[i0:i0 + lmax + 1 - m] *= f[m:]
    return out


def alm2cl(alms1, alms2=None, lmax=None, mmax=None, lmax_out=None):
    """C_l = [a_l0 b_l0 + 2 sum_{m>0} Re(a_lm b_lm^*)] / (2 l + 1), accumulated over m in increasing order"""
    a = np.asarray(alms1)
    b = a if alms2 is None else np.asarray(alms2)
    assert a.ndim == 1 and a.shape == b.shape
    lmax_in = Alm.getlmax(a.size)
    assert lmax_in >= 0
    acc = np.zeros(lmax_in + 1)
    for m in range(lmax_in + 1):
        i0 = Alm.getidx(lmax_in, m, m)
        sl = slice(i0, i0 + lmax_in + 1 - m)
        acc[m:] += (1. if m == 0 else 2.) * (a[sl].real * b[sl].real + a[sl].imag * b[sl].imag)
    acc /= 2. * np.arange(lmax_in + 1) + 1.
    if lmax_out is None:
        lmax_out = lmax_in
    ret = np.zeros(lmax_out + 1)
    n = min(lmax_out, lmax_in) + 1
    ret[:n] = acc[:n]
    return ret


def gauss_beam(fwhm, lmax=512, pol=False):
    """exp(-l (l + 1) sigma^2 / 2), sigma = fwhm / sqrt(8 ln 2) (temperature beam; the reference never asks for pol=True)"""
    assert not pol
    sigma = fwhm / np.sqrt(8. * np.log(2.))
    ell = np.arange(lmax + 1)
    return np.exp(-0.5 * ell * (ell + 1.) * sigma ** 2)


def synalm(cls, lmax=None, rng=None):
    """Gaussian alm of spectrum cls.  These are the fixtures' INPUTS: the draws are taken in the order the committed fixtures were made
    with (n real parts, n imaginary parts, then lmax + 1 fresh values for the real m = 0 entries)."""
    cls = np.asarray(cls, dtype=float)
    if lmax is None:
        lmax = cls.size - 1
    rng = np.random.default_rng() if rng is None else rng
    n = Alm.getsize(lmax)
    re = rng.standard_normal(n)
    im = rng.standard_normal(n)
    a = (re + 1j * im) * np.sqrt(0.5)
    a[:lmax + 1] = rng.standard_normal(lmax + 1)
    return almxfl(a, np.sqrt(np.maximum(cls[:lmax + 1], 0.)))


# ---- RING geometry (HEALPix definition) -----------------------------------------------------------------------
def _ring_of_pixels(nside):
    """(ring index i = 1 .. 4 nside - 1, position j = 1 .. n_i in the ring) of every RING pixel"""
    nside = int(nside)
    npix, ncap = 12 * nside * nside, 2 * nside * (nside - 1)
    ring, pos = np.empty(npix, dtype=np.int64), np.empty(npix, dtype=np.int64)
    start = 0
    for i in range(1, 4 * nside):
        n = 4 * i if i < nside else (4 * nside if i <= 3 * nside else 4 * (4 * nside - i))
        ring[start:start + n] = i
        pos[start:start + n] = np.arange(1, n + 1)
        start += n
    assert start == npix and ncap >= 0
    return ring, pos


def pix2ang(nside, ipix=None):
    """(theta, phi) of the pixel centres, RING ordering"""
    nside = int(nside)
    ring, j = _ring_of_pixels(nside)
    i = ring.astype(float)
    z = np.where(ring < nside, 1. - i * i / (3. * nside * nside),
                 np.where(ring <= 3 * nside, 4. / 3. - 2. * i / (3. * nside), -(1. - (4. * nside - i) ** 2 / (3. * nside * nside))))
    nr = np.where(ring < nside, ring, np.where(ring <= 3 * nside, nside, 4 * nside - ring))
    shift = np.where((ring < nside) | (ring > 3 * nside), 0.5, np.where((ring + nside) % 2 == 1, 1.0, 0.5))
    phi = (j - shift) * np.pi / (2. * nr)
    theta = np.arccos(z)
    if ipix is None:
        return theta, phi
    return theta[np.asarray(ipix)], phi[np.asarray(ipix)]


def pix2vec(nside, ipix=None):
    th, ph = pix2ang(nside, ipix)
    return np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)


def ang2pix(nside, theta, phi):
    """RING index of the pixel containing (theta, phi): the standard HEALPix ang2pix_ring"""
    nside = int(nside)
    z = np.cos(np.asarray(theta, dtype=float))
    za = np.abs(z)
    tt = np.mod(np.asarray(phi, dtype=float), 2. * np.pi) / (0.5 * np.pi)  # in [0, 4)
    ncap, npix = 2 * nside * (nside - 1), 12 * nside * nside
    out = np.empty(z.shape, dtype=np.int64)
    eq = za <= 2. / 3.
    # equatorial belt
    t1 = nside * (0.5 + tt[eq])
    t2 = nside * z[eq] * 0.75
    jp = np.floor(t1 - t2).astype(np.int64)   # ascending edge line
    jm = np.floor(t1 + t2).astype(np.int64)   # descending edge line
    ir = nside + 1 + jp - jm                  # ring counted from z = 2 / 3
    kshift = 1 - (ir & 1)
    ip = (jp + jm - nside + kshift + 1) // 2
    ip = np.mod(ip, 4 * nside)
    out[eq] = ncap + (ir - 1) * 4 * nside + ip
    # polar caps
    pc = ~eq
    tp = tt[pc] - np.floor(tt[pc])
    tmp = nside * np.sqrt(3. * (1. - za[pc]))
    jp = np.floor(tp * tmp).astype(np.int64)
    jm = np.floor((1. - tp) * tmp).astype(np.int64)
    ir = jp + jm + 1
    ip = np.floor(tt[pc] * ir).astype(np.int64)
    ip = np.mod(ip, 4 * ir)
    north = z[pc] > 0
    out[pc] = np.where(north, 2 * ir * (ir - 1) + ip, npix - 2 * ir * (ir + 1) + ip)
    return out


def ud_grade(map_in, nside_out, pess=False, order_in='RING', order_out=None, power=None, dtype=None):
    """Degrade a RING map: mean over the (nside_in / nside_out)^2 children of every parent pixel, times (nside_out / nside_in)^power
    (power = -2: the SUM of the children -- what the reference uses for inverse-noise maps).  A child belongs to the parent that
    contains its centre (the HEALPix hierarchy nests pixel centres inside their parents), found with ang2pix -- no NESTED arithmetic."""
    m = np.asarray(map_in, dtype=float)
    nside_in = npix2nside(m.size)
    assert order_in == 'RING' and order_out in (None, 'RING')
    if nside_out == nside_in:
        return m.copy()
    assert nside_out < nside_in and nside_in % nside_out == 0, 'only degrading'
    th, ph = pix2ang(nside_in)
    parent = ang2pix(nside_out, th, ph)
    nchild = (nside_in // nside_out) ** 2
    counts = np.bincount(parent, minlength=nside2npix(nside_out))
    assert np.all(counts == nchild), 'children not evenly assigned'
    out = np.bincount(parent, weights=m, minlength=nside2npix(nside_out)) / nchild
    if power is not None:
        out *= (float(nside_out) / float(nside_in)) ** power
    return out
