"""healpy-compatible host helpers (index math, geometry, l-filters, spectra).

The reference calls ``healpy`` directly for these (``almxfl`` 147 sites,
``Alm.getlmax`` 32, ``nside2npix`` 22, ``alm2cl`` 14, ``gauss_beam`` 8 ...).  healpy is a third-party
dependency that is absent here, so the public HEALPix / healpy definitions are restated (SURVEY.md
Appendix A.1-A.3).  None of this is the SHT arithmetic: that lives in ``plancklens_amd.shts`` (HIP).

Conventions (healpy):
  * alm: 1-D complex128, m-major triangular, idx(l, m) = m (2 lmax + 1 - m) / 2 + l, mmax = lmax.
  * maps: 1-D float64, RING ordering, npix = 12 nside^2.
"""
import numpy as np

UNSEEN = -1.6375e30


class Alm(object):
    """Index helpers of healpy.Alm (reference uses getlmax/getsize/getidx, e.g. plancklens/utils.py:26-34)."""

    @staticmethod
    def getsize(lmax, mmax=None):
        if mmax is None or mmax < 0 or mmax > lmax:
            mmax = lmax
        return mmax * (2 * lmax + 1 - mmax) // 2 + lmax + 1

    @staticmethod
    def getlmax(s, mmax=None):
        if mmax is not None and mmax >= 0:
            x = (2 * s + mmax ** 2 - mmax - 2) / (2 * mmax + 2)
        else:
            x = (-3 + np.sqrt(1 + 8 * s)) / 2
        if x != np.floor(x):
            return -1
        return int(x)

    @staticmethod
    def getidx(lmax, l, m):
        return m * (2 * lmax + 1 - m) // 2 + l

    @staticmethod
    def getlm(lmax, i=None):
        if i is None:
            i = np.arange(Alm.getsize(lmax))
        i = np.asarray(i)
        m = (np.ceil(((2 * lmax + 1) - np.sqrt((2 * lmax + 1) ** 2 - 8 * (i - lmax))) / 2)).astype(int)
        l = i - m * (2 * lmax + 1 - m) // 2
        return l, m


_LIDX_CACHE = {}


def _ell_of_index(lmax):
    """l value of every entry of a (lmax, mmax=lmax) healpy alm array (cached)."""
    if lmax not in _LIDX_CACHE:
        ls = np.concatenate([np.arange(m, lmax + 1, dtype=np.int32) for m in range(lmax + 1)])
        if len(_LIDX_CACHE) > 8:
            _LIDX_CACHE.clear()
        _LIDX_CACHE[lmax] = ls
    return _LIDX_CACHE[lmax]


def nside2npix(nside):
    return 12 * int(nside) ** 2


def npix2nside(npix):
    nside = int(np.round(np.sqrt(npix / 12.)))
    if 12 * nside * nside != npix:
        raise ValueError("Wrong pixel number (it is not 12*nside**2)")
    return nside


def nside2pixarea(nside, degrees=False):
    a = 4. * np.pi / nside2npix(nside)
    return a * (180. / np.pi) ** 2 if degrees else a


def nside2resol(nside, arcmin=False):
    r = np.sqrt(nside2pixarea(nside))
    return r * 180. * 60. / np.pi if arcmin else r


def isnsideok(nside):
    return nside > 0 and (nside & (nside - 1)) == 0


def almxfl(alm, fl, mmax=None, inplace=False):
    """alm_lm <- fl_l alm_lm; fl shorter than lmax + 1 is zero-extended (healpy semantics)."""
    alm = np.asarray(alm) if inplace else np.array(alm, dtype=complex, copy=True)
    lmax = Alm.getlmax(alm.size, mmax)
    assert lmax >= 0, 'wrong alm size'
    fl = np.asarray(fl)
    f = np.zeros(lmax + 1, dtype=fl.dtype if np.iscomplexobj(fl) else float)
    n = min(lmax + 1, fl.size)
    f[:n] = fl[:n]
    alm *= f[_ell_of_index(lmax)]
    return alm


def alm2cl(alms1, alms2=None, lmax=None, mmax=None, lmax_out=None):
    """C_l = 1 / (2l + 1) sum_m w_m Re(a_lm b_lm^*), w_0 = 1, w_m>0 = 2 (single-spectrum form only)."""
    a = np.asarray(alms1)
    b = a if alms2 is None else np.asarray(alms2)
    assert a.ndim == 1 and b.ndim == 1 and a.size == b.size
    lmax_in = Alm.getlmax(a.size)
    assert lmax_in >= 0
    if lmax_out is None:
        lmax_out = lmax_in
    prod = a.real * b.real + a.imag * b.imag
    ls = _ell_of_index(lmax_in)
    w = np.full(a.size, 2.)
    w[:lmax_in + 1] = 1.
    cl = np.bincount(ls, weights=prod * w, minlength=lmax_in + 1) / (2. * np.arange(lmax_in + 1) + 1.)
    ret = np.zeros(lmax_out + 1)
    n = min(lmax_out, lmax_in) + 1
    ret[:n] = cl[:n]
    return ret


def gauss_beam(fwhm, lmax=512, pol=False):
    sigma = fwhm / np.sqrt(8. * np.log(2.))
    ell = np.arange(lmax + 1)
    g = np.exp(-0.5 * ell * (ell + 1.) * sigma ** 2)
    if not pol:
        return g
    f2 = np.exp(2. * sigma ** 2)
    f1 = np.exp(0.5 * sigma ** 2)
    return np.stack([g, g * f2, g * f2, g * f1], axis=1)


def pixwin(nside, pol=False, lmax=None, datapath=None):
    """healpy.pixwin: the HEALPix pixel window function, read from the `pixel_window_n%04d.fits` tables that ship with
    healpy / HEALPix (columns TEMPERATURE, POLARIZATION).  The tables are data, not code, and cannot be regenerated here
    (they come from the HEALPix facility, extrapolated above nside 128); they are looked for in `datapath`,
    $PLENS_HEALPIX_DATA, $HEALPY_DATAPATH and $HEALPIX/data.  Without them this raises (hard part 6)."""
    import os
    from . import fitsio
    fname = 'pixel_window_n%04d.fits' % nside
    dirs = [datapath, os.environ.get('PLENS_HEALPIX_DATA'), os.environ.get('HEALPY_DATAPATH')]
    if os.environ.get('HEALPIX'):
        dirs.append(os.path.join(os.environ['HEALPIX'], 'data'))
    for d in dirs:
        if d and os.path.exists(os.path.join(d, fname)):
            cols, _ = fitsio.read_bintable(os.path.join(d, fname), hdu=1)
            low = {k.upper(): v for k, v in cols.items()}
            pw_t = np.asarray(low['TEMPERATURE'], dtype=np.float64).ravel()
            pw_p = np.asarray(low.get('POLARIZATION', low['TEMPERATURE']), dtype=np.float64).ravel()
            n = (3 * nside - 1 if lmax is None else lmax) + 1
            assert n <= pw_t.size, 'pixel window table of nside %d stops at l = %d' % (nside, pw_t.size - 1)
            return (pw_t[:n], pw_p[:n]) if pol else pw_t[:n]
    raise NotImplementedError('pixwin needs the window-function tables shipped with healpy (%s): point PLENS_HEALPIX_DATA, '
                              'HEALPY_DATAPATH or HEALPIX to a directory that holds them' % fname)


# ---------------------------------------------------------------------------------------------
# RING geometry
# ---------------------------------------------------------------------------------------------
def ring_info(nside):
    """Per-ring (i = 1 .. 4 nside - 1) arrays: cos(theta), sin(theta), nphi, phi0, first pixel."""
    nside = int(nside)
    i = np.arange(1, 4 * nside, dtype=np.int64)
    north = np.minimum(i, 4 * nside - i)
    cap = north < nside
    nphi = np.where(cap, 4 * north, 4 * nside)
    z = np.where(cap, 1. - north.astype(float) ** 2 / (3. * nside ** 2), 4. / 3. - 2. * north / (3. * nside))
    # sin(theta) without cancellation in the caps: 1 - z^2 = (1 - z)(1 + z)
    omz = np.where(cap, north.astype(float) ** 2 / (3. * nside ** 2), 1. - z)
    sth = np.sqrt(omz * (1. + z))
    cth = np.where(i > 2 * nside, -z, z)
    shifted = np.where(cap, True, ((north - nside) & 1) == 0)
    phi0 = np.where(shifted, np.pi / nphi, 0.)
    ofs = np.concatenate([[0], np.cumsum(nphi)[:-1]])
    return cth, sth, nphi.astype(np.int64), phi0, ofs.astype(np.int64)


def pix2ang(nside, ipix=None):
    """(theta, phi) of RING pixels (all of them if ipix is None)."""
    cth, sth, nphi, phi0, ofs = ring_info(nside)
    npix = nside2npix(nside)
    ring = np.repeat(np.arange(nphi.size), nphi)
    j = np.arange(npix) - ofs[ring]
    th = np.arctan2(sth, cth)[ring]
    ph = phi0[ring] + j * (2. * np.pi / nphi[ring])
    if ipix is None:
        return th, ph
    return th[ipix], ph[ipix]


def pix2ring(nside, ipix=None):
    """ring number 1 ... 4 nside - 1 (north to south) of RING pixels (all of them if ipix is None)"""
    nphi = ring_info(nside)[2]
    ring = np.repeat(np.arange(1, nphi.size + 1), nphi)
    return ring if ipix is None else ring[ipix]


def pix2vec(nside, ipix=None):
    cth, sth, nphi, phi0, ofs = ring_info(nside)
    npix = nside2npix(nside)
    ring = np.repeat(np.arange(nphi.size), nphi)
    j = np.arange(npix) - ofs[ring]
    ph = phi0[ring] + j * (2. * np.pi / nphi[ring])
    x, y, z = sth[ring] * np.cos(ph), sth[ring] * np.sin(ph), cth[ring]
    if ipix is None:
        return x, y, z
    return x[ipix], y[ipix], z[ipix]


# ---------------------------------------------------------------------------------------------
# RING <-> NESTED (integer work, needed by ud_grade; reference: qcinv/opfilt_tt.py:172-181)
# ---------------------------------------------------------------------------------------------
_JRLL = np.array([2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4])
_JPLL = np.array([1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7])


def _compress_bits(v):
    v = v & 0x5555555555555555
    v = (v | (v >> 1)) & 0x3333333333333333
    v = (v | (v >> 2)) & 0x0F0F0F0F0F0F0F0F
    v = (v | (v >> 4)) & 0x00FF00FF00FF00FF
    v = (v | (v >> 8)) & 0x0000FFFF0000FFFF
    v = (v | (v >> 16)) & 0x00000000FFFFFFFF
    return v


def nest2ring(nside, ipnest):
    nside = int(nside)
    ipnest = np.asarray(ipnest, dtype=np.int64)
    npface = nside * nside
    face = ipnest // npface
    p = ipnest % npface
    ix = _compress_bits(p)
    iy = _compress_bits(p >> 1)
    jr = _JRLL[face] * nside - ix - iy - 1
    nl4 = 4 * nside
    ncap = 2 * nside * (nside - 1)
    npix = 12 * nside * nside
    north = jr < nside
    south = jr > 3 * nside
    nr = np.where(north, jr, np.where(south, nl4 - jr, nside))
    n_before = np.where(north, 2 * nr * (nr - 1),
                        np.where(south, npix - 2 * (nr + 1) * nr, ncap + (jr - nside) * nl4))
    kshift = np.where(north | south, 0, (jr - nside) & 1)
    jp = (_JPLL[face] * nr + ix - iy + 1 + kshift) // 2
    jp = np.where(jp > nl4, jp - nl4, jp)
    jp = np.where(jp < 1, jp + nl4, jp)
    return n_before + jp - 1


def ud_grade(map_in, nside_out, pess=False, order_in='RING', order_out=None, power=None, dtype=None):
    """Degrade (only) a RING map: children average x (nside_out/nside_in)^power  (healpy semantics;
    with power=-2 this is the SUM of the children)."""
    map_in = np.asarray(map_in, dtype=float)
    nside_in = npix2nside(map_in.size)
    assert order_in == 'RING' and order_out in (None, 'RING')
    if nside_out == nside_in:
        return map_in.copy()
    assert nside_out < nside_in and isnsideok(nside_out) and isnsideok(nside_in), 'only degrading implemented'
    rat = (nside_in // nside_out) ** 2
    m_nest = np.empty_like(map_in)
    m_nest[:] = map_in[nest2ring(nside_in, np.arange(map_in.size))]
    out_nest = m_nest.reshape(nside2npix(nside_out), rat).mean(axis=1)
    if power is not None:
        out_nest *= (float(nside_out) / float(nside_in)) ** power
    out = np.empty_like(out_nest)
    out[nest2ring(nside_out, np.arange(out_nest.size))] = out_nest
    return out


def synalm(cls, lmax=None, rng=None):
    """Gaussian alm with spectrum cls (single field; healpy-style normalisation)."""
    rng = np.random.default_rng() if rng is None else rng
    cls = np.asarray(cls, dtype=float)
    if lmax is None:
        lmax = cls.size - 1
    n = Alm.getsize(lmax)
    alm = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * np.sqrt(0.5)
    alm[:lmax + 1] = rng.standard_normal(lmax + 1)
    return almxfl(alm, np.sqrt(np.maximum(cls[:lmax + 1], 0.)), inplace=True)


# ---------------------------------------------------------------------------------------------
# on-disk formats (reference cache files: filt_simple.py:97-99, qest.py:17,201)
# ---------------------------------------------------------------------------------------------
def write_alm(filename, alms, lmax=-1, mmax=-1, overwrite=False, **kwargs):
    """healpy.write_alm layout: one binary table, columns index = l^2 + l + m + 1, real, imag."""
    from . import fitsio
    alms = np.asarray(alms)
    lmax_in = Alm.getlmax(alms.size)
    assert lmax_in >= 0
    if lmax < 0 or lmax > lmax_in:
        lmax = lmax_in
    if mmax < 0 or mmax > lmax:
        mmax = lmax
    l, m = Alm.getlm(lmax_in)
    keep = (l <= lmax) & (m <= mmax)
    idx = (l[keep].astype(np.int64) ** 2 + l[keep] + m[keep] + 1).astype(np.int32)
    fitsio.write_bintable(filename, [('index', idx), ('real', alms.real[keep]), ('imag', alms.imag[keep])],
                          extname='xtension', overwrite=overwrite)


def read_alm(filename, hdu=1, return_mmax=False):
    from . import fitsio
    cols, _ = fitsio.read_bintable(filename, hdu=hdu)
    names = list(cols.keys())
    idx = cols[names[0]].astype(np.int64)
    l = np.floor(np.sqrt(idx - 1)).astype(np.int64)
    m = idx - l * l - l - 1
    lmax, mmax = int(l.max()), int(m.max())
    alm = np.zeros(Alm.getsize(lmax, mmax), dtype=complex)
    i = m * (2 * lmax + 1 - m) // 2 + l
    alm.real[i] = cols[names[1]]
    alm.imag[i] = cols[names[2]]
    return (alm, mmax) if return_mmax else alm


def write_map(filename, m, nest=False, dtype=None, overwrite=False, **kwargs):
    from . import fitsio
    m = np.asarray(m, dtype=np.float64 if dtype is None else dtype)
    maps = [m] if m.ndim == 1 else list(m)
    npix = maps[0].size
    nside = npix2nside(npix)
    rep = 1024 if npix % 1024 == 0 else 1
    names = ['TEMPERATURE', 'Q_POLARISATION', 'U_POLARISATION'] if len(maps) == 3 else ['I_STOKES%d' % i for i in range(len(maps))]
    if len(maps) == 1:
        names = ['TEMPERATURE']
    cols = [(n, mm.reshape(-1, rep) if rep > 1 else mm) for n, mm in zip(names, maps)]
    fitsio.write_bintable(filename, cols, extname='xtension', overwrite=overwrite,
                          extra=[('PIXTYPE', 'HEALPIX'), ('ORDERING', 'NESTED' if nest else 'RING'), ('NSIDE', nside),
                                 ('FIRSTPIX', 0), ('LASTPIX', npix - 1), ('INDXSCHM', 'IMPLICIT')])


def read_map(filename, field=0, dtype=None, nest=False, hdu=1, verbose=False, **kwargs):
    from . import fitsio
    cols, hdr = fitsio.read_bintable(filename, hdu=hdu)
    assert str(hdr.get('ORDERING', 'RING')).strip() == ('NESTED' if nest else 'RING'), 'ordering conversion not implemented'
    names = list(cols.keys())
    fields = [field] if np.isscalar(field) else (range(len(names)) if field is None else list(field))
    out = [np.asarray(cols[names[f]], dtype=np.float64 if dtype is None else dtype).ravel() for f in fields]
    return out[0] if np.isscalar(field) else out
