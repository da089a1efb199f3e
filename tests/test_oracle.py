"""Pins the CPU oracle (test infrastructure) -- SURVEY.md section 8(c) substitutes (i)-(iii).

The reference has no SHT test, so the oracle is pinned by closed-form known answers, by the reference's
own Fortran Wigner module (built from /root/reference into oracle/_ref/ when present), by a brute-force
pixel-by-pixel evaluation of the definition, by adjointness, and by long-double vs scaled-double agreement.
"""
import ctypes
import os

import numpy as np
import pytest

from plancklens_amd import hp
from helpers import random_alm, alm_dot, relrms, alm_size

HERE = os.path.dirname(os.path.abspath(__file__))
REFSO = os.path.join(os.path.dirname(HERE), 'oracle', '_ref', 'libwigners_ref.so')


def test_monopole_dipole(oracle):
    """a_00 = sqrt(4pi) -> 1; xyz_to_alm dipole (template_removal.py:153-158)."""
    nside, lmax = 8, 16
    for mode in (0, 1):
        alm = np.zeros(alm_size(lmax), complex)
        alm[0] = np.sqrt(4 * np.pi)
        assert np.abs(oracle.alm2map(alm, nside, mode=mode) - 1).max() < 1e-14
        x, y, z = 0.3, -0.7, 1.1
        alm[:] = 0
        alm[1] = z * np.sqrt(4 * np.pi / 3)
        alm[hp.Alm.getidx(lmax, 1, 1)] = (-x + 1j * y) * np.sqrt(2 * np.pi / 3)
        vx, vy, vz = hp.pix2vec(nside)
        assert np.abs(oracle.alm2map(alm, nside, mode=mode) - (x * vx + y * vy + z * vz)).max() < 1e-14
    # A.5 (iii): equal-area pixels -> map2alm of the constant map gives exactly sqrt(4pi)
    a = oracle.map2alm(np.ones(12 * nside ** 2), lmax)
    assert abs(a[0] - np.sqrt(4 * np.pi)) < 1e-14


@pytest.mark.parametrize('spin', [0, 1, 2, 3])
def test_brute_force_and_modes(oracle, spin):
    """Definition (pixel by pixel) == separable long double == scaled double."""
    rng = np.random.default_rng(10 + spin)
    nside, lmax = 4, 11
    g, c = random_alm(rng, lmax, spin), random_alm(rng, lmax, spin)
    if spin == 0:
        ref = oracle.brute_alm2map_spin([g], nside, 0, lmax)
        m0 = oracle.alm2map(g, nside, mode=0, use_pairs=False)
        m1 = oracle.alm2map(g, nside, mode=1)
        assert relrms(m0, ref) < 1e-13 and relrms(m1, ref) < 1e-13
    else:
        ref = oracle.brute_alm2map_spin([g, c], nside, spin, lmax)
        m0 = oracle.alm2map_spin([g, c], nside, spin, lmax, mode=0, use_pairs=False)
        m1 = oracle.alm2map_spin([g, c], nside, spin, lmax, mode=1)
        for i in range(2):
            assert relrms(m0[i], ref[i]) < 1e-13 and relrms(m1[i], ref[i]) < 1e-13


@pytest.mark.parametrize('spin', [0, 1, 2, 3])
def test_adjointness(oracle, spin):
    """<map, alm2map(a)> = npix/4pi <map2alm(map), a>  (opfilt_tt.py:190)."""
    rng = np.random.default_rng(20 + spin)
    nside, lmax = 8, 20
    npix = 12 * nside ** 2
    for mode in (0, 1):
        if spin == 0:
            a = random_alm(rng, lmax)
            t = rng.standard_normal(npix)
            lhs = np.sum(t * oracle.alm2map(a, nside, mode=mode))
            rhs = npix / (4 * np.pi) * alm_dot(oracle.map2alm(t, lmax, mode=mode), a, lmax)
        else:
            g, c = random_alm(rng, lmax, spin), random_alm(rng, lmax, spin)
            q, u = rng.standard_normal(npix), rng.standard_normal(npix)
            mq, mu = oracle.alm2map_spin([g, c], nside, spin, lmax, mode=mode)
            ga, ca = oracle.map2alm_spin([q, u], spin, lmax, mode=mode)
            lhs = np.sum(q * mq + u * mu)
            rhs = npix / (4 * np.pi) * (alm_dot(ga, g, lmax) + alm_dot(ca, c, lmax))
        assert abs(lhs - rhs) < 1e-12 * abs(lhs)


@pytest.mark.parametrize('spin', [0, 1, 2, 3])
def test_scaled_vs_longdouble_high_m(oracle, spin):
    """Scaled double arithmetic (mode 1) in the regime where sin^m(theta) underflows IEEE double."""
    rng = np.random.default_rng(30 + spin)
    nside, lmax = 16, 47
    cth, sth, nphi, phi0, ofs = oracle.ring_geometry(nside)
    # long lmax on few rings: rings near the pole, lmax = 1500 -> sin^m underflows for m > ~ 400
    lmax = 1500
    rings = np.array([0, 1, 2, 5, 9, 15, 23, 31])
    ncomp = 1 if spin == 0 else 2
    alm = np.stack([random_alm(rng, lmax, spin) for _ in range(ncomp)])
    pair = np.ones(rings.size, dtype=np.int32)
    p0 = oracle.legendre(0, 0, spin, lmax, lmax, cth[rings], sth[rings], pair, alm=alm)
    p1 = oracle.legendre(0, 1, spin, lmax, lmax, cth[rings], sth[rings], pair, alm=alm)
    assert relrms(p1, p0) < 1e-12
    a0 = oracle.legendre(1, 0, spin, lmax, lmax, cth[rings], sth[rings], pair, phase=p0)
    a1 = oracle.legendre(1, 1, spin, lmax, lmax, cth[rings], sth[rings], pair, phase=p0)
    assert relrms(a1, a0) < 1e-12


@pytest.mark.skipif(not os.path.exists(REFSO), reason='oracle/_ref not built (needs /root/reference + amdflang)')
@pytest.mark.parametrize('spin', [0, 1, 2, 3])
def test_against_reference_fortran_wigners(oracle, spin):
    """_slambda_lm from the reference's wigners.f90 (wigners.f90:566-624),
    _sY_lm(theta, 0) = (-1)^s sqrt((2l+1)/4pi) d^l_{m,-s}(theta); wignerpos(cl, x, s1, s2) returns
    sum_l cl_l (2l+1)/(4pi) d^l_{s1 s2}(x)."""
    lib = ctypes.CDLL(REFSO)
    dp = ctypes.POINTER(ctypes.c_double)
    ip = ctypes.POINTER(ctypes.c_int)

    def wigd(l, m, n, x, lmax):
        cl = np.zeros(lmax + 1)
        cl[l] = 1.
        xi = np.zeros(x.size)
        nx, lm_, s1, s2 = (ctypes.c_int(v) for v in (x.size, lmax, m, n))
        lib.wignerpos_(xi.ctypes.data_as(dp), ctypes.byref(nx), ctypes.byref(lm_), cl.ctypes.data_as(dp),
                       x.ctypes.data_as(dp), ctypes.byref(s1), ctypes.byref(s2))
        return xi * 4 * np.pi / (2 * l + 1)

    sg = (-1.) ** spin
    # (lmax, orders m, step in l, tolerance): the small case of round 1, and lmax = 300 -- every l from max(m, s) on a
    # coarse grid up to 299 and orders up to m = 290, where the seeds are ~1e-90 at x = 0.88 (the double-precision Fortran
    # recursion is itself good to ~1e-13 there)
    for lmax, ms, lstep, tol in ((24, (0, 1, 2, 5, 17), 5, 1e-13), (300, (0, 1, 3, 40, 150, 290), 37, 1e-12)):
        x = np.array([-0.93, -0.41, 0.07, 0.55, 0.88])
        for m in ms:
            for l in range(max(m, spin), lmax, lstep):  # wignerpos needs lmax > lmin (SURVEY.md 8(c) caveat)
                nrm = np.sqrt((2 * l + 1) / (4 * np.pi))
                lam_p = sg * nrm * wigd(l, m, -spin, x, lmax)
                lam_m = sg * nrm * wigd(l, m, spin, x, lmax)
                for ix, xx in enumerate(x):
                    fp, fm = oracle.lambda_lm(spin, m, lmax, xx, np.sqrt(1 - xx * xx))
                    if spin == 0:
                        assert abs(fp[l] - lam_p[ix]) < tol, (lmax, m, l, xx)
                    else:
                        assert abs(fp[l] - (-0.5) * (lam_p[ix] + sg * lam_m[ix])) < tol, (lmax, m, l, xx)
                        assert abs(fm[l] - (-0.5) * (lam_p[ix] - sg * lam_m[ix])) < tol, (lmax, m, l, xx)


def test_healpix_pixel_centres_literal_constants(oracle):
    """Geometry pinned by constants that do not come from this repository: the base-resolution pixel centres of the HEALPix
    definition (Gorski et al. 2005: z = 2/3, 0, -2/3; phi = (k + 1/2) pi/2 in the polar rings, k pi/2 on the equator),
    the nside = 2 ring table written out by hand from the same definition, and the (theta, phi) values printed in healpy's
    own documentation of pix2ang for nside = 16 (pixels 1440, 427, 1520, 0, 3068).  Both the product's hp.py and the
    oracle's ring_geometry must reproduce them."""
    # nside = 1
    z1 = np.repeat([2. / 3., 0., -2. / 3.], 4)
    phi1 = np.array([0.25, 0.75, 1.25, 1.75, 0., 0.5, 1., 1.5, 0.25, 0.75, 1.25, 1.75]) * np.pi
    # nside = 2: rings of 4, 8, 8, 8, 8, 8, 4 pixels at z = 11/12, 2/3, 1/3, 0, -1/3, -2/3, -11/12
    z2 = np.concatenate([np.full(n, z) for n, z in zip((4, 8, 8, 8, 8, 8, 4), (11. / 12, 2. / 3, 1. / 3, 0., -1. / 3, -2. / 3, -11. / 12))])
    odd8, even8 = np.arange(1, 16, 2) * np.pi / 8, np.arange(0, 16, 2) * np.pi / 8
    cap4 = np.array([2., 6., 10., 14.]) * np.pi / 8
    phi2 = np.concatenate([cap4, odd8, even8, odd8, even8, odd8, cap4])
    for nside, z, phi in ((1, z1, phi1), (2, z2, phi2)):
        th, ph = hp.pix2ang(nside)
        assert np.abs(np.cos(th) - z).max() < 1e-15 and np.abs(ph - phi).max() < 1e-15
        cth, sth, nphi, phi0, ofs = oracle.ring_geometry(nside)
        zz = np.repeat(cth, nphi)
        pp = np.concatenate([p0 + 2 * np.pi * np.arange(n) / n for p0, n in zip(phi0, nphi)])
        assert np.abs(zz - z).max() < 1e-15 and np.abs(pp - phi).max() < 1e-15
        assert np.abs(np.repeat(sth, nphi) - np.sqrt(1 - z * z)).max() < 1e-15
    th, ph = hp.pix2ang(16)
    idx = [1440, 427, 1520, 0, 3068]
    assert np.abs(th[idx] - np.array([1.52911759, 0.78550497, 1.57079633, 0.05103658, 3.09055608])).max() < 5e-9
    assert np.abs(ph[idx] - np.array([0., 0.78539816, 1.61988371, 0.78539816, 0.78539816])).max() < 5e-9


def test_gradient_spin1(oracle):
    """with _1X_lm = -(G + iC) and the Goldberg edth, edth T = -(d_theta + i/sin d_phi) T
    = sum sqrt(l(l+1)) T_lm _1Y_lm, so alm2map_spin([-sqrt(l(l+1)) T_lm, 0], 1) (qest.py:592-593) is edth T
    = -(d_theta T, d_phi T / sin theta); equivalently G = +sqrt(l(l+1)) phi_lm gives +grad(phi), the lensing
    deflection convention.  For the dipole T = cos(theta): (+sin theta, 0)."""
    nside, lmax = 8, 4
    alm = np.zeros(alm_size(lmax), complex)
    alm[1] = np.sqrt(4 * np.pi / 3)  # T = cos(theta)
    fl = -np.sqrt(np.arange(lmax + 1.) * np.arange(1, lmax + 2.))
    g = hp.almxfl(alm, fl)
    re, im = oracle.alm2map_spin([g, np.zeros_like(g)], nside, 1, lmax)
    th, ph = hp.pix2ang(nside)
    assert np.abs(re - np.sin(th)).max() < 1e-13 and np.abs(im).max() < 1e-13
    # T = x = sin(theta) cos(phi): -(d_theta T, d_phi T / sin) = (-cos(theta) cos(phi), +sin(phi))
    alm[:] = 0
    alm[hp.Alm.getidx(lmax, 1, 1)] = -np.sqrt(2 * np.pi / 3)
    g = hp.almxfl(alm, fl)
    re, im = oracle.alm2map_spin([g, np.zeros_like(g)], nside, 1, lmax)
    assert np.abs(re + np.cos(th) * np.cos(ph)).max() < 1e-13 and np.abs(im - np.sin(ph)).max() < 1e-13


@pytest.mark.parametrize('nside,lmax', [(1, 2), (4, 11), (6, 20), (8, 16), (16, 47)])
def test_c_ring_fft_stage_equals_numpy_stage(oracle, nside, lmax):
    """oracle.ring_fft_c (threaded C radix-2 / Bluestein, used by the CPU baseline) against the numpy pocketfft route the
    other tests pin, both directions, incl. aliasing (lmax = 3 nside - 1) and a ring count that is not a power of two."""
    rng = np.random.default_rng(nside + lmax)
    c, s, pair, slots = oracle._pair_geometry(nside, True)
    ph = rng.standard_normal((slots.size, lmax + 1)) + 1j * rng.standard_normal((slots.size, lmax + 1))
    ph[:, 0] = ph[:, 0].real
    ph[slots < 0] = 0.
    a, b = oracle._phase2map(ph, nside, lmax, slots), oracle.ring_fft_c(0, nside, lmax, slots, phase=ph, nthreads=2)
    assert relrms(b, a) < 1e-13
    m = rng.standard_normal(12 * nside ** 2)
    a, b = oracle._map2phase(m, nside, lmax, slots), oracle.ring_fft_c(1, nside, lmax, slots, m=m, nthreads=2)
    assert relrms(b[slots >= 0], a[slots >= 0]) < 1e-13


def test_preallocated_outputs_of_the_c_stages(oracle):
    """legendre(..., out=) and ring_fft_c(..., out=) (the preallocated result arrays bench.py's cpu_baseline reuses) give exactly what the
    allocating calls give, also when the arrays held other data before."""
    so = oracle
    nside, lmax = 16, 40
    rng = np.random.default_rng(3)
    c, s, pair, slots = so._pair_geometry(nside, True)
    nalm = so.alm_size(lmax)
    for spin in (0, 2):
        nc = 1 if spin == 0 else 2
        alm = rng.standard_normal((nc, nalm)) + 1j * rng.standard_normal((nc, nalm))
        alm[:, :lmax + 1] = alm[:, :lmax + 1].real
        ref = so.legendre(0, 1, spin, lmax, lmax, c, s, pair, alm=alm)
        buf = np.full(ref.shape, 7. + 3j)
        assert so.legendre(0, 1, spin, lmax, lmax, c, s, pair, alm=alm, out=buf) is buf and np.array_equal(buf, ref)
        maps = rng.standard_normal((nc, 12 * nside ** 2))
        ph = np.zeros((nc, slots.size, lmax + 1), dtype=complex)
        for i in range(nc):
            r = so.ring_fft_c(1, nside, lmax, slots, m=maps[i])
            assert so.ring_fft_c(1, nside, lmax, slots, m=maps[i], out=ph[i]) is not None and np.array_equal(ph[i], r)
            assert np.allclose(r, so._map2phase(maps[i], nside, lmax, slots), rtol=0, atol=1e-13 * np.abs(r).max())
        aref = so.legendre(1, 1, spin, lmax, lmax, c, s, pair, phase=ph)
        abuf = np.full(aref.shape, 1. - 2j)
        so.legendre(1, 1, spin, lmax, lmax, c, s, pair, phase=ph, out=abuf)
        assert np.array_equal(abuf, aref)
        # mode 1 (vectorised, scaled double) against mode 0 (long double) on the same inputs
        assert np.abs(ref - so.legendre(0, 0, spin, lmax, lmax, c, s, pair, alm=alm)).max() < 1e-12 * np.abs(ref).max()
        assert np.abs(aref - so.legendre(1, 0, spin, lmax, lmax, c, s, pair, phase=ph)).max() < 1e-12 * np.abs(aref).max()


def test_oracle_host_helpers_agree_with_the_products():
    """oracle/hp_oracle.py (what the golden generator hands the reference as `healpy`) and plancklens_amd/hp.py (the product's host
    helpers) are written independently; on random inputs they must agree -- bit for bit where the arithmetic is a fixed sequence
    (index maps, almxfl, alm2cl, gauss_beam, synalm), to a few ulp for the degrade (different summation order) and the pixel angles."""
    from oracle import hp_oracle as oh
    from plancklens_amd import hp
    rng = np.random.default_rng(77)
    for lmax in (0, 1, 7, 40, 129):
        n = oh.Alm.getsize(lmax)
        assert n == hp.Alm.getsize(lmax) and oh.Alm.getlmax(n) == lmax == hp.Alm.getlmax(n) and oh.Alm.getlmax(n + 1) == -1
        l1, m1 = oh.Alm.getlm(lmax)
        l2, m2 = hp.Alm.getlm(lmax)
        assert np.array_equal(l1, l2) and np.array_equal(m1, m2) and np.array_equal(oh.Alm.getidx(lmax, l1, m1), np.arange(n))
        a = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        for fl in (rng.standard_normal(lmax + 1), rng.standard_normal(max(1, lmax // 2)), rng.standard_normal(lmax + 9)):
            assert np.array_equal(oh.almxfl(a, fl), hp.almxfl(a, fl))
        assert np.array_equal(oh.alm2cl(a), hp.alm2cl(a)) and np.array_equal(oh.alm2cl(a, b), hp.alm2cl(a, b))
        assert np.array_equal(oh.alm2cl(a, b, lmax_out=lmax + 3), hp.alm2cl(a, b, lmax_out=lmax + 3))
        cl = rng.uniform(0.1, 2., lmax + 1)
        assert np.array_equal(oh.synalm(cl, lmax, np.random.default_rng(5)), hp.synalm(cl, lmax, np.random.default_rng(5)))
    for fwhm in (0.01, 0.1):
        assert np.array_equal(oh.gauss_beam(fwhm, lmax=300), hp.gauss_beam(fwhm, lmax=300))
    for nside in (1, 2, 8, 32):
        assert oh.nside2npix(nside) == hp.nside2npix(nside) and oh.npix2nside(12 * nside ** 2) == nside
        assert oh.nside2pixarea(nside, degrees=True) == hp.nside2pixarea(nside, degrees=True)
        t1, p1 = oh.pix2ang(nside)
        t2, p2 = hp.pix2ang(nside)
        assert np.max(np.abs(t1 - t2)) < 1e-14 and np.max(np.abs(p1 - p2)) < 1e-14
        assert np.array_equal(oh.ang2pix(nside, t2, p2), np.arange(12 * nside ** 2))  # every centre lies in its own pixel
        m = rng.standard_normal(12 * nside ** 2)
        nout = nside
        while nout >= 1:
            for power in (None, -2):
                x, y = oh.ud_grade(m, nout, power=power), hp.ud_grade(m, nout, power=power)
                assert np.max(np.abs(x - y)) <= 1e-14 * np.max(np.abs(y))
            nout //= 2
