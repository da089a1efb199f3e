"""The template projection of the CG operators in harmonic space, checked on the CPU with the oracle's transforms (no GPU):
    B^t Y^t [N^-1 - N^-1 T (T^t N^-1 T)^-1 T^t N^-1] Y B x  =  B^t Y^t N^-1 Y B x - V (T^t N^-1 T)^-1 V^t x,   V = B^t Y^t N^-1 T
with Y = alm2map, Y^t = (npix / 4 pi) map2alm its adjoint and the real scalar product of alm vectors (weights 1 for m = 0, 2 above).
The left side is what plancklens/qcinv/opfilt_tt.py:67-73,196-205 applies; the right side is what the product runs
(plancklens_amd/qcinv/template_removal.harmonic_matrices + pl_lowrank_update_b)."""
import numpy as np

from helpers import random_alm, relrms


def _rdot(a, b, lmax):
    w = np.full(a.size, 2.)
    w[:lmax + 1] = 1.
    return float(np.sum(w * (a.real * b.real + a.imag * b.imag)))


def test_harmonic_space_projection_is_the_pixel_space_one(oracle):
    nside, lmax = 8, 16
    npix = 12 * nside ** 2
    rng = np.random.default_rng(5)
    ell = np.arange(lmax + 1.)
    bl = np.exp(-ell * (ell + 1.) * 1e-3)
    fac = npix / (4. * np.pi)
    ninv = (0.5 + rng.random(npix)) * (rng.random(npix) > 0.2)

    def Y(a):
        return oracle.alm2map(a, nside, lmax=lmax)

    def almxfl(a, fl):
        out = a.copy()
        i = 0
        for m in range(lmax + 1):
            n = lmax + 1 - m
            out[i:i + n] *= fl[m:]
            i += n
        return out

    def Yt(m):
        return oracle.map2alm(m, lmax=lmax) * fac

    # templates: monopole, a dipole-like map and a random map
    z = np.cos(np.linspace(0.05, np.pi - 0.05, npix))
    tmpl = np.stack([np.ones(npix), z, rng.standard_normal(npix)])
    tniti = np.linalg.inv(tmpl @ (ninv[None, :] * tmpl).T)
    x = random_alm(rng, lmax)

    # pixel space (the reference's form)
    m = Y(almxfl(x, bl))
    w = ninv * m
    w -= ninv * (tmpl.T @ (tniti @ (tmpl @ w)))
    lhs = almxfl(Yt(w), bl)

    # harmonic space
    v = np.stack([almxfl(Yt(ninv * t), bl) for t in tmpl])
    c = np.array([_rdot(vk, x, lmax) for vk in v])
    rhs = almxfl(Yt(ninv * m), bl) - (tniti @ c) @ v
    assert relrms(rhs, lhs) < 1e-12
    # the adjoint relation the identity rests on: <Y^t p, a> = <p, Y a> for maps p and alm a (real scalar products)
    p = rng.standard_normal(npix)
    assert abs(_rdot(Yt(p), x, lmax) - float(p @ Y(x))) < 1e-10 * abs(float(p @ Y(x)))
