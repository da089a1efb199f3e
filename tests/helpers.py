"""Shared test helpers (seeded inputs, norms)."""
import numpy as np


def alm_size(lmax):
    return (lmax + 1) * (lmax + 2) // 2


def random_alm(rng, lmax, lmin=0):
    """Unit-variance alm, real m=0 column, zero below lmin."""
    n = alm_size(lmax)
    a = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    a[:lmax + 1] = a[:lmax + 1].real
    ls = np.concatenate([np.arange(m, lmax + 1) for m in range(lmax + 1)])
    a[ls < lmin] = 0.
    return a


def alm_dot(a, b, lmax):
    """sum_lm over all m (including negative) of Re(a b*)."""
    w = np.full(a.size, 2.)
    w[:lmax + 1] = 1.
    return np.sum(w * (a.real * b.real + a.imag * b.imag))


def relrms(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    return np.sqrt(np.sum(np.abs(a - b) ** 2) / max(np.sum(np.abs(b) ** 2), 1e-300))


def cinv_golden_inputs(alm2map, alm2map_spin, nside=512, lmax=1024):
    """Inputs of tests/golden/cinv_golden.npz (the reference's own filt_cinv.cinv_t / cinv_p at the smallest size they accept),
    as a recipe: a galactic-cut + point-source mask, inhomogeneous inverse-variance maps and seeded data maps.  The transforms are
    arguments so that the generator (oracle SHTs, tests/golden/make_golden.py) and the tests share one definition; the maps are
    25 MB each and are therefore re-made, not stored (the fixture stores a checksum of every input)."""
    from plancklens_amd import hp
    npix = 12 * nside ** 2
    th, ph = hp.pix2ang(nside, np.arange(npix))
    z = np.cos(th)
    mask = (np.abs(z) > 0.18 + 0.05 * np.cos(2 * ph)).astype(float)           # wavy galactic cut, fsky ~ 0.8
    rng = np.random.default_rng(4242)
    for zc, pc in zip(rng.uniform(-1, 1, 24), rng.uniform(0, 2 * np.pi, 24)):  # 24 discs of 1.5 degrees
        cosd = z * zc + np.sqrt((1 - z ** 2) * (1 - zc ** 2)) * np.cos(ph - pc)
        mask[cosd > np.cos(1.5 * np.pi / 180)] = 0.
    ell = np.arange(lmax + 1.)
    cl = {'tt': np.where(ell >= 2, 6e3 / np.maximum(ell, 1.) ** 2 * np.exp(-(ell / 900.) ** 2), 0.),
          'ee': np.where(ell >= 2, 1.5 * np.exp(-((ell - 400.) / 450.) ** 2) + 0.02, 0.),
          'bb': np.where(ell >= 2, 2e-3 * np.ones_like(ell), 0.)}
    transf = hp.gauss_beam(12. / 60. / 180. * np.pi, lmax=lmax)
    vamin = np.sqrt(hp.nside2pixarea(nside, degrees=True)) * 60
    nlev_t, nlev_p = 35., 55.
    scan = 1. + 0.5 * z ** 2 + 0.3 * np.sin(th) * np.sin(2 * ph)             # smooth hit-count modulation, 0.7 .. 1.8
    ninv_t = mask * scan * (vamin / nlev_t) ** 2
    ninv_p = mask * (0.8 + 0.4 * np.sin(th) ** 2 * np.cos(ph) ** 2) * (vamin / nlev_p) ** 2
    t = hp.synalm(cl['tt'], lmax, rng)
    e, b = hp.synalm(cl['ee'], lmax, rng), hp.synalm(cl['bb'], lmax, rng)
    tmap = np.asarray(alm2map(hp.almxfl(t, transf), nside, lmax=lmax))
    q, u = alm2map_spin([hp.almxfl(e, transf), hp.almxfl(b, transf)], nside, 2, lmax)
    with np.errstate(divide='ignore'):
        sig_t = np.where(ninv_t > 0, 1. / np.sqrt(np.where(ninv_t > 0, ninv_t, 1.)), 0.)
        sig_p = np.where(ninv_p > 0, 1. / np.sqrt(np.where(ninv_p > 0, ninv_p, 1.)), 0.)
    tmap = (tmap + sig_t * rng.standard_normal(npix) + 50. + 30. * z) * mask   # a monopole and a dipole for the marginalisation to remove
    qmap = (np.asarray(q) + sig_p * rng.standard_normal(npix)) * mask
    umap = (np.asarray(u) + sig_p * rng.standard_normal(npix)) * mask
    return {'nside': nside, 'lmax': lmax, 'cl': cl, 'transf': transf, 'ninv_t': ninv_t, 'ninv_p': ninv_p, 'tmap': tmap,
            'qmap': qmap, 'umap': umap}
