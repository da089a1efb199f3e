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
