"""Gauss-Legendre quadrature and Wigner small-d series on the host (numpy), interface of the reference's Fortran module
plancklens/wigners/wigners.f90 (`get_xgwg` :132-184, `wignerpos` :566-624, `wignercoeff` :628-685) as it is called from
utils_spin.wignerc (utils_spin.py:52-93).  One-dimensional and cheap (a few 10^7 flops per call): the analytic response
and N0 calculations (qresp.get_response, nhl.get_nhl) do not need the GPU, they need the Fortran dependency gone.

    wignerpos(cl, x, s1, s2)      = sum_l cl_l (2l + 1) / (4 pi) d^l_{s1 s2}(x)        (x = cos theta)
    wignercoeff(xi, x, s1, s2, L) = 2 pi sum_i xi_i d^l_{s1 s2}(x_i),  l = 0 .. L       (xi includes the quadrature weights)

d^l_{s1 s2} is obtained by the standard upward three-term recursion in l from its closed form at l0 = max(|s1|, |s2|); the
sign convention is pinned against the reference's Fortran build in tests/test_wigners.py (golden vectors)."""
import numpy as np
from scipy.special import gammaln


def get_xgwg(x1, x2, n):
    """Gauss-Legendre nodes and weights on [x1, x2] (ascending nodes).  Newton iterations on P_n from the
    Tricomi initial guess, all nodes at once."""
    n = int(n)
    k = np.arange(1, n + 1, dtype=float)
    th = np.pi * (k - 0.25) / (n + 0.5)
    z = (1. - (n - 1.) / (8. * n ** 3) - 1. / (384. * n ** 4) * (39. - 28. / np.sin(th) ** 2)) * np.cos(th)
    for _ in range(100):
        p0, p1 = np.ones_like(z), z.copy()
        for j in range(2, n + 1):
            p0, p1 = p1, ((2. * j - 1.) * z * p1 - (j - 1.) * p0) / j
        pp = n * (z * p1 - p0) / (z * z - 1.)
        dz = p1 / pp
        z = z - dz
        if np.max(np.abs(dz)) < 1e-15:
            break
    p0, p1 = np.ones_like(z), z.copy()
    for j in range(2, n + 1):
        p0, p1 = p1, ((2. * j - 1.) * z * p1 - (j - 1.) * p0) / j
    pp = n * (z * p1 - p0) / (z * z - 1.)
    w = 2. / ((1. - z * z) * pp * pp)
    xm, xl = 0.5 * (x2 + x1), 0.5 * (x2 - x1)
    x = xm - xl * z  # z is descending (theta ascending), so x ascends
    return x, xl * w


def _d_start(s1, s2, x):
    """d^{l0}_{s1 s2}(x) at l0 = max(|s1|, |s2|), with the symmetries d_{m n} = (-1)^{m - n} d_{n m} = d_{-n, -m} used to
    bring the larger index to the top: d^j_{j n} = sqrt(C(2j, j + n)) cos^{j + n}(th/2) (-sin(th/2))^{j - n}."""
    m, n, sgn = s1, s2, 1.
    if abs(n) > abs(m):           # d_{m n} = (-1)^{m - n} d_{n m}
        m, n = n, m
        sgn *= (-1.) ** (s1 - s2)
    if m < 0:                     # d_{m n} = (-1)^{m - n} d_{-m, -n}
        sgn *= (-1.) ** (m - n)
        m, n = -m, -n
    j = m
    lognorm = 0.5 * (gammaln(2 * j + 1) - gammaln(j + n + 1) - gammaln(j - n + 1))
    ch, sh = np.sqrt(0.5 * (1. + x)), np.sqrt(0.5 * (1. - x))
    return sgn * np.exp(lognorm) * ch ** (j + n) * (-sh) ** (j - n)


def _d_series(lmax, s1, s2, x):
    """Generator of (l, d^l_{s1 s2}(x)) for l = l0 .. lmax."""
    l0 = max(abs(s1), abs(s2))
    if l0 > lmax:
        return
    dm1 = np.zeros_like(x)
    d = _d_start(s1, s2, x)
    yield l0, d
    mn = float(s1 * s2)
    for l in range(l0, lmax):
        # (l + 1 - |.|): l (l+1)-type recursion of d^l_{mn}; for l = 0 (s1 = s2 = 0) it reduces to P_1 = x
        if l == 0:
            dn = x * d
        else:
            c1 = np.sqrt(((l + 1.) ** 2 - s1 * s1) * ((l + 1.) ** 2 - s2 * s2))
            c0 = np.sqrt((l * l - s1 * s1) * (l * l - s2 * s2)) if l > l0 else 0.
            dn = ((2. * l + 1.) * (l * (l + 1.) * x - mn) * d - (l + 1.) * c0 * dm1) / (l * c1)
        dm1, d = d, dn
        yield l + 1, d


def wignerpos(cl, x, s1, s2):
    """sum_l cl_l (2l + 1) / (4 pi) d^l_{s1 s2}(x)."""
    cl = np.asarray(cl, dtype=float)
    x = np.asarray(x, dtype=float)
    xi = np.zeros_like(x)
    for l, d in _d_series(len(cl) - 1, int(s1), int(s2), x):
        if cl[l] != 0.:
            xi += (cl[l] * (2. * l + 1.) / (4. * np.pi)) * d
    return xi


def wignercoeff(xi, x, s1, s2, lmax):
    """2 pi sum_i xi_i d^l_{s1 s2}(x_i) for l = 0 .. lmax (xi already holds the quadrature weights)."""
    xi = np.asarray(xi, dtype=float)
    x = np.asarray(x, dtype=float)
    cl = np.zeros(int(lmax) + 1, dtype=float)
    for l, d in _d_series(int(lmax), int(s1), int(s2), x):
        cl[l] = 2. * np.pi * np.dot(xi, d)
    return cl
