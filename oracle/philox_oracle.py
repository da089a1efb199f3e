"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the device-side standard-normal generator (pl_map_add_normal / pl_alm_unit_phases,
include/plshts.h): Philox4x32-10 (Salmon, Moraes, Dror & Shaw 2011; the public algorithm of the Random123 library, whose
known-answer vectors tests/test_sims.py checks) -> two 53-bit uniforms -> Box-Muller.  The reference draws its phases with numpy's
generator on the host (plancklens/sims/phas.py:125-195); the device generator is this repository's own, so what is pinned here is its
definition, bit for bit on the integer part and to rounding on the transcendental part.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
import numpy as np

_M0, _M1, _W0, _W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr: four (arrays of) 32-bit words, key: two 32-bit words -> the four output words (uint64 arrays holding 32-bit values)"""
    c = [np.asarray(x, dtype=np.uint64) for x in ctr]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    for _ in range(10):
        p0, p1 = np.uint64(_M0) * c[0], np.uint64(_M1) * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ k0, p1 & _MASK, (p0 >> np.uint64(32)) ^ c[3] ^ k1, p0 & _MASK]
        k0, k1 = (k0 + np.uint64(_W0)) & _MASK, (k1 + np.uint64(_W1)) & _MASK
    return c


def normal_pairs(key, npairs, tag):
    """(n_2p, n_2p+1) for p < npairs: the deviates of positions 2 p and 2 p + 1 under the 64-bit `key`; tag 0 = pixels, 1 = alm"""
    p = np.arange(npairs, dtype=np.uint64)
    r = philox4x32_10([p & _MASK, p >> np.uint64(32), np.full(npairs, tag, dtype=np.uint64), np.zeros(npairs, dtype=np.uint64)],
                      (int(key) & 0xFFFFFFFF, (int(key) >> 32) & 0xFFFFFFFF))
    a = (r[0] | (r[1] << np.uint64(32))) >> np.uint64(11)
    b = (r[2] | (r[3] << np.uint64(32))) >> np.uint64(11)
    u1 = (a + np.uint64(1)).astype(np.float64) * 2. ** -53
    u2 = b.astype(np.float64) * 2. ** -53
    rad = np.sqrt(-2. * np.log(u1))
    return rad * np.cos(2. * np.pi * u2), rad * np.sin(2. * np.pi * u2)


def normals(key, n):
    """the deviates of positions 0 .. n - 1 (what pl_map_add_normal adds, times sigma)"""
    c, s = normal_pairs(key, (n + 1) // 2, 0)
    return np.stack([c, s], axis=1).reshape(-1)[:n]


def unit_phases(key, lmax):
    """pl_alm_unit_phases: complex128 alm, healpy layout"""
    n = (lmax + 1) * (lmax + 2) // 2
    c, s = normal_pairs(key, n, 1)
    alm = (c + 1j * s) * np.sqrt(0.5)
    alm[:lmax + 1] = c[:lmax + 1]
    return alm
