"""Host utilities mirroring the hot-path subset of plancklens/utils.py (alm_copy :19-35, clhash :115-124,
mchash :126-130, cli :132-138, hash_check :144-180, camb_clfile :308-333, enumerate_progress :94-113)."""
import hashlib
import sys
import time

import numpy as np

from . import hp


def alm_copy(alm, lmax=None):
    """Copy of a healpy alm array, optionally truncated to a smaller lmax (utils.py:19-35)."""
    alm = np.asarray(alm)
    alm_lmax = hp.Alm.getlmax(alm.size)
    assert alm_lmax >= 0, alm.size
    if lmax is None or lmax == alm_lmax:
        return np.copy(alm)
    assert lmax <= alm_lmax, (lmax, alm_lmax)
    ret = np.zeros(hp.Alm.getsize(lmax), dtype=complex)
    for m in range(lmax + 1):
        o, i = hp.Alm.getidx(lmax, m, m), hp.Alm.getidx(alm_lmax, m, m)
        ret[o:o + lmax - m + 1] = alm[i:i + lmax - m + 1]
    return ret


def enumerate_progress(lst, label=''):
    """Progress bar over an iterable, yielding (index, value) (utils.py:94-113)."""
    t0 = time.time()
    n = len(lst)
    for i, v in enumerate(lst):
        yield i, v
        if n and int(100. * (i + 1) / n) > int(100. * i / n):
            dt = time.time() - t0
            sys.stdout.write("\r [%02d:%02d:%02d] %s %s> %02d%%" % (dt // 3600, (dt % 3600) // 60, dt % 60, label,
                                                                   int(10. * (i + 1) / n) * '-', int(100. * (i + 1) / n)))
            sys.stdout.flush()
    sys.stdout.write("\n")
    sys.stdout.flush()


def clhash(cl, dtype=np.float16):
    """SHA-1 of the array cast to low precision (machine-independent cache keys, utils.py:115-124)."""
    with np.errstate(over='ignore'):  # values beyond the float16 range hash as inf, as in the reference; no warning per hash
        return hashlib.sha1(np.copy(np.asarray(cl).astype(dtype), order='C')).hexdigest()


def mchash(cl):
    """Order-independent hash of an integer array (simulation indices, utils.py:126-130)."""
    return hashlib.sha1(np.copy(np.sort(cl), order='C')).hexdigest()


def cli(cl):
    """Pseudo-inverse of a non-negative array (utils.py:132-138)."""
    cl = np.asarray(cl)
    ret = np.zeros_like(cl)
    pos = cl > 0
    ret[pos] = 1. / cl[pos]
    return ret


def joincls(cls_list):
    n = min(len(cl) for cl in cls_list)
    return np.prod(np.array([cl[:n] for cl in cls_list]), axis=0)


def hash_check(hash1, hash2, ignore=('lib_dir', 'prefix'), keychain=(), fn=None):
    """Asserts equality of two (nested) hash dictionaries (utils.py:144-180)."""
    keys1 = [k for k in hash1.keys() if k not in ignore]
    keys2 = [k for k in hash2.keys() if k not in ignore]
    for key in set(keys1).union(keys2):
        if key not in hash1 or key not in hash2:
            raise KeyError("Cannot find key %s in hashdict %s" % (key, fn))
        v1, v2 = hash1[key], hash2[key]
        where = 'hash check failed (%s) at %s' % (fn, '/'.join(list(keychain) + [str(key)]))
        assert type(v1) == type(v2), where + ': types %s vs %s' % (type(v1), type(v2))
        if isinstance(v1, dict):
            hash_check(v1, v2, ignore=ignore, keychain=tuple(keychain) + (str(key),), fn=fn)
        elif isinstance(v1, np.ndarray):
            assert np.allclose(v1, v2), where + ': unequal arrays'
        else:
            assert v1 == v2, where + ': %s vs %s' % (v1, v2)


def camb_clfile(fname, lmax=None):
    """CAMB lensedCls / lenspotentialCls text file -> dict of C_l arrays (D_l factors removed), utils.py:308-333."""
    cols = np.loadtxt(fname).transpose()
    ell = cols[0].astype(int)
    if lmax is None:
        lmax = ell[-1]
    assert ell[-1] >= lmax, (ell[-1], lmax)
    sel = ell <= lmax
    w = ell * (ell + 1) / (2. * np.pi)
    cls = {}
    for i, k in enumerate(['tt', 'ee', 'bb', 'te']):
        cls[k] = np.zeros(lmax + 1)
        cls[k][ell[sel]] = cols[i + 1][sel] / w[sel]
    if len(cols) > 5:
        el = ell[sel].astype(float)
        wpp = el ** 2 * (el + 1) ** 2 / (2. * np.pi)
        wpx = np.sqrt(el ** 3 * (el + 1.) ** 3) / (2. * np.pi)
        for i, (k, wk) in enumerate(zip(['pp', 'pt', 'pe'], [wpp, wpx, wpx])):
            cls[k] = np.zeros(lmax + 1)
            cls[k][ell[sel]] = cols[5 + i][sel] / wk
    return cls


class stats(object):
    """Running mean / covariance over simulations (utils.py:181-260, the part qecl uses)."""

    def __init__(self, size, xcoord=None, docov=True):
        self.N = 0
        self.size = size
        self.sum = np.zeros(self.size)
        if docov:
            self.mom = np.zeros((self.size, self.size))
        self.xcoord = xcoord
        self.docov = docov

    def add(self, v):
        assert v.shape == (self.size,), "input not understood"
        self.sum += v
        if self.docov:
            self.mom += np.outer(v, v)
        self.N += 1

    def mean(self):
        assert self.N > 0
        return self.sum / float(self.N)

    avg = mean

    def cov(self):
        assert self.docov and self.N > 0
        if self.N == 1:
            return np.zeros((self.size, self.size))
        mean = self.mean()
        return self.mom / (self.N - 1.) - self.N / (self.N - 1.) * np.outer(mean, mean)

    def sigmas(self):
        return np.sqrt(np.diagonal(self.cov()))

    def sigmas_on_mean(self):
        assert self.N > 0
        return self.sigmas() / np.sqrt(self.N)

    def corrcoeffs(self):
        sig = self.sigmas()
        return self.cov() / np.outer(sig, sig)

    def inverse(self, bias_p=None):
        """inverse covariance, de-biased for the finite number of samples ((N - size - 2) / (N - 1) unless bias_p is given)"""
        assert self.N > self.size, "Non invertible cov.matrix"
        if bias_p is None:
            bias_p = (self.N - self.size - 2.) / (self.N - 1)
        return bias_p * np.linalg.inv(self.cov())

    def get_chisq(self, data):
        """(data - mean)^t C^-1 (data - mean)"""
        assert data.size == self.size, (data.size, self.size)
        dx = data - self.mean()
        return float(dx @ self.inverse() @ dx)

    def get_chisq_pte(self, data):
        from scipy.stats import chi2
        return chi2.sf(self.get_chisq(data), self.N - 1)

    def rebin_that_nooverlap(self, orig_coord, lmins, lmaxs, weights=None):
        """stats of the weighted bin averages over the non-overlapping bins [lmins[k], lmaxs[k]] of the coordinate orig_coord"""
        lmins, lmaxs = np.asarray(lmins), np.asarray(lmaxs)
        assert orig_coord.size == self.size and lmins.size == lmaxs.size, "Incompatible input"
        assert np.all(np.diff(lmins) > 0.) and np.all(np.diff(lmaxs) > 0.), "This only for non overlapping bins."
        w = np.ones(self.size) if weights is None else np.asarray(weights)
        assert w.size == self.size and self.size > lmins.size, "incompatible input"
        tmat = np.zeros((lmins.size, self.size))
        for k, (lo, hi) in enumerate(zip(lmins, lmaxs)):
            sel = (orig_coord >= lo) & (orig_coord <= hi)
            if np.any(sel):
                tmat[k, sel] = w[sel] / np.sum(w[sel])
        ret = stats(lmins.size, xcoord=0.5 * (lmins[:-1] + lmaxs[1:]))
        ret.sum, ret.mom, ret.N = tmat @ self.sum, tmat @ self.mom @ tmat.T, self.N
        return ret


def cl_inverse(cls):
    """Per-multipole pseudo-inverse of the symmetric T, E, B spectral matrix given as a dictionary ('tt', 'ee', 'bb',
    'te', 'tb', 'eb'; missing entries are zero, shorter arrays are zero-padded); returns the non-zero entries of the
    inverse under the same keys (utils.py:336-365)."""
    lmax = max(len(cl) for cl in cls.values()) - 1
    order = ['t', 'e', 'b']
    mat = np.zeros((lmax + 1, 3, 3))
    for i, a in enumerate(order):
        for j, b in enumerate(order[i:], start=i):
            cl = np.asarray(cls.get(a + b, [0.]), dtype=float)
            n = min(len(cl), lmax + 1)
            mat[:n, i, j] = cl[:n]
            mat[:n, j, i] = cl[:n]
    inv = np.linalg.pinv(mat)
    ret = {}
    for k, (i, j) in zip(['tt', 'ee', 'bb', 'te', 'tb', 'eb'], [(0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2)]):
        if np.any(inv[:, i, j]):
            ret[k] = inv[:, i, j].copy()
    return ret


def extcl(lmax, cl):
    """cl zero-padded or truncated to lmax + 1 entries (utils.py:367-373)"""
    cl = np.asarray(cl)
    if len(cl) > lmax:
        return cl[:lmax + 1]
    out = np.zeros(lmax + 1)
    out[:len(cl)] = cl
    return out


def _cldict2arr(cls_dict):
    """(3, 3, lmax + 1) array of a T, E, B spectra dictionary (symmetric; missing spectra zero)"""
    n = max(len(cl) for cl in cls_dict.values())
    return np.array([[extcl(n - 1, cls_dict.get(a + b, cls_dict.get(b + a, np.zeros(1)))) for b in 'teb'] for a in 'teb'], dtype=float)


def cls_dot(cls_list, ret_dict=False):
    """Per-multipole product of T, E, B spectral matrices, each a dictionary or a (3, 3, lmax + 1) array (utils.py:383-416);
    ret_dict: the non-zero entries of the upper triangle as a dictionary."""
    mats = [_cldict2arr(c) if isinstance(c, dict) else np.asarray(c) for c in cls_list]
    ret = mats[-1]
    for m in mats[-2::-1]:
        ret = np.einsum('ikl,kjl->ijl', m, ret)
    if not ret_dict:
        return ret
    out = {}
    for k, (i, j) in zip(['tt', 'ee', 'bb', 'te', 'tb', 'eb'], [(0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2)]):
        if np.any(ret[i, j]):
            out[k] = ret[i, j].copy()
    return out


def alm2rlm(alm):
    """complex alm -> real harmonic coefficients (utils.py:37-52; the dense preconditioners' layout, qcinv.dense.alm2rlm)"""
    from .qcinv import dense
    return dense.alm2rlm(np.asarray(alm))


def rlm2alm(rlm):
    from .qcinv import dense
    return dense.rlm2alm(np.asarray(rlm))
