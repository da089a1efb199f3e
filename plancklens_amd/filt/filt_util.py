"""Wrappers around filtering libraries, API of plancklens/filt/filt_util.py (`library_ftl` :39-103, `library_fml` :106-182,
`library_shuffle` :186-236).  Pure index / l- and m-weight bookkeeping; device-resident alms are passed through."""
import numpy as np

from .. import dev, hp


class library_ftl(object):
    """Rescales the filtered alms of `ivfs` by isotropic functions of l (filt_util.py:39-103)."""

    def __init__(self, ivfs, lmax, lfilt_t, lfilt_e, lfilt_b):
        assert len(lfilt_t) > lmax and len(lfilt_e) > lmax and len(lfilt_b) > lmax
        self.ivfs = ivfs
        self.lmax = lmax
        self.lfilt_t, self.lfilt_e, self.lfilt_b = lfilt_t, lfilt_e, lfilt_b
        self.lib_dir = ivfs.lib_dir

    def hashdict(self):
        from .. import utils
        return {'ivfs': self.ivfs.hashdict(), 'filt_t': utils.clhash(self.lfilt_t[:self.lmax + 1]),
                'filt_e': utils.clhash(self.lfilt_e[:self.lmax + 1]), 'filt_b': utils.clhash(self.lfilt_b[:self.lmax + 1])}

    def get_fmask(self):
        return self.ivfs.get_fmask()

    def get_tal(self, a):
        return self.ivfs.get_tal(a)

    def get_ftl(self):
        return self.ivfs.get_ftl()[:self.lmax + 1] * self.lfilt_t[:self.lmax + 1]

    def get_fel(self):
        return self.ivfs.get_fel()[:self.lmax + 1] * self.lfilt_e[:self.lmax + 1]

    def get_fbl(self):
        return self.ivfs.get_fbl()[:self.lmax + 1] * self.lfilt_b[:self.lmax + 1]

    def _resc(self, alm, fl):
        from ..utils import alm_copy
        return hp.almxfl(alm_copy(alm, lmax=self.lmax), fl[:self.lmax + 1], inplace=True)

    def get_sim_tlm(self, idx):
        return self._resc(self.ivfs.get_sim_tlm(idx), self.lfilt_t)

    def get_sim_elm(self, idx):
        return self._resc(self.ivfs.get_sim_elm(idx), self.lfilt_e)

    def get_sim_blm(self, idx):
        return self._resc(self.ivfs.get_sim_blm(idx), self.lfilt_b)

    def get_sim_tmliklm(self, idx):
        return self._resc(self.ivfs.get_sim_tmliklm(idx), self.lfilt_t)

    def get_sim_emliklm(self, idx):
        return self._resc(self.ivfs.get_sim_emliklm(idx), self.lfilt_e)

    def get_sim_bmliklm(self, idx):
        return self._resc(self.ivfs.get_sim_bmliklm(idx), self.lfilt_b)


class library_fml(object):
    """Rescales the filtered alms of `ivfs` by functions of the azimuthal order: a_lm -> f_m a_lm (filt_util.py:106-182).
    As in the reference, the inverse-variance filtered E and B alms take the *temperature* weights mfilt_t (filt_util.py:169-173;
    the Wiener-filtered ones take their own) -- results parity first."""

    def __init__(self, ivfs, lmax, mfilt_t, mfilt_e, mfilt_b):
        assert len(mfilt_t) > lmax and len(mfilt_e) > lmax and len(mfilt_b) > lmax
        self.ivfs = ivfs
        self.lmax = lmax
        self.mfilt_t, self.mfilt_e, self.mfilt_b = mfilt_t, mfilt_e, mfilt_b
        self.lib_dir = ivfs.lib_dir

    def hashdict(self):
        from .. import utils
        return {'ivfs': self.ivfs.hashdict(), 'filt_t': utils.clhash(self.mfilt_t[:self.lmax + 1]),
                'filt_e': utils.clhash(self.mfilt_e[:self.lmax + 1]), 'filt_b': utils.clhash(self.mfilt_b[:self.lmax + 1])}

    def get_fmask(self):
        return self.ivfs.get_fmask()

    def get_tal(self, a):
        return self.ivfs.get_tal(a)

    @staticmethod
    def almxfm(alm, fm, lmax):
        """copy of alm at band-limit lmax with every entry multiplied by fm[m]"""
        from ..utils import alm_copy
        ret = alm_copy(alm, lmax=lmax)
        ret *= np.asarray(fm)[hp.Alm.getlm(lmax)[1]]
        return ret

    def _isotropic_equivalent(self, fl, fm):
        """f_l sqrt(<f_m>_l), <f_m>_l the mean of f_|m| over the 2l + 1 orders of l (the root: applies at the spectrum level)"""
        f = np.asarray(fm[:self.lmax + 1], dtype=float)
        mean = (2. * np.cumsum(f) - f[0]) / (2. * np.arange(self.lmax + 1) + 1.)
        return fl[:self.lmax + 1] * np.sqrt(mean)

    def get_ftl(self):
        return self._isotropic_equivalent(self.ivfs.get_ftl(), self.mfilt_t)

    def get_fel(self):
        return self._isotropic_equivalent(self.ivfs.get_fel(), self.mfilt_e)

    def get_fbl(self):
        return self._isotropic_equivalent(self.ivfs.get_fbl(), self.mfilt_b)

    def get_sim_tlm(self, idx):
        return self.almxfm(self.ivfs.get_sim_tlm(idx), self.mfilt_t, self.lmax)

    def get_sim_elm(self, idx):
        return self.almxfm(self.ivfs.get_sim_elm(idx), self.mfilt_t, self.lmax)

    def get_sim_blm(self, idx):
        return self.almxfm(self.ivfs.get_sim_blm(idx), self.mfilt_t, self.lmax)

    def get_sim_tmliklm(self, idx):
        return self.almxfm(self.ivfs.get_sim_tmliklm(idx), self.mfilt_t, self.lmax)

    def get_sim_emliklm(self, idx):
        return self.almxfm(self.ivfs.get_sim_emliklm(idx), self.mfilt_e, self.lmax)

    def get_sim_bmliklm(self, idx):
        return self.almxfm(self.ivfs.get_sim_bmliklm(idx), self.mfilt_b, self.lmax)


class library_shuffle(object):
    """Filtering library with remapped simulation indices: idx -> idxs[idx] (filt_util.py:186-236).
    This is what makes ivfs1 != ivfs2 in the ds / ss estimator pairs (qest.py:327-332)."""

    def __init__(self, ivfs, idxs):
        self.ivfs = ivfs
        self.idxs = idxs

    def hashdict(self):
        return {'ivfs': self.ivfs.hashdict(), 'idxs': self.idxs}

    def get_fmask(self):
        return self.ivfs.get_fmask()

    def get_tal(self, a):
        return self.ivfs.get_tal(a)

    def get_ftl(self):
        return self.ivfs.get_ftl()

    def get_fel(self):
        return self.ivfs.get_fel()

    def get_fbl(self):
        return self.ivfs.get_fbl()

    def get_sim_alm_dev(self, name, idx):
        getter = getattr(self.ivfs, 'get_sim_alm_dev', None)
        return None if getter is None else getter(name, self.idxs[idx])

    def get_sim_tlm(self, idx):
        return self.ivfs.get_sim_tlm(self.idxs[idx])

    def get_sim_elm(self, idx):
        return self.ivfs.get_sim_elm(self.idxs[idx])

    def get_sim_blm(self, idx):
        return self.ivfs.get_sim_blm(self.idxs[idx])

    def get_sim_tmliklm(self, idx):
        return self.ivfs.get_sim_tmliklm(self.idxs[idx])

    def get_sim_emliklm(self, idx):
        return self.ivfs.get_sim_emliklm(self.idxs[idx])

    def get_sim_bmliklm(self, idx):
        return self.ivfs.get_sim_bmliklm(self.idxs[idx])
