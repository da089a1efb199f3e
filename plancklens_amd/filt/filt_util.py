"""Wrappers around filtering libraries, API of plancklens/filt/filt_util.py (`library_ftl` :39-103, `library_fml` :106-182,
`library_shuffle` :186-236): the same `ivfs` protocol (get_sim_tlm / elm / blm, get_sim_tmliklm / emliklm / bmliklm, get_ftl / fel /
fbl, get_tal, get_fmask, hashdict) seen through a re-weighting of the alms or a re-mapping of the simulation index.

One base class forwards the protocol; a wrapper only says what it does to an alm of a given field (`_alm`), to the isotropic
filter of a field (`_fl`) and to a simulation index (`_idx`).  Pure index / l- and m-weight bookkeeping on host arrays;
device-resident alms are passed through where the wrapped library offers them."""
import numpy as np

from .. import hp, utils

_ALMS = ('tlm', 'elm', 'blm', 'tmliklm', 'emliklm', 'bmliklm')


class _ivfs_view(object):
    def __init__(self, ivfs):
        self.ivfs = ivfs
        self.lib_dir = getattr(ivfs, 'lib_dir', None)

    # what a wrapper changes
    def _idx(self, idx):
        return idx

    def _alm(self, name, alm):
        return alm

    def _fl(self, field, fl):
        return fl

    def filter_sims(self, idxs, **kwargs):
        """block filtering of the wrapped library where it offers it (filt_cinv.library_cinv_sepTP.filter_sims); True if done"""
        inner = getattr(self.ivfs, 'filter_sims', None)
        return bool(inner is not None and inner([self._idx(i) for i in idxs], **kwargs))

    # the protocol, forwarded
    def get_fmask(self):
        return self.ivfs.get_fmask()

    def get_tal(self, a):
        return self.ivfs.get_tal(a)

    def get_ftl(self):
        return self._fl('t', self.ivfs.get_ftl())

    def get_fel(self):
        return self._fl('e', self.ivfs.get_fel())

    def get_fbl(self):
        return self._fl('b', self.ivfs.get_fbl())


def _forward(name):
    def get(self, idx):
        return self._alm(name, getattr(self.ivfs, 'get_sim_' + name)(self._idx(idx)))
    get.__name__ = 'get_sim_' + name
    get.__doc__ = "the wrapped library's get_sim_%s through this wrapper" % name
    return get


for _n in _ALMS:
    setattr(_ivfs_view, 'get_sim_' + _n, _forward(_n))


class _reweighted(_ivfs_view):
    """alms cut to a new band-limit and multiplied by per-field weight arrays"""

    def __init__(self, ivfs, lmax, wt, we, wb):
        assert min(len(wt), len(we), len(wb)) > lmax
        super(_reweighted, self).__init__(ivfs)
        self.lmax = lmax
        self._w = {'t': wt, 'e': we, 'b': wb}

    def hashdict(self):
        ret = {'ivfs': self.ivfs.hashdict()}
        ret.update({'filt_' + f: utils.clhash(self._w[f][:self.lmax + 1]) for f in 'teb'})
        return ret


class library_ftl(_reweighted):
    """Rescales the filtered alms of `ivfs` by isotropic functions of l: a_lm -> f_l a_lm (filt_util.py:39-103)."""

    def __init__(self, ivfs, lmax, lfilt_t, lfilt_e, lfilt_b):
        super(library_ftl, self).__init__(ivfs, lmax, lfilt_t, lfilt_e, lfilt_b)
        self.lfilt_t, self.lfilt_e, self.lfilt_b = lfilt_t, lfilt_e, lfilt_b

    def _alm(self, name, alm):
        return hp.almxfl(utils.alm_copy(alm, lmax=self.lmax), self._w[name[0]][:self.lmax + 1], inplace=True)

    def _fl(self, field, fl):
        return fl[:self.lmax + 1] * self._w[field][:self.lmax + 1]


class library_fml(_reweighted):
    """Rescales the filtered alms of `ivfs` by functions of the azimuthal order: a_lm -> f_m a_lm (filt_util.py:106-182).
    As in the reference, the inverse-variance filtered E and B alms take the *temperature* weights (filt_util.py:169-173; the
    Wiener-filtered ones take their own) -- results parity first."""
    _WEIGHT_OF = {'tlm': 't', 'elm': 't', 'blm': 't', 'tmliklm': 't', 'emliklm': 'e', 'bmliklm': 'b'}

    def __init__(self, ivfs, lmax, mfilt_t, mfilt_e, mfilt_b):
        super(library_fml, self).__init__(ivfs, lmax, mfilt_t, mfilt_e, mfilt_b)
        self.mfilt_t, self.mfilt_e, self.mfilt_b = mfilt_t, mfilt_e, mfilt_b

    @staticmethod
    def almxfm(alm, fm, lmax):
        """copy of alm at band-limit lmax with every entry multiplied by fm[m]"""
        ret = utils.alm_copy(alm, lmax=lmax)
        ret *= np.asarray(fm)[hp.Alm.getlm(lmax)[1]]
        return ret

    def _alm(self, name, alm):
        return self.almxfm(alm, self._w[self._WEIGHT_OF[name]], self.lmax)

    def _fl(self, field, fl):
        """f_l sqrt(<f_m>_l), <f_m>_l the mean of f_|m| over the 2l + 1 orders of l (the root: applies at the spectrum level)"""
        f = np.asarray(self._w[field][:self.lmax + 1], dtype=float)
        return fl[:self.lmax + 1] * np.sqrt((2. * np.cumsum(f) - f[0]) / (2. * np.arange(self.lmax + 1) + 1.))


class library_shuffle(_ivfs_view):
    """Filtering library with remapped simulation indices: idx -> idxs[idx] (filt_util.py:186-236).
    This is what makes ivfs1 != ivfs2 in the ds / ss estimator pairs (qest.py:327-332)."""

    def __init__(self, ivfs, idxs):
        super(library_shuffle, self).__init__(ivfs)
        self.idxs = idxs

    def hashdict(self):
        return {'ivfs': self.ivfs.hashdict(), 'idxs': self.idxs}

    def _idx(self, idx):
        return self.idxs[idx]

    def get_sim_alm_dev(self, name, idx):
        getter = getattr(self.ivfs, 'get_sim_alm_dev', None)
        return None if getter is None else getter(name, self.idxs[idx])
