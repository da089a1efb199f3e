"""Wrappers of simulation libraries, API of plancklens/sims/utils.py (`sim_lib_shuffle` :3-18, `sim_lib_add_sim` :20-56,
`sim_lib_add_dat` :59-95).  Maps may be numpy arrays or device tensors; the sums are formed out of place (the wrapped libraries'
arrays are never modified)."""
import numpy as np


class sim_lib_shuffle(object):
    """sim idx of this library is sim shuffle_dict[idx] of the wrapped one."""

    def __init__(self, sim_lib, shuffle_dict):
        self.sim_lib = sim_lib
        self._shuffle = shuffle_dict

    def hashdict(self):
        return {'sim_lib': self.sim_lib.hashdict(), 'shuffle': self._shuffle}

    def get_sim_tmap(self, idx):
        return self.sim_lib.get_sim_tmap(int(self._shuffle[idx]))

    def get_sim_pmap(self, idx):
        return self.sim_lib.get_sim_pmap(int(self._shuffle[idx]))

    def hint_pair(self, idx0, idx1):
        """forwarded (see sims.maps.cmb_maps.hint_pair)"""
        if hasattr(self.sim_lib, 'hint_pair'):
            self.sim_lib.hint_pair(int(self._shuffle[idx0]), int(self._shuffle[idx1]))


class _sim_lib_add(object):
    """Weighted sum of the maps of several libraries for the indices `self._adds(idx)` selects; the first library alone
    (times its weight) for the others."""
    _tag = None

    def __init__(self, sim_libs, weights=None):
        self.w = weights if weights is not None else np.ones(len(sim_libs))
        self.sim_libs = sim_libs

    def _adds(self, idx):
        raise NotImplementedError

    def _libs(self, idx):
        n = len(self.sim_libs) if self._adds(idx) else 1
        return list(zip(self.sim_libs[:n], self.w[:n]))

    def get_sim_tmap(self, idx):
        terms = [s.get_sim_tmap(idx) * w for s, w in self._libs(idx)]
        t = terms[0]
        for x in terms[1:]:
            t = t + x
        return t

    def get_sim_pmap(self, idx):
        q = u = None
        for s, w in self._libs(idx):
            _q, _u = s.get_sim_pmap(idx)
            q = _q * w if q is None else q + _q * w
            u = _u * w if u is None else u + _u * w
        return q, u

    def hashdict(self):
        ret = {'lib': self._tag}
        for i, (s, w) in enumerate(zip(self.sim_libs, self.w)):
            ret['sim_lib ' + str(i)] = s.hashdict()
            ret['w ' + str(i)] = w
        return ret


class sim_lib_add_sim(_sim_lib_add):
    """Added simulation libraries; the sum only for simulation (idx >= 0) indices, the first library alone for the data."""
    _tag = 'add_sim'

    def _adds(self, idx):
        return idx >= 0


class sim_lib_add_dat(_sim_lib_add):
    """Added simulation libraries; the sum only for the data (idx < 0), the first library alone for simulations."""
    _tag = 'add_dat'

    def _adds(self, idx):
        return idx < 0
