"""Index-remapping wrapper of simulation libraries, API of plancklens/sims/utils.py (`sim_lib_shuffle`)."""


class sim_lib_shuffle(object):
    """sim idx of this library is sim idxs[idx] of the wrapped one."""

    def __init__(self, sim_lib, idxs):
        self.sim_lib = sim_lib
        self.idxs = idxs

    def hashdict(self):
        return {'sim_lib': self.sim_lib.hashdict(), 'shuffled_idxs': self.idxs}

    def get_sim_tmap(self, idx):
        return self.sim_lib.get_sim_tmap(self.idxs[idx])

    def get_sim_pmap(self, idx):
        return self.sim_lib.get_sim_pmap(self.idxs[idx])
