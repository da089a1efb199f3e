"""Constructor-level stand-in for plancklens/n1/n1.py:102-136 (`library_n1`).

The N1 lensing bias is post-processing above the hot path ("must at least construct cheaply"; its
arithmetic lives in the reference's n1f Fortran extension, DESIGN.md "Out of scope").  Parameter files instantiate the
library next to the estimator libraries (params/idealized_example.py:127), so the constructor is provided with the
reference's arguments, directory layout and hash file; asking it for an N1 raises."""
import os
import pickle as pk

import numpy as np

from ..helpers import mpi, sql
from ..utils import clhash, hash_check


def _default_lps(lmaxphi):
    """multipoles at which the flat-sky integral is sampled (n1.py:104-116): dense at low L, sparser above"""
    lps = [1] + list(range(2, 111, 10))
    lps += list(range(lps[-1] + 30, 580, 30))
    lps += list(range(lps[-1] + 100, lmaxphi // 2, 100))
    lps += list(range(lps[-1] + 300, lmaxphi, 300))
    if lps[-1] != lmaxphi:
        lps.append(lmaxphi)
    return np.array(lps)


class library_n1(object):
    def __init__(self, lib_dir, cltt, clte, clee, lmaxphi=2500, dL=10, lps=None):
        self.lps = _default_lps(lmaxphi) if lps is None else np.asarray(lps)
        self.dL = dL
        self.cltt, self.clte, self.clee = cltt, clte, clee
        self.lmaxphi = self.lps[-1]
        self.n1 = {}
        self.lib_dir = lib_dir
        fn = os.path.join(lib_dir, 'n1_hash.pk')
        if mpi.rank == 0:
            if not os.path.exists(lib_dir):
                os.makedirs(lib_dir)
            if not os.path.exists(fn):
                pk.dump(self.hashdict(), open(fn, 'wb'), protocol=2)
        mpi.barrier()
        hash_check(self.hashdict(), pk.load(open(fn, 'rb')), fn=fn)
        self.npdb = sql.npdb(os.path.join(lib_dir, 'npdb.db'))

    def hashdict(self):
        return {'cltt': clhash(self.cltt), 'clte': clhash(self.clte), 'clee': clhash(self.clee), 'dL': self.dL, 'lps': self.lps}

    def get_n1(self, *args, **kwargs):
        raise NotImplementedError('the N1 bias needs the n1f Fortran kernels of the reference (plancklens/n1/n1f.f90); '
                                  'it is outside the hot path this package implements (DESIGN.md, out of scope)')

    get_n1_x_p = get_n1
