"""Random-phase libraries, API of plancklens/sims/phas.py (`lib_phas` :178-195, `pix_lib_phas` :137-155).

The reference stores numpy RNG states in sqlite so that a simulation can be regenerated; here the same
guarantee (get_sim(idx, idf) is a pure function of (library seed, idf, idx)) comes from counter-based seeding
of numpy's Philox-free default generator: default_rng([seed, idf, idx]).  The seed is part of hashdict().

`pix_lib_phas_dev` / `lib_phas_dev` are device-side variants: the same interface and the same
pure-function guarantee, with the numbers drawn on the GPU by torch's counter-based Philox generator seeded from
(library seed, idf, idx) and returned as device tensors -- at GPU reconstruction speeds the host draw of 3 npix normals
and their upload would be the bottleneck of a Monte-Carlo run.  The streams differ from the numpy ones (and so the
realisations), which the hash records ('rng' entry).
"""
import os
import pickle as pk

import numpy as np

from .. import hp, utils
from ..helpers import mpi


class _seeded_lib(object):
    def __init__(self, lib_dir, seed=None, nsims_max=None):
        self.lib_dir = lib_dir
        self.nmax = nsims_max
        if mpi.rank == 0 and not os.path.exists(lib_dir):
            os.makedirs(lib_dir)
        mpi.barrier()
        fn_seed = os.path.join(lib_dir, 'seed.pk')
        if mpi.rank == 0 and not os.path.exists(fn_seed):
            pk.dump(int(np.random.SeedSequence().entropy % (2 ** 31)) if seed is None else int(seed), open(fn_seed, 'wb'), protocol=2)
        mpi.barrier()
        self.seed = pk.load(open(fn_seed, 'rb'))
        assert seed is None or seed == self.seed, 'library at %s was created with another seed' % lib_dir
        fn_hash = os.path.join(lib_dir, 'sim_hash.pk')
        if mpi.rank == 0 and not os.path.exists(fn_hash):
            pk.dump(self.hashdict(), open(fn_hash, 'wb'), protocol=2)
        mpi.barrier()
        utils.hash_check(pk.load(open(fn_hash, 'rb')), self.hashdict(), ignore=['lib_dir'], fn=fn_hash)

    def _rng(self, idf, idx):
        if self.nmax is not None:
            assert idx < self.nmax
        return np.random.default_rng([self.seed, int(idf), int(idx)])


class pix_lib_phas(_seeded_lib):
    """Unit-variance white pixel maps (phas.py:137-155)."""

    def __init__(self, lib_dir, nfields, shape, seed=None, **kwargs):
        self.nfields = nfields
        self.shape = shape
        super(pix_lib_phas, self).__init__(lib_dir, seed=seed, **kwargs)

    def get_sim(self, idx, idf=None, phas_only=False):
        if idf is not None:
            assert idf < self.nfields, (idf, self.nfields)
            return None if phas_only else self._rng(idf, idx).standard_normal(self.shape)
        return np.array([self.get_sim(idx, idf=i) for i in range(self.nfields)])

    def hashdict(self):
        return {'nfields': self.nfields, 'shape': self.shape, 'seed': self.seed}


class lib_phas(_seeded_lib):
    """Unit-variance harmonic phases: complex normal alm with real m = 0 column (phas.py:157-195)."""

    def __init__(self, lib_dir, nfields, lmax, seed=None, **kwargs):
        self.nfields = nfields
        self.lmax = lmax
        super(lib_phas, self).__init__(lib_dir, seed=seed, **kwargs)

    def get_sim(self, idx, idf=None, phas_only=False):
        if idf is not None:
            assert idf < self.nfields, (idf, self.nfields)
            if phas_only:
                return None
            rng = self._rng(idf, idx)
            n = hp.Alm.getsize(self.lmax)
            alm = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) / np.sqrt(2.)
            alm[:self.lmax + 1] = np.sqrt(2.) * alm[:self.lmax + 1].real
            return alm
        return np.array([self.get_sim(idx, idf=i) for i in range(self.nfields)])

    def hashdict(self):
        return {'nfields': self.nfields, 'lmax': self.lmax, 'seed': self.seed}


def _dev_generator(seed, idf, idx):
    import torch
    g = torch.Generator(device='cuda')
    # SplitMix-style mix of the three integers into one 63-bit seed (distinct streams for distinct (seed, idf, idx))
    z = (int(seed) * 0x9E3779B97F4A7C15 + int(idf) * 0xBF58476D1CE4E5B9 + int(idx) * 0x94D049BB133111EB + 0x2545F4914F6CDD1D) & (2 ** 64 - 1)
    z ^= z >> 31
    g.manual_seed(z & (2 ** 63 - 1))
    return g


class pix_lib_phas_dev(pix_lib_phas):
    """pix_lib_phas drawn on the device: get_sim returns float64 CUDA tensors."""

    def get_sim(self, idx, idf=None, phas_only=False):
        import torch
        if idf is not None:
            assert idf < self.nfields, (idf, self.nfields)
            if self.nmax is not None:
                assert idx < self.nmax
            if phas_only:
                return None
            return torch.randn(self.shape, generator=_dev_generator(self.seed, idf, idx), dtype=torch.float64, device='cuda')
        return torch.stack([self.get_sim(idx, idf=i) for i in range(self.nfields)])

    def hashdict(self):
        return {'nfields': self.nfields, 'shape': self.shape, 'seed': self.seed, 'rng': 'torch-philox-cuda'}


class lib_phas_dev(lib_phas):
    """lib_phas drawn on the device: get_sim returns complex128 CUDA tensors (unit variance, real m = 0 column)."""

    def get_sim(self, idx, idf=None, phas_only=False):
        import torch
        if idf is not None:
            assert idf < self.nfields, (idf, self.nfields)
            if self.nmax is not None:
                assert idx < self.nmax
            if phas_only:
                return None
            n = hp.Alm.getsize(self.lmax)
            ri = torch.randn((n, 2), generator=_dev_generator(self.seed, idf, idx), dtype=torch.float64, device='cuda')
            ri *= np.sqrt(0.5)
            ri[:self.lmax + 1, 0] *= np.sqrt(2.)
            ri[:self.lmax + 1, 1] = 0.
            return torch.view_as_complex(ri)
        return torch.stack([self.get_sim(idx, idf=i) for i in range(self.nfields)])

    def hashdict(self):
        return {'nfields': self.nfields, 'lmax': self.lmax, 'seed': self.seed, 'rng': 'torch-philox-cuda'}
