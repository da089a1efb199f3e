"""Random-phase libraries, API of plancklens/sims/phas.py (`rng_db` :13-62, `sim_lib` :65-123, `pix_lib_phas` :126-155,
`lib_phas` :157-195).

Reference semantics (the default classes `lib_phas`, `pix_lib_phas`): a simulation is *the state of numpy's global legacy generator
at the moment the simulation was first asked for*, kept per index in an sqlite table (`rngdb.db`, one directory per field:
`pha_%04d`, `pix_pha_%04d`) beside a `sim_hash.pk`; asking again restores that state and redraws.  An existing `$PLENS` tree
therefore yields here the phases it yields in the reference, and a fresh library driven by the same `np.random.seed` produces the
same sequence (the draw advances the global generator exactly as the reference's does).  Table layout, file names and the draw
order (real parts, then imaginary parts, / sqrt 2, m = 0 column made real with unit variance) are that contract.

Extensions: `lib_phas_seeded` / `pix_lib_phas_seeded` make get_sim(idx, idf) a pure function of (library seed, idf, idx)
(`default_rng([seed, idf, idx])`, no database, order-independent, what a sharded run wants), and `lib_phas_dev` /
`pix_lib_phas_dev` draw on the GPU with torch's counter-based Philox generator seeded the same way and
return device tensors -- at GPU reconstruction speeds the host draw of 3 npix normals and their upload would be the bottleneck
of a Monte-Carlo run.  Their streams differ from numpy's (and so the realisations), which the hash records.
"""
import os
import pickle as pk
import sqlite3

import numpy as np

from .. import hp, utils
from ..helpers import mpi

_RNG_COLUMNS = ('type', 'pos', 'has_gauss', 'cached_gaussian', 'keys')  # np.random.get_state(): (name, keys[624], pos, has_gauss, cached)


class rng_db(object):
    """Generator states of numpy's legacy RandomState in an sqlite file: table rngdb(id, type, pos, has_gauss, cached_gaussian,
    keys), the 624 key words stored as one '_'-joined string (phas.py:13-62)."""

    def __init__(self, fname, idtype="INTEGER"):
        if mpi.rank == 0 and not os.path.exists(fname):
            con = sqlite3.connect(fname, detect_types=sqlite3.PARSE_DECLTYPES, timeout=3600)
            con.execute("create table rngdb (id %s PRIMARY KEY, type STRING, pos INTEGER, has_gauss INTEGER,cached_gaussian REAL, keys STRING)" % idtype)
            con.commit()
            con.close()
        mpi.barrier()
        self.con = sqlite3.connect(fname, timeout=3600., detect_types=sqlite3.PARSE_DECLTYPES)

    def get(self, idx):
        row = self.con.execute("SELECT %s FROM rngdb WHERE id=?" % ', '.join(_RNG_COLUMNS), (int(idx),)).fetchone()
        if row is None:
            return None
        typ, pos, has_gauss, cached, keys = row
        return [typ, np.array(keys.split('_'), dtype=np.uint64).astype(np.uint32), pos, has_gauss, cached]

    def add(self, idx, state):
        """stores the state under idx unless idx is taken (reported, not raised, like the reference)"""
        try:
            assert self.get(idx) is None
            name, keys, pos, has_gauss, cached = state
            self.con.execute("INSERT INTO rngdb (id, %s) VALUES (?,?,?,?,?,?)" % ', '.join(_RNG_COLUMNS),
                             (int(idx), name, int(pos), int(has_gauss), float(cached), '_'.join(str(int(k)) for k in keys)))
            self.con.commit()
        except Exception:
            print("rng_db::rngdb add failed!")

    def delete(self, idx):
        try:
            if self.get(idx) is not None:
                self.con.execute("DELETE FROM rngdb WHERE id=?", (int(idx),))
                self.con.commit()
        except Exception:
            print("rng_db::rngdb delete %s failed!" % idx)


class sim_lib(object):
    """Simulations defined by stored generator states (phas.py:65-123): subclasses give hashdict() and
    _build_sim_from_rng(state, **kwargs)."""

    def __init__(self, lib_dir, get_state_func=np.random.get_state, nsims_max=None):
        if mpi.rank == 0 and not os.path.exists(lib_dir):
            os.makedirs(lib_dir)
        self.nmax = nsims_max
        fn_hash = os.path.join(lib_dir, 'sim_hash.pk')
        if mpi.rank == 0 and not os.path.exists(fn_hash):
            pk.dump(self.hashdict(), open(fn_hash, 'wb'), protocol=2)
        mpi.barrier()
        utils.hash_check(pk.load(open(fn_hash, 'rb')), self.hashdict(), ignore=['lib_dir'], fn=fn_hash)
        self._rng_db = rng_db(os.path.join(lib_dir, 'rngdb.db'), idtype='INTEGER')
        self._get_rng_state = get_state_func

    def get_sim(self, idx, **kwargs):
        """Simulation idx; the first request records the generator's current state under idx."""
        assert self.nmax is None or idx < self.nmax
        if not self.is_stored(idx):
            self._rng_db.add(idx, self._get_rng_state())
        return self._build_sim_from_rng(self._rng_db.get(idx), **kwargs)

    def has_nmax(self):
        return self.nmax is not None

    def is_stored(self, idx):
        return self._rng_db.get(idx) is not None

    def is_full(self):
        return self.has_nmax() and all(self.is_stored(i) for i in range(self.nmax))

    def is_empty(self):
        assert self.nmax is not None
        return not any(self.is_stored(i) for i in range(self.nmax))

    def hashdict(self):
        assert 0, 'override this'

    def _build_sim_from_rng(self, rng_state, **kwargs):
        assert 0, 'override this'


def _restore(state):
    """numpy's global legacy generator put into `state` (as the reference does: the draw that follows advances the global stream)"""
    np.random.set_state((state[0], np.asarray(state[1], dtype=np.uint32), int(state[2]), int(state[3]), float(state[4])))
    return np.random


class _pix_lib_phas(sim_lib):
    def __init__(self, lib_dir, shape, **kwargs):
        self.shape = shape
        super(_pix_lib_phas, self).__init__(lib_dir, **kwargs)

    def _build_sim_from_rng(self, rng_state, phas_only=False):
        return _restore(rng_state).standard_normal(self.shape)

    def hashdict(self):
        return {'shape': self.shape}


class _lib_phas(sim_lib):
    def __init__(self, lib_dir, lmax, **kwargs):
        self.lmax = lmax
        super(_lib_phas, self).__init__(lib_dir, **kwargs)

    def _build_sim_from_rng(self, rng_state, phas_only=False):
        rng = _restore(rng_state)
        n = hp.Alm.getsize(self.lmax)
        re = rng.standard_normal(n)
        alm = (re + 1j * rng.standard_normal(n)) / np.sqrt(2.)
        if phas_only:
            return None
        alm[:self.lmax + 1] = np.sqrt(2.) * alm[:self.lmax + 1].real  # the m = 0 entries open the array
        return alm

    def hashdict(self):
        return {'lmax': self.lmax}


class _fields(object):
    """nfields independent single-field libraries in sub-directories `<prefix>_%04d` of lib_dir"""

    def __init__(self, lib_dir, nfields, make, prefix):
        self.nfields = nfields
        self._libs = {i: make(os.path.join(lib_dir, '%s_%04d' % (prefix, i))) for i in range(nfields)}

    def __getitem__(self, i):
        return self._libs[i]

    def is_full(self):
        return bool(np.all([lib.is_full() for lib in self._libs.values()]))

    def get_sim(self, idx, idf=None, phas_only=False):
        if idf is not None:
            assert idf < self.nfields, (idf, self.nfields)
            return self._libs[idf].get_sim(idx, phas_only=phas_only)
        return np.array([self._libs[i].get_sim(idx, phas_only=phas_only) for i in range(self.nfields)])


class pix_lib_phas(_fields):
    """Unit-variance white pixel maps, nfields of them per simulation (phas.py:137-155)."""

    def __init__(self, lib_dir, nfields, shape, **kwargs):
        self.shape = shape
        super(pix_lib_phas, self).__init__(lib_dir, nfields, lambda d: _pix_lib_phas(d, shape, **kwargs), 'pix_pha')
        self.lib_pix = self._libs

    def hashdict(self):
        return {'nfields': self.nfields, 'shape': self.shape}


class lib_phas(_fields):
    """Unit-variance harmonic phases: complex normal alm with real m = 0 column (phas.py:157-195)."""

    def __init__(self, lib_dir, nfields, lmax, **kwargs):
        self.lmax = lmax
        super(lib_phas, self).__init__(lib_dir, nfields, lambda d: _lib_phas(d, lmax, **kwargs), 'pha')
        self.lib_phas = self._libs

    def hashdict(self):
        return {'nfields': self.nfields, 'lmax': self.lmax}


# ---- counter-seeded variants (extension): no database, get_sim a pure function of (seed, idf, idx) ------------------------------
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

    def is_full(self):
        return True


class pix_lib_phas_seeded(_seeded_lib):
    """pix_lib_phas without stored states."""

    def __init__(self, lib_dir, nfields, shape, seed=None, **kwargs):
        self.nfields = nfields
        self.shape = shape
        super(pix_lib_phas_seeded, self).__init__(lib_dir, seed=seed, **kwargs)

    def get_sim(self, idx, idf=None, phas_only=False):
        if idf is not None:
            assert idf < self.nfields, (idf, self.nfields)
            return None if phas_only else self._rng(idf, idx).standard_normal(self.shape)
        return np.array([self.get_sim(idx, idf=i) for i in range(self.nfields)])

    def hashdict(self):
        return {'nfields': self.nfields, 'shape': self.shape, 'seed': self.seed}


class lib_phas_seeded(_seeded_lib):
    """lib_phas without stored states."""

    def __init__(self, lib_dir, nfields, lmax, seed=None, **kwargs):
        self.nfields = nfields
        self.lmax = lmax
        super(lib_phas_seeded, self).__init__(lib_dir, seed=seed, **kwargs)

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


def _dev_key(seed, idf, idx):
    """64-bit Philox key of (library seed, field, simulation): a SplitMix-style mix, distinct streams for distinct triples"""
    z = (int(seed) * 0x9E3779B97F4A7C15 + int(idf) * 0xBF58476D1CE4E5B9 + int(idx) * 0x94D049BB133111EB + 0x2545F4914F6CDD1D) & (2 ** 64 - 1)
    z ^= z >> 31
    z = (z * 0xD6E8FEB86659FD93) & (2 ** 64 - 1)
    z ^= z >> 32
    return z


class pix_lib_phas_dev(pix_lib_phas_seeded):
    """pix_lib_phas drawn on the device by the library's own counter-based generator (pl_map_add_normal: Philox4x32-10 + Box-Muller, a pure
    function of (seed, field, index, pixel)): get_sim returns float64 CUDA tensors; `add_scaled` adds scale x that realisation to a map in
    the one pass that reads and writes it -- no tensor of deviates, no second pass (sims/maps.py:46-77,136-173 of the reference add a
    host array)."""

    def _key(self, idx, idf):
        assert idf < self.nfields, (idf, self.nfields)
        if self.nmax is not None:
            assert idx < self.nmax
        return _dev_key(self.seed, idf, idx)

    def add_scaled(self, m, idx, idf, scale, out=None):
        """out = m + scale x (realisation idx of field idf); out defaults to m (in place); m None: out = scale x realisation"""
        import ctypes
        import torch
        from .. import _lib, dev
        n = int(np.prod(self.shape))
        out = m if out is None else out
        assert out is not None and out.is_cuda and out.dtype == torch.float64 and out.is_contiguous() and out.numel() == n, 'float64 device map of the library shape'
        assert m is None or (m.is_cuda and m.dtype == torch.float64 and m.is_contiguous() and m.numel() == n)
        _lib.check(_lib.lib().pl_map_add_normal(n, None if m is None else m.data_ptr(), out.data_ptr(), float(scale), ctypes.c_uint64(self._key(idx, idf)),
                                               dev.stream_ptr()))
        return out

    def get_sim(self, idx, idf=None, phas_only=False):
        import torch
        if idf is not None:
            key = self._key(idx, idf)
            if phas_only:
                return None
            del key
            return self.add_scaled(None, idx, idf, 1.0, out=torch.empty(self.shape, dtype=torch.float64, device='cuda'))
        return torch.stack([self.get_sim(idx, idf=i) for i in range(self.nfields)])

    def hashdict(self):
        return {'nfields': self.nfields, 'shape': self.shape, 'seed': self.seed, 'rng': 'plshts-philox4x32-10-boxmuller'}


class lib_phas_dev(lib_phas_seeded):
    """lib_phas drawn on the device (pl_alm_unit_phases): get_sim returns complex128 CUDA tensors (unit variance, real m = 0 column).  The
    fields of the most recent simulation are kept: a sky library asks for each of them once per correlated field."""

    def get_sim(self, idx, idf=None, phas_only=False):
        import ctypes
        import torch
        from .. import _lib, dev
        if idf is not None:
            assert idf < self.nfields, (idf, self.nfields)
            if self.nmax is not None:
                assert idx < self.nmax
            if phas_only:
                return None
            memo = self.__dict__.setdefault('_memo', {})
            if (idx, idf) not in memo:
                for k in [k for k in memo if k[0] != idx]:
                    del memo[k]
                out = torch.empty(hp.Alm.getsize(self.lmax), dtype=torch.complex128, device='cuda')
                _lib.check(_lib.lib().pl_alm_unit_phases(int(self.lmax), out.data_ptr(), ctypes.c_uint64(_dev_key(self.seed, idf, idx)), dev.stream_ptr()))
                memo[(idx, idf)] = out
            return memo[(idx, idf)]  # (read-only by convention: the sky libraries combine the fields out of place)
        return torch.stack([self.get_sim(idx, idf=i) for i in range(self.nfields)])

    def hashdict(self):
        return {'nfields': self.nfields, 'lmax': self.lmax, 'seed': self.seed, 'rng': 'plshts-philox4x32-10-boxmuller'}
