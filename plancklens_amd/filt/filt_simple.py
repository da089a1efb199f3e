"""Non-iterative CMB filtering on the MI355X, behind the API of plancklens/filt/filt_simple.py
(`library_sepTP` :16-183, `library_fullsky_sepTP` :346-407, `library_fullsky_alms_sepTP` :409-470,
`library_apo_sepTP` :473-535).

Xb = F_l b_l^-1 map2alm(map) runs as ONE device call: the map is uploaded (or already resident), the spin-0 /
spin-2 analysis is done by the HIP kernels and the l-filter is fused into its epilogue.  The filtered alms
of the most recent simulations stay resident in HBM (`get_sim_alm_dev`) so that the quadratic estimators
do not go through PCIe or disk between the filter and the estimator; disk caching (same FITS file names as
the reference) is kept behind `cache`.
"""
from __future__ import print_function

import os
import pickle as pk

import numpy as np
import torch

from .. import dev, hp, shts, utils
from ..helpers import mpi


class library_sepTP(object):
    """Template for separately filtered T and P (filt_simple.py:16-183).  Subclasses provide `_apply_ivf_t`,
    `_apply_ivf_p`, `get_ftl/fel/fbl`, `get_tal`, `get_fmask`, `hashdict`."""

    _dev_slots = 4  # simulations whose filtered alms are kept resident on the device

    def __init__(self, lib_dir, sim_lib, cl_weights, soltn_lib=None, cache=True):
        self.lib_dir = lib_dir
        self.sim_lib = sim_lib
        self.cl = cl_weights
        self.soltn_lib = soltn_lib
        self.cache = cache
        self._dev_cache = {}
        fn_hash = os.path.join(lib_dir, 'filt_hash.pk')
        if mpi.rank == 0:
            if not os.path.exists(lib_dir):
                os.makedirs(lib_dir)
            if not os.path.exists(fn_hash):
                pk.dump(self.hashdict(), open(fn_hash, 'wb'), protocol=2)
        mpi.barrier()
        utils.hash_check(pk.load(open(fn_hash, 'rb')), self.hashdict(), fn=fn_hash)

    def hashdict(self):
        assert 0, 'override this'

    def get_fmask(self):
        assert 0, 'override this'

    def _apply_ivf_t(self, tmap, soltn=None):
        assert 0, 'override this'

    def _apply_ivf_p(self, pmap, soltn=None):
        assert 0, 'override this'

    def get_ftl(self):
        assert 0, 'override this'

    def get_fel(self):
        assert 0, 'override this'

    def get_fbl(self):
        assert 0, 'override this'

    def get_tal(self, a):
        assert 0, 'override this'

    # ---- file names of the reference's cache -------------------------------------------------------
    def _fn(self, a, idx):
        return os.path.join(self.lib_dir, ('sim_%04d_%slm.fits' % (idx, a)) if idx >= 0 else 'dat_%slm.fits' % a)

    # ---- device-resident route (extension used by qest.lib_filt2map) --------------------------------
    def _dev_entry(self, idx):
        if idx not in self._dev_cache:
            while len(self._dev_cache) >= self._dev_slots:
                self._dev_cache.pop(next(iter(self._dev_cache)))
            self._dev_cache[idx] = {}
        return self._dev_cache[idx]

    def _soltn_t(self, idx):
        return None if self.soltn_lib is None else self.soltn_lib.get_sim_tmliklm(idx)

    def _soltn_p(self, idx):
        if self.soltn_lib is None:
            return None
        return np.array([self.soltn_lib.get_sim_emliklm(idx), self.soltn_lib.get_sim_bmliklm(idx)])

    def get_sim_alm_dev(self, name, idx):
        """Device tensor of 'tlm' | 'elm' | 'blm' | 'tmliklm' | 'emliklm' | 'bmliklm' for simulation idx."""
        if name.endswith('mliklm'):
            a = name[0]
            return dev.almxfl(self.get_sim_alm_dev(a + 'lm', idx), self.cl[a + a])
        a = name[0]
        ent = self._dev_entry(idx)
        if a not in ent:
            if self.cache and os.path.exists(self._fn(a, idx)):
                ent[a] = dev.to_dev(hp.read_alm(self._fn(a, idx)), torch.complex128)
            elif a == 't':
                tmap = self.sim_lib.get_sim_tmap(idx)
                ent['t'] = dev.to_dev(self._apply_ivf_t(dev.to_dev(tmap, torch.float64), soltn=self._soltn_t(idx)))
                if self.cache:
                    hp.write_alm(self._fn('t', idx), dev.to_host(ent['t']), overwrite=True)
            else:
                pmap = self.sim_lib.get_sim_pmap(idx)
                if isinstance(pmap[0], torch.Tensor):
                    qu = [dev.to_dev(pmap[0], torch.float64), dev.to_dev(pmap[1], torch.float64)]
                else:  # host arrays land in the two rows of one device array: the layout the spin transform takes, no stacking copy
                    buf = torch.empty((2, np.size(pmap[0])), dtype=torch.float64, device=dev.device())
                    buf[0].copy_(torch.from_numpy(np.ascontiguousarray(pmap[0], dtype=np.float64)))
                    buf[1].copy_(torch.from_numpy(np.ascontiguousarray(pmap[1], dtype=np.float64)))
                    qu = [buf[0], buf[1]]
                e, b = self._apply_ivf_p(qu, soltn=self._soltn_p(idx))
                ent['e'], ent['b'] = dev.to_dev(e), dev.to_dev(b)
                if self.cache:
                    hp.write_alm(self._fn('e', idx), dev.to_host(ent['e']), overwrite=True)
                    hp.write_alm(self._fn('b', idx), dev.to_host(ent['b']), overwrite=True)
        return ent[a]

    # ---- the reference's getters (host arrays) -------------------------------------------------------
    def get_sim_tlm(self, idx):
        """Inverse-variance filtered temperature alm of simulation idx (filt_simple.py:84-99)."""
        return dev.to_host(self.get_sim_alm_dev('tlm', idx))

    def get_sim_elm(self, idx):
        return dev.to_host(self.get_sim_alm_dev('elm', idx))

    def get_sim_blm(self, idx):
        return dev.to_host(self.get_sim_alm_dev('blm', idx))

    def get_sim_tmliklm(self, idx):
        """Wiener-filtered temperature alm C^TT_l Tb_lm (filt_simple.py:149-159)."""
        return hp.almxfl(self.get_sim_tlm(idx), self.cl['tt'])

    def get_sim_emliklm(self, idx):
        return hp.almxfl(self.get_sim_elm(idx), self.cl['ee'])

    def get_sim_bmliklm(self, idx):
        return hp.almxfl(self.get_sim_blm(idx), self.cl['bb'])


class library_jTP(object):
    """Template for jointly filtered T and P (filt_simple.py:187-343).  Subclasses provide `_apply_ivf` (three maps ->
    three alms), `get_fal`, `get_fmask`, `hashdict`.  The Wiener-filtered legs mix the fields through the
    cross-spectra of `cl_weights` (te, tb, eb when present)."""

    _dev_slots = 4

    def __init__(self, lib_dir, sim_lib, cl_weights, soltn_lib=None, cache=True):
        assert np.all([k in cl_weights.keys() for k in ['tt', 'ee', 'bb']])
        self.lib_dir = lib_dir
        self.sim_lib = sim_lib
        self.cl = cl_weights
        self.soltn_lib = soltn_lib
        self.cache = cache
        self._dev_cache = {}
        fn_hash = os.path.join(lib_dir, 'filt_hash.pk')
        if mpi.rank == 0:
            if not os.path.exists(lib_dir):
                os.makedirs(lib_dir)
            if not os.path.exists(fn_hash):
                pk.dump(self.hashdict(), open(fn_hash, 'wb'), protocol=2)
        mpi.barrier()
        utils.hash_check(pk.load(open(fn_hash, 'rb')), self.hashdict(), fn=fn_hash)

    def hashdict(self):
        assert 0, 'override this'

    def get_fmask(self):
        assert 0, 'override this'

    def _apply_ivf(self, tqumap, soltn=None):
        assert 0, 'override this'

    def get_fal(self):
        assert 0, 'override this'

    def _fn(self, a, idx):
        return os.path.join(self.lib_dir, ('sim_%04d_%slm.fits' % (idx, a)) if idx >= 0 else 'dat_%slm.fits' % a)

    def _dev_entry(self, idx):
        if idx not in self._dev_cache:
            while len(self._dev_cache) >= self._dev_slots:
                self._dev_cache.pop(next(iter(self._dev_cache)))
            self._dev_cache[idx] = {}
        return self._dev_cache[idx]

    def get_sim_alm_dev(self, name, idx):
        """Device tensor of 'tlm' | 'elm' | 'blm' | 'tmliklm' | 'emliklm' | 'bmliklm' for simulation idx."""
        a = name[0]
        if name.endswith('mliklm'):
            ret = dev.almxfl(self.get_sim_alm_dev(a + 'lm', idx), self.cl[a + a])
            for b in 'teb':
                if b != a:
                    cl = self.cl.get(a + b, self.cl.get(b + a, None))
                    if cl is not None:
                        ret = ret + dev.almxfl(self.get_sim_alm_dev(b + 'lm', idx), cl)
            return ret
        ent = self._dev_entry(idx)
        if a not in ent:
            if self.cache and os.path.exists(self._fn(a, idx)):
                ent[a] = dev.to_dev(hp.read_alm(self._fn(a, idx)), torch.complex128)
            else:
                T = self.sim_lib.get_sim_tmap(idx)
                Q, U = self.sim_lib.get_sim_pmap(idx)
                soltn = None
                if self.soltn_lib is not None:
                    soltn = (self.soltn_lib.get_sim_tmliklm(idx), self.soltn_lib.get_sim_emliklm(idx), self.soltn_lib.get_sim_bmliklm(idx))
                alms = self._apply_ivf([dev.to_dev(m, torch.float64) for m in (T, Q, U)], soltn=soltn)
                for f, alm in zip('teb', alms):
                    ent[f] = dev.to_dev(alm)
                    if self.cache:
                        hp.write_alm(self._fn(f, idx), dev.to_host(ent[f]), overwrite=True)
        return ent[a]

    def _get_alms(self, a, idx):
        """inverse-variance filtered alm of field a in 'teb' (the reference's private getter, filt_simple.py:266-342, kept by name: one
        joint filter application serves all three fields, here through the device cache of get_sim_alm_dev)"""
        assert a in ['t', 'e', 'b'], a
        return dev.to_host(self.get_sim_alm_dev(a + 'lm', idx))

    def get_sim_tlm(self, idx):
        return self._get_alms('t', idx)

    def get_sim_elm(self, idx):
        return self._get_alms('e', idx)

    def get_sim_blm(self, idx):
        return self._get_alms('b', idx)

    def get_sim_tmliklm(self, idx):
        return dev.to_host(self.get_sim_alm_dev('tmliklm', idx))

    def get_sim_emliklm(self, idx):
        return dev.to_host(self.get_sim_alm_dev('emliklm', idx))

    def get_sim_bmliklm(self, idx):
        return dev.to_host(self.get_sim_alm_dev('bmliklm', idx))


def _as_transf_dict(transf):
    d = transf if isinstance(transf, dict) else {'t': transf, 'e': transf, 'b': transf}
    assert all(k in d.keys() for k in 'teb')
    return d


class _iso_filter_mixin(object):
    """Shared pieces of the isotropic filters: F_l / b_l weights and their accessors."""

    def _setup_fl(self, transfd, ftl, fel, fbl):
        self.ftl, self.fel, self.fbl = ftl, fel, fbl
        self.lmax_fl = int(np.max([len(ftl), len(fel), len(fbl)])) - 1
        self.transf = transfd

    def _tal(self, a):
        t = self.transf[a] if isinstance(self.transf, dict) else self.transf
        return utils.cli(t)

    def get_tal(self, a):
        assert a.lower() in ['t', 'e', 'b']
        return self._tal(a.lower())

    def get_ftl(self):
        return np.copy(self.ftl)

    def get_fel(self):
        return np.copy(self.fel)

    def get_fbl(self):
        return np.copy(self.fbl)

    def _weight(self, a):
        f = {'t': self.ftl, 'e': self.fel, 'b': self.fbl}[a]
        return f * self._tal(a)[:len(f)]


class library_fullsky_sepTP(_iso_filter_mixin, library_sepTP):
    """Full-sky isotropic filtering Xb_lm = F^X_l / b_l map2alm(map)_lm (filt_simple.py:346-407)."""

    def __init__(self, lib_dir, sim_lib, nside, transf, cl_len, ftl, fel, fbl, cache=False):
        self._setup_fl(_as_transf_dict(transf), ftl, fel, fbl)
        self.nside = nside
        super(library_fullsky_sepTP, self).__init__(lib_dir, sim_lib, cl_len, cache=cache)

    def hashdict(self):
        return {'sim_lib': self.sim_lib.hashdict(), 'transf': utils.clhash(self.transf['t']),
                'cl_len': {k: utils.clhash(self.cl[k]) for k in ['tt', 'ee', 'bb']},
                'ftl': utils.clhash(self.ftl), 'fel': utils.clhash(self.fel), 'fbl': utils.clhash(self.fbl)}

    def get_fmask(self):
        return np.ones(hp.nside2npix(self.nside), dtype=float)

    def _mask(self, m):
        return m

    def _apply_ivf_t(self, tmap, soltn=None):
        n = tmap.numel() if isinstance(tmap, torch.Tensor) else len(tmap)
        assert n == hp.nside2npix(self.nside), (n, self.nside)
        return shts.map2alm(self._mask(tmap), lmax=self.lmax_fl, iter=0, fl=self._weight('t'))

    def _apply_ivf_p(self, pmap, soltn=None):
        n = pmap[0].numel() if isinstance(pmap[0], torch.Tensor) else len(pmap[0])
        assert n == hp.nside2npix(self.nside) and len(pmap) == 2
        fe, fb = self._weight('e'), self._weight('b')
        maps = [self._mask(m) for m in pmap]
        if len(fe) == len(fb) and np.all(fe == fb):
            return shts.map2alm_spin(maps, 2, lmax=self.lmax_fl, fl=fe)
        elm, blm = shts.map2alm_spin(maps, 2, lmax=self.lmax_fl)
        if isinstance(elm, torch.Tensor):
            return dev.almxfl(elm, fe), dev.almxfl(blm, fb)
        return hp.almxfl(elm, fe), hp.almxfl(blm, fb)


class library_fullsky_alms_sepTP(_iso_filter_mixin, library_sepTP):
    """Full-sky isotropic filtering with harmonic-space inputs: sim_lib.get_sim_tmap / get_sim_pmap return alms
    (filt_simple.py:409-470)."""

    def __init__(self, lib_dir, sim_lib, transf, cl_len, ftl, fel, fbl, cache=False):
        self._setup_fl(_as_transf_dict(transf), ftl, fel, fbl)
        super(library_fullsky_alms_sepTP, self).__init__(lib_dir, sim_lib, cl_len, cache=cache)

    def hashdict(self):
        return {'sim_lib': self.sim_lib.hashdict(), 'transf': utils.clhash(self.transf['t']),
                'cl_len': {k: utils.clhash(self.cl[k]) for k in ['tt', 'ee', 'bb']},
                'ftl': utils.clhash(self.ftl), 'fel': utils.clhash(self.fel), 'fbl': utils.clhash(self.fbl)}

    def get_fmask(self):
        return np.array([1.])  # compatibility only, as in the reference

    def _apply_ivf_t(self, tlm, soltn=None):
        if isinstance(tlm, torch.Tensor):
            return dev.almxfl(tlm.to(torch.complex128), self._weight('t'))
        return hp.almxfl(tlm, self._weight('t'))

    def _apply_ivf_p(self, eblm, soltn=None):
        if isinstance(eblm[0], torch.Tensor):
            return (dev.almxfl(eblm[0].to(torch.complex128), self._weight('e')),
                    dev.almxfl(eblm[1].to(torch.complex128), self._weight('b')))
        return hp.almxfl(eblm[0], self._weight('e')), hp.almxfl(eblm[1], self._weight('b'))


class library_apo_sepTP(library_fullsky_sepTP):
    """Isotropic filtering of apodised-mask maps (filt_simple.py:473-535)."""

    def __init__(self, lib_dir, sim_lib, apomask_path, cl_len, transf, ftl, fel, fbl, cache=False):
        assert len(transf) >= np.max([len(ftl), len(fel), len(fbl)])
        assert np.all([k in cl_len.keys() for k in ['tt', 'ee', 'bb']])
        assert os.path.exists(apomask_path)
        self.apomask_path = apomask_path
        self._setup_fl(transf, ftl, fel, fbl)
        self._fmask_dev = None
        self.nside = hp.npix2nside(hp.read_map(apomask_path).size)
        library_sepTP.__init__(self, lib_dir, sim_lib, cl_len, cache=cache)

    def hashdict(self):
        return {'sim_lib': self.sim_lib.hashdict(), 'apomask': self.apomask_path, 'transf': utils.clhash(self.transf),
                'cl_len': {k: utils.clhash(self.cl[k]) for k in ['tt', 'ee', 'bb']},
                'ftl': utils.clhash(self.ftl), 'fel': utils.clhash(self.fel), 'fbl': utils.clhash(self.fbl)}

    def get_fmask(self):
        return hp.read_map(self.apomask_path)

    def _mask(self, m):
        if isinstance(m, torch.Tensor):
            if self._fmask_dev is None:
                self._fmask_dev = dev.to_dev(self.get_fmask(), torch.float64)
            return dev.map_mul(m, self._fmask_dev)
        return m * self.get_fmask()
