"""CMB map simulation libraries, API of plancklens/sims/maps.py (`cmb_maps` :13-77, `cmb_maps_noisefree` :88-99, `cmb_maps_nlev`
:100-173, `cmb_maps_harmonicspace` :177-275): sky alm x transfer function -> alm2map / alm2map_spin (GPU) + noise."""
import os
import pickle as pk

import numpy as np

from .. import dev, hp, shts
from ..helpers import mpi
from ..utils import clhash, hash_check


class cmb_maps(object):
    """Sky library + transfer function -> T, Q, U maps (maps.py:13-77).  With device_maps=True (extension) the
    maps are returned as device tensors and never leave HBM."""

    def __init__(self, sims_cmb_len, cl_transf, nside=2048, cl_transf_P=None, lib_dir=None, device_maps=False):
        self.sims_cmb_len = sims_cmb_len
        self.cl_transf_T = cl_transf
        self.cl_transf_P = np.copy(cl_transf) if cl_transf_P is None else cl_transf_P
        self.nside = nside
        self.device_maps = device_maps
        if lib_dir is not None:
            fn_hash = os.path.join(lib_dir, 'sim_hash.pk')
            if mpi.rank == 0 and not os.path.exists(fn_hash):
                if not os.path.exists(lib_dir):
                    os.makedirs(lib_dir)
                pk.dump(self.hashdict(), open(fn_hash, 'wb'), protocol=2)
            mpi.barrier()
            hash_check(self.hashdict(), pk.load(open(fn_hash, 'rb')), fn=fn_hash)

    def hashdict(self):
        ret = {'sims_cmb_len': self.sims_cmb_len.hashdict(), 'nside': self.nside, 'cl_transf': clhash(self.cl_transf_T)}
        if not np.all(self.cl_transf_P == self.cl_transf_T):
            ret['cl_transf_P'] = clhash(self.cl_transf_P)
        return ret

    def _out(self, t):
        return t if self.device_maps else dev.to_host(t)

    def _add_noise(self, m, idx, idf, out=None):
        """sky map m (device tensor, may be overwritten) + the noise of field idf -> device tensor; `out`: written there instead"""
        res = m + dev.to_dev(self._noise_term(idx, idf))
        return res if out is None else out.copy_(res)

    def _tsky(self, idx):
        held = self.__dict__.get('_pair_held_t')
        if held is not None and held[0] == idx:  # made together with the previous simulation (hint_pair)
            self._pair_held_t = None
            return held[1]
        tlm = dev.to_dev(self.sims_cmb_len.get_sim_tlm(idx))
        nxt = self.__dict__.get('_pair_next_t')
        self._pair_next_t = None
        if nxt is not None and nxt[0] == idx and self.device_maps:
            # the temperature sky of the announced next simulation in the same call (pl_alm2map_batch2 with spin 0: one Legendre recursion for
            # both on fine grids; bit-identical maps)
            t2 = dev.to_dev(self.sims_cmb_len.get_sim_tlm(nxt[1]))
            both = shts.alm2map_batch2(tlm, t2, self.nside, fl=self.cl_transf_T)
            self._pair_held_t = (nxt[1], both[1])
            return both[0]
        return shts.alm2map(tlm, self.nside, fl=self.cl_transf_T)

    def _psky(self, idx):
        held = self.__dict__.get('_pair_held')
        if held is not None and held[0] == idx:  # made together with the previous simulation (hint_pair)
            self._pair_held = None
            return held[1]
        elm = dev.to_dev(self.sims_cmb_len.get_sim_elm(idx))
        blm = dev.to_dev(self.sims_cmb_len.get_sim_blm(idx))
        lmax = hp.Alm.getlmax(elm.numel())
        nxt = self.__dict__.get('_pair_next')
        self._pair_next = None
        if nxt is not None and nxt[0] == idx and self.device_maps:
            # the sky of the announced next simulation on the same Legendre recursion (pl_alm2map_batch2: bit-identical maps)
            e2, b2 = dev.to_dev(self.sims_cmb_len.get_sim_elm(nxt[1])), dev.to_dev(self.sims_cmb_len.get_sim_blm(nxt[1]))
            (Q, U), (Q2, U2) = shts.alm2map_spin_batch2([elm, blm], [e2, b2], self.nside, 2, lmax, fl=self.cl_transf_P)
            self._pair_held = (nxt[1], (Q2, U2))
            return Q, U
        return shts.alm2map_spin([elm, blm], self.nside, 2, lmax, fl=self.cl_transf_P)

    def get_sim_tmap(self, idx):
        return self._out(self._add_noise(self._tsky(idx), idx, 0))

    def get_sim_pmap(self, idx):
        Q, U = self._psky(idx)
        return self._out(self._add_noise(Q, idx, 1)), self._out(self._add_noise(U, idx, 2))

    # the same maps written into buffers of the caller (device float64 [npix]; an extension for consumers with fixed input slots --
    # the replayed graph of qest.library._pair_graph): with device noise the pass that adds the noise is the one that fills the slot
    def get_sim_tmap_into(self, idx, out):
        self._add_noise(self._tsky(idx), idx, 0, out=out)

    def get_sim_pmap_into(self, idx, outq, outu):
        Q, U = self._psky(idx)
        self._add_noise(Q, idx, 1, out=outq)
        self._add_noise(U, idx, 2, out=outu)

    def hint_pair(self, idx0, idx1):
        """The caller is about to ask for the maps of idx0 and then idx1 (a mean-field loop serving simulations in pairs): with device
        maps the polarization sky syntheses of the two then share one Legendre recursion, and so do the temperature ones.  A hint only: any other order of calls
        gives the same maps one by one."""
        self._pair_next = (idx0, idx1)
        self._pair_next_t = (idx0, idx1)

    def _noise_term(self, idx, idf):
        return (self.get_sim_tnoise, self.get_sim_qnoise, self.get_sim_unoise)[idf](idx)

    def get_sim_tnoise(self, idx):
        assert 0, 'subclass this'

    def get_sim_qnoise(self, idx):
        assert 0, 'subclass this'

    def get_sim_unoise(self, idx):
        assert 0, 'subclass this'


class cmb_maps_noisefree(cmb_maps):
    """Sky maps without noise (maps.py:88-99)."""

    def __init__(self, sims_cmb_len, cl_transf, nside=2048, cl_transf_P=None, device_maps=False):
        super(cmb_maps_noisefree, self).__init__(sims_cmb_len, cl_transf, nside=nside, cl_transf_P=cl_transf_P, device_maps=device_maps)

    def get_sim_tnoise(self, idx):
        return np.zeros(hp.nside2npix(self.nside))

    get_sim_qnoise = get_sim_tnoise
    get_sim_unoise = get_sim_tnoise


class cmb_maps_nlev(cmb_maps):
    """Homogeneous white noise of nlev_t / nlev_p muK-arcmin on top of the sky maps (maps.py:100-173)."""

    def __init__(self, sims_cmb_len, cl_transf, nlev_t, nlev_p, nside, lib_dir=None, pix_lib_phas=None, device_maps=False):
        if pix_lib_phas is None:
            assert lib_dir is not None
            from . import phas
            if mpi.size > 1:
                print('cmb_maps_nlev: default noise phases are the state-recording pix_lib_phas (the reference\'s semantics: realisations depend on the '
                      'order of first requests and on each rank\'s generator); a rank-sharded run should pass phas.pix_lib_phas_seeded or _dev')
            pix_lib_phas = phas.pix_lib_phas(lib_dir, 3, (hp.nside2npix(nside),))
        assert pix_lib_phas.shape == (hp.nside2npix(nside),), (pix_lib_phas.shape, (hp.nside2npix(nside),))
        self.pix_lib_phas = pix_lib_phas
        self.nlev_t = nlev_t
        self.nlev_p = nlev_p
        super(cmb_maps_nlev, self).__init__(sims_cmb_len, cl_transf, nside=nside, lib_dir=lib_dir, device_maps=device_maps)

    def hashdict(self):
        ret = {'sims_cmb_len': self.sims_cmb_len.hashdict(), 'nside': self.nside, 'cl_transf': clhash(self.cl_transf_T),
               'nlev_t': self.nlev_t, 'nlev_p': self.nlev_p, 'pixphas': self.pix_lib_phas.hashdict()}
        if not np.all(self.cl_transf_P == self.cl_transf_T):
            ret['cl_transf_P'] = clhash(self.cl_transf_P)
        return ret

    def _vamin(self):
        return np.sqrt(hp.nside2pixarea(self.nside, degrees=True)) * 60

    def _scale(self, idf):
        return (self.nlev_t if idf == 0 else self.nlev_p) / self._vamin()

    def _add_noise(self, m, idx, idf, out=None):
        """device phase library with `add_scaled` (phas.pix_lib_phas_dev): sigma n(0, 1) is added by the generator kernel in the one pass
        that reads the sky map and writes the result (pl_map_add_normal) -- in place, or into `out`; host phases: the noise map is added"""
        if hasattr(self.pix_lib_phas, 'add_scaled'):
            return self.pix_lib_phas.add_scaled(m.contiguous(), idx, idf, self._scale(idf), out=out)
        res = m + dev.to_dev(self._scale(idf) * self.pix_lib_phas.get_sim(idx, idf=idf))
        return res if out is None else out.copy_(res)

    def get_sim_tnoise(self, idx):
        return self.nlev_t / self._vamin() * self.pix_lib_phas.get_sim(idx, idf=0)

    def get_sim_qnoise(self, idx):
        return self.nlev_p / self._vamin() * self.pix_lib_phas.get_sim(idx, idf=1)

    def get_sim_unoise(self, idx):
        return self.nlev_p / self._vamin() * self.pix_lib_phas.get_sim(idx, idf=2)


class cmb_maps_harmonicspace(object):
    """Sky alm x per-field transfer function + isotropic (possibly coloured) noise drawn in harmonic space from a `lib_phas` with at
    least three fields (maps.py:177-275).  Returns alms -- what filt_simple.library_fullsky_alms_sepTP filters -- or, with nside,
    maps synthesised on the GPU."""

    def __init__(self, sims_cmb_len, cls_transf, cls_noise, noise_phas, lib_dir=None, nside=None):
        assert noise_phas.nfields >= 3, noise_phas.nfields
        self.sims_cmb_len = sims_cmb_len
        self.cls_transf = cls_transf
        self.cls_noise = cls_noise
        self.phas = noise_phas
        self.nside = nside
        if hasattr(sims_cmb_len, 'lmax'):
            assert sims_cmb_len.lmax == noise_phas.lmax, 'band-limits of the sky (%s) and of the noise phases (%s) differ' % (sims_cmb_len.lmax, noise_phas.lmax)
        if lib_dir is not None:
            fn_hash = os.path.join(lib_dir, 'sim_hash.pk')
            if mpi.rank == 0 and not os.path.exists(fn_hash):
                if not os.path.exists(lib_dir):
                    os.makedirs(lib_dir)
                pk.dump(self.hashdict(), open(fn_hash, 'wb'), protocol=2)
            mpi.barrier()
            hash_check(self.hashdict(), pk.load(open(fn_hash, 'rb')), fn=fn_hash)

    def hashdict(self):
        ret = {'sims_cmb_len': self.sims_cmb_len.hashdict(), 'phas': self.phas.hashdict()}
        ret.update({'noise' + k: clhash(v) for k, v in self.cls_noise.items()})
        ret.update({'transf' + k: clhash(v) for k, v in self.cls_transf.items()})
        return ret

    def _noise(self, idx, field, idf):
        assert field in self.cls_noise, field
        return hp.almxfl(self.phas.get_sim(idx, idf), np.sqrt(self.cls_noise[field]))

    def get_sim_tnoise(self, idx):
        return self._noise(idx, 't', 0)

    def get_sim_enoise(self, idx):
        return self._noise(idx, 'e', 1)

    def get_sim_bnoise(self, idx):
        return self._noise(idx, 'b', 2)

    def get_sim_tmap(self, idx):
        assert 't' in self.cls_transf
        tlm = hp.almxfl(self.sims_cmb_len.get_sim_tlm(idx), self.cls_transf['t']) + self.get_sim_tnoise(idx)
        return tlm if not self.nside else shts.alm2map(tlm, self.nside)

    def get_sim_pmap(self, idx):
        assert 'e' in self.cls_transf and 'b' in self.cls_transf
        elm = hp.almxfl(self.sims_cmb_len.get_sim_elm(idx), self.cls_transf['e']) + self.get_sim_enoise(idx)
        blm = hp.almxfl(self.sims_cmb_len.get_sim_blm(idx), self.cls_transf['b']) + self.get_sim_bnoise(idx)
        if self.nside is not None:
            return shts.alm2map_spin([elm, blm], self.nside, 2, hp.Alm.getlmax(elm.size))
        return elm, blm
