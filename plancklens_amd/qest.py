"""Quadratic estimators on the MI355X, behind the API of plancklens/qest.py.

Same classes, methods, keys, cache-file names and normalisations as the reference (`library` qest.py:50-438,
`lib_filt2map` :441-530, `lib_filt2map_sepTP` :533-638, `eval_qe` :19-39).  What differs is where the work
happens: the filtered alms are uploaded once per leg, every spin-0/1/2/3 transform, l-filter and pixel
product runs on the GPU (plancklens_amd.shts / dev, HIP kernels), the l-weights are fused into the
transforms, and only the final gradient / curl alm comes back to the host.

Formulas as coded in the reference, with Tb = inverse-variance filtered,
T^WF = C^TT Tb (+ C^TE Eb for the MV estimator):
  ptt / xtt : (G, C) = -sqrt(L(L+1)) map2alm_spin_1[ Tb(n) alm2map_spin_1(-sqrt(l(l+1)) T^WF_lm, 0) ]
  p_p / x_p : d = (Qb - iUb)(3G + i 3C) - (Qb + iUb)(1G - i 1C), (Qb, Ub) = alm2map_spin_2(Eb/2, Bb/2),
              sG = alm2map_spin_s(w^s_l (E^WF, B^WF)), w^3 = sqrt((l-2)(l+3)), w^1 = sqrt((l+2)(l-1));
              (G, C) = -sqrt(L(L+1)) map2alm_spin_1(Re d, Im d)
  p / x     : p_p form with E^WF += C^TE Tb, plus ptt form with T^WF += C^TE Eb
"""
from __future__ import print_function

import collections
import os
import pickle as pk

import numpy as np
import torch

from . import dev, hp, options, shts
from . import utils as ut
from .helpers import mpi

_write_alm = lambda fn, alm: hp.write_alm(fn, alm, overwrite=True)


def eval_qe(qe_key, lmax_ivf, cls_weight, get_alm, nside, lmax_qlm, verbose=True, get_alm2=None, transf=None):
    """Generic spin-weight route (qest.py:19-39): gradient and curl of the estimator `qe_key` built from the
    leg decomposition of qresp.get_qes and evaluated by utils_qe.qe_eval on the GPU."""
    from . import qresp, utils_qe as uqe
    qe_list = qresp.get_qes(qe_key, lmax_ivf, cls_weight, transf=transf)
    return uqe.qe_eval(qe_list, nside, get_alm, lmax_qlm, verbose=verbose, get_alm2=get_alm2)


def library_jtTP(lib_dir, ivfs1, ivfs2, nside, lmax_qlm=None, resplib=None, **kwargs):
    return library(lib_dir, ivfs1, ivfs2, nside, lmax_qlm=lmax_qlm, resplib=resplib, **kwargs)


def library_sepTP(lib_dir, ivfs1, ivfs2, clte, nside, lmax_qlm=None, resplib=None, **kwargs):
    return library(lib_dir, ivfs1, ivfs2, nside, clte=clte, lmax_qlm=lmax_qlm, resplib=resplib, **kwargs)


def _lens_weight(lmax):
    """-sqrt(L (L + 1)), L = 0 .. lmax (qest.py:260,282,463,592)."""
    ell = np.arange(lmax + 1, dtype=float)
    return -np.sqrt(ell * (ell + 1.))


def _spin_weight(spin, lmax):
    """sqrt((l+2)(l-1)) for spin 1, sqrt((l-2)(l+3)) for spin 3, first `spin` entries zeroed (qest.py:494-501)."""
    if spin == 1:
        fl = np.arange(2, lmax + 3, dtype=float) * np.arange(-1, lmax)
    elif spin == 3:
        fl = np.arange(-2, lmax - 1, dtype=float) * np.arange(3, lmax + 4)
    else:
        assert 0, spin
    fl[:spin] = 0.
    return np.sqrt(fl)


class library(object):
    """QE library from two inverse-variance filtered CMB libraries (qest.py:50-438).

        Args:
            lib_dir: QE estimates are cached there (same file names as the reference).
            ivfs1, ivfs2: filtering instances of the first and second leg.
            nside: resolution of the real-space products.
            clte (optional): TE spectrum, builds X^WF from Xb for separately filtered T and P.
            lmax_qlm (optional): output band-limit (defaults to 3 nside - 1).
            resplib (optional): response library, only for the bias-hardened keys.
            cache (extension, default True): set False to skip all disk IO of the estimates.
    """

    def __init__(self, lib_dir, ivfs1, ivfs2, nside, clte=None, lmax_qlm=None, resplib=None, cache=True):
        if lmax_qlm is None:
            lmax_qlm = 3 * nside - 1
        self.lib_dir = lib_dir
        self.prefix = lib_dir
        self.cache = cache
        self.lmax_qlm = {'T': lmax_qlm, 'P': lmax_qlm, 'PS': lmax_qlm}
        if clte is None:
            self.f2map1, self.f2map2 = lib_filt2map(ivfs1, nside), lib_filt2map(ivfs2, nside)
        else:
            self.f2map1, self.f2map2 = lib_filt2map_sepTP(ivfs1, nside, clte), lib_filt2map_sepTP(ivfs2, nside, clte)
        assert self.lmax_qlm['T'] == self.lmax_qlm['P'], 'implement this'
        fnhash = os.path.join(self.lib_dir, "qe_sim_hash.pk")
        if mpi.rank == 0 and not os.path.exists(fnhash):
            if not os.path.exists(self.lib_dir):
                os.makedirs(self.lib_dir)
            pk.dump(self.hashdict(), open(fnhash, 'wb'), protocol=2)
        mpi.barrier()
        ut.hash_check(pk.load(open(fnhash, 'rb')), self.hashdict(), fn=fnhash)
        fn_fsky = os.path.join(lib_dir, 'fskies.dat')
        if mpi.rank == 0 and not os.path.exists(fn_fsky):
            masks = {1: self.get_mask(1), 2: self.get_mask(2)}
            with open(fn_fsky, 'w') as f:
                for lab in [11, 12, 22]:
                    f.write('%4s %.5f \n' % (lab, np.mean(masks[lab // 10] * masks[lab % 10])))
        mpi.barrier()
        self.fskies = {}
        with open(fn_fsky) as f:
            for line in f:
                key, val = line.split()
                self.fskies[int(key)] = float(val)
        self.fsky11, self.fsky12, self.fsky22 = self.fskies[11], self.fskies[12], self.fskies[22]
        self.resplib = resplib
        self.keys_fund = ['ptt', 'xtt', 'p_p', 'x_p', 'p', 'x', 'stt', 's', 'ftt', 'f_p', 'f', 'dtt', 'ntt', 'a_p',
                          'pte', 'pet', 'ptb', 'pbt', 'pee', 'peb', 'pbe', 'pbb',
                          'xte', 'xet', 'xtb', 'xbt', 'xee', 'xeb', 'xbe', 'xbb']
        self.keys = self.keys_fund + ['p_tp', 'x_tp', 'p_te', 'p_tb', 'p_eb', 'x_te', 'x_tb', 'x_eb', 'ptt_bh_n',
                                      'ptt_bh_s', 'ptt_bh_f', 'ptt_bh_d', 'dtt_bh_p', 'stt_bh_p', 'ftt_bh_d', 'p_bh_s']
        self.keys_remaps = {'s': 'stt'}
        self._mem = {}
        self._last_dev = None  # device tensors (G, C) of the most recent gradient / curl evaluation ...
        self._last_dev_key = None  # ... and its (gradient key, idx): lets the mean-field sum stay on the device

    def hashdict(self):
        return {'f2map1': self.f2map1.hashdict(), 'f2map2': self.f2map2.hashdict()}

    # ring-FFT stages of the leg syntheses of the single-simulation MV route on side lanes (shts.lane); a measured non-gain kept as a
    # tested code path of the lane mechanism, switched per instance, never from the environment
    pipeline_lanes = False

    def get_fundkeys(self, k_list):
        """Fundamental estimators needed to build the (possibly derived) keys of k_list (qest.py:122-141)."""
        ret = []
        for k in (k_list if isinstance(k_list, list) else [k_list]):
            if k in self.keys_fund:
                ret.append(k)
            elif '_tp' in k:
                ret += [k[0] + 'tt', k[0] + '_p']
            elif 'tt_bh_' in k:
                l, f = k.split('_bh_')
                ret += [l, f + 'tt']
            elif k in ['p_te', 'p_tb', 'p_eb', 'x_te', 'x_tb', 'x_eb']:
                ret += [k[0] + k[2] + k[3], k[0] + k[3] + k[2]]
        return list(collections.OrderedDict.fromkeys(ret))

    def get_fsky(self, id):
        assert id in [11, 22, 12], id
        return self.fskies[id]

    def get_lmax_qlm(self, k):
        assert self.lmax_qlm['T'] == self.lmax_qlm['P']
        return self.lmax_qlm['T']

    def get_mask(self, leg):
        assert leg in [1, 2]
        return self.f2map1.ivfs.get_fmask() if leg == 1 else self.f2map2.ivfs.get_fmask()

    # ---- cache -----------------------------------------------------------------------------------
    def _fname(self, k, idx):
        return os.path.join(self.lib_dir, 'sim_%s_%04d.fits' % (k, idx) if idx != -1 else 'dat_%s.fits' % k)

    def _has(self, k, idx):
        return (k, idx) in self._mem or (self.cache and os.path.exists(self._fname(k, idx)))

    def _store(self, k, idx, alm):
        if self.cache:
            _write_alm(self._fname(k, idx), alm)
        else:
            self._mem[(k, idx)] = alm

    def _load(self, k, idx):
        if (k, idx) in self._mem:
            return dev.resolve(self._mem[(k, idx)])  # may still be on its way to the host (dev.host_future)
        return hp.read_alm(self._fname(k, idx))

    # ---- public getters ----------------------------------------------------------------------------
    def get_sim_qlm(self, k, idx, lmax=None):
        """QE estimate for key k and simulation idx, computed and cached on first request (qest.py:155-201)."""
        k = self.keys_remaps.get(k, k)
        if lmax is None:
            lmax = self.get_lmax_qlm(k)
        assert lmax <= self.get_lmax_qlm(k)
        if k in ['p_tp', 'x_tp', 'f_tp', 's_tp']:
            return self.get_sim_qlm('%stt' % k[0], idx, lmax=lmax) + self.get_sim_qlm('%s_p' % k[0], idx, lmax=lmax)
        if k in ['p_te', 'p_tb', 'p_eb', 'x_te', 'x_tb', 'x_eb']:
            return self.get_sim_qlm(k[0] + k[2] + k[3], idx, lmax=lmax) + self.get_sim_qlm(k[0] + k[3] + k[2], idx, lmax=lmax)
        if '_bh_' in k:
            kQE, wL = self._bh_weights(k)
            lmax = self.get_lmax_qlm(kQE)
            ksrc = k.split('_bh_')[1] + kQE[1:]
            return self.get_sim_qlm(kQE, idx, lmax=lmax) - hp.almxfl(self.get_sim_qlm(ksrc, idx, lmax=lmax), wL)
        assert k in self.keys_fund, (k, self.keys_fund)
        if not self._has(k, idx):
            if k in ['ptt', 'xtt']: self._build_sim_Tgclm(idx)
            elif k in ['p_p', 'x_p']: self._build_sim_Pgclm(idx)
            elif k in ['p', 'x']: self._build_sim_MVgclm(idx)
            elif k in ['f']: self._build_sim_f(idx)
            elif k in ['stt']: self._build_sim_stt(idx)
            elif k in ['ftt']: self._build_sim_ftt(idx)
            elif k in ['f_p']: self._build_sim_f_p(idx)
            elif k in ['ntt']: self._build_sim_ntt(idx)
            elif k in ['a_p']: self._build_sim_a_p(idx)
            elif k[0] in 'px' and len(k) == 3 and k[1] in 'teb' and k[2] in 'teb':
                self._build_sim_xfiltMVgclm(idx, k)
            else:
                assert 0, k
        if not self.cache and lmax == self.get_lmax_qlm(k):
            return self._load(k, idx)  # in-memory mode: the stored array is handed over as is
        return ut.alm_copy(self._load(k, idx), lmax=lmax)

    def get_sim_qlms(self, k, idxs, lmax=None):
        """[get_sim_qlm(k, idx) for idx in idxs] (an addition to the reference's API for Monte-Carlo loops): simulations that are
        not cached yet are evaluated two at a time where the library can pair them (see _pair_getter), results as usual."""
        idxs = list(idxs)
        k_ = self.keys_remaps.get(k, k)
        full = self.get_lmax_qlm(k_) if lmax is None else lmax
        pair = self._pair_getter(k_, full)
        if pair is not None:
            todo = [i for i in idxs if not self._has(k_, i)]
            for a, b in zip(todo[0::2], todo[1::2]):
                pair(a, b)  # (evaluates and stores both simulations' gradient and curl entries)
        return [self.get_sim_qlm(k, i, lmax=lmax) for i in idxs]

    def get_dat_qlm(self, k, **kwargs):
        return self.get_sim_qlm(k, -1, **kwargs)

    def _bh_weights(self, k):
        assert self.resplib is not None, 'resplib arg necessary for this'
        kQE, ksource = k.split('_bh_')
        assert len(ksource) == 1, (ksource, kQE)
        assert self.get_lmax_qlm(kQE) == self.get_lmax_qlm(ksource + kQE[1:]), 'fix this (easy)'
        wL = self.resplib.get_response(kQE, ksource) * ut.cli(self.resplib.get_response(ksource + kQE[1:], ksource))
        return kQE, wL

    def get_sim_qlm_mf(self, k, mc_sims, lmax=None, collective=False):
        """Mean field: average of the estimates over mc_sims, cached (qest.py:206-246).

        A local computation by default, as in the reference: the calling rank loops over all of mc_sims, so the call is safe
        inside rank-sharded loops (qecl.get_sim_qcl under jobs[rank::size]).  collective=True is a COLLECTIVE call -- every rank
        of the job must make it with the same arguments: the simulations are sharded jobs[rank::size], the running sum stays on the
        device, one all-reduce (RCCL over xGMI) completes it and rank 0 writes the cache file.  The driver
        (examples/run_qlms.py -mfdd) and bench.py use that form before any sharded loop needs the mean field."""
        k = self.keys_remaps.get(k, k)
        if lmax is None:
            lmax = self.get_lmax_qlm(k)
        assert lmax <= self.get_lmax_qlm(k)
        if k in ['p_tp', 'x_tp']:
            return self.get_sim_qlm_mf('%stt' % k[0], mc_sims, lmax=lmax, collective=collective) + self.get_sim_qlm_mf('%s_p' % k[0], mc_sims, lmax=lmax, collective=collective)
        if k in ['p_te', 'p_tb', 'p_eb', 'x_te', 'x_tb', 'x_eb']:
            return self.get_sim_qlm_mf(k[0] + k[2] + k[3], mc_sims, lmax=lmax, collective=collective) \
                   + self.get_sim_qlm_mf(k[0] + k[3] + k[2], mc_sims, lmax=lmax, collective=collective)
        if '_bh_' in k:
            kQE, wL = self._bh_weights(k)
            lmax = self.get_lmax_qlm(kQE)
            ksrc = k.split('_bh_')[1] + kQE[1:]
            return self.get_sim_qlm_mf(kQE, mc_sims, lmax=lmax, collective=collective) - hp.almxfl(self.get_sim_qlm_mf(ksrc, mc_sims, lmax=lmax, collective=collective), wL)
        assert k in self.keys_fund, (k, self.keys_fund)
        fname = os.path.join(self.lib_dir, 'simMF_k1%s_%s.fits' % (k, ut.mchash(mc_sims)))
        if (not self.cache and ('mf', fname) in self._mem):
            return ut.alm_copy(self._mem[('mf', fname)], lmax=lmax)
        if not (self.cache and os.path.exists(fname)):
            this_mcs = np.unique(mc_sims)
            if len(this_mcs) == 0:
                return np.zeros(hp.Alm.getsize(lmax), dtype=complex)
            from . import parallel
            like = torch.zeros(hp.Alm.getsize(lmax), dtype=torch.complex128, device='cuda')
            MF = dev.to_host(parallel.mean_field(lambda idx: self._get_sim_qlm_dev(k, idx, lmax), this_mcs, like,
                                                 get_pair=self._pair_getter(k, lmax), collective=collective))
            if self.cache:
                if mpi.rank == 0 or not collective:
                    _write_alm(fname, MF)
                    print("Cached ", fname)
                if collective:
                    mpi.barrier()
            else:
                self._mem[('mf', fname)] = MF
            return ut.alm_copy(MF, lmax=lmax)
        return ut.alm_copy(hp.read_alm(fname), lmax=lmax)

    def _pair_getter(self, k, lmax):
        """(idx0, idx1) -> the two device estimates, evaluated together (the leg syntheses of the two simulations share their Legendre
        recursions, pl_alm2map_batch2), or None when this key / library does not pair: 'p' / 'x' (minimum variance), 'p_p' / 'x_p'
        (polarization) and 'ptt' / 'xtt' (temperature: the filter stage only) of a same-legs library at its full band-limit.
        options.opts.batch2 = False disables."""
        k = self.keys_remaps.get(k, k)
        fam = self._GC_FAMILY.get(k)
        if (fam is None or k not in self.keys_fund or lmax != self.get_lmax_qlm(k) or not self._same_legs()
                or not options.opts.batch2):
            return None
        build = {'p': self._build_sim_MVgclm_pair, 'p_p': self._build_sim_Pgclm_pair, 'ptt': self._build_sim_Tgclm_pair}[fam[0]]
        which = fam[1]

        def get_pair(idx0, idx1):
            if self._has(k, idx0) or self._has(k, idx1):
                return self._get_sim_qlm_dev(k, idx0, lmax), self._get_sim_qlm_dev(k, idx1, lmax)
            sim_lib = getattr(self.f2map1.ivfs, 'sim_lib', None)
            if fam[0] != 'ptt' and hasattr(sim_lib, 'hint_pair'):
                sim_lib.hint_pair(idx0, idx1)  # simulation libraries that make their maps on the device pair the sky syntheses too
            (r0, r1) = build(idx0, idx1)
            return r0[which], r1[which]
        return get_pair

    # gradient key of the evaluation that serves (gradient, curl) key pairs: ('p', 'x') come out of one MV evaluation, ...
    _GC_FAMILY = {'p': ('p', 0), 'x': ('p', 1), 'ptt': ('ptt', 0), 'xtt': ('ptt', 1), 'p_p': ('p_p', 0), 'x_p': ('p_p', 1)}

    def _get_sim_qlm_dev(self, k, idx, lmax):
        """get_sim_qlm as a device tensor: the evaluation (and its cache entry) is the public one; when it has just run its
        device result is handed over instead of waiting for the host copy and uploading it again."""
        k = self.keys_remaps.get(k, k)
        fam = self._GC_FAMILY.get(k)
        if fam is not None and k in self.keys_fund and lmax == self.get_lmax_qlm(k) and self._same_legs() and not self._has(k, idx):
            self._last_dev_key = None
            {'p': self._build_sim_MVgclm, 'ptt': self._build_sim_Tgclm, 'p_p': self._build_sim_Pgclm}[fam[0]](idx)
            if self._last_dev_key == (fam[0], idx, False):
                return self._last_dev[fam[1]]
        return dev.to_dev(self.get_sim_qlm(k, idx, lmax=lmax), torch.complex128)

    # ---- estimators (device) -----------------------------------------------------------------------
    def _legs(self, swapped):
        return (self.f2map2, self.f2map1) if swapped else (self.f2map1, self.f2map2)

    def _gc_from_product(self, re, im, lmax_key):
        """-sqrt(L(L+1)) map2alm_spin_1(re, im), weight fused into the analysis (qest.py:259-262,280-284)."""
        lmax = self.lmax_qlm[lmax_key]
        G, C = shts.map2alm_spin([re, im], 1, lmax=lmax, fl=_lens_weight(lmax))
        return G, C

    def _t_product(self, idx, k, swapped=False, xfilt1=None, xfilt2=None):
        """Tb(n) x spin-1 gradient leg: the real-space product of the T estimator (qest.py:254-257)."""
        f2map1, f2map2 = self._legs(swapped)
        xf1, xf2 = (xfilt2, xfilt1) if swapped else (xfilt1, xfilt2)
        tmap = f2map1.get_irestmap(idx, xfilt=xf1)
        G, C = f2map2.get_gtmap(idx, k=k, xfilt=xf2)
        return dev.qe_lens_product((tmap, G, C), None)  # both components in one pass over the three maps

    def _p_product(self, idx, k, swapped=False, xfilt1=None, xfilt2=None):
        """(Qb - iUb)(3G + i 3C) - (Qb + iUb)(1G - i 1C): the real-space product of the P estimator (qest.py:273-278)."""
        f2map1, f2map2 = self._legs(swapped)
        xf1, xf2 = (xfilt2, xfilt1) if swapped else (xfilt1, xfilt2)
        repmap, impmap = f2map1.get_irespmap(idx, xfilt=xf1)
        g3, c3 = f2map2.get_gpmap(idx, 3, k=k, xfilt=xf2)
        g1, c1 = f2map2.get_gpmap(idx, 1, k=k, xfilt=xf2)
        return dev.qe_lens_product(None, (repmap, impmap, g3, c3, g1, c1))  # both terms in one pass over the six leg maps

    def _get_sim_Tgclm_dev(self, idx, k, swapped=False, xfilt1=None, xfilt2=None):
        G, C = self._t_product(idx, k, swapped=swapped, xfilt1=xfilt1, xfilt2=xfilt2)
        return self._gc_from_product(G, C, 'T')

    def _get_sim_Pgclm_dev(self, idx, k, swapped=False, xfilt1=None, xfilt2=None):
        dre, dim = self._p_product(idx, k, swapped=swapped, xfilt1=xfilt1, xfilt2=xfilt2)
        return self._gc_from_product(dre, dim, 'P')

    def _get_sim_Tgclm(self, idx, k, swapped=False, xfilt1=None, xfilt2=None, defer=False):
        """T-only lensing gradient / curl (qest.py:248-263).  defer: host copies as dev.host_future objects."""
        G, C = self._get_sim_Tgclm_dev(idx, k, swapped=swapped, xfilt1=xfilt1, xfilt2=xfilt2)
        if xfilt1 is None and xfilt2 is None:
            self._last_dev, self._last_dev_key = (G, C), ('ptt', idx, swapped)
        if defer and G.numel() >= self._DEFER_MIN_ENTRIES:
            return dev.host_future(G), dev.host_future(C)
        return dev.to_host(G), dev.to_host(C)

    def _get_sim_Pgclm(self, idx, k, swapped=False, xfilt1=None, xfilt2=None, defer=False):
        """Polarization-only lensing gradient / curl (qest.py:265-285).  defer: host copies as dev.host_future objects."""
        G, C = self._get_sim_Pgclm_dev(idx, k, swapped=swapped, xfilt1=xfilt1, xfilt2=xfilt2)
        if xfilt1 is None and xfilt2 is None:
            self._last_dev, self._last_dev_key = (G, C), ('p_p', idx, swapped)
        if defer and G.numel() >= self._DEFER_MIN_ENTRIES:
            return dev.host_future(G), dev.host_future(C)
        return dev.to_host(G), dev.to_host(C)

    def _get_sim_MVgclm(self, idx, k, swapped=False, defer=False):
        """Minimum-variance estimator = P part + T part (qest.py:318-322).  The reference analyses the two
        product maps separately and adds the alms; map2alm_spin is linear, so the product maps are summed on the
        device and analysed once (one spin-1 transform less, results equal to rounding).
        defer: return the host copies as dev.host_future objects (the copies overlap whatever is issued next)."""
        assert k == 'p'
        f2map1, f2map2 = self._legs(swapped)
        # The five leg syntheses are independent: inside a lane the FFT stage of a synthesis runs on a side stream (own plan
        # fork and phase buffer) while the current stream goes on with the Legendre stage of the next one; joined before
        # the pixel product.
        lanes = self.pipeline_lanes  # measured +-2 % at nside 2048: off
        ln = (lambda i: shts.lane(i if lanes else 0))
        with ln(1):
            tmap = f2map1.get_irestmap(idx)
        with ln(3):
            rep, imp = f2map1.get_irespmap(idx)
        with ln(4):
            g3, c3 = f2map2.get_gpmap(idx, 3, k='p')
        if lanes:
            with ln(2):
                gt, ct = f2map2.get_gtmap(idx, k='p')
            with ln(5):
                g1, c1 = f2map2.get_gpmap(idx, 1, k='p')
        else:  # the two spin-1 legs share one Legendre recursion
            (gt, ct), (g1, c1) = f2map2.get_gt_gp1maps(idx, k='p')
        shts.join_lanes()
        dre, dim = dev.qe_lens_product((tmap, gt, ct), (rep, imp, g3, c3, g1, c1))  # all nine leg maps in one pass
        del tmap, gt, ct, rep, imp, g3, c3, g1, c1
        G, C = self._gc_from_product(dre, dim, 'P')
        self._last_dev, self._last_dev_key = (G, C), ('p', idx, swapped)
        if defer and G.numel() >= self._DEFER_MIN_ENTRIES:
            return dev.host_future(G), dev.host_future(C)
        return dev.to_host(G), dev.to_host(C)

    # ---- two simulations at a time: the leg syntheses of a pair share Legendre recursions ---------------------------------------------
    # Each family has a device part (`_pair_dev_*`: filtered alms resident -> device (G, C) of both simulations, no host interaction:
    # what `_pair_graph` captures) and the common host part (`_host_pair`).
    def _pair_dev_p(self, idx0, idx1, emit=None):
        """_get_sim_MVgclm of two simulations whose spin-2 and spin-3 leg syntheses share their Legendre recursions
        (lib_filt2map.get_irespmap_batch2 / get_gpmap_batch2; same legs on both sides only)."""
        f2map1, f2map2 = self._legs(False)
        idxs = (idx0, idx1)
        tmaps = f2map1.get_irestmap_batch2(idx0, idx1)  # (the two inverse-variance filtered T maps on one Legendre recursion)
        resp = f2map1.get_irespmap_batch2(idx0, idx1)
        gp3 = f2map2.get_gpmap_batch2(idx0, idx1, 3, k='p')
        out = []
        for j, idx in enumerate(idxs):
            (gt, ct), (g1, c1) = f2map2.get_gt_gp1maps(idx, k='p')
            dre, dim = dev.qe_lens_product((tmaps[j], gt, ct), (resp[j][0], resp[j][1], gp3[j][0], gp3[j][1], g1, c1))
            del gt, ct, g1, c1
            out.append(tuple(self._gc_from_product(dre, dim, 'P')))
            if emit is not None:
                emit(*out[-1])
        return out

    def _pair_dev_p_p(self, idx0, idx1, emit=None):
        """_get_sim_Pgclm of two simulations: the spin-2, spin-3 and spin-1 leg syntheses each serve both on one Legendre recursion
        (lib_filt2map.get_irespmap_batch2 / get_gpmap_batch2; same legs on both sides only); maps bit-identical to the one-by-one
        evaluation."""
        f2map1, f2map2 = self._legs(False)
        resp = f2map1.get_irespmap_batch2(idx0, idx1)
        gp3 = f2map2.get_gpmap_batch2(idx0, idx1, 3, k='p_p')
        gp1 = f2map2.get_gpmap_batch2(idx0, idx1, 1, k='p_p')
        out = []
        for j in (0, 1):
            dre, dim = dev.qe_lens_product(None, (resp[j][0], resp[j][1], gp3[j][0], gp3[j][1], gp1[j][0], gp1[j][1]))
            out.append(tuple(self._gc_from_product(dre, dim, 'P')))
            if emit is not None:
                emit(*out[-1])
        return out

    def _pair_dev_ptt(self, idx0, idx1, emit=None):
        """_get_sim_Tgclm of two simulations: their gradient legs (gradient-only spin-1 syntheses, 8 FMAs per step each) share one
        Legendre recursion (lib_filt2map.get_gtmap_pair, pl_alm2map_grad_pair: 12 for the two; maps bit-identical to the one-by-one
        evaluation), and the filter stage of both is issued before either estimator."""
        f2map1, f2map2 = self._legs(False)
        for i in (idx0, idx1):
            f2map1._alm('tlm', i)
        gts = f2map2.get_gtmap_pair(idx0, idx1, k='ptt')  # both gradient legs on one recursion (None: one by one)
        out = []
        for j, idx in enumerate((idx0, idx1)):
            if gts is None:
                out.append(tuple(self._get_sim_Tgclm_dev(idx, 'ptt')))
            else:
                if j == 0:
                    tmaps = f2map1.get_irestmap_batch2(idx0, idx1)
                out.append(tuple(self._gc_from_product(*dev.qe_lens_product((tmaps[j], gts[j][0], gts[j][1]), None), 'T')))
            if emit is not None:
                emit(*out[-1])
        return out

    def _host_pair(self, G, C, defer):
        """(G host, C host, G device, C device); the host entries are dev.host_future objects when `defer`"""
        if defer:  # (small results: copies nobody waits for until they are read -- no helper thread, see dev.host_future)
            big = G.numel() >= self._DEFER_MIN_ENTRIES
            return (dev.host_future(G, threaded=big), dev.host_future(C, threaded=big), G, C)
        return (dev.to_host(G), dev.to_host(C), G, C)

    def _pair(self, fam, idx0, idx1, defer):
        """per simulation (G host, C host, G device, C device)"""
        gcs = self._pair_graph(fam, idx0, idx1)  # one replayed HIP graph where the libraries allow it (None: not here)
        if gcs is not None:
            out = [self._host_pair(G, C, defer) for G, C in gcs]
        else:  # eager: a simulation's results start for the host as soon as they exist -- the copies of the first run beside the (FMA-bound)
            out = []  # Legendre kernels of the second instead of beside the bandwidth-bound first stages of the next pair
            getattr(self, '_pair_dev_' + fam)(idx0, idx1, emit=lambda G, C: out.append(self._host_pair(G, C, defer)))
        self._last_dev, self._last_dev_key = (out[1][2], out[1][3]), (fam, idx1, False)
        return out

    def _get_sim_MVgclm_pair(self, idx0, idx1, defer=False):
        return self._pair('p', idx0, idx1, defer)

    def _get_sim_Pgclm_pair(self, idx0, idx1, defer=False):
        return self._pair('p_p', idx0, idx1, defer)

    def _get_sim_Tgclm_pair(self, idx0, idx1, defer=False):
        return self._pair('ptt', idx0, idx1, defer)

    # ---- a pair of reconstructions as ONE replayed HIP graph ---------------------------------------------------------------------------
    # From the input maps to the device (G, C) of both simulations a pair is ~400 kernel launches (two filters, five leg syntheses with
    # their ring-FFT classes on side streams, the products, two analyses) with no host-side data dependence: launched from one Python
    # thread they make the rate depend on how quiet the host is (8 ranks share one host), and at small sizes (nside 512: 1.7 ms per
    # reconstruction for 0.1 ms of arithmetic) the launches ARE the cost.  After `graph_after` eager evaluations the sequence is captured
    # once (torch.cuda.CUDAGraph: libplshts launches on torch's current stream, its side-stream forks / joins are captured with it)
    # and replayed; per pair the host then copies / adopts the input maps, replays, and starts the device -> host copies.  The captured
    # code is the eager code (`_pair_body`), so the results are the eager ones bit for bit (tests/test_gpu_qe.py, test_gpu_fullsize.py).
    # The reference has no counterpart: its loop (qest.py:238-244, examples/run_qlms.py:66-74) is CPU code.
    use_graph = True   # per instance; options.opts.qe_graph = False switches the route off for the process
    graph_min_nside = None  # smallest grid served through the replayed graph (None: options.opts.qe_graph_min_nside); see _pair_graph_ok
    graph_fallbacks = 0  # captures of this library that failed and fell back to eager launches for good (process totals: options.stats)
    graph_after = 2    # eager pair evaluations before the capture (workspaces grown, filters uploaded, code objects loaded)

    _PAIR_FIELDS = {'p': 'tqu', 'p_p': 'qu', 'ptt': 't'}

    def _pair_graph_ok(self, fam, idx0, idx1):
        ivfs = self.f2map1.ivfs
        if not self.use_graph or not options.opts.qe_graph or self.pipeline_lanes or shts.lane_active():
            return False
        if torch.cuda.is_current_stream_capturing() or not self._same_legs() or self.cache:
            return False
        # the filter must be a pure device function of the maps: isotropic filter classes without a file cache or a starting-point library
        if not (hasattr(ivfs, '_apply_ivf_t') and hasattr(ivfs, '_apply_ivf_p') and hasattr(ivfs, '_dev_entry') and hasattr(ivfs, 'nside')):
            return False
        # A size threshold for the replayed route (default 0: none).  One hipGraphLaunch of a pair's ~400 nodes costs the host about what 400
        # launches cost, so at nside 512 (1.4 ms of GPU work per pair) the two routes are equal within the noise of a shared host: 0.90-0.94 ms per
        # reconstruction eager, 0.77-1.29 replayed (profiles/round6_h_small_inputs_ab.txt); from nside 1024 on the replay's immunity to a busy host counts.
        min_nside = options.opts.qe_graph_min_nside if self.graph_min_nside is None else self.graph_min_nside
        if ivfs.nside < min_nside:
            return False
        if getattr(ivfs, 'cache', True) or getattr(ivfs, 'soltn_lib', None) is not None or not hasattr(ivfs, 'sim_lib'):
            return False
        if type(self.f2map1) not in (lib_filt2map, lib_filt2map_sepTP):
            return False
        for idx in (idx0, idx1):  # filtered alms already resident (another key ran first): the eager route reuses them
            if idx in ivfs._dev_cache and not ivfs._dev_cache[idx].get('_graph_static', False):
                return False
        # per-stage HIP-event timing (pl_profile_enable) records events on the launch stream: not inside a captured region
        return not any(getattr(pl, '_profiling', False) for pl in shts._PLANS.values())

    def _pair_inputs(self, fam, idxs):
        """the input maps of the pair in slot order: per simulation T (if used), then Q, U (if used)"""
        sim_lib = self.f2map1.ivfs.sim_lib
        maps = []
        for idx in idxs:
            if 't' in self._PAIR_FIELDS[fam]:
                maps.append(sim_lib.get_sim_tmap(idx))
            if 'q' in self._PAIR_FIELDS[fam]:
                maps += list(sim_lib.get_sim_pmap(idx))
        return maps

    def _pair_body(self, fam, idx0, idx1, maps):
        """filter of both simulations (entries of the filter library's device cache, as get_sim_alm_dev makes them), then the paired
        estimator: device work only.  maps: device tensors in slot order (Q, U the two rows of one array)."""
        ivfs = self.f2map1.ivfs
        it = iter(maps)
        ents = []
        for idx in (idx0, idx1):
            ent = ivfs._dev_entry(idx)
            ent.clear()
            if 't' in self._PAIR_FIELDS[fam]:
                ent['t'] = dev.to_dev(ivfs._apply_ivf_t(next(it), soltn=None))
            if 'q' in self._PAIR_FIELDS[fam]:
                e, b = ivfs._apply_ivf_p([next(it), next(it)], soltn=None)
                ent['e'], ent['b'] = dev.to_dev(e), dev.to_dev(b)
            ents.append(ent)
        return getattr(self, '_pair_dev_' + fam)(idx0, idx1), ents

    def _pair_graph(self, fam, idx0, idx1):
        if not self._pair_graph_ok(fam, idx0, idx1):
            return None
        import gc
        ivfs = self.f2map1.ivfs
        npix = hp.nside2npix(ivfs.nside)
        nslots = 2 * len(self._PAIR_FIELDS[fam])
        st = self.__dict__.setdefault('_pair_graphs', {}).setdefault((fam, shts.context(), torch.cuda.current_device()),
                                                                    {'calls': 0, 'graph': None, 'held': None})
        sim_lib = ivfs.sim_lib
        # a simulation library that writes its maps into buffers of the caller (sims.maps.cmb_maps with device maps) fills the static
        # input slots itself: the pass that adds the noise is the one that writes the slot
        into = (isinstance(st['graph'], torch.cuda.CUDAGraph) and getattr(sim_lib, 'device_maps', False)
                and hasattr(sim_lib, 'get_sim_tmap_into') and hasattr(sim_lib, 'get_sim_pmap_into'))
        maps = None
        if not into:
            maps = self._pair_inputs(fam, (idx0, idx1))
            if len(maps) != nslots or any((m.numel() if isinstance(m, torch.Tensor) else np.size(m)) != npix for m in maps):
                return None

        def evict():  # cache entries of the filter library that alias the graph's static alms: stale once the inputs change
            for idx in [i for i, ent in ivfs._dev_cache.items() if ent.get('_graph_static', False)]:
                ivfs._dev_cache.pop(idx)

        # Inputs of the captured launches.  Indirect (default): the graph's analyses read their maps through a table of device addresses
        # (shts.MapRef, pl_map2alm_ind) -- a replay on other device-resident maps rewrites 8 bytes per map instead of copying 8 npix; only host
        # arrays (uploaded) and simulation libraries that write into buffers of the caller get slots of the graph's own.  Otherwise (filters
        # that touch the maps before the transform, options.opts.qe_indirect off): static slots, device inputs copied into them.
        from .filt import filt_simple as _fs
        indirect = bool(options.opts.qe_indirect) and getattr(type(ivfs), '_mask', None) is _fs.library_fullsky_sepTP._mask  # (the maps go straight into the transforms)
        if isinstance(st['graph'], torch.cuda.CUDAGraph):
            indirect = st.get('indirect', False)  # (as captured)

        def own_slot(k):
            own = st.setdefault('own', {})
            if k not in own:
                own[k] = torch.empty(npix, dtype=torch.float64, device=dev.device())
            return own[k]

        def fill_indirect():
            addrs, hold = list(st['addr']), []
            if into:
                k = 0
                for idx in (idx0, idx1):
                    if 't' in self._PAIR_FIELDS[fam]:
                        sim_lib.get_sim_tmap_into(idx, own_slot(k))
                        addrs[k] = own_slot(k).data_ptr()
                        k += 1
                    if 'q' in self._PAIR_FIELDS[fam]:
                        sim_lib.get_sim_pmap_into(idx, own_slot(k), own_slot(k + 1))
                        addrs[k], addrs[k + 1] = own_slot(k).data_ptr(), own_slot(k + 1).data_ptr()
                        k += 2
            else:
                for k, m in enumerate(maps):
                    if isinstance(m, torch.Tensor) and m.is_cuda and m.dtype == torch.float64 and m.is_contiguous() and m.data_ptr() % 16 == 0:
                        t = m.reshape(-1)
                        addrs[k] = t.data_ptr()
                        hold.append(t)  # alive until the next pair's inputs are in place: the replay that reads it is enqueued by then
                    elif isinstance(m, torch.Tensor):
                        own_slot(k).copy_(m.reshape(-1), non_blocking=True)
                        addrs[k] = own_slot(k).data_ptr()
                    else:
                        own_slot(k).copy_(torch.from_numpy(np.ascontiguousarray(m, dtype=np.float64).reshape(-1)))
                        addrs[k] = own_slot(k).data_ptr()
            if addrs != st['addr']:
                # (by a one-workgroup kernel that takes the addresses by value: stream-ordered behind the previous replay, nothing on the host to keep alive)
                shts.store_addresses(addrs, st['ptab'])
                st['addr'] = addrs
            st['held'] = hold

        def fill(slots):
            """input maps into the static slots.  A device tensor that a `stable_maps` simulation library hands out again (same
            storage, shape and version: maps resident in HBM) is already there from the last replay; host arrays are uploaded straight
            into their slot."""
            if into:
                it = iter(slots)
                for idx in (idx0, idx1):
                    if 't' in self._PAIR_FIELDS[fam]:
                        sim_lib.get_sim_tmap_into(idx, next(it))
                    if 'q' in self._PAIR_FIELDS[fam]:
                        sim_lib.get_sim_pmap_into(idx, next(it), next(it))
                st['tags'] = [None] * nslots
                return
            stable = getattr(sim_lib, 'stable_maps', False)
            for k, (slot, m) in enumerate(zip(slots, maps)):
                if isinstance(m, torch.Tensor):
                    tag = (m.data_ptr(), m._version, tuple(m.shape), m.dtype) if stable else None
                    if tag is not None and st['tags'][k] == tag:
                        continue
                    slot.copy_(m.reshape(-1), non_blocking=True)
                    st['tags'][k] = tag
                else:
                    slot.copy_(torch.from_numpy(np.ascontiguousarray(m, dtype=np.float64).reshape(-1)))
                    st['tags'][k] = None

        if st['graph'] is None or st['graph'] is False:
            st['calls'] += 1
            if st['graph'] is False or st['calls'] <= self.graph_after:
                evict()
                dmaps = self._pair_dev_maps(fam, maps)
                return self._pair_body(fam, idx0, idx1, dmaps)[0]
            try:
                evict()
                st['indirect'] = indirect
                if indirect:  # inputs through a pointer table: entries = the slot order of _pair_inputs
                    st['ptab'] = torch.zeros(nslots, dtype=torch.int64, device=dev.device())
                    st['addr'] = [0] * nslots
                    st['in'] = [shts.MapRef(st['ptab'], k, npix) for k in range(nslots)]
                    fill_indirect()
                else:
                    # static input slots: per simulation [T] and / or the two rows of one (2, npix) array for (Q, U)
                    st['in'], st['tags'] = [], [None] * nslots
                    for _ in range(2):
                        if 't' in self._PAIR_FIELDS[fam]:
                            st['in'].append(torch.empty(npix, dtype=torch.float64, device=dev.device()))
                        if 'q' in self._PAIR_FIELDS[fam]:
                            qu = torch.empty((2, npix), dtype=torch.float64, device=dev.device())
                            st['in'] += [qu[0], qu[1]]
                    fill(st['in'])
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                gc_was_on = gc.isenabled()
                gc.disable()  # (finalisers of unrelated garbage make HIP calls that are illegal while capturing: qcinv.multigrid)
                try:
                    with torch.cuda.graph(g, capture_error_mode='thread_local'):
                        gcs, ents = self._pair_body(fam, -(10 ** 9) - 1, -(10 ** 9) - 2, st['in'])
                finally:
                    if gc_was_on:
                        gc.enable()
                for i in (-(10 ** 9) - 1, -(10 ** 9) - 2):
                    ivfs._dev_cache.pop(i, None)
                st['out'] = gcs
                st['ents'] = [dict(ent, _graph_static=True) for ent in ents]
                st['graph'] = g
                st['first'] = True
                options.count('qe_graph_captures')
            except Exception as e:  # capture is an optimisation: stay eager for good -- counted, and the reason kept, for whoever asks
                self.graph_fallbacks += 1
                options.count('qe_graph_fallbacks', 'qest.library %s pair: %s' % (fam, str(e).split('\n')[0]))
                if options.opts.debug:
                    import traceback
                    traceback.print_exc()
                torch.cuda.synchronize()
                for i in (-(10 ** 9) - 1, -(10 ** 9) - 2):
                    ivfs._dev_cache.pop(i, None)
                st['graph'] = False
                return self._pair_body(fam, idx0, idx1, self._pair_dev_maps(fam, maps))[0]
        evict()
        if st.pop('first', False):
            pass  # (the inputs were put in place for the capture)
        elif st.get('indirect', False):
            fill_indirect()
        else:
            fill(st['in'])
        st['graph'].replay()
        for idx, ent in zip((idx0, idx1), st['ents']):
            # the filtered alms of the pair stay available to further keys UNTIL THE NEXT REPLAY of this graph (they alias its static buffers:
            # entries marked _graph_static; a caller that keeps one across pairs must clone it)
            ivfs._dev_entry(idx).update(ent)
        # The results leave the graph's static outputs at once (4 x 33.6 MB at lmax 2048: ~0.1 ms of copies per pair): what the caller holds --
        # the device (G, C) it sums or keeps as _last_dev, and the source of the device -> host copies on the copy stream -- is its own
        # memory, valid for as long as it is referenced, and the next replay need not wait for the previous pair's results to have crossed
        # PCIe (it used to: the copy stream read the static outputs, 4 x 0.6 ms per pair with the launch stream idle behind them).
        return [(G.clone(), C.clone()) for G, C in st['out']]

    def _pair_dev_maps(self, fam, maps):
        """input maps as device tensors for the eager body: (Q, U) as the two rows of one array (what the spin transform takes)"""
        out, k = [], 0
        for _ in range(2):
            if 't' in self._PAIR_FIELDS[fam]:
                out.append(dev.to_dev(maps[k], torch.float64).reshape(-1))
                k += 1
            if 'q' in self._PAIR_FIELDS[fam]:
                q, u = maps[k], maps[k + 1]
                k += 2
                if isinstance(q, torch.Tensor) and isinstance(u, torch.Tensor):
                    out += [dev.to_dev(q, torch.float64).reshape(-1), dev.to_dev(u, torch.float64).reshape(-1)]
                else:
                    buf = torch.empty((2, np.size(q)), dtype=torch.float64, device=dev.device())
                    buf[0].copy_(torch.from_numpy(np.ascontiguousarray(q, dtype=np.float64).reshape(-1)))
                    buf[1].copy_(torch.from_numpy(np.ascontiguousarray(u, dtype=np.float64).reshape(-1)))
                    out += [buf[0], buf[1]]
        return out

    def _scalar_from_product(self, prod, fac, lmax_key):
        lmax = self.get_lmax_qlm(lmax_key)
        return dev.to_host(shts.map2alm(prod, lmax=lmax, iter=0)) * fac

    def _get_sim_stt(self, idx, swapped=False):
        """Point-source estimator -1/2 map2alm(Tb1 Tb2) (qest.py:287-291)."""
        f1, f2 = self._legs(swapped)
        return self._scalar_from_product(dev.map_mul(f1.get_irestmap(idx), f2.get_irestmap(idx)), -0.5, 'PS')

    def _get_sim_ntt(self, idx, swapped=False):
        """Noise-inhomogeneity estimator on beam-deconvolved maps (qest.py:293-298)."""
        f1, f2 = self._legs(swapped)
        t1 = f1.get_wirestmap(idx, f1.ivfs.get_tal('t')[:])
        t2 = f2.get_wirestmap(idx, f2.ivfs.get_tal('t')[:])
        return self._scalar_from_product(dev.map_mul(t1, t2), -0.5, 'T')

    def _get_sim_ftt(self, idx, joint=False, swapped=False):
        """Temperature modulation estimator -map2alm(Tb T^WF) (qest.py:300-304)."""
        f1, f2 = self._legs(swapped)
        return self._scalar_from_product(dev.map_mul(f1.get_irestmap(idx), f2.get_tmap(idx, joint=joint)), -1., 'T')

    def _get_sim_f_p(self, idx, joint=False, swapped=False):
        """Polarization modulation estimator -2 map2alm(Qb Q^WF + Ub U^WF) (qest.py:306-310)."""
        f1, f2 = self._legs(swapped)
        Q1, U1 = f1.get_irespmap(idx)
        Q2, U2 = f2.get_pmap(idx, joint=joint)
        return self._scalar_from_product(Q1 * Q2 + U1 * U2, -2., 'P')

    def _get_sim_a_p(self, idx, joint=False, swapped=False):
        """Polarization rotation estimator -4 map2alm(Qb U^WF - Ub Q^WF) (qest.py:312-316)."""
        f1, f2 = self._legs(swapped)
        Q1, U1 = f1.get_irespmap(idx)
        Q2, U2 = f2.get_pmap(idx, joint=joint)
        return self._scalar_from_product(Q1 * U2 - U1 * Q2, -4., 'P')

    # ---- builders: symmetrise when the legs differ, then cache ---------------------------------------
    def _same_legs(self):
        return self.f2map1.ivfs == self.f2map2.ivfs

    def _sym_gc(self, fun, idx, *args, **kwargs):
        G, C = fun(idx, *args, **kwargs)
        if not self._same_legs():
            _G, _C = fun(idx, *args, swapped=True, **kwargs)
            G, C = 0.5 * (G + _G), 0.5 * (C + _C)
        return G, C

    # results smaller than this (complex entries) are copied with a blocking call: below ~lmax 700 a reconstruction is a
    # millisecond of launch-bound work and the copy-stream / helper-thread hand-off costs more than the 2 MB copy it hides
    # (nside = lmax = 512 'ptt': 1.65 ms per reconstruction blocking, 4.2 ms deferred)
    _DEFER_MIN_ENTRIES = 1 << 18

    def _defer_ok(self):
        """in-memory results of a same-legs library may be stored while still crossing PCIe (resolved by _load);
        options.opts.async_d2h = False: blocking copies"""
        return self._same_legs() and not self.cache and options.opts.async_d2h

    def _build_sim_Tgclm(self, idx):
        G, C = self._get_sim_Tgclm(idx, 'ptt', defer=True) if self._defer_ok() else self._sym_gc(self._get_sim_Tgclm, idx, 'ptt')
        self._store('ptt', idx, G)
        self._store('xtt', idx, C)

    def _build_sim_Pgclm(self, idx):
        G, C = self._get_sim_Pgclm(idx, 'p_p', defer=True) if self._defer_ok() else self._sym_gc(self._get_sim_Pgclm, idx, 'p_p')
        self._store('p_p', idx, G)
        self._store('x_p', idx, C)

    def _build_sim_MVgclm(self, idx):
        if self._defer_ok():
            G, C = self._get_sim_MVgclm(idx, 'p', defer=True)
        else:
            G, C = self._sym_gc(self._get_sim_MVgclm, idx, 'p')
        self._store('p', idx, G)
        self._store('x', idx, C)

    def _build_sim_MVgclm_pair(self, idx0, idx1):
        """both simulations' ('p', 'x') entries from one paired evaluation; returns their device (G, C)"""
        res = self._get_sim_MVgclm_pair(idx0, idx1, defer=self._defer_ok())
        for idx, (G, C, _, _) in zip((idx0, idx1), res):
            self._store('p', idx, G)
            self._store('x', idx, C)
        return [(r[2], r[3]) for r in res]

    def _build_sim_Pgclm_pair(self, idx0, idx1):
        """both simulations' ('p_p', 'x_p') entries from one paired evaluation; returns their device (G, C)"""
        res = self._get_sim_Pgclm_pair(idx0, idx1, defer=self._defer_ok())
        for idx, (G, C, _, _) in zip((idx0, idx1), res):
            self._store('p_p', idx, G)
            self._store('x_p', idx, C)
        return [(r[2], r[3]) for r in res]

    def _build_sim_Tgclm_pair(self, idx0, idx1):
        """both simulations' ('ptt', 'xtt') entries; returns their device (G, C)"""
        res = self._get_sim_Tgclm_pair(idx0, idx1, defer=self._defer_ok())
        for idx, (G, C, _, _) in zip((idx0, idx1), res):
            self._store('ptt', idx, G)
            self._store('xtt', idx, C)
        return [(r[2], r[3]) for r in res]

    def _build_sim_f(self, idx):
        G = self._get_sim_f_p(idx, joint=True)
        if not self._same_legs():
            G = 0.5 * (G + self._get_sim_f_p(idx, joint=True, swapped=True))
        GT = self._get_sim_ftt(idx, joint=True)
        if not self._same_legs():
            GT = 0.5 * (GT + self._get_sim_ftt(idx, joint=True, swapped=True))
        self._store('f', idx, G + GT)

    def _build_sim_xfiltMVgclm(self, idx, k):
        """Single field-pair estimators V X_1 W Y_2 from the MV machinery with 0/1 field selectors
        (qest.py:372-402)."""
        assert k[0] in 'px' and k[1] in 'teb' and k[2] in 'teb', k
        xfilt1 = {f: (k[-2] == f) * np.ones(10000) for f in ['t', 'e', 'b']}
        xfilt2 = {f: (k[-1] == f) * np.ones(10000) for f in ['t', 'e', 'b']}
        G, C = self._sym_gc(self._get_sim_Pgclm, idx, 'p', xfilt1=xfilt1, xfilt2=xfilt2)
        GT, CT = self._sym_gc(self._get_sim_Tgclm, idx, 'p', xfilt1=xfilt1, xfilt2=xfilt2)
        self._store('p' + k[1:], idx, G + GT)
        self._store('x' + k[1:], idx, C + CT)

    def _build_sim_stt(self, idx):
        self._store('stt', idx, self._get_sim_stt(idx))  # symmetric in its legs

    def _build_sim_ntt(self, idx):
        self._store('ntt', idx, self._get_sim_ntt(idx))

    def _build_sim_ftt(self, idx):
        fLM = self._get_sim_ftt(idx)
        if not self._same_legs():
            fLM = 0.5 * (fLM + self._get_sim_ftt(idx, swapped=True))
        self._store('ftt', idx, fLM)

    def _build_sim_f_p(self, idx):
        fLM = self._get_sim_f_p(idx)
        if not self._same_legs():
            fLM = 0.5 * (fLM + self._get_sim_f_p(idx, swapped=True))
        self._store('f_p', idx, fLM)

    def _build_sim_a_p(self, idx):
        fLM = self._get_sim_a_p(idx)
        if not self._same_legs():
            # the reference symmetrises with _get_sim_f_p here (qest.py:435); parity first
            fLM = 0.5 * (fLM + self._get_sim_f_p(idx, swapped=True))
        self._store('a_p', idx, fLM)


class lib_filt2map(object):
    """Filtered alms -> real-space legs of the estimators, on the GPU (qest.py:441-530; joint T-P filtering)."""

    def __init__(self, ivfs, nside):
        self.ivfs = ivfs
        self.nside = nside

    def hashdict(self):
        return {'ivfs': self.ivfs.hashdict(), 'nside': self.nside}

    # device copies of the filtered alms; an ivfs may hand over device tensors itself (get_sim_alm_dev)
    def _alm(self, name, idx):
        getter = getattr(self.ivfs, 'get_sim_alm_dev', None)
        if getter is not None:
            t = getter(name, idx)
            if t is not None:
                return t
        return dev.to_dev(getattr(self.ivfs, 'get_sim_' + name)(idx), torch.complex128)

    @staticmethod
    def _lmax(alm):
        return hp.Alm.getlmax(alm.numel())

    def prefetch_filtered(self, idx):
        """Makes the filtered T, E, B alms of simulation idx resident on the device, the T and P filters running on two
        lanes; True when that was possible without host round trips (filter library with the device route and without a
        file cache), i.e. when every later device operation may assume the alms complete on the current stream."""
        get = getattr(self.ivfs, 'get_sim_alm_dev', None)
        if get is None or getattr(self.ivfs, 'cache', True) or not hasattr(self.ivfs, '_apply_ivf_t'):
            return False
        with shts.lane(1):
            get('tlm', idx)
        with shts.lane(2):
            get('elm', idx)
        shts.join_lanes()
        return True

    def _gt_alm(self, idx, k=None, xfilt=None):
        """gradient alm of the spin-1 temperature leg (its curl is zero), or None when it vanishes identically"""
        assert xfilt is None, 'not implemented'
        return self._alm('tmliklm', idx)

    def _gp_alms(self, idx, k=None, xfilt=None):
        """(G, C) of the spin-1 / spin-3 polarization legs (C may be None: no curl), or None when they vanish identically"""
        assert xfilt is None, 'not implemented'
        G, C = self._alm('emliklm', idx), self._alm('bmliklm', idx)
        assert G.numel() == C.numel()
        return G, C

    def get_gtmap(self, idx, k=None, xfilt=None):
        """alm2map_spin_1(-sqrt(l(l+1)) T^WF_lm, 0) (qest.py:453-464)."""
        mlik = self._gt_alm(idx, k=k, xfilt=xfilt)
        if mlik is None:
            return self._zeros()
        lmax = self._lmax(mlik)
        return shts.alm2map_spin([mlik, None], self.nside, 1, lmax, fl=_lens_weight(lmax))  # no curl: gradient-only synthesis

    def get_gtmap_pair(self, idx0, idx1, k=None):
        """get_gtmap of two simulations on one Legendre recursion (pl_alm2map_grad_pair); None where that form does not apply"""
        m0, m1 = self._gt_alm(idx0, k=k), self._gt_alm(idx1, k=k)
        if m0 is None or m1 is None or self._lmax(m0) != self._lmax(m1) or shts._lane_active() or not options.opts.batch2 \
                or not (isinstance(m0, torch.Tensor) and m0.is_cuda and isinstance(m1, torch.Tensor) and m1.is_cuda):
            return None
        lmax = self._lmax(m0)
        return shts.alm2map_spin_grad_pair(m0, m1, self.nside, 1, lmax, fl=_lens_weight(lmax))

    def get_gt_gp1maps(self, idx, k=None):
        """(gt, ct), (g1, c1): the two spin-1 legs of the minimum-variance estimator, get_gtmap and get_gpmap(spin 1), on one
        Legendre recursion when both exist with the same band-limit (pl_alm2map_pair)."""
        mlik, gc = self._gt_alm(idx, k=k), self._gp_alms(idx, k=k)
        if mlik is None or gc is None or gc[1] is None or self._lmax(mlik) != self._lmax(gc[0]) or shts._lane_active():
            return self.get_gtmap(idx, k=k), self.get_gpmap(idx, 1, k=k)
        lmax = self._lmax(mlik)
        p1, t1 = shts.alm2map_spin_pair(list(gc), mlik, self.nside, 1, lmax, fl=_spin_weight(1, lmax), fl2=_lens_weight(lmax))
        return t1, p1

    def _zeros(self):
        z = torch.zeros(hp.nside2npix(self.nside), dtype=torch.float64, device=dev.device())
        return [z, z.clone()]

    def get_tmap(self, idx, joint=False):
        return shts.alm2map(self._alm('tmliklm', idx), self.nside)

    def get_pmap(self, idx, joint=False):
        G, C = self._alm('emliklm', idx), self._alm('bmliklm', idx)
        return shts.alm2map_spin([G, C], self.nside, 2, self._lmax(G))

    def get_gpmap(self, idx, spin, k=None, xfilt=None):
        """alm2map_spin_s(w^s_l (E^WF, B^WF)), s = 1, 3 (qest.py:481-504)."""
        assert spin in [1, 3]
        gc = self._gp_alms(idx, k=k, xfilt=xfilt)
        if gc is None:
            return self._zeros()
        lmax = self._lmax(gc[0])
        return shts.alm2map_spin(list(gc), self.nside, spin, lmax, fl=_spin_weight(spin, lmax))  # C None: gradient-only synthesis

    def get_irestmap(self, idx, xfilt=None):
        if xfilt is not None:
            assert isinstance(xfilt, dict) and 't' in xfilt.keys()
            if not np.any(xfilt['t']):
                return torch.zeros(hp.nside2npix(self.nside), dtype=torch.float64, device=dev.device())
        reslm = self._alm('tlm', idx)
        return shts.alm2map(reslm, self.nside, lmax=self._lmax(reslm), fl=None if xfilt is None else xfilt['t'])

    def get_wirestmap(self, idx, wl):
        reslm = self._alm('tlm', idx)
        return shts.alm2map(reslm, self.nside, lmax=self._lmax(reslm), fl=wl)

    def get_irespmap(self, idx, xfilt=None):
        """(Qb, Ub) = alm2map_spin_2(Eb / 2, Bb / 2) (qest.py:521-530)."""
        e, b = self._alm('elm', idx), self._alm('blm', idx)
        assert e.numel() == b.numel()
        lmax = self._lmax(e)
        if xfilt is not None:
            assert isinstance(xfilt, dict) and 'e' in xfilt.keys() and 'b' in xfilt.keys()
            e, b = dev.almxfl(e, xfilt['e']), dev.almxfl(b, xfilt['b'])
        return shts.alm2map_spin([e, b], self.nside, 2, lmax, fl=0.5 * np.ones(lmax + 1))

    def get_irestmap_batch2(self, idx0, idx1):
        """get_irestmap of two simulations: their scalar syntheses in one call (pl_alm2map_batch2 with spin 0: one Legendre recursion for both on
        fine grids); maps bit-identical to the one-by-one evaluation, which serves every case the pair form does not."""
        t0, t1 = self._alm('tlm', idx0), self._alm('tlm', idx1)
        if (not options.opts.batch2 or shts._lane_active() or not (isinstance(t0, torch.Tensor) and t0.is_cuda and isinstance(t1, torch.Tensor) and t1.is_cuda)
                or t0.numel() != t1.numel()):
            return [self.get_irestmap(idx0), self.get_irestmap(idx1)]
        out = shts.alm2map_batch2(t0, t1, self.nside, lmax=self._lmax(t0))
        return [out[0], out[1]]

    # ---- two simulations on one Legendre recursion (pl_alm2map_batch2): same maps, bit for bit, at 0.8 of the time -----------
    def get_irespmap_batch2(self, idx0, idx1):
        """get_irespmap of two simulations: ((Qb0, Ub0), (Qb1, Ub1))"""
        eb = [(self._alm('elm', i), self._alm('blm', i)) for i in (idx0, idx1)]
        lmax = self._lmax(eb[0][0])
        if any(self._lmax(x) != lmax for pair in eb for x in pair):
            return self.get_irespmap(idx0), self.get_irespmap(idx1)
        return shts.alm2map_spin_batch2(list(eb[0]), list(eb[1]), self.nside, 2, lmax, fl=0.5 * np.ones(lmax + 1))

    def get_gpmap_batch2(self, idx0, idx1, spin, k=None):
        """get_gpmap of two simulations: ((G0, C0), (G1, C1)) spin-s maps"""
        assert spin in [1, 3]
        gcs = [self._gp_alms(i, k=k) for i in (idx0, idx1)]
        if any(gc is None or gc[1] is None for gc in gcs) or self._lmax(gcs[0][0]) != self._lmax(gcs[1][0]):
            return self.get_gpmap(idx0, spin, k=k), self.get_gpmap(idx1, spin, k=k)
        lmax = self._lmax(gcs[0][0])
        return shts.alm2map_spin_batch2(list(gcs[0]), list(gcs[1]), self.nside, spin, lmax, fl=_spin_weight(spin, lmax))


class lib_filt2map_sepTP(lib_filt2map):
    """Same for separately filtered T and P: X^WF is built here from Xb with C^TE (qest.py:533-638)."""

    def __init__(self, ivfs, nside, clte):
        super(lib_filt2map_sepTP, self).__init__(ivfs, nside)
        self.clte = clte

    def hashdict(self):
        return {'ivfs': self.ivfs.hashdict(), 'nside': self.nside, 'clte': ut.clhash(self.clte)}

    def get_tmap(self, idx, joint=False):
        tlm = self._alm('tmliklm', idx)
        if joint:
            tlm = tlm + dev.almxfl(self._alm('elm', idx), self.clte)
        return shts.alm2map(tlm, self.nside)

    def get_pmap(self, idx, joint=False):
        G, C = self._alm('emliklm', idx), self._alm('bmliklm', idx)
        if joint:
            G = G + dev.almxfl(self._alm('tlm', idx), self.clte)
        return shts.alm2map_spin([G, C], self.nside, 2, self._lmax(G))

    # ---- the Wiener-filtered legs in one launch each -------------------------------------------------------------------------------------
    # For a filter library whose Wiener-filtered fields are C^XX_l Xb_lm (filt_simple.library_sepTP and its subclasses) the legs
    # T^WF (+ C^TE Eb) and (E^WF (+ C^TE Tb), B^WF) are linear combinations of the filtered alms: one pl_alm_lincomb launch instead of
    # hp.almxfl per field plus an addition (qest.py:582-589,613-618), rounded the same way.  The polarization pair serves the spin-3 and
    # the spin-1 leg: it is kept for the two most recent simulations.
    def _wf_direct(self):
        from .filt import filt_simple
        return (isinstance(self.ivfs, filt_simple.library_sepTP) and type(self.ivfs).get_sim_alm_dev is filt_simple.library_sepTP.get_sim_alm_dev
                and all(kk in self.ivfs.cl for kk in ('tt', 'ee', 'bb')))

    def _memo_leg(self, key, make):
        memo = self.__dict__.setdefault('_leg_memo', collections.OrderedDict())
        if key not in memo:
            while len(memo) >= 4:
                memo.popitem(last=False)
            memo[key] = make()
        return memo[key]

    def _gt_alm(self, idx, k=None, xfilt=None):
        """Gradient alm of the spin-1 leg of T^WF (+ C^TE Eb for k = 'p'), with optional 0/1 field selectors (qest.py:566-595)."""
        assert k in ['ptt', 'p'], k
        if xfilt is None and self._wf_direct():
            t = self._alm('tlm', idx)
            terms = [(t, self.ivfs.cl['tt'])] + ([(self._alm('elm', idx), self.clte)] if k == 'p' else [])
            if all(isinstance(a, torch.Tensor) and a.is_cuda and a.numel() == t.numel() for a, _ in terms):
                return self._memo_leg(('gt', k, idx, t.data_ptr()), lambda: dev.alm_lincomb([terms])[0])
        if xfilt is not None:
            assert isinstance(xfilt, dict) and 't' in xfilt.keys()
            if k == 'p':
                assert 'e' in xfilt.keys()
        need_t = xfilt is None or np.any(xfilt['t'])
        mlik = None
        if need_t:
            mlik = self._alm('tmliklm', idx)
            if xfilt is not None:
                mlik = dev.almxfl(mlik, xfilt['t'])
        if k == 'p' and (xfilt is None or np.any(xfilt['e'])):
            if xfilt is None:  # T^WF + C^TE Eb in one pass
                mlik = dev.almxfl_add(mlik, self._alm('elm', idx), self.clte)
            else:
                telm = dev.almxfl(dev.almxfl(self._alm('elm', idx), self.clte), xfilt['e'])
                mlik = telm if mlik is None else mlik + telm
        if mlik is None or (xfilt is not None and not bool(torch.any(mlik != 0))):
            return None
        return mlik

    def _gp_alms(self, idx, k=None, xfilt=None):
        """(G, C) of the spin-1 / spin-3 legs of (E^WF (+ C^TE Tb for k = 'p'), B^WF) (qest.py:597-638)."""
        assert k in ['p_p', 'p'], k
        if xfilt is None and self._wf_direct():
            e, b = self._alm('elm', idx), self._alm('blm', idx)
            gterms = [(e, self.ivfs.cl['ee'])] + ([(self._alm('tlm', idx), self.clte)] if k == 'p' else [])
            if all(isinstance(a, torch.Tensor) and a.is_cuda and a.numel() == e.numel() for a, _ in gterms + [(b, None)]):
                return self._memo_leg(('gp', k, idx, e.data_ptr()), lambda: tuple(dev.alm_lincomb([gterms, [(b, self.ivfs.cl['bb'])]])))
        if xfilt is not None:
            assert isinstance(xfilt, dict) and all(f in xfilt.keys() for f in 'teb')
        need_p = xfilt is None or np.any(xfilt['e']) or np.any(xfilt['b'])
        G = C = None
        if need_p:
            G, C = self._alm('emliklm', idx), self._alm('bmliklm', idx)
            if xfilt is not None:
                G, C = dev.almxfl(G, xfilt['e']), dev.almxfl(C, xfilt['b'])
        if k == 'p' and (xfilt is None or np.any(xfilt['t'])):
            if xfilt is None:  # E^WF + C^TE Tb in one pass
                G = dev.almxfl_add(G, self._alm('tlm', idx), self.clte)
            else:
                G_t = dev.almxfl(dev.almxfl(self._alm('tlm', idx), self.clte), xfilt['t'])
                G = G_t if G is None else G + G_t
        if G is None or (xfilt is not None and not (bool(torch.any(G != 0)) or (C is not None and bool(torch.any(C != 0))))):
            return None
        return G, C
