"""Quadratic-estimator power spectra behind the names of plancklens/qecl.py (`library` :13-148, `average` :151-224).

C_L = 1 / ((2L + 1) fsky) sum_M (qA_LM - <qA>_LM) conj(qB_LM - <qB>_LM), cached per (keys, simulation) in the sqlite
database `cldb.db`, sky fractions in `fskies.dat` -- file names, hash dictionary and call signatures of the reference, so
that existing spectra directories stay readable.  The work itself is device work here: the mean-field subtraction and
the spectrum are one pass of pl_alm2cl over device-resident estimates (which the estimator library hands over without a
host round trip when it has just computed them), and the overlap integrals of the four analysis masks are taken on the
GPU when masks are full-resolution maps.
"""
from __future__ import print_function

import os
import pickle as pk

import numpy as np

from . import hp, utils
from .helpers import mpi, sql


def _overlap_fractions(masks):
    """{10 i + j: <m_i m_j>, 1234: <m_1 m_2 m_3 m_4>} for the masks of the four legs (labels 1..4), as written to fskies.dat."""
    assert len(set(np.shape(m) for m in masks)) == 1, 'the four leg masks must have one shape'
    if all(np.size(m) == 1 or np.all(np.asarray(m) == 1.) for m in masks):  # full-sky libraries: nothing to integrate
        out = {10 * i + j: 1.0 for i in range(1, 5) for j in range(i, 5)}
        out[1234] = 1.0
        return out
    try:  # 50-million-pixel products: on the device when there is one
        import torch
        from . import dev
        ms = [dev.to_dev(np.asarray(m, dtype=np.float64)) for m in masks]
        mean = lambda t: float(t.mean())
    except Exception:
        ms = [np.asarray(m, dtype=np.float64) for m in masks]
        mean = lambda t: float(np.mean(t))
    out = {10 * (i + 1) + (j + 1): mean(ms[i] * ms[j]) for i in range(4) for j in range(i, 4)}
    out[1234] = mean(ms[0] * ms[1] * ms[2] * ms[3])
    return out


class library(object):
    """Mean-field subtracted cross-spectra of estimator library qeA with qeB; mc_sims_mf[0::2] build the mean field of the
    first, mc_sims_mf[1::2] of the second, so that the subtraction adds no noise bias."""

    def __init__(self, lib_dir, qeA, qeB, mc_sims_mf):
        self.lib_dir, self.prefix = lib_dir, lib_dir
        self.qeA, self.qeB = qeA, qeB
        self.mc_sims_mf = mc_sims_mf
        fn_fsky, fn_hash = os.path.join(lib_dir, 'fskies.dat'), os.path.join(lib_dir, 'qcl_sim_hash.pk')
        if mpi.rank == 0:
            os.makedirs(lib_dir, exist_ok=True)
            if not os.path.exists(fn_fsky):
                fr = _overlap_fractions([qeA.get_mask(1), qeA.get_mask(2), qeB.get_mask(1), qeB.get_mask(2)])
                with open(fn_fsky, 'w') as f:
                    f.writelines('%4s %.5f \n' % (lab, fr[lab]) for lab in sorted(fr))
            if not os.path.exists(fn_hash):
                with open(fn_hash, 'wb') as f:
                    pk.dump(self.hashdict(), f, protocol=2)
        mpi.barrier()
        with open(fn_hash, 'rb') as f:
            utils.hash_check(pk.load(f), self.hashdict(), fn=fn_hash)
        self.npdb = sql.npdb(os.path.join(lib_dir, 'cldb.db'))
        with open(fn_fsky) as f:
            self.fskies = {int(k): float(v) for k, v in (line.split() for line in f if line.strip())}
        self.fsky1234, self.fsky11, self.fsky12, self.fsky22 = (self.fskies[k] for k in (1234, 11, 12, 22))

    def hashdict(self):
        return {'qeA': self.qeA.hashdict(), 'qeB': self.qeB.hashdict(), 'mc_sims_mf': self._mcmf_hash()}

    def _mcmf_hash(self):
        return utils.mchash(self.mc_sims_mf)

    def get_lmaxqcl(self, k1, k2):
        return min(self.qeA.get_lmax_qlm(k1), self.qeB.get_lmax_qlm(k2))

    def _entry(self, k1, k2, lmax_qcl, idx):
        """database key of one spectrum: the file name the reference would use"""
        assert idx >= -1, idx
        tag = '%04d' % idx if idx >= 0 else 'dat'
        return os.path.join(self.lib_dir, 'sim_qcl_k1%s_k2%s_lmax%s_%s_%s.dat' % (k1, k2, lmax_qcl, tag, self._mcmf_hash()))

    def _spectrum(self, k1, k2, idx, lmax):
        """unnormalised (no 1 / fsky) spectrum of the mean-field subtracted estimates"""
        same = k1 == k2 and self.qeA is self.qeB
        # the device route forms the spectrum with pl_alm2cl; a subclass that overrides the reference's hook _alm2clfsky1234
        # (qecl.py:147-148: e.g. a mask-deconvolved spectrum) gets host arrays through its hook, as in the reference
        dev_route = (hasattr(self.qeA, '_get_sim_qlm_dev') and hasattr(self.qeB, '_get_sim_qlm_dev')
                     and type(self)._alm2clfsky1234 is library._alm2clfsky1234)
        if dev_route:
            from . import dev
            import torch
            a = self.qeA._get_sim_qlm_dev(k1, idx, lmax)
            b = a if same else self.qeB._get_sim_qlm_dev(k2, idx, lmax)
            a = a - dev.to_dev(self.qeA.get_sim_qlm_mf(k1, self.mc_sims_mf[0::2], lmax=lmax), torch.complex128)
            b = b - dev.to_dev(self.qeB.get_sim_qlm_mf(k2, self.mc_sims_mf[1::2], lmax=lmax), torch.complex128)
            return dev.to_host(dev.alm2cl(a.contiguous(), b.contiguous()))
        a = self.qeA.get_sim_qlm(k1, idx, lmax=lmax)
        b = a if same else self.qeB.get_sim_qlm(k2, idx, lmax=lmax)
        a = a - self.qeA.get_sim_qlm_mf(k1, self.mc_sims_mf[0::2], lmax=lmax)
        b = b - self.qeB.get_sim_qlm_mf(k2, self.mc_sims_mf[1::2], lmax=lmax)
        return self._alm2clfsky1234(a, b, k1, k2)

    def get_sim_qcl(self, k1, idx, k2=None, lmax=None, recache=False, calc=True):
        """Spectrum of estimator k1 (library A) with k2 (library B; k1 if omitted) on simulation idx (-1: the data).
        calc=False only reads the cache (and fails if the entry is absent); `recache` is accepted and, as in the reference
        (qecl.py:110-111), has no effect."""
        k2 = k1 if k2 is None else k2
        assert k1 in self.qeA.keys and k2 in self.qeB.keys, (k1, k2)
        assert idx not in self.mc_sims_mf, 'simulation %s belongs to the mean-field set' % idx
        lmax_qcl = self.get_lmaxqcl(k1, k2)
        lmax_out = lmax or lmax_qcl
        assert lmax_out <= lmax_qcl
        entry = self._entry(k1, k2, lmax_qcl, idx)
        cl = self.npdb.get(entry)
        if calc and cl is None:
            cl = self._spectrum(k1, k2, idx, lmax_qcl)
            self.npdb.add(entry, cl)
            cl = self.npdb.get(entry)
        return cl[:lmax_out + 1] / self.fskies[1234]

    def load_sim_qcl(self, k1, idx, k2=None, lmax=None):
        return self.get_sim_qcl(k1, idx, k2=k2, lmax=lmax, calc=False)

    def get_dat_qcl(self, k1, k2=None, lmax=None):
        return self.get_sim_qcl(k1, -1, k2=k2, lmax=lmax)

    def get_sim_stats_qcl(self, k1, mc_sims, k2=None, recache=False):
        """utils.stats (mean and scatter, no covariance) of the spectra over mc_sims, pickled next to the database"""
        k2 = k1 if k2 is None else k2
        fn = os.path.join(self.lib_dir, 'sim_qcl_stats_%s_%s_%s.pk' % (k1, k2, utils.mchash(mc_sims)))
        if recache or not os.path.exists(fn):
            st = utils.stats(self.get_lmaxqcl(k1, k2) + 1, docov=False)
            for _, idx in utils.enumerate_progress(mc_sims, label='sim_stats qcl (k1,k2)=' + str((k1, k2))):
                st.add(self.get_sim_qcl(k1, idx, k2=k2))
            with open(fn, 'wb') as f:
                pk.dump(st, f, protocol=2)
        with open(fn, 'rb') as f:
            return pk.load(f)

    def _alm2clfsky1234(self, qlm1, qlm2, k1, k2):
        return hp.alm2cl(qlm1, alms2=qlm2)


class average(object):
    """Arithmetic mean of the spectra of several qecl libraries (e.g. data splits); same getters as `library`."""

    def __init__(self, lib_dir, qcls_lib):
        self.lib_dir, self.qclibs = lib_dir, list(qcls_lib)
        fn_hash = os.path.join(lib_dir, 'qeclav_hash.pk')
        if mpi.rank == 0:
            os.makedirs(lib_dir, exist_ok=True)
            if not os.path.exists(fn_hash):
                with open(fn_hash, 'wb') as f:
                    pk.dump(self.hashdict(), f, protocol=2)
        mpi.barrier()
        with open(fn_hash, 'rb') as f:
            utils.hash_check(pk.load(f), self.hashdict(), fn=fn_hash)
        self.mc_sims_mf = np.unique(np.concatenate([np.asarray(q.mc_sims_mf) for q in self.qclibs]))

    def hashdict(self):
        return {'qcl_lib %s' % i: q.hashdict() for i, q in enumerate(self.qclibs)}

    def get_lmaxqcl(self, k1, k2):
        return int(min(q.get_lmaxqcl(k1, k2) for q in self.qclibs))

    def get_sim_qcl(self, k1, idx, k2=None, lmax=None):
        if lmax is None:
            lmax = self.get_lmaxqcl(k1, k1 if k2 is None else k2)
        return sum(q.get_sim_qcl(k1, idx, k2=k2, lmax=lmax) for q in self.qclibs) / len(self.qclibs)

    def get_dat_qcl(self, k1, k2=None, lmax=None):
        return self.get_sim_qcl(k1, -1, k2=k2, lmax=lmax)

    def get_sim_stats_qcl(self, k1, mc_sims, k2=None, recache=False, lmax=None):
        """utils.stats of the averaged spectra over mc_sims, pickled in lib_dir (qecl.py:205-222)"""
        k2 = k1 if k2 is None else k2
        lmax = self.get_lmaxqcl(k1, k2) if lmax is None else lmax
        assert lmax <= self.get_lmaxqcl(k1, k2)
        fn = os.path.join(self.lib_dir, 'sim_qcl_stats_%s_%s_%s_%s.pk' % (k1, k2, lmax, utils.mchash(mc_sims)))
        if recache or not os.path.exists(fn):
            st = utils.stats(lmax + 1, docov=False)
            for _, idx in utils.enumerate_progress(mc_sims, label='sim_stats qcl (k1,k2)=' + str((k1, k2))):
                st.add(self.get_sim_qcl(k1, idx, k2=k2, lmax=lmax))
            with open(fn, 'wb') as f:
                pk.dump(st, f, protocol=2)
        with open(fn, 'rb') as f:
            return pk.load(f)
