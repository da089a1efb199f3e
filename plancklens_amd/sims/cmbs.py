"""Gaussian CMB sky libraries, API of plancklens/sims/cmbs.py (`sims_cmb_unl` :25-101, `sims_cmb_unl_fixed_phi` :236-261)."""
import numpy as np

from .. import hp, utils


def _get_fields(cls):
    """Field letters present in the spectra dictionary, in the reference's order p, t, e, b, o."""
    ret = [f for f in ['p', 't', 'e', 'b', 'o'] if (f + f) in cls.keys()]
    for k in cls.keys():
        for f in k:
            if f not in ret:
                ret.append(f)
    return ret


class sims_cmb_unl(object):
    """Correlated Gaussian alms with spectra cls_unl from unit phases: alm_i = sum_j (C_l^{1/2})_ij pha_j."""

    def __init__(self, cls_unl, lib_pha):
        lmax = lib_pha.lmax
        fields = _get_fields(cls_unl)
        nf = len(fields)
        cmat = np.zeros((lmax + 1, nf, nf))
        for i, f1 in enumerate(fields):
            for j, f2 in enumerate(fields):
                if j >= i and (f1 + f2) in cls_unl.keys():
                    cmat[:, i, j] = cmat[:, j, i] = cls_unl[f1 + f2][:lmax + 1]
        t, v = np.linalg.eigh(cmat)
        assert np.all(t >= -1e-12 * np.abs(t).max()), 'spectral matrix not positive semidefinite'
        self.rmat = np.einsum('lij,lj,lkj->lik', v, np.sqrt(np.maximum(t, 0.)), v)
        self._cl_hash = {k: utils.clhash(cls_unl[k]) for k in cls_unl.keys()}
        self.lmax = lmax
        self.lib_pha = lib_pha
        self.fields = fields

    def hashdict(self):
        ret = dict(self._cl_hash)
        ret['phas'] = self.lib_pha.hashdict()
        return ret

    def _get_sim_alm(self, idx, idf):
        pha = self.lib_pha.get_sim(idx, idf=0)
        if not isinstance(pha, np.ndarray):  # device phases (phas.lib_phas_dev): the alms are built and stay on the GPU
            from .. import dev
            ret = dev.almxfl(pha, self.rmat[:, idf, 0])
            for i in range(1, len(self.fields)):
                if np.any(self.rmat[:, idf, i]):
                    ret = ret + dev.almxfl(self.lib_pha.get_sim(idx, idf=i), self.rmat[:, idf, i])
            return ret
        ret = hp.almxfl(pha, self.rmat[:, idf, 0])
        for i in range(1, len(self.fields)):
            ret += hp.almxfl(self.lib_pha.get_sim(idx, idf=i), self.rmat[:, idf, i])
        return ret

    def get_sim_alm(self, idx, field):
        assert field in self.fields, self.fields
        return self._get_sim_alm(idx, self.fields.index(field))

    def get_sim_plm(self, idx):
        return self.get_sim_alm(idx, 'p')

    def get_sim_olm(self, idx):
        return self.get_sim_alm(idx, 'o')

    def get_sim_tlm(self, idx):
        return self.get_sim_alm(idx, 't')

    def get_sim_elm(self, idx):
        return self.get_sim_alm(idx, 'e')

    def get_sim_blm(self, idx):
        return self.get_sim_alm(idx, 'b')


class sims_cmb_unl_fixed_phi(sims_cmb_unl):
    """sims_cmb_unl whose lensing potential is the same for every index: `plm` if given, that of simulation 0 otherwise
    (cmbs.py:236-261)."""

    def __init__(self, cls_unl, lib_pha, plm=None):
        super(sims_cmb_unl_fixed_phi, self).__init__(cls_unl, lib_pha)
        self.fixed_plm = sims_cmb_unl._get_sim_alm(self, 0, self.fields.index('p')) if plm is None else plm

    def _get_sim_alm(self, idx, idf):
        if idf == self.fields.index('p'):
            return self.fixed_plm
        return sims_cmb_unl._get_sim_alm(self, idx, idf)
