"""GPU parity of the library classes the first fixture set did not reach, against outputs of the reference's own Python
(tests/golden/lib_golden.npz, written by tests/golden/make_golden.py `lib`): qest.library_jtTP over a jointly filtered library
(lib_filt2map, qest.py:441-530), the 'ntt' estimator, a bias-hardened key with its mean field (qest.py:155-246),
filt_simple.library_apo_sepTP (filt_simple.py:473-535) and filt_util.library_ftl (filt_util.py:39-103).
Tolerance on qlm: relative rms < 1e-8 (north_star); filtered alms 1e-11."""
import os

import numpy as np
import pytest

from helpers import relrms

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-8


class _gold_sims(object):
    """the maps of the first fixture set (same seeded recipe in both generator functions)"""

    def __init__(self):
        self.g = np.load(os.path.join(HERE, 'golden', 'qe_golden.npz'))

    def hashdict(self):
        return {'gold': 1}

    def get_sim_tmap(self, idx):
        return self.g['tmap_%d' % idx]

    def get_sim_pmap(self, idx):
        return self.g['qmap_%d' % idx], self.g['umap_%d' % idx]


@pytest.fixture(scope='module')
def gold():
    import torch
    assert torch.cuda.is_available()
    g = np.load(os.path.join(HERE, 'golden', 'lib_golden.npz'))
    cl = {k: g['cl_' + k] for k in ['tt', 'ee', 'bb', 'te']}
    return g, cl, _gold_sims(), int(g['nside']), int(g['lmax_ivf']), int(g['lmax_qlm'])


def test_joint_filter_library_and_jtTP_estimators(gold, tmp_path):
    import torch
    from plancklens_amd import dev, qest, shts, utils
    from plancklens_amd.filt import filt_simple
    g, cl, sims, nside, lmax_ivf, lmax_qlm = gold
    fal = {k: g['jt_fal_' + k] for k in ['tt', 'ee', 'bb', 'te']}
    transf = g['transf']

    class iso_jTP(filt_simple.library_jTP):
        """isotropic 3 x 3 filter with a TE block: the subclass the fixture generator builds on the reference's template"""

        def hashdict(self):
            return {'sims': self.sim_lib.hashdict(), 'fal': {k: utils.clhash(v) for k, v in fal.items()}}

        def get_fmask(self):
            return np.ones(12 * nside ** 2)

        def get_fal(self):
            return {k: v.copy() for k, v in fal.items()}

        def _apply_ivf(self, tqumap, soltn=None):
            bi = utils.cli(transf)
            t = shts.map2alm(tqumap[0], lmax=lmax_ivf, iter=0, fl=bi)
            e, b = shts.map2alm_spin([tqumap[1], tqumap[2]], 2, lmax=lmax_ivf, fl=bi)
            return (dev.almxfl(t, fal['tt']) + dev.almxfl(e, fal['te']), dev.almxfl(t, fal['te']) + dev.almxfl(e, fal['ee']),
                    dev.almxfl(b, fal['bb']))

    ivfs = iso_jTP(str(tmp_path / 'ivfs_j'), sims, cl, cache=True)
    for a in 'teb':
        assert relrms(getattr(ivfs, 'get_sim_%slm' % a)(0), g['jt_%slm_0' % a]) < 1e-11, a
    assert relrms(ivfs.get_sim_tmliklm(0), g['jt_tmliklm_0']) < 1e-11 and relrms(ivfs.get_sim_emliklm(0), g['jt_emliklm_0']) < 1e-11
    qlms = qest.library_jtTP(str(tmp_path / 'qlms_j'), ivfs, ivfs, nside, lmax_qlm=lmax_qlm)
    for k in ['p', 'x', 'ptt', 'p_p', 'stt']:
        assert relrms(qlms.get_sim_qlm(k, 0), g['jt_%s_0' % k]) < TOL, k


def test_ntt_and_bias_hardened_keys(gold, tmp_path):
    from plancklens_amd import qest, qresp
    from plancklens_amd.filt import filt_simple
    g, cl, sims, nside, lmax_ivf, lmax_qlm = gold
    ivfs = filt_simple.library_fullsky_sepTP(str(tmp_path / 'ivfs'), sims, nside, g['transf'], cl, g['ftl'], g['fel'], g['fbl'], cache=False)
    resp = qresp.resp_lib_simple(str(tmp_path / 'resp'), lmax_ivf, cl, cl, {'t': g['ftl'], 'e': g['fel'], 'b': g['fbl']}, lmax_qlm)
    qlms = qest.library_sepTP(str(tmp_path / 'qlms'), ivfs, ivfs, cl['te'], nside, lmax_qlm=lmax_qlm, resplib=resp)
    assert relrms(qlms.get_sim_qlm('ntt', 0), g['dd_ntt_0']) < TOL
    assert relrms(qlms.get_sim_qlm('ptt_bh_s', 0), g['dd_ptt_bh_s_0']) < TOL
    assert relrms(qlms.get_sim_qlm_mf('ptt_bh_s', np.array([0, 1])), g['dd_mf_ptt_bh_s']) < TOL


def test_apodised_mask_filter_library(gold, tmp_path):
    from plancklens_amd import hp
    from plancklens_amd.filt import filt_simple
    g, cl, sims, nside, lmax_ivf, lmax_qlm = gold
    apo_path = str(tmp_path / 'apomask.fits')
    hp.write_map(apo_path, g['apomask'])
    ivfs = filt_simple.library_apo_sepTP(str(tmp_path / 'ivfs_apo'), sims, apo_path, cl, g['transf'], g['ftl'], g['fel'], g['fbl'], cache=False)
    assert relrms(ivfs.get_sim_tlm(1), g['apo_tlm_1']) < 1e-11
    assert relrms(ivfs.get_sim_elm(1), g['apo_elm_1']) < 1e-11 and relrms(ivfs.get_sim_blm(1), g['apo_blm_1']) < 1e-11
    assert relrms(ivfs.get_sim_tmliklm(1), g['apo_tmliklm_1']) < 1e-11
    assert np.all(ivfs.get_fmask() == g['apomask'])


def test_rescaled_filter_library(gold, tmp_path):
    from plancklens_amd import qest
    from plancklens_amd.filt import filt_simple, filt_util
    g, cl, sims, nside, lmax_ivf, lmax_qlm = gold
    ivfs = filt_simple.library_fullsky_sepTP(str(tmp_path / 'ivfs'), sims, nside, g['transf'], cl, g['ftl'], g['fel'], g['fbl'], cache=False)
    ivfs_f = filt_util.library_ftl(ivfs, int(g['ftl_lmax']), g['ftl_lt'], g['ftl_le'], g['ftl_lb'])
    for a in 'teb':
        assert relrms(getattr(ivfs_f, 'get_sim_%slm' % a)(0), g['ftl_%slm_0' % a]) < 1e-11, a
    assert relrms(ivfs_f.get_sim_emliklm(0), g['ftl_emliklm_0']) < 1e-11
    assert np.allclose(ivfs_f.get_fel(), g['ftl_get_fel'], rtol=1e-15, atol=0)
    qlms = qest.library_sepTP(str(tmp_path / 'qlms_f'), ivfs_f, ivfs_f, cl['te'], nside, lmax_qlm=lmax_qlm)
    assert relrms(qlms.get_sim_qlm('p', 0), g['ftl_p_0']) < TOL
