"""Simulation-input classes against outputs of the reference's own classes (tests/golden/sims_golden.npz, written by
tests/golden/make_golden.py sims from /root/reference): phase libraries with the reference's generator-state database, Gaussian
skies, harmonic-space map library, m-dependent rescaling of a filtering library.  Host logic only (no GPU)."""
import os

import numpy as np
import pytest

from helpers import relrms

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'sims_golden.npz')


@pytest.fixture(scope='module')
def g():
    return np.load(GOLD)


def _plant(g, root, name, sub, nfields):
    """an existing library on disk: the reference's own rngdb.db / sim_hash.pk files, byte for byte"""
    for i in range(nfields):
        d = os.path.join(root, '%s_%04d' % (sub, i))
        os.makedirs(d)
        open(os.path.join(d, 'rngdb.db'), 'wb').write(g['file_%s_%d_rngdb' % (name, i)].tobytes())
        open(os.path.join(d, 'sim_hash.pk'), 'wb').write(g['file_%s_%d_hash' % (name, i)].tobytes())


def _libs(g, tmp_path):
    from plancklens_amd.sims import phas
    _plant(g, str(tmp_path / 'pha'), 'pha', 'pha', 4)
    _plant(g, str(tmp_path / 'pix'), 'pix', 'pix_pha', 3)
    lmax, nside = int(g['lmax']), int(g['nside'])
    return phas.lib_phas(str(tmp_path / 'pha'), 4, lmax), phas.pix_lib_phas(str(tmp_path / 'pix'), 3, (12 * nside ** 2,))


def test_existing_phase_libraries_yield_the_reference_phases(g, tmp_path):
    """plancklens/sims/phas.py:13-195: the stored numpy generator states of an existing library reproduce its simulations --
    whatever the state of the global generator now, and in any request order."""
    np.random.seed(1)
    lp, pp = _libs(g, tmp_path)
    assert lp.is_full() is False and lp[1].is_stored(0) and not lp[1].is_stored(2)
    for idx in (1, 0):
        assert np.array_equal(lp.get_sim(idx), g['pha_%d' % idx])
        assert np.array_equal(pp.get_sim(idx), g['pix_%d' % idx])
    assert np.array_equal(lp.get_sim(0, idf=3), g['pha_0'][3]) and np.array_equal(pp.get_sim(1, idf=2), g['pix_1'][2])
    assert lp.get_sim(0, idf=1, phas_only=True) is None
    assert lp.hashdict() == {'nfields': 4, 'lmax': int(g['lmax'])}


def test_fresh_phase_libraries_follow_the_global_generator_like_the_reference(g, tmp_path):
    """a new library driven by the same np.random.seed and the same request order records the same states"""
    from plancklens_amd.sims import phas
    lmax, nside = int(g['lmax']), int(g['nside'])
    np.random.seed(4242)
    lp = phas.lib_phas(str(tmp_path / 'pha'), 4, lmax)
    pp = phas.pix_lib_phas(str(tmp_path / 'pix'), 3, (12 * nside ** 2,))
    for idx in (0, 1):
        assert np.array_equal(lp.get_sim(idx), g['pha_%d' % idx])
        assert np.array_equal(pp.get_sim(idx), g['pix_%d' % idx])
    st = lp[2]._rng_db.get(1)
    assert st[0] == 'MT19937' and st[1].dtype == np.uint32 and st[1].size == 624
    lp[2]._rng_db.delete(1)
    assert not lp[2].is_stored(1)
    lim = phas.pix_lib_phas(str(tmp_path / 'lim'), 1, (4,), nsims_max=2)
    lim.get_sim(0), lim.get_sim(1)
    assert lim.is_full()
    with pytest.raises(AssertionError):
        lim.get_sim(2)


def test_gaussian_skies_and_harmonic_space_maps(g, tmp_path):
    """cmbs.sims_cmb_unl / sims_cmb_unl_fixed_phi (cmbs.py:25-101,236-261) and maps.cmb_maps_harmonicspace in alm mode
    (maps.py:177-275) on the reference's phases"""
    from plancklens_amd.sims import cmbs, maps
    lp, _ = _libs(g, tmp_path)
    cls = {k[4:]: g[k] for k in g.files if k.startswith('cls_')}
    sky = cmbs.sims_cmb_unl(cls, lp)
    assert list(sky.fields) == [str(f) for f in g['sky_fields']]
    for f in 'pteb':
        assert relrms(sky.get_sim_alm(1, f), g['sky_%slm_1' % f]) < 1e-13
    fixed = cmbs.sims_cmb_unl_fixed_phi(cls, lp)
    assert relrms(fixed.get_sim_plm(1), g['fixed_plm_1']) < 1e-13 and relrms(fixed.get_sim_tlm(1), g['fixed_tlm_1']) < 1e-13
    assert np.array_equal(fixed.get_sim_plm(1), fixed.get_sim_plm(0))
    hs = maps.cmb_maps_harmonicspace(sky, {k: g['hs_transf_' + k] for k in 'teb'}, {k: g['hs_noise_' + k] for k in 'teb'}, lp,
                                     lib_dir=str(tmp_path / 'hs'))
    assert relrms(hs.get_sim_tmap(0), g['hs_tlm_0']) < 1e-13
    e, b = hs.get_sim_pmap(0)
    assert relrms(e, g['hs_elm_0']) < 1e-13 and relrms(b, g['hs_blm_0']) < 1e-13
    assert relrms(sky.get_sim_tlm(0), sky.get_sim_tlm(0)) == 0.  # the getters do not modify the library's alms


def test_library_fml_vs_reference(g):
    """filt_util.library_fml (filt_util.py:106-182), including its use of the temperature weights for the filtered E and B alms"""
    from plancklens_amd import hp
    from plancklens_amd.filt import filt_util
    lmax_i, lmax_f = int(g['fml_lmax_in']), int(g['fml_lmax'])

    class stub_ivfs(object):
        lib_dir = None

        def hashdict(self):
            return {'stub': 1}

        def _a(self, idx, k):
            rng = np.random.default_rng(100 * idx + k)
            a = rng.standard_normal(hp.Alm.getsize(lmax_i)) + 1j * rng.standard_normal(hp.Alm.getsize(lmax_i))
            a[:lmax_i + 1] = a[:lmax_i + 1].real
            return a

        def get_fmask(self):
            return np.ones(48)

        def get_tal(self, a):
            return np.ones(lmax_i + 1)

        def get_ftl(self):
            return 1. / (1. + np.arange(lmax_i + 1.))

        def get_fel(self):
            return 2. / (2. + np.arange(lmax_i + 1.))

        def get_fbl(self):
            return 3. / (3. + np.arange(lmax_i + 1.))
    names = ['tlm', 'elm', 'blm', 'tmliklm', 'emliklm', 'bmliklm']
    for k, name in enumerate(names):
        setattr(stub_ivfs, 'get_sim_' + name, (lambda self, idx, k=k: self._a(idx, k)))
    fml = filt_util.library_fml(stub_ivfs(), lmax_f, g['fml_mt'], g['fml_me'], g['fml_mb'])
    for name in names:
        assert relrms(getattr(fml, 'get_sim_' + name)(3), g['fml_%s_3' % name]) < 1e-14, name
    for name in ['ftl', 'fel', 'fbl']:
        assert np.allclose(getattr(fml, 'get_' + name)(), g['fml_' + name], rtol=1e-14, atol=0)
    assert set(fml.hashdict().keys()) == {'ivfs', 'filt_t', 'filt_e', 'filt_b'}


class _fake_lib(object):
    def __init__(self, tag, npix=48):
        self.tag, self.npix = tag, npix

    def hashdict(self):
        return {'fake': self.tag}

    def get_sim_tmap(self, idx):
        return np.random.default_rng(100 * self.tag + idx + 7).standard_normal(self.npix)

    def get_sim_pmap(self, idx):
        r = np.random.default_rng(1000 * self.tag + idx + 7)
        return r.standard_normal(self.npix), r.standard_normal(self.npix)


def test_sim_lib_add_sim_and_add_dat():
    """sims.utils.sim_lib_add_sim / sim_lib_add_dat (sims/utils.py:20-95): weighted sums of libraries for simulation (idx >= 0) resp.
    data (idx < 0) indices only, the first library alone otherwise; hash dictionaries with the reference's keys; nested as the
    reference's smicadx12 parameter file nests them -- and, where the reference is present, equal to its classes."""
    from plancklens_amd.sims import utils as su
    libs, w = [_fake_lib(1), _fake_lib(2), _fake_lib(3)], np.array([1., -0.5, 2.])
    a, d = su.sim_lib_add_sim(libs, weights=w), su.sim_lib_add_dat(libs, weights=w)
    for idx in (-1, 0, 3):
        full_t = sum(l.get_sim_tmap(idx) * x for l, x in zip(libs, w))
        full_q = sum(l.get_sim_pmap(idx)[0] * x for l, x in zip(libs, w))
        first_t, first_u = libs[0].get_sim_tmap(idx) * w[0], libs[0].get_sim_pmap(idx)[1] * w[0]
        assert np.allclose(a.get_sim_tmap(idx), full_t if idx >= 0 else first_t, rtol=1e-15, atol=1e-15)
        assert np.allclose(d.get_sim_tmap(idx), full_t if idx < 0 else first_t, rtol=1e-15, atol=1e-15)
        assert np.allclose(a.get_sim_pmap(idx)[0], full_q if idx >= 0 else libs[0].get_sim_pmap(idx)[0] * w[0], rtol=1e-15, atol=1e-15)
        assert np.allclose(d.get_sim_pmap(idx)[1], first_u if idx >= 0 else sum(l.get_sim_pmap(idx)[1] * x for l, x in zip(libs, w)), rtol=1e-15, atol=1e-15)
    assert a.hashdict() == {'lib': 'add_sim', 'sim_lib 0': {'fake': 1}, 'w 0': 1., 'sim_lib 1': {'fake': 2}, 'w 1': -0.5, 'sim_lib 2': {'fake': 3}, 'w 2': 2.}
    assert su.sim_lib_add_dat(libs[:2]).hashdict() == {'lib': 'add_dat', 'sim_lib 0': {'fake': 1}, 'w 0': 1., 'sim_lib 1': {'fake': 2}, 'w 1': 1.}
    nested = su.sim_lib_add_dat([su.sim_lib_add_sim(libs[:2]), su.sim_lib_shuffle(libs[2], {-1: 5})])
    assert np.allclose(nested.get_sim_tmap(-1), libs[0].get_sim_tmap(-1) + libs[2].get_sim_tmap(5))
    assert np.allclose(nested.get_sim_tmap(2), libs[0].get_sim_tmap(2) + libs[1].get_sim_tmap(2))
    assert su.sim_lib_shuffle(libs[2], {-1: 5}).hashdict() == {'sim_lib': {'fake': 3}, 'shuffle': {-1: 5}}
    if os.path.exists('/root/reference/plancklens/sims/utils.py'):  # the reference's own classes on the same libraries
        import importlib.util
        spec = importlib.util.spec_from_file_location('ref_sims_utils', '/root/reference/plancklens/sims/utils.py')
        ru = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ru)
        for ours, theirs in ((a, ru.sim_lib_add_sim(libs, weights=w)), (d, ru.sim_lib_add_dat(libs, weights=w))):
            assert ours.hashdict() == theirs.hashdict()
            for idx in (-1, 0, 2):
                assert np.allclose(ours.get_sim_tmap(idx), theirs.get_sim_tmap(idx), rtol=1e-15, atol=1e-15)
                assert all(np.allclose(x, y, rtol=1e-15, atol=1e-15) for x, y in zip(ours.get_sim_pmap(idx), theirs.get_sim_pmap(idx)))


def test_philox_restatement_against_the_published_known_answers():
    """oracle/philox_oracle.py (the definition of the device-side generator pl_map_add_normal / pl_alm_unit_phases) reproduces the
    known-answer vectors of Philox4x32-10 published with the Random123 library (Salmon et al. 2011: kat_vectors), and its deviates
    have the right moments; distinct keys / tags / positions give distinct streams."""
    from oracle import philox_oracle as po
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, out in kat:
        assert tuple(int(x) for x in po.philox4x32_10(ctr, key)) == out
    n = po.normals(0x1234567890ABCDEF, 200001)
    assert n.size == 200001 and abs(n.mean()) < 0.01 and abs(n.std() - 1.) < 0.01 and abs(np.mean(n ** 4) - 3.) < 0.1
    assert abs(np.mean(n[0:200000:2] * n[1:200000:2])) < 0.01  # the two deviates of a pair are uncorrelated
    assert not np.array_equal(n[:100], po.normals(0x1234567890ABCDEE, 100))
    a = po.unit_phases(99, 40)
    assert a.size == 41 * 42 // 2 and np.all(a[:41].imag == 0) and abs(np.mean(np.abs(a[41:]) ** 2) - 1.) < 0.1
    c, s = po.normal_pairs(99, 41, 1)
    assert np.array_equal(a[:41].real, c) and not np.array_equal(po.normal_pairs(99, 41, 0)[0], c)
