"""GPU parity of the filter -> quadratic-estimator chain (plancklens_amd.filt.filt_simple + plancklens_amd.qest)
against (a) the outputs of the reference's own Python stored in tests/golden/qe_golden.npz and (b) the oracle chain
on seeded inputs.  Tolerance on qlm: relative rms < 1e-8 (north_star); observed ~1e-13."""
import os

import numpy as np
import pytest

from helpers import relrms

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'qe_golden.npz')
TOL = 1e-8


class _gold_sims(object):
    def __init__(self, g):
        self.g = g

    def hashdict(self):
        return {'gold': 1}

    def get_sim_tmap(self, idx):
        return self.g['tmap_%d' % idx]

    def get_sim_pmap(self, idx):
        return self.g['qmap_%d' % idx], self.g['umap_%d' % idx]


@pytest.fixture(scope='module')
def setup(tmp_path_factory):
    import torch
    assert torch.cuda.is_available()
    from plancklens_amd import qest
    from plancklens_amd.filt import filt_simple, filt_util
    g = np.load(GOLD)
    tmp = str(tmp_path_factory.mktemp('qe'))
    cl = {k: g['cl_' + k] for k in ['tt', 'ee', 'bb', 'te']}
    nside, lmax_qlm = int(g['nside']), int(g['lmax_qlm'])
    ivfs = filt_simple.library_fullsky_sepTP(os.path.join(tmp, 'ivfs'), _gold_sims(g), nside, g['transf'], cl, g['ftl'], g['fel'],
                                             g['fbl'], cache=True)
    qdd = qest.library_sepTP(os.path.join(tmp, 'qdd'), ivfs, ivfs, cl['te'], nside, lmax_qlm=lmax_qlm)
    ivfs_s = filt_util.library_shuffle(ivfs, {0: 1, 1: 0})
    qds = qest.library_sepTP(os.path.join(tmp, 'qds'), ivfs, ivfs_s, cl['te'], nside, lmax_qlm=lmax_qlm)
    return g, ivfs, qdd, qds, cl, tmp


def test_filter_vs_reference(setup):
    g, ivfs = setup[0], setup[1]
    for idx in (0, 1):
        assert relrms(ivfs.get_sim_tlm(idx), g['tlm_%d' % idx]) < 1e-11
        assert relrms(ivfs.get_sim_elm(idx), g['elm_%d' % idx]) < 1e-11
        assert relrms(ivfs.get_sim_blm(idx), g['blm_%d' % idx]) < 1e-11
    from plancklens_amd import hp
    assert relrms(ivfs.get_sim_tmliklm(0), hp.almxfl(g['tlm_0'], g['cl_tt'])) < 1e-11


@pytest.mark.parametrize('key', ['ptt', 'xtt', 'p_p', 'x_p', 'p', 'x', 'stt', 'ftt', 'f_p', 'a_p', 'pte', 'peb', 'p_tp', 'p_eb'])
def test_qlm_vs_reference(setup, key):
    g, qdd = setup[0], setup[2]
    assert relrms(qdd.get_sim_qlm(key, 0), g['dd_%s_0' % key]) < TOL


def test_symmetrised_and_meanfield_vs_reference(setup):
    g, qdd, qds = setup[0], setup[2], setup[3]
    for key in ['ptt', 'p_p', 'p', 'x', 'ftt', 'f_p']:
        assert relrms(qds.get_sim_qlm(key, 0), g['ds_%s_0' % key]) < TOL
    assert relrms(qdd.get_sim_qlm_mf('p', np.array([0, 1])), g['dd_mf_p']) < TOL
    assert qdd.get_sim_qlm('p', 0, lmax=20).size == 21 * 22 // 2


def test_cache_files_like_reference(setup):
    """sim_{key}_%04d.fits / sim_%04d_{t,e,b}lm.fits, read back with the FITS reader (qest.py:184,201)."""
    g, ivfs, qdd, tmp = setup[0], setup[1], setup[2], setup[5]
    qdd.get_sim_qlm('p', 0)
    from plancklens_amd import hp
    assert os.path.exists(os.path.join(tmp, 'qdd', 'sim_p_0000.fits')) and os.path.exists(os.path.join(tmp, 'qdd', 'sim_x_0000.fits'))
    assert os.path.exists(os.path.join(tmp, 'ivfs', 'sim_0000_tlm.fits')) and os.path.exists(os.path.join(tmp, 'qdd', 'qe_sim_hash.pk'))
    assert relrms(hp.read_alm(os.path.join(tmp, 'qdd', 'sim_p_0000.fits')), g['dd_p_0']) < TOL
    assert qdd.fsky11 == 1.0


def test_mv_vs_oracle_chain_medium_size(oracle):
    """nside 256, lmax 384: the whole device chain from maps against the oracle chain (qe_oracle)."""
    import tempfile
    from oracle import qe_oracle as qo
    from plancklens_amd import qest
    from plancklens_amd.filt import filt_simple
    rng = np.random.default_rng(11)
    nside, lmax = 256, 384
    ell = np.arange(lmax + 1.)
    cl = {k: 1e3 / (1. + ell) ** 2.5 for k in ['tt', 'ee', 'bb']}
    cl['ee'] = 0.05 * cl['tt']; cl['bb'] = 0.002 * cl['tt']; cl['te'] = 0.1 * cl['tt']
    maps = rng.standard_normal((3, 12 * nside ** 2))
    fl = 1. / (cl['tt'] + 0.01); fl[:10] = 0
    transf = np.exp(-ell * (ell + 1) * 1e-6)

    class sims(object):
        def hashdict(self): return {'s': 0}
        def get_sim_tmap(self, idx): return maps[0]
        def get_sim_pmap(self, idx): return maps[1], maps[2]
    t, e, b = qo.filter_maps(maps[0], maps[1], maps[2], lmax, fl, fl, fl, transf)
    Go, Co = qo.qe_sepTP('p', (t, e, b), (t, e, b), cl, nside, lmax)
    with tempfile.TemporaryDirectory() as tmp:
        ivfs = filt_simple.library_fullsky_sepTP(os.path.join(tmp, 'i'), sims(), nside, transf, cl, fl, fl, fl, cache=False)
        ql = qest.library_sepTP(os.path.join(tmp, 'q'), ivfs, ivfs, cl['te'], nside, lmax_qlm=lmax, cache=False)
        assert relrms(ql.get_sim_qlm('p', 0), Go) < TOL and relrms(ql.get_sim_qlm('x', 0), Co) < TOL


@pytest.mark.parametrize('key', ['ptt', 'p_p', 'p'])
def test_generic_route_eval_qe_vs_reference(setup, key):
    """qest.eval_qe (qresp.get_qes + utils_qe.qe_eval on the GPU) against the reference's eval_qe outputs."""
    from plancklens_amd import qest
    g, ivfs, cl = setup[0], setup[1], setup[4]
    get_alm = lambda a: {'t': ivfs.get_sim_tlm, 'e': ivfs.get_sim_elm, 'b': ivfs.get_sim_blm}[a](0)
    G, C = qest.eval_qe(key, int(g['lmax_ivf']), cl, get_alm, int(g['nside']), int(g['lmax_qlm']), verbose=False)
    assert relrms(G, g['gen_%s_G' % key]) < TOL and relrms(C, g['gen_%s_C' % key]) < TOL


def test_qecl_vs_reference(setup, tmp_path):
    from plancklens_amd import qecl
    g, qdd = setup[0], setup[2]
    qcls = qecl.library(str(tmp_path / 'qcls'), qdd, qdd, np.array([]))
    assert relrms(qcls.get_sim_qcl('p', 0), g['qcl_p_0']) < 1e-7
    assert relrms(qcls.get_sim_qcl('ptt', 0, k2='p_p'), g['qcl_ptt_p_p_0']) < 1e-7
    assert os.path.exists(str(tmp_path / 'qcls' / 'cldb.db'))
    assert relrms(qcls.get_sim_qcl('p', 0, lmax=10), g['qcl_p_0'][:11]) < 1e-7  # served from the sqlite cache


class _gauss_sims(object):
    """Seeded Gaussian T, Q, U skies (TE-correlated) x beam + white noise, synthesised on the GPU."""

    def __init__(self, nside, lmax, cls, transf, nlev_t, nlev_p):
        self.nside, self.lmax, self.cls, self.transf, self.nlev_t, self.nlev_p = nside, lmax, cls, transf, nlev_t, nlev_p

    def hashdict(self):
        return {'gauss': self.nside, 'lmax': self.lmax}

    def _alms(self, idx):
        from plancklens_amd import hp
        rng = np.random.default_rng(4000 + idx)
        one = np.ones(self.lmax + 1)
        u1, u2, u3 = (hp.synalm(one, self.lmax, rng) for _ in range(3))
        tt, ee, bb, te = (self.cls[k][:self.lmax + 1] for k in ['tt', 'ee', 'bb', 'te'])
        r = te * np.where(tt > 0, 1. / np.sqrt(np.where(tt > 0, tt, 1.)), 0.)
        return (hp.almxfl(u1, np.sqrt(tt)), hp.almxfl(u1, r) + hp.almxfl(u2, np.sqrt(np.maximum(ee - r ** 2, 0.))),
                hp.almxfl(u3, np.sqrt(bb)))

    def _noise(self, idx, f):
        from plancklens_amd import hp
        rng = np.random.default_rng(9000 + 3 * idx + f)
        vamin = np.sqrt(hp.nside2pixarea(self.nside, degrees=True)) * 60
        return (self.nlev_t if f == 0 else self.nlev_p) / vamin * rng.standard_normal(12 * self.nside ** 2)

    def get_sim_tmap(self, idx):
        from plancklens_amd import hp, shts
        return shts.alm2map(hp.almxfl(self._alms(idx)[0], self.transf), self.nside, lmax=self.lmax) + self._noise(idx, 0)

    def get_sim_pmap(self, idx):
        from plancklens_amd import hp, shts
        _, e, b = self._alms(idx)
        q, u = shts.alm2map_spin([hp.almxfl(e, self.transf), hp.almxfl(b, self.transf)], self.nside, 2, self.lmax)
        return q + self._noise(idx, 1), u + self._noise(idx, 2)


def test_qe_power_of_gaussian_skies_matches_analytic_n0(tmp_path):
    """for Gaussian skies whose spectra are the filter's, the power of the GPU-made (unnormalised)
    estimator averaged over simulations is the semi-analytical N0 of nhl.get_nhl (which tests/test_resp.py pins to the
    response and to the reference).  Catches any sign / normalisation slip between the QE kernels and the analytic
    side (factors of 2, signs, missing weights): 10 simulations, bands of 20 multipoles, 12 % tolerance on the band ratios
    (the estimator power is a 4-point function, its scatter is larger than the Gaussian mode-counting estimate)."""
    from plancklens_amd import hp, nhl, qest, utils
    from plancklens_amd.filt import filt_simple
    nside, lmax, nsims = 64, 128, 10
    cls_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'plancklens_amd', 'data', 'cls',
                            'FFP10_wdipole_lensedCls.dat')
    cl = utils.camb_clfile(cls_path, lmax=lmax)
    transf = hp.gauss_beam(60. / 60. / 180. * np.pi, lmax=lmax)
    nlev_t, nlev_p = 300., 40.
    arcmin = np.pi / 180. / 60.
    ftl = utils.cli(cl['tt'][:lmax + 1] + (nlev_t * arcmin) ** 2 * utils.cli(transf ** 2))
    fel = utils.cli(cl['ee'][:lmax + 1] + (nlev_p * arcmin) ** 2 * utils.cli(transf ** 2))
    fbl = utils.cli(cl['bb'][:lmax + 1] + (nlev_p * arcmin) ** 2 * utils.cli(transf ** 2))
    for f in (ftl, fel, fbl):
        f[:2] = 0
    sims = _gauss_sims(nside, lmax, cl, transf, nlev_t, nlev_p)
    ivfs = filt_simple.library_fullsky_sepTP(str(tmp_path / 'ivfs'), sims, nside, transf, cl, ftl, fel, fbl, cache=False)
    qlms = qest.library_sepTP(str(tmp_path / 'qlms'), ivfs, ivfs, cl['te'], nside, lmax_qlm=lmax)
    cls_ivfs = {'tt': ftl, 'ee': fel, 'bb': fbl, 'te': cl['te'][:lmax + 1] * ftl * fel}
    for key in ('ptt', 'p_p', 'p'):
        pg = np.mean([hp.alm2cl(qlms.get_sim_qlm(key, i)) for i in range(nsims)], axis=0)
        pc = np.mean([hp.alm2cl(qlms.get_sim_qlm('x' + key[1:], i)) for i in range(nsims)], axis=0)
        GG, CC, _, _ = nhl.get_nhl(key, key, cl, cls_ivfs, lmax, lmax, lmax_out=lmax)
        for lo in range(10, 110, 20):
            sl = slice(lo, lo + 20)
            assert abs(np.sum(pg[sl]) / np.sum(GG[sl]) - 1.) < 0.12, (key, 'G', lo, np.sum(pg[sl]) / np.sum(GG[sl]))
            assert abs(np.sum(pc[sl]) / np.sum(CC[sl]) - 1.) < 0.12, (key, 'C', lo, np.sum(pc[sl]) / np.sum(CC[sl]))


def test_mv_estimator_with_pipelined_lanes(setup):
    """library.pipeline_lanes: the FFT stage of each leg synthesis runs on a side stream with a fork of the plan (pl_plan_fork, own
    phase buffer) while the current stream continues with the next Legendre stage -- same numbers as the plain path."""
    from plancklens_amd import qest
    g, ivfs, cl, tmp = setup[0], setup[1], setup[4], setup[5]
    q = qest.library_sepTP(os.path.join(tmp, 'qdd_lanes'), ivfs, ivfs, cl['te'], int(g['nside']), lmax_qlm=int(g['lmax_qlm']), cache=False)
    ref = q.get_sim_qlm('p', 0)
    q._mem.clear()
    q.pipeline_lanes = True
    out = q.get_sim_qlm('p', 0)
    assert out is not ref and relrms(out, ref) < 1e-14 and relrms(out, g['dd_p_0']) < TOL


def test_paired_simulations_equal_single_evaluations(setup, tmp_path):
    """The mean-field loop serves a same-legs library two simulations at a time (their spin-2 and spin-3 leg syntheses share
    Legendre recursions, pl_alm2map_batch2): estimates and mean field must equal the one-at-a-time evaluation bit for bit."""
    from plancklens_amd import options, qest
    from plancklens_amd.filt import filt_simple
    g, cl = setup[0], setup[4]
    nside, lmax_qlm = int(g['nside']), int(g['lmax_qlm'])
    res = {}
    for tag, pairing in (('pair', True), ('single', False)):
        with options.override(batch2=pairing):
            ivfs = filt_simple.library_fullsky_sepTP(str(tmp_path / ('ivfs_' + tag)), _gold_sims(g), nside, g['transf'], cl, g['ftl'], g['fel'],
                                                     g['fbl'], cache=False)
            q = qest.library_sepTP(str(tmp_path / ('q_' + tag)), ivfs, ivfs, cl['te'], nside, lmax_qlm=lmax_qlm, cache=False)
            mf = q.get_sim_qlm_mf('p', np.array([0, 1]))
            res[tag] = (mf, q.get_sim_qlm('p', 0), q.get_sim_qlm('p', 1), q.get_sim_qlm('x', 1))
            # round 4: the polarization-only and temperature-only keys pair as well ('p_p': all three leg syntheses on shared recursions)
            for k, kx in (('p_p', 'x_p'), ('ptt', 'xtt')):
                res[tag] += (q.get_sim_qlm_mf(k, np.array([0, 1])), q.get_sim_qlm(k, 0), q.get_sim_qlm(k, 1), q.get_sim_qlm(kx, 0))
    for a, b in zip(res['pair'], res['single']):
        assert np.array_equal(a, b)
    assert relrms(res['pair'][0], g['dd_mf_p']) < TOL and relrms(res['pair'][2], g['dd_p_1']) < TOL
    assert relrms(res['pair'][5], g['dd_p_p_0']) < TOL and relrms(res['pair'][9], g['dd_ptt_0']) < TOL and relrms(res['pair'][11], g['dd_xtt_0']) < TOL


@pytest.mark.parametrize('route', ['indirect_host', 'indirect_device', 'slots_device'])
def test_pair_graph_replay_equals_eager(setup, tmp_path, route, monkeypatch):
    """qest.library._pair_graph: a pair of reconstructions (filter -> legs -> product -> analysis) captured into one HIP graph after
    `graph_after` eager evaluations and replayed from then on.  Eager evaluations, the capturing call and pure replays must give the
    same gradient / curl / mean field bit for bit, for all three paired families, with inputs that change between replays; the
    filtered alms of a replayed pair stay available to further keys (filter-library device cache) until the next replay.
    Routes of the inputs into the captured launches (round 6): through a table of device addresses (pl_map2alm_ind, the default) with host
    arrays (uploaded into slots of the graph's own) or with device tensors that are new objects at new addresses on every call (the table is
    rewritten, nothing is copied); and the static-slot route (options.opts.qe_indirect off: device tensors copied into fixed slots)."""
    import torch
    from plancklens_amd import dev, options, qest
    from plancklens_amd.filt import filt_simple
    g, cl = setup[0], setup[4]
    nside, lmax_qlm = int(g['nside']), int(g['lmax_qlm'])
    monkeypatch.setattr(options.opts, 'qe_indirect', route != 'slots_device')
    on_dev = (lambda a: dev.to_dev(np.ascontiguousarray(a))) if route != 'indirect_host' else (lambda a: a)

    class sims(_gold_sims):  # more simulations out of the two golden ones (2, 3, ...: fields of either, rescaled)
        def get_sim_tmap(self, idx):
            return on_dev(self.g['tmap_%d' % (idx % 2)] * (1. + 0.25 * (idx // 2)))

        def get_sim_pmap(self, idx):
            return on_dev(self.g['qmap_%d' % ((idx + idx // 2) % 2)]), on_dev(self.g['umap_%d' % (idx % 2)] * (1. - 0.125 * (idx // 2)))

    def make(tag, use_graph):
        ivfs = filt_simple.library_fullsky_sepTP(str(tmp_path / ('ivfs_' + tag)), sims(g), nside, g['transf'], cl, g['ftl'], g['fel'], g['fbl'], cache=False)
        q = qest.library_sepTP(str(tmp_path / ('q_' + tag)), ivfs, ivfs, cl['te'], nside, lmax_qlm=lmax_qlm, cache=False)
        q.use_graph, q.graph_after, q.graph_min_nside = use_graph, 1, 0  # (the graph route also on this small grid)
        return ivfs, q
    ivfs_e, qe = make('eager', False)
    ivfs_g, qg = make('graph', True)
    for fam, kx in (('p', 'x'), ('p_p', 'x_p'), ('ptt', 'xtt')):
        ref = {}
        pairs = ((0, 1), (2, 3), (4, 5), (6, 7), (6, 7))
        for pair in pairs[:-1]:
            qe._mem.clear()
            mf = qe.get_sim_qlm_mf(fam, np.array(pair))
            ref[pair] = [mf] + [qe.get_sim_qlm(k_, i) for i in pair for k_ in (fam, kx)]
        # 1 eager evaluation, the capture (+ first replay), then replays with new inputs and once more with the same
        for rep, pair in enumerate(pairs):
            qg._mem.clear()
            mf = qg.get_sim_qlm_mf(fam, np.array(pair))
            out = [mf] + [qg.get_sim_qlm(k_, i) for i in pair for k_ in (fam, kx)]
            for a, b in zip(out, ref[pair]):
                assert np.array_equal(a, b), (fam, rep, pair)
        st = [v for (f_, _, _), v in qg._pair_graphs.items() if f_ == fam]
        assert len(st) == 1 and isinstance(st[0]['graph'], torch.cuda.CUDAGraph) and st[0]['calls'] == 2, (fam, st)
        assert st[0]['indirect'] == (route != 'slots_device')
        if route == 'indirect_device':
            assert not st[0].get('own'), 'device-resident inputs are read in place: no slot of the graph\'s own'
    # after the last replay of ('ptt', (6, 7)) the filtered T alms of 6 and 7 are cache entries of the filter library (static buffers of the
    # graph): a further key for those simulations reuses them, and its result is the eager library's
    assert ivfs_g._dev_cache[7].get('_graph_static') and 't' in ivfs_g._dev_cache[7]
    assert np.array_equal(qg.get_sim_qlm('stt', 7), qe.get_sim_qlm('stt', 7))
    assert relrms(qg.get_sim_qlm('p', 0), g['dd_p_0']) < TOL
