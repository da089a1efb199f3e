"""GPU parity against the round-3 fixtures made from the reference's own Python (tests/golden/make_golden.py `sims` and `cg2`):
map simulators on the reference's phases (maps.cmb_maps_nlev / cmb_maps_noisefree / cmb_maps_harmonicspace, maps.py:13-275), the
cross-filtered estimator keys of qest (_build_sim_xfiltMVgclm, qest.py:372-402), and the conjugate-gradient operators for the
noise models the first CG fixtures did not reach: (QQ, QU, UU) polarization noise, marginalised Q / U template maps
(opfilt_pp.py:272-303), the four-map joint filter (opfilt_tp.py:306-327)."""
import os

import numpy as np
import pytest

from helpers import relrms
from test_sims import _libs

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-8


@pytest.fixture(scope='module')
def gs():
    import torch
    assert torch.cuda.is_available()
    return np.load(os.path.join(HERE, 'golden', 'sims_golden.npz'))


@pytest.fixture(scope='module')
def g2():
    return np.load(os.path.join(HERE, 'golden', 'cg2_golden.npz'))


def test_map_simulators_on_the_reference_phases(gs, tmp_path):
    from plancklens_amd import dev
    from plancklens_amd.sims import cmbs, maps
    g = gs
    lp, pp = _libs(g, tmp_path)
    nside = int(g['nside'])
    cls = {k[4:]: g[k] for k in g.files if k.startswith('cls_')}
    sky = cmbs.sims_cmb_unl(cls, lp)
    nl = maps.cmb_maps_nlev(sky, g['transf'], 50., 70., nside, pix_lib_phas=pp)
    assert relrms(nl.get_sim_tmap(0), g['nlev_tmap_0']) < 1e-12
    q, u = nl.get_sim_pmap(0)
    assert relrms(q, g['nlev_qmap_0']) < 1e-12 and relrms(u, g['nlev_umap_0']) < 1e-12
    nld = maps.cmb_maps_nlev(sky, g['transf'], 50., 70., nside, pix_lib_phas=pp, device_maps=True)  # the same maps, left in HBM
    assert relrms(dev.to_host(nld.get_sim_tmap(0)), g['nlev_tmap_0']) < 1e-12
    nf = maps.cmb_maps_noisefree(sky, g['transf'], nside=nside, cl_transf_P=g['transf'] ** 2)
    assert relrms(nf.get_sim_tmap(1), g['nf_tmap_1']) < 1e-12
    q, u = nf.get_sim_pmap(1)
    assert relrms(q, g['nf_qmap_1']) < 1e-12 and relrms(u, g['nf_umap_1']) < 1e-12
    hs = maps.cmb_maps_harmonicspace(sky, {k: g['hs_transf_' + k] for k in 'teb'}, {k: g['hs_noise_' + k] for k in 'teb'}, lp, nside=nside)
    assert relrms(hs.get_sim_tmap(0), g['hs_tmap_0']) < 1e-12
    q, u = hs.get_sim_pmap(0)
    assert relrms(q, g['hs_qmap_0']) < 1e-12 and relrms(u, g['hs_umap_0']) < 1e-12


class _gold_sims(object):
    def __init__(self):
        self.g = np.load(os.path.join(HERE, 'golden', 'qe_golden.npz'))

    def hashdict(self):
        return {'gold': 1}

    def get_sim_tmap(self, idx):
        return self.g['tmap_%d' % idx]

    def get_sim_pmap(self, idx):
        return self.g['qmap_%d' % idx], self.g['umap_%d' % idx]


def test_cross_filtered_estimator_keys_vs_reference(gs, tmp_path):
    """'pte', 'peb', ...: one field kept on each leg (qest.py:372-402), same-leg library and a shuffled one (legs swapped and
    averaged), plus the derived 'p_eb' = 'peb' + 'pbe' (qest.py:170-171)."""
    from plancklens_amd import qest
    from plancklens_amd.filt import filt_simple, filt_util
    g = gs
    nside, lmax_qlm = int(g['q_nside']), int(g['q_lmax_qlm'])
    cl = {k: g['q_cl_' + k] for k in ['tt', 'ee', 'bb', 'te']}
    sims = _gold_sims()
    assert sims.g['tmap_0'].size == 12 * nside ** 2
    ivfs = filt_simple.library_fullsky_sepTP(str(tmp_path / 'ivfs'), sims, nside, g['q_transf'], cl, g['q_ftl'], g['q_fel'], g['q_fbl'], cache=False)
    ivfs_s = filt_util.library_shuffle(ivfs, {0: 1, 1: 0})
    qdd = qest.library_sepTP(str(tmp_path / 'qdd'), ivfs, ivfs, cl['te'], nside, lmax_qlm=lmax_qlm, cache=False)
    qds = qest.library_sepTP(str(tmp_path / 'qds'), ivfs, ivfs_s, cl['te'], nside, lmax_qlm=lmax_qlm, cache=False)
    for k in ['pte', 'pet', 'pee', 'peb', 'pbe', 'ptb', 'xeb', 'xte']:
        assert relrms(qdd.get_sim_qlm(k, 0), g['xf_dd_%s_0' % k]) < TOL, k
    for k in ['pte', 'peb', 'xbe']:
        assert relrms(qds.get_sim_qlm(k, 0), g['xf_ds_%s_0' % k]) < TOL, k
    assert relrms(qdd.get_sim_qlm('p_eb', 0), g['xf_dd_p_eb_0']) < TOL


def test_polarization_operators_with_qu_noise_and_templates(g2):
    import torch
    from plancklens_amd import dev
    from plancklens_amd.qcinv import opfilt_pp
    from plancklens_amd.qcinv.util_alm import eblm
    g = g2
    cl = {k: g['cl_' + k] for k in ['tt', 'ee', 'bb']}
    x = eblm([dev.to_dev(g['xe']), dev.to_dev(g['xb'])])
    x0 = (x.elm.clone(), x.blm.clone())
    f3 = opfilt_pp.alm_filter_ninv([g['nqq'], g['nqu'], g['nuu']], g['transf'])
    r = opfilt_pp.fwd_op(cl, f3)(x)
    assert relrms(dev.to_host(r.elm), g['pp3_fwd_e']) < 1e-11 and relrms(dev.to_host(r.blm), g['pp3_fwd_b']) < 1e-11
    pr = opfilt_pp.calc_prep([g['qmap'], g['umap']], cl, f3)
    assert relrms(dev.to_host(pr.elm), g['pp3_prep_e']) < 1e-11 and relrms(dev.to_host(pr.blm), g['pp3_prep_b']) < 1e-11
    # the same operator on a block of two right-hand sides (pl_cg_fwd_pp_qu_b), and step by step (map_qu_weight in apply_map)
    xb = eblm([torch.stack([x.elm, x.blm * 0.5]).contiguous(), torch.stack([x.blm, x.elm * 2.]).contiguous()])
    rb = opfilt_pp.fwd_op(cl, f3)(xb)
    assert bool((rb.elm[0] == r.elm).all()) and bool((rb.blm[0] == r.blm).all())
    r1 = opfilt_pp.fwd_op(cl, f3)(eblm([xb.elm[1].contiguous(), xb.blm[1].contiguous()]))
    assert bool((rb.elm[1] == r1.elm).all()) and bool((rb.blm[1] == r1.blm).all())
    steps = f3._apply_alm_steps(x)
    from plancklens_amd.qcinv.opfilt_pp import _apply_2x2
    steps = _apply_2x2(opfilt_pp.fwd_op(cl, f3).s_inv_filt.slinv, x, add_to=steps)
    assert relrms(dev.to_host(steps.elm), g['pp3_fwd_e']) < 1e-11 and relrms(dev.to_host(steps.blm), g['pp3_fwd_b']) < 1e-11
    fm = opfilt_pp.alm_filter_ninv([g['nqq']], g['transf'], marge_qmaps=[g['tq0'], g['tq1']], marge_umaps=[g['tu0']])
    r = opfilt_pp.fwd_op(cl, fm)(x)
    assert relrms(dev.to_host(r.elm), g['ppm_fwd_e']) < 1e-11 and relrms(dev.to_host(r.blm), g['ppm_fwd_b']) < 1e-11
    # (that was the projection as a rank-3 update in harmonic space, pl_lowrank_update_b on the stacked (E, B) vectors; the pixel-space form:)
    from plancklens_amd import options
    with options.override(tproj_harm=False):
        assert not fm.one_call_ok(x)
        r0 = opfilt_pp.fwd_op(cl, fm)(x)
    assert fm.one_call_ok(x)
    assert relrms(dev.to_host(r0.elm), g['ppm_fwd_e']) < 1e-11 and relrms(dev.to_host(r0.blm), g['ppm_fwd_b']) < 1e-11
    assert relrms(dev.to_host(r.elm), dev.to_host(r0.elm)) < 1e-12 and relrms(dev.to_host(r.blm), dev.to_host(r0.blm)) < 1e-12
    pr = opfilt_pp.calc_prep([g['qmap'], g['umap']], cl, fm)
    assert relrms(dev.to_host(pr.elm), g['ppm_prep_e']) < 1e-11 and relrms(dev.to_host(pr.blm), g['ppm_prep_b']) < 1e-11
    assert bool((x.elm == x0[0]).all()) and bool((x.blm == x0[1]).all())
    # inside an iteration nothing may come back to the host: the operator is capturable into a HIP graph
    torch.cuda.synchronize()
    for f in (f3, fm):
        op = opfilt_pp.fwd_op(cl, f)
        op(x)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            out = op(x)
        gr.replay()
        torch.cuda.synchronize()
        ref = op(x)
        assert bool((out.elm == ref.elm).all()) and bool((out.blm == ref.blm).all())


def test_joint_operator_with_four_noise_maps(g2):
    from plancklens_amd import dev
    from plancklens_amd.qcinv import opfilt_tp
    from plancklens_amd.qcinv.util_alm import teblm
    g = g2
    cl = {k: g['cl_' + k] for k in ['tt', 'ee', 'bb', 'te']}
    f4 = opfilt_tp.alm_filter_ninv([g['ntt'], g['nqq'], g['nqu'], g['nuu']], g['transf'], marge_monopole=True, marge_dipole=True)
    xt = teblm([dev.to_dev(g['tp4_xt']), dev.to_dev(g['tp4_xe']), dev.to_dev(g['tp4_xb'])])
    r = opfilt_tp.fwd_op(cl, f4)(xt)
    for a, k in ((r.tlm, 'tp4_fwd_t'), (r.elm, 'tp4_fwd_e'), (r.blm, 'tp4_fwd_b')):
        assert relrms(dev.to_host(a), g[k]) < 1e-11, k
    pr = opfilt_tp.calc_prep([g['tmap'], g['qmap'], g['umap']], cl, f4)
    for a, k in ((pr.tlm, 'tp4_prep_t'), (pr.elm, 'tp4_prep_e'), (pr.blm, 'tp4_prep_b')):
        assert relrms(dev.to_host(a), g[k]) < 1e-11, k
