"""Pins the oracle's numpy restatement of the filter -> QE chain (oracle/qe_oracle.py) against outputs of the
reference's own Python (tests/golden/qe_golden.npz, made by tests/golden/make_golden.py).  No GPU."""
import os

import numpy as np
import pytest

from helpers import relrms

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'qe_golden.npz')


@pytest.fixture(scope='module')
def gold():
    return np.load(GOLD)


def test_filter_matches_reference(oracle, gold):
    from oracle import qe_oracle as qo
    t, e, b = qo.filter_maps(gold['tmap_0'], gold['qmap_0'], gold['umap_0'], int(gold['lmax_ivf']), gold['ftl'], gold['fel'],
                             gold['fbl'], gold['transf'])
    assert relrms(t, gold['tlm_0']) < 1e-13 and relrms(e, gold['elm_0']) < 1e-13 and relrms(b, gold['blm_0']) < 1e-13


@pytest.mark.parametrize('key', ['ptt', 'p_p', 'p'])
def test_qe_matches_reference_both_routes(oracle, gold, key):
    from oracle import qe_oracle as qo
    alms = (gold['tlm_0'], gold['elm_0'], gold['blm_0'])
    cls = {k: gold['cl_' + k] for k in ['tt', 'ee', 'bb', 'te']}
    G, C = qo.qe_sepTP(key, alms, alms, cls, int(gold['nside']), int(gold['lmax_qlm']))
    assert relrms(G, gold['dd_%s_0' % key]) < 1e-12
    assert relrms(C, gold['dd_x%s_0' % key[1:]]) < 1e-12
    # the reference's generic route (qest.eval_qe) gives the same estimator
    assert relrms(G, gold['gen_%s_G' % key]) < 1e-12 and relrms(C, gold['gen_%s_C' % key]) < 1e-12


@pytest.mark.parametrize('key', ['stt', 'ftt', 'f_p', 'a_p'])
def test_scalar_qe_matches_reference(oracle, gold, key):
    from oracle import qe_oracle as qo
    alms = (gold['tlm_0'], gold['elm_0'], gold['blm_0'])
    cls = {k: gold['cl_' + k] for k in ['tt', 'ee', 'bb', 'te']}
    q = qo.qe_scalar(key, alms, alms, cls, int(gold['nside']), int(gold['lmax_qlm']))
    assert relrms(q, gold['dd_%s_0' % key]) < 1e-12


def test_symmetrised_estimator_matches_reference(oracle, gold):
    """legs from different simulations: average with the legs swapped (qest.py:327-332)."""
    from oracle import qe_oracle as qo
    a0 = (gold['tlm_0'], gold['elm_0'], gold['blm_0'])
    a1 = (gold['tlm_1'], gold['elm_1'], gold['blm_1'])
    cls = {k: gold['cl_' + k] for k in ['tt', 'ee', 'bb', 'te']}
    G1, C1 = qo.qe_sepTP('p', a0, a1, cls, int(gold['nside']), int(gold['lmax_qlm']))
    G2, C2 = qo.qe_sepTP('p', a1, a0, cls, int(gold['nside']), int(gold['lmax_qlm']))
    assert relrms(0.5 * (G1 + G2), gold['ds_p_0']) < 1e-12 and relrms(0.5 * (C1 + C2), gold['ds_x_0']) < 1e-12
