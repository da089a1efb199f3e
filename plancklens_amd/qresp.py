"""Quadratic-estimator weights, hot-path part of plancklens/qresp.py (`get_qes` :50-101, `get_resp_legs` :104-133,
`get_covresp` :135-163, `qe_spin_data` :165-181).  The analytic responses (`get_response`, `resp_lib_simple`) need the
Wigner-series module and are out of scope."""
import numpy as np

from . import utils_qe as uqe
from . import utils_spin as uspin


def _clinv(cl):
    ret = np.zeros_like(cl)
    ii = np.where(cl != 0)
    ret[ii] = 1. / cl[ii]
    return ret


def get_qes(qe_key, lmax, cls_weight, lmax2=None, transf=None):
    """List of utils_qe.qe terms defining the estimator `qe_key` by its action on the filtered spin-weight alms."""
    if lmax2 is None:
        lmax2 = lmax
    if qe_key[0] in ['p', 'x', 'a', 'f', 's']:
        if qe_key in ['ptt', 'xtt', 'att', 'ftt', 'stt']:
            s_lefts = [0]
        elif qe_key in ['p_p', 'x_p', 'a_p', 'f_p']:
            s_lefts = [-2, 2]
        else:
            s_lefts = [0, -2, 2]
        qes = []
        for s_left in s_lefts:
            for sin in s_lefts:
                sout = -s_left
                s_qe, irr1, cl_sosi, cL_out = get_covresp(qe_key[0], sout, sin, cls_weight, lmax2, transf=transf)
                if np.any(cl_sosi):
                    lega = uqe.qeleg(s_left, s_left, 0.5 * (1. + (s_left == 0)) * np.ones(lmax + 1, dtype=float))
                    legb = uqe.qeleg(sin, sout + s_qe, 0.5 * (1. + (sin == 0)) * 2 * cl_sosi)
                    qes.append(uqe.qe(lega, legb, cL_out))
        if len(qe_key) == 1 or qe_key[1:] in ['tt', '_p']:
            return uqe.qe_simplify(qes)
        if qe_key[1:] in ['te', 'et', 'tb', 'bt', 'ee', 'eb', 'be', 'bb']:
            return uqe.qe_simplify(uqe.qe_proj(qes, qe_key[1], qe_key[2]))
        if qe_key[1:] in ['_te', '_tb', '_eb']:
            return uqe.qe_simplify(uqe.qe_proj(qes, qe_key[2], qe_key[3]) + uqe.qe_proj(qes, qe_key[3], qe_key[2]))
        assert 0, 'qe key %s  not recognized' % qe_key
    if qe_key in ['ntt']:
        lega = uqe.qeleg(0, 0, 1 * _clinv(transf[:lmax + 1]))
        legb = uqe.qeleg(0, 0, 0.5 * _clinv(transf[:lmax + 1]))
        return uqe.qe_simplify([uqe.qe(lega, legb, lambda L: np.ones(len(L), dtype=float))])
    if qe_key in ['ktt']:
        ls = np.arange(1, lmax + 3)
        dlnDldlnl = ls[:-1] * np.diff(np.log(cls_weight['tt'][ls] * ls * (ls + 1)))
        lega = uqe.qeleg(0, 0, np.ones(lmax + 1, dtype=float))
        legb = uqe.qeleg(0, 0, 0.5 * cls_weight['tt'][:lmax + 1] * dlnDldlnl)
        return uqe.qe_simplify([uqe.qe(lega, legb, lambda L: -L * (L + 1.))])
    assert 0, qe_key + ' not implemented'


def get_resp_legs(source, lmax):
    """{s: (source spin r, response to +r, response to -r, G/C -> potential scaling)} for s in 0, -2, 2."""
    if source in ['p', 'x']:
        return {s: (1, -0.5 * uspin.get_spin_lower(s, lmax), -0.5 * uspin.get_spin_raise(s, lmax),
                    lambda ell: uspin.get_spin_raise(0, np.max(ell))[ell]) for s in [0, -2, 2]}
    if source == 'f':
        return {s: (0, 0.5 * np.ones(lmax + 1, dtype=float), 0.5 * np.ones(lmax + 1, dtype=float),
                    lambda ell: np.ones(len(ell), dtype=float)) for s in [0, -2, 2]}
    if source in ['a', 'a_p']:
        ret = {s: (0, -np.sign(s) * 1j * np.ones(lmax + 1, dtype=float), -np.sign(s) * 1j * np.ones(lmax + 1, dtype=float),
                   lambda ell: np.ones(len(ell), dtype=float)) for s in [-2, 2]}
        ret[0] = (0, np.zeros(lmax + 1, dtype=float), np.zeros(lmax + 1, dtype=float), lambda ell: np.ones(len(ell), dtype=float))
        return ret
    assert 0, source + ' response legs not implemented'


def get_covresp(source, s1, s2, cls, lmax, transf=None):
    """Response of the spin-(s1, s2) covariance to the anisotropy source (field representation or point sources)."""
    if source in ['p', 'x', 'f', 'a', 'a_p']:
        s_source, prR, mrR, cL_scal = get_resp_legs(source, lmax)[s1]
        coupl = uspin.spin_cls(s1, s2, cls)[:lmax + 1]
        return s_source, prR * coupl, mrR * coupl, cL_scal
    if source in ['stt', 's']:
        cond = s1 == 0 and s2 == 0
        w = 0.25 * cond * np.ones(lmax + 1, dtype=float)
        return 0, w, w.copy(), lambda ell: np.ones(len(ell), dtype=float)
    assert 0, 'source ' + source + ' cov. response not implemented'


def qe_spin_data(qe_key):
    """(output spin, 'G' | 'C', input spins, base key) of an estimator key."""
    if qe_key in ['ntt']:
        return 0, 'G', [0], 'n'
    qes = get_qes(qe_key, 10, {k: np.ones(11 + 4, dtype=float) for k in ['tt', 'te', 'ee', 'bb']})
    spins_out = [q.leg_a.spin_ou + q.leg_b.spin_ou for q in qes]
    spins_in = np.unique(np.abs([q.leg_a.spin_in for q in qes] + [q.leg_b.spin_in for q in qes]))
    assert len(np.unique(spins_out)) == 1, spins_out
    assert spins_out[0] >= 0, spins_out[0]
    return spins_out[0], 'C' if qe_key[0] == 'x' else 'G', spins_in, 'p' if qe_key[0] == 'x' else qe_key[0]
