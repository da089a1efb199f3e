"""Generic spin-weight quadratic-estimator engine, API of plancklens/utils_qe.py (`qeleg` :5-32, `qeleg_multi` :35-77,
`qe` :79-90, `qe_eval` :92-132, `qe_proj` :135-177, `qe_simplify` :180-204, `qe_compress` :207-226).

A QE is a list of terms  leg_a(n) x leg_b(n)  with leg(n) = sum_lm cl_l _{s_in}Xb_lm _{s_out}Y_lm(n); the legs are
synthesised, multiplied and analysed on the GPU.  This is the reference's second, independent route to every
estimator."""
import numpy as np
import torch

from . import dev, hp
from . import utils as ut
from . import utils_spin as uspin


class qeleg(object):
    def __init__(self, spin_in, spin_out, cl):
        self.spin_in = spin_in
        self.spin_ou = spin_out
        self.cl = cl

    def __eq__(self, leg):
        if self.spin_in != leg.spin_in or self.spin_ou != leg.spin_ou or self.get_lmax() != leg.get_lmax():
            return False
        return np.all(self.cl == leg.cl)

    def __mul__(self, other):
        return qeleg(self.spin_in, self.spin_ou, self.cl * other)

    def __add__(self, other):
        assert self.spin_in == other.spin_in and self.spin_ou == other.spin_ou
        lmax = max(self.get_lmax(), other.get_lmax())
        cl = np.zeros(lmax + 1, dtype=np.result_type(self.cl, other.cl))
        cl[:len(self.cl)] += self.cl
        cl[:len(other.cl)] += other.cl
        return qeleg(self.spin_in, self.spin_ou, cl)

    def copy(self):
        return qeleg(self.spin_in, self.spin_ou, np.copy(self.cl))

    def get_lmax(self):
        return len(self.cl) - 1


def _almxfl_any(alm, fl):
    """l-filter that may be complex (polarization-rotation weights are imaginary): torch gather on the device."""
    lmax = hp.Alm.getlmax(alm.numel())
    fl = np.asarray(fl)
    f = np.zeros(lmax + 1, dtype=complex if np.iscomplexobj(fl) else float)
    n = min(lmax + 1, fl.size)
    f[:n] = fl[:n]
    if np.iscomplexobj(f):
        return alm * dev.to_dev(f, torch.complex128)[dev.lidx(lmax)]
    return dev.almxfl(alm, f)


class qeleg_multi(object):
    def __init__(self, spins_in, spin_out, cls):
        assert isinstance(spins_in, list) and isinstance(cls, list) and len(spins_in) == len(cls)
        self.spins_in = spins_in
        self.cls = cls
        self.spin_ou = spin_out

    def __iadd__(self, leg):
        assert leg.spin_ou == self.spin_ou, (leg.spin_ou, self.spin_ou)
        self.spins_in.append(leg.spin_in)
        self.cls.append(np.copy(leg.cl))
        return self

    def __call__(self, get_alm, nside):
        """Complex spin-weight map of the leg (device tensor): X_lm = glm + i clm is assembled with the reference's
        sign rules, then handed to alm2map_spin (utils_qe.py:50-73)."""
        lmax = self.get_lmax()
        n = hp.Alm.getsize(lmax)
        glm = torch.zeros(n, dtype=torch.complex128, device=dev.device())
        clm = torch.zeros(n, dtype=torch.complex128, device=dev.device())
        has_c = False
        for si, cl in zip(self.spins_in, self.cls):
            assert si in [0, -2, 2], str(si) + ' input spin not implemented'
            if abs(si) == 2:
                g, c = dev.to_dev(get_alm('e'), torch.complex128), dev.to_dev(get_alm('b'), torch.complex128)
            else:
                g, c = -dev.to_dev(get_alm('t'), torch.complex128), None
            sgn_g = -(-1) ** si if si < 0 else -1
            sgn_c = (-1) ** si if si < 0 else -1
            glm += _almxfl_any(dev.alm_copy(g, lmax), sgn_g * cl)
            if c is not None and bool(torch.any(c != 0)):
                clm += _almxfl_any(dev.alm_copy(c, lmax), sgn_c * cl)
                has_c = True
        glm *= -1
        if self.spin_ou > 0:
            clm *= -1
        so = abs(self.spin_ou)
        if so > 0:
            red, imd = uspin.alm2map_spin([glm, clm], nside, so, lmax)
        else:
            red = uspin.alm2map_spin([glm, clm], nside, 0, lmax)[0]
            imd = torch.zeros_like(red)
        if self.spin_ou < 0 and self.spin_ou % 2 == 1:
            red = -red
        if self.spin_ou < 0 and self.spin_ou % 2 == 0:
            imd = -imd
        return torch.complex(red, imd)

    def get_lmax(self):
        return int(np.max([len(cl) for cl in self.cls])) - 1


class qe(object):
    def __init__(self, leg_a, leg_b, cL):
        assert leg_a.spin_ou + leg_b.spin_ou >= 0
        self.leg_a = leg_a
        self.leg_b = leg_b
        self.cL = cL

    def get_lmax_a(self):
        return self.leg_a.get_lmax()

    def get_lmax_b(self):
        return self.leg_b.get_lmax()


def qe_eval(qe_list, nside, get_alm, lmax_qlm, verbose=True, get_alm2=None):
    """Gradient and curl alm (host arrays) of the QE defined by qe_list (utils_qe.py:92-132)."""
    if get_alm2 is None:
        get_alm2 = get_alm
    symmetrize = not (get_alm2 is get_alm)
    qes = qe_compress(qe_list, verbose=verbose)
    qe_spin = qes[0][0].spin_ou + qes[0][1].spin_ou
    cL_out = qes[0][-1](np.arange(lmax_qlm + 1))
    assert qe_spin >= 0, qe_spin
    for q in qes[1:]:
        assert np.all(q[-1](np.arange(lmax_qlm + 1)) == cL_out)
        assert q[0].spin_ou + q[1].spin_ou == qe_spin
    d = torch.zeros(hp.nside2npix(nside), dtype=torch.complex128, device=dev.device())
    for i, q in enumerate(qes):
        if verbose:
            print("QE %s out of %s :" % (i + 1, len(qes)))
            print("in-spins 1st leg and out-spin", q[0].spins_in, q[0].spin_ou)
            print("in-spins 2nd leg and out-spin", q[1].spins_in, q[1].spin_ou)
        d += q[0](get_alm, nside) * q[1](get_alm2, nside)
        if symmetrize:
            d += q[0](get_alm2, nside) * q[1](get_alm, nside)
    re, im = d.real.contiguous(), d.imag.contiguous()
    if qe_spin > 0:
        glm, clm = uspin.map2alm_spin([re, im], qe_spin, lmax=lmax_qlm)
    else:
        glm, clm = uspin.map2alm_spin([re, im], 0, lmax=lmax_qlm)[0], torch.zeros(hp.Alm.getsize(lmax_qlm), dtype=torch.complex128, device=dev.device())
    if symmetrize:
        glm = glm * 0.5
        clm = clm * 0.5
    glm = _almxfl_any(glm, cL_out)
    if bool(torch.any(clm != 0)):
        clm = _almxfl_any(clm, cL_out)
    return dev.to_host(glm), dev.to_host(clm)


def qe_proj(qe_list, a, b):
    """Terms of qe_list whose first leg uses only field `a` and second leg only field `b` (utils_qe.py:135-177)."""
    assert a in ['t', 'e', 'b'] and b in ['t', 'e', 'b']
    l_in = [0] if a == 't' else [-2, 2]
    r_in = [0] if b == 't' else [-2, 2]
    qes_ret = []
    for q in qe_list:
        si, ri = q.leg_a.spin_in, q.leg_b.spin_in
        if si in l_in and ri in r_in:
            leg_a, leg_b = q.leg_a.copy(), q.leg_b.copy()
            if si == 0 and ri == 0:
                qes_ret.append(qe(leg_a, leg_b, q.cL))
            elif si == 0 and abs(ri) > 0:
                sgn = 1 if b == 'e' else -1
                qes_ret.append(qe(leg_a, leg_b * 0.5, q.cL))
                leg_b.spin_in *= -1
                qes_ret.append(qe(leg_a, leg_b * 0.5 * sgn, q.cL))
            elif ri == 0 and abs(si) > 0:
                sgn = 1 if a == 'e' else -1
                qes_ret.append(qe(leg_a * 0.5, leg_b, q.cL))
                leg_a.spin_in *= -1
                qes_ret.append(qe(leg_a * 0.5 * sgn, leg_b, q.cL))
            elif abs(ri) > 0 and abs(si) > 0:
                sgna = 1 if a == 'e' else -1
                sgnb = 1 if b == 'e' else -1
                qes_ret.append(qe(leg_a * 0.5, leg_b * 0.5, q.cL))
                leg_b.spin_in *= -1
                qes_ret.append(qe(leg_a * 0.5, leg_b * 0.5 * sgnb, q.cL))
                leg_a.spin_in *= -1
                qes_ret.append(qe(leg_a * 0.5 * sgna, leg_b * 0.5 * sgnb, q.cL))
                leg_b.spin_in *= -1
                qes_ret.append(qe(leg_a * 0.5 * sgna, leg_b * 0.5, q.cL))
            else:
                assert 0, (si, ri)
    return qe_simplify(qes_ret)


def qe_simplify(qe_list, _swap=False, verbose=False):
    """Co-adds terms with identical first leg and compatible second leg, then the same with the legs swapped."""
    skip = []
    qes_ret = []
    qes = [qe(q.leg_b.copy(), q.leg_a.copy(), q.cL) for q in qe_list] if _swap else qe_list
    for i, qe1 in enumerate(qes):
        if i in skip:
            continue
        leg_a, leg_b = qe1.leg_a.copy(), qe1.leg_b.copy()
        for j, qe2 in enumerate(qes[i + 1:]):
            if qe2.leg_a == leg_a and qe2.leg_b.spin_in == qe1.leg_b.spin_in and qe2.leg_b.spin_ou == qe1.leg_b.spin_ou:
                Ls = np.arange(max(qe1.leg_b.get_lmax(), qe2.leg_b.get_lmax()) + 1)
                if np.all(qe1.cL(Ls) == qe2.cL(Ls)):
                    leg_b = leg_b + qe2.leg_b
                    skip.append(j + i + 1)
        if np.any(leg_a.cl) and np.any(leg_b.cl):
            qes_ret.append(qe(leg_a, leg_b, qe1.cL))
    if verbose and len(skip) > 0:
        print("%s terms down from %s" % (len(qes_ret), len(qes)))
    if not _swap:
        return qe_simplify(qes_ret, _swap=True, verbose=verbose)
    return [qe(q.leg_b.copy(), q.leg_a.copy(), q.cL) for q in qes_ret]


def qe_compress(qes, verbose=True):
    """Merges terms with identical first leg into multi-input second legs: fewer spin transforms."""
    skip = []
    out = []
    for i, qi in enumerate(qes):
        if i in skip:
            continue
        lega = qi.leg_a
        lega_m = qeleg_multi([qi.leg_a.spin_in], qi.leg_a.spin_ou, [qi.leg_a.cl])
        legb_m = qeleg_multi([qi.leg_b.spin_in], qi.leg_b.spin_ou, [qi.leg_b.cl])
        for j, qj in enumerate(qes[i + 1:]):
            if qj.leg_a == lega and legb_m.spin_ou == qj.leg_b.spin_ou:
                legb_m += qj.leg_b
                skip.append(i + 1 + j)
        out.append((lega_m, legb_m, qi.cL))
    if len(skip) > 0 and verbose:
        print("%s alm2map_spin transforms now required, down from %s" % (2 * (len(qes) - len(skip)), 2 * len(qes)))
    return out
