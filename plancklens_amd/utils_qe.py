"""Generic spin-weight quadratic-estimator engine behind the names of plancklens/utils_qe.py (`qeleg` :5-32,
`qeleg_multi` :35-77, `qe` :79-90, `qe_eval` :92-132, `qe_proj` :135-177, `qe_simplify` :180-204, `qe_compress` :207-226).

A quadratic estimator is a list of terms  _a leg(n) x _b leg(n) -> analysed at spin a + b, times c_L, with
    _{s_out} leg(n) = sum_lm  w_l  _{s_in}Xb_lm  _{s_out}Y_lm(n),
Xb the inverse-variance filtered alms (_0Xb = Tb, _{+-2}Xb = -(Eb +- i Bb)).  This is the reference's second, independent
route to every estimator; here the term algebra is organised around hashable leg signatures
(co-adding and merging are dictionary groupings instead of pairwise scans) and the evaluation around device planes: a leg
is a pair of float64 maps (real, imaginary) made by one spin transform, products are accumulated by the complex-product
kernel pl_map_cmul, and the result is analysed once.
"""
import itertools

import numpy as np
import torch

from . import dev, hp
from . import utils_spin as uspin


# ---------------------------------------------------------------------------------------------------------------------
# terms
# ---------------------------------------------------------------------------------------------------------------------
class qeleg(object):
    """One input field component (spin_in) weighted by cl and synthesised at spin_ou."""
    __slots__ = ('spin_in', 'spin_ou', 'cl')

    def __init__(self, spin_in, spin_out, cl):
        self.spin_in, self.spin_ou, self.cl = spin_in, spin_out, cl

    def signature(self):
        """hashable identity of the leg: spins and the exact weights"""
        return (self.spin_in, self.spin_ou, self.get_lmax(), (np.asarray(self.cl, dtype=complex) + 0.).tobytes())  # + 0.: -0. == 0.

    def __eq__(self, other):
        return (self.spin_in, self.spin_ou, self.get_lmax()) == (other.spin_in, other.spin_ou, other.get_lmax()) and bool(np.all(self.cl == other.cl))

    __hash__ = None

    def __mul__(self, fac):
        return qeleg(self.spin_in, self.spin_ou, self.cl * fac)

    def __add__(self, other):
        assert (self.spin_in, self.spin_ou) == (other.spin_in, other.spin_ou), 'only legs of equal spins add up'
        long_, short = (self.cl, other.cl) if len(self.cl) >= len(other.cl) else (other.cl, self.cl)
        cl = np.array(long_, dtype=np.result_type(self.cl, other.cl))
        cl[:len(short)] += short
        return qeleg(self.spin_in, self.spin_ou, cl)

    def copy(self):
        return qeleg(self.spin_in, self.spin_ou, np.copy(self.cl))

    def get_lmax(self):
        return len(self.cl) - 1


class qe(object):
    """leg_a(n) leg_b(n), analysed at spin leg_a.spin_ou + leg_b.spin_ou >= 0 and multiplied by cL(L)."""

    def __init__(self, leg_a, leg_b, cL):
        assert leg_a.spin_ou + leg_b.spin_ou >= 0
        self.leg_a, self.leg_b, self.cL = leg_a, leg_b, cL

    def get_lmax_a(self):
        return self.leg_a.get_lmax()

    def get_lmax_b(self):
        return self.leg_b.get_lmax()

    def swapped(self):
        return qe(self.leg_b.copy(), self.leg_a.copy(), self.cL)


# how -(G + iC) = w_l _{s_in}Xb_lm reads in terms of the filtered T, E, B alms: (field, 'G' | 'C', sign)
_GC_OF_SPIN_IN = {0: (('t', 'G', -1.),), 2: (('e', 'G', 1.), ('b', 'C', 1.)), -2: (('e', 'G', 1.), ('b', 'C', -1.))}


def _weighted(alm, fl, lmax):
    """fl_l alm_lm truncated to lmax; fl may be complex (the polarization-rotation weights are imaginary)."""
    alm = dev.alm_copy(alm, lmax)
    fl = np.asarray(fl)
    f = np.zeros(lmax + 1, dtype=fl.dtype if np.iscomplexobj(fl) else float)
    n = min(lmax + 1, fl.size)
    f[:n] = fl[:n]
    if np.iscomplexobj(f):
        return alm * dev.to_dev(f, torch.complex128)[dev.lidx(lmax)]
    return dev.almxfl(alm, f)


class qeleg_multi(object):
    """Several weighted inputs synthesised by one transform at spin_ou (utils_qe.py:35-77)."""

    def __init__(self, spins_in, spin_out, cls):
        assert isinstance(spins_in, list) and isinstance(cls, list) and len(spins_in) == len(cls)
        self.spins_in, self.cls, self.spin_ou = spins_in, cls, spin_out

    def __iadd__(self, leg):
        assert leg.spin_ou == self.spin_ou, (leg.spin_ou, self.spin_ou)
        self.spins_in.append(leg.spin_in)
        self.cls.append(np.copy(leg.cl))
        return self

    def get_lmax(self):
        return max(len(cl) for cl in self.cls) - 1

    def planes(self, get_alm, nside):
        """(real, imaginary) float64 device maps of  sum_inputs sum_lm w_l _{s_in}Xb_lm _{s_ou}Y_lm(n).
        For s_ou > 0 the gradient / curl pair of hp.alm2map_spin is read off -(G + iC) = sum w _{s_in}Xb; a negative s_ou
        follows from  sum A_lm _{-s}Y_lm = (-1)^s conj(sum At_lm _sY_lm), At_lm = (-1)^m conj(A_l-m), which flips the
        sign of C; s_ou = 0 is the scalar transform of -G (utils_spin.alm2map_spin)."""
        lmax, s = self.get_lmax(), abs(self.spin_ou)
        gc = {'G': None, 'C': None}
        fields = {}
        for si, cl in zip(self.spins_in, self.cls):
            assert si in _GC_OF_SPIN_IN, str(si) + ' input spin not implemented'
            for f, part, sign in _GC_OF_SPIN_IN[si]:
                if f not in fields:
                    fields[f] = dev.to_dev(get_alm(f), torch.complex128)
                if part == 'C' and not bool(torch.any(fields[f] != 0)):
                    continue  # no curl input: the gradient-only transform does
                term = _weighted(fields[f], sign * cl, lmax)
                gc[part] = term if gc[part] is None else gc[part] + term
        glm = gc['G'] if gc['G'] is not None else torch.zeros(hp.Alm.getsize(lmax), dtype=torch.complex128, device=dev.device())
        clm = gc['C']
        if clm is not None and self.spin_ou < 0:
            clm = -clm
        if s == 0:
            re = uspin.alm2map_spin([glm, clm], nside, 0, lmax)[0]
            return re, torch.zeros_like(re)
        re, im = uspin.alm2map_spin([glm, clm], nside, s, lmax)
        if self.spin_ou < 0:  # (-1)^s x complex conjugate
            re, im = (re, -im) if s % 2 == 0 else (-re, im)
        return re, im

    def __call__(self, get_alm, nside):
        """complex map of the leg (device tensor), as the reference returns it (utils_qe.py:50-73)"""
        re, im = self.planes(get_alm, nside)
        return torch.complex(re, im)


# ---------------------------------------------------------------------------------------------------------------------
# term algebra
# ---------------------------------------------------------------------------------------------------------------------
def _cl_signature(cL, lmax):
    return (np.asarray(cL(np.arange(lmax + 1)), dtype=complex) + 0.).tobytes()


def _coadd_second_legs(terms):
    """Terms sharing the first leg, the spins of the second and the output weights become one term whose second leg carries
    the summed weights; terms with an identically vanishing leg drop out.  First-appearance order is kept."""
    groups = {}
    lmax_b = max([t.leg_b.get_lmax() for t in terms] + [0])
    for t in terms:
        key = (t.leg_a.signature(), t.leg_b.spin_in, t.leg_b.spin_ou, _cl_signature(t.cL, lmax_b))
        if key in groups:
            groups[key][1] = groups[key][1] + t.leg_b
        else:
            groups[key] = [t.leg_a.copy(), t.leg_b.copy(), t.cL]
    return [qe(a, b, cL) for a, b, cL in groups.values() if np.any(a.cl) and np.any(b.cl)]


def qe_simplify(qe_list, _swap=False, verbose=False):
    """Co-adds terms that differ only by the weights of one leg: first over second legs, then -- legs exchanged -- over first
    legs (utils_qe.py:180-204)."""
    if _swap:  # kept for call compatibility: one exchanged pass only
        return [t.swapped() for t in _coadd_second_legs([t.swapped() for t in qe_list])]
    once = _coadd_second_legs(list(qe_list))
    twice = [t.swapped() for t in _coadd_second_legs([t.swapped() for t in once])]
    if verbose and len(twice) < len(qe_list):
        print("%s terms down from %s" % (len(twice), len(qe_list)))
    return twice


def _field_components(leg, field):
    """The part of a leg that involves one field only.  A spin +-2 input is -(E +- iB): E alone (B alone) is the half sum
    (half difference) of the +2 and -2 inputs, so the leg splits into two legs of opposite input spin."""
    if field == 't':
        return [leg.copy()] if leg.spin_in == 0 else []
    if leg.spin_in == 0:
        return []
    sgn = 1. if field == 'e' else -1.
    return [leg * 0.5, qeleg(-leg.spin_in, leg.spin_ou, leg.cl * (0.5 * sgn))]


def qe_proj(qe_list, a, b):
    """The estimator restricted to field `a` on its first leg and field `b` on its second (utils_qe.py:135-177)."""
    assert a in ['t', 'e', 'b'] and b in ['t', 'e', 'b']
    terms = []
    for t in qe_list:
        for la, lb in itertools.product(_field_components(t.leg_a, a), _field_components(t.leg_b, b)):
            terms.append(qe(la, lb, t.cL))
    return qe_simplify(terms)


def qe_compress(qes, verbose=True):
    """Terms with the same first leg and the same output spin of the second are evaluated with one multi-input second leg:
    fewer spin transforms (utils_qe.py:207-226).  Returns (leg_a, leg_b, cL) triplets of qeleg_multi."""
    groups = {}
    for t in qes:
        key = (t.leg_a.signature(), t.leg_b.spin_ou)
        if key in groups:
            legb = groups[key][1]
            legb += t.leg_b  # in place: one more input of the same transform
        else:
            groups[key] = (qeleg_multi([t.leg_a.spin_in], t.leg_a.spin_ou, [t.leg_a.cl]),
                           qeleg_multi([t.leg_b.spin_in], t.leg_b.spin_ou, [t.leg_b.cl]), t.cL)
    if verbose and len(groups) < len(qes):
        print("%s alm2map_spin transforms now required, down from %s" % (2 * len(groups), 2 * len(qes)))
    return list(groups.values())


# ---------------------------------------------------------------------------------------------------------------------
# evaluation
# ---------------------------------------------------------------------------------------------------------------------
def qe_eval(qe_list, nside, get_alm, lmax_qlm, verbose=True, get_alm2=None):
    """Gradient and curl alm (host arrays) of the estimator defined by qe_list (utils_qe.py:92-132): the leg maps of every
    compressed term are multiplied into one complex product map on the device, analysed once at the estimator's spin and
    weighted by c_L.  With a second getter the estimator is symmetrised over its two inputs."""
    symmetrize = get_alm2 is not None and get_alm2 is not get_alm
    getters = [(get_alm, get_alm2), (get_alm2, get_alm)] if symmetrize else [(get_alm, get_alm)]
    terms = qe_compress(qe_list, verbose=verbose)
    Ls = np.arange(lmax_qlm + 1)
    spin = terms[0][0].spin_ou + terms[0][1].spin_ou
    cL_out = terms[0][2](Ls)
    assert spin >= 0, spin
    for la, lb, cL in terms[1:]:
        assert la.spin_ou + lb.spin_ou == spin and np.all(cL(Ls) == cL_out), 'terms of one estimator share spin and output weights'
    npix = hp.nside2npix(nside)
    dre = torch.zeros(npix, dtype=torch.float64, device=dev.device())
    dim = torch.zeros(npix, dtype=torch.float64, device=dev.device())
    for i, (la, lb, _) in enumerate(terms):
        if verbose:
            print("QE %s out of %s :" % (i + 1, len(terms)))
            print("in-spins 1st leg and out-spin", la.spins_in, la.spin_ou)
            print("in-spins 2nd leg and out-spin", lb.spins_in, lb.spin_ou)
        for ga, gb in getters:
            ar, ai = la.planes(ga, nside)
            br, bi = lb.planes(gb, nside)
            dev.map_cmul(ar, ai, 1., br, bi, 1., 1., dre, dim, True)  # d += leg_a x leg_b
    if spin > 0:
        glm, clm = uspin.map2alm_spin([dre, dim], spin, lmax=lmax_qlm)
    else:
        glm, clm = uspin.map2alm_spin([dre, dim], 0, lmax=lmax_qlm)[0], None
    norm = 0.5 if symmetrize else 1.
    glm = _weighted(glm, norm * cL_out, lmax_qlm)
    if clm is None or not bool(torch.any(clm != 0)):
        return dev.to_host(glm), np.zeros(hp.Alm.getsize(lmax_qlm), dtype=complex)
    return dev.to_host(glm), dev.to_host(_weighted(clm, norm * cL_out, lmax_qlm))
