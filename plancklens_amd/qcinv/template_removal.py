"""Templates marginalised by the pixel-space inverse-noise operator, API of plancklens/qcinv/template_removal.py
(`template_monopole` :116-129, `template_dipole` :132-150, `template_map` :36-60, `template_qmap` / `template_umap` :55-108,
`xyz_to_alm` / `alm_to_xyz` :153-166).  The reference evaluates the dipole through lmax = 1 SHTs; here the (x, y, z) pixel maps are built
once per nside on the device and the dot products are plain device reductions (same numbers: the lmax = 1
transform of the reference is sum_p m_p (x, y, z)_p up to the normalisation it divides out again)."""
import numpy as np
import torch

from .. import dev, hp
from .util import read_map

_XYZ = {}


def _xyz_dev(nside):
    key = (nside, torch.cuda.current_device())
    if key not in _XYZ:
        x, y, z = hp.pix2vec(nside)
        _XYZ[key] = dev.to_dev(np.stack([x, y, z]), torch.float64)
    return _XYZ[key]


def xyz_to_alm(xyz):
    assert len(xyz) == 3
    alm = np.zeros(3, dtype=complex)
    alm[1] = +xyz[2] * np.sqrt(4. * np.pi / 3.)
    alm[2] = (-xyz[0] + 1.j * xyz[1]) * np.sqrt(2. * np.pi / 3.)
    return alm


def alm_to_xyz(alm):
    assert len(alm) == 3
    return np.array([-alm[2].real / np.sqrt(2. * np.pi / 3.), +alm[2].imag / np.sqrt(2. * np.pi / 3.),
                     +alm[1].real / np.sqrt(4. * np.pi / 3.)])


class template(object):
    nmodes = 0

    def apply(self, m, coeffs):
        assert 0, 'override this'

    def apply_mode(self, m, mode):
        assert 0 <= mode < self.nmodes
        tcoeffs = np.zeros(self.nmodes)
        tcoeffs[mode] = 1.0
        self.apply(m, tcoeffs)

    def accum(self, m, coeffs):
        assert 0, 'override this'

    def dot(self, m):
        ret = []
        for i in range(self.nmodes):
            tmap = m.clone()
            self.apply_mode(tmap, i)
            ret.append(float(tmap.sum()))
        return ret


class template_map(template):
    """One arbitrary pixel-space template."""

    def __init__(self, tmap):
        self.nmodes = 1
        self.map = dev.to_dev(read_map(tmap), torch.float64)

    def apply(self, m, coeffs):
        assert len(coeffs) == self.nmodes
        m *= self.map * float(coeffs[0])

    def accum(self, m, coeffs):
        assert len(coeffs) == self.nmodes
        m += self.map * float(coeffs[0])

    def dot(self, m):
        return [float(torch.dot(self.map, m))]


class template_monopole(template):
    def __init__(self):
        self.nmodes = 1

    def apply(self, m, coeffs):
        assert len(coeffs) == self.nmodes
        m *= float(coeffs[0])

    def accum(self, m, coeffs):
        m += float(coeffs[0])

    def dot(self, m):
        return [float(m.sum())]


class template_dipole(template):
    def __init__(self):
        self.nmodes = 3

    def _combo(self, tmap, coeffs):
        xyz = _xyz_dev(hp.npix2nside(tmap.numel()))
        return float(coeffs[0]) * xyz[0] + float(coeffs[1]) * xyz[1] + float(coeffs[2]) * xyz[2]

    def apply(self, tmap, coeffs):
        assert len(coeffs) == self.nmodes
        tmap *= self._combo(tmap, coeffs)

    def accum(self, tmap, coeffs):
        assert len(coeffs) == self.nmodes
        tmap += self._combo(tmap, coeffs)

    def dot(self, tmap):
        xyz = _xyz_dev(hp.npix2nside(tmap.numel()))
        return list(dev.to_host(xyz @ tmap))


class _template_pol(template):
    """One polarization template living in the Q (comp 0) or U (comp 1) map of a (Q, U) pair
    (template_removal.py:55-108: template_qmap, template_umap).  opfilt_pp marginalises these through a device matrix
    (alm_filter_ninv._proj_matrices_p); the methods here are the reference's element-wise interface."""
    comp = 0

    def __init__(self, m):
        self.nmodes = 1
        self.map = m

    def _m(self, like):
        t = dev.to_dev(read_map(self.map), torch.float64)
        return t if isinstance(like, torch.Tensor) else dev.to_host(t)

    def apply(self, pmap, coeffs):
        assert len(coeffs) == self.nmodes
        if len(pmap) == 2:  # (Q, U): the template's component times the template, the other component zero
            pmap[self.comp] *= self._m(pmap[self.comp]) * float(coeffs[0])
            pmap[1 - self.comp] *= 0.
        else:  # the template's component alone
            assert len(pmap) == 1
            pmap[0] *= self._m(pmap[0]) * float(coeffs[0])

    def accum(self, pmap, coeffs):
        assert len(pmap) == 2 and len(coeffs) == self.nmodes
        pmap[self.comp] += self._m(pmap[self.comp]) * float(coeffs[0])

    def dot(self, pmap):
        m = pmap[self.comp if len(pmap) == 2 else 0]
        return [float((self._m(m) * m).sum())]


class template_qmap(_template_pol):
    comp = 0


class template_umap(_template_pol):
    comp = 1

