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



def harmonic_matrices(templates, n_inv, map2alm, lmax, fl_out, pt_nn1_p_inv):
    """Scalar templates in harmonic space, for the rank-nmodes form of the projection inside a CG operator:
        B^t Y^t [N^-1 - N^-1 T (T^t N^-1 T)^-1 T^t N^-1] Y B x = B^t Y^t N^-1 Y B x - V (T^t N^-1 T)^-1 V^t x,   V = B^t Y^t N^-1 T
    (the bracket is what opfilt_tt.py:196-205 applies to the map; Y^t the adjoint of the synthesis, (npix / 4 pi) map2alm, both factors in
    fl_out).  Returns two real (nmodes, 2 nalm) device matrices for dev.lowrank_update: V with the weights of the real scalar product of
    alm vectors folded in (1 for m = 0, 2 above), and (T^t N^-1 T)^-1 V.  One analysis transform per mode."""
    rows = []
    for t in templates:
        for i in range(t.nmodes):
            tmap = n_inv.clone()
            t.apply_mode(tmap, i)
            vlm = dev.to_dev(map2alm(tmap, lmax=lmax, iter=0, fl=fl_out), torch.complex128).contiguous()
            rows.append(torch.view_as_real(vlm).reshape(-1))
    v = torch.stack(rows).contiguous()
    w = torch.full((v.shape[1] // 2,), 2., dtype=torch.float64, device=v.device)
    w[:lmax + 1] = 1.
    pinv = dev.to_dev(np.ascontiguousarray(pt_nn1_p_inv), torch.float64)
    return (v * w.repeat_interleave(2).unsqueeze(0)).contiguous(), torch.mm(pinv, v).contiguous()
