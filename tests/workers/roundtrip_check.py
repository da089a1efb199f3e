"""Prints a checksum of the CG operators on all-generic grids; run with PLSHTS_DEBUG=1 PLSHTS_CG_ROUNDTRIP=0 and =1: the sums must be equal
(k_ring_roundtrip is bit-identical to k_phase2map + k_map2phase)."""
import hashlib
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from plancklens_amd import dev, hp
from plancklens_amd.qcinv import opfilt_pp, opfilt_tt
from plancklens_amd.qcinv.util_alm import eblm

h = hashlib.sha256()
for nside, lmax, nb in [(8, 16, 1), (16, 47, 3), (32, 64, 2), (64, 100, 1), (128, 256, 4), (256, 512, 1), (256, 512, 2), (48, 96, 1)]:
    rng = np.random.default_rng(nside + lmax)
    npix = 12 * nside ** 2
    ell = np.arange(lmax + 1.)
    bl = np.exp(-ell * (ell + 1.) * 1e-6)
    x, y, z = hp.pix2vec(nside, np.arange(npix))
    ninv = (1. + 0.3 * x) * (np.abs(z) > 0.3) * (1. + rng.random(npix))
    cl = {'tt': 1. / (ell + 3.) ** 2, 'ee': .1 / (ell + 3.) ** 2, 'bb': .01 / (ell + 3.) ** 2}
    nalm = (lmax + 1) * (lmax + 2) // 2

    def ralm():
        a = rng.standard_normal((nb, nalm)) + 1j * rng.standard_normal((nb, nalm))
        a[:, :lmax + 1] = a[:, :lmax + 1].real
        t = torch.from_numpy(a).cuda()
        return t if nb > 1 else t[0].contiguous()
    for marge in (False, True):
        nf = opfilt_tt.alm_filter_ninv(ninv, bl, marge_monopole=marge, marge_dipole=marge)
        h.update(dev.to_host(opfilt_tt.fwd_op(cl, nf)(ralm())).tobytes())
    r = opfilt_pp.fwd_op(cl, opfilt_pp.alm_filter_ninv([ninv], bl))(eblm([ralm(), ralm()]))
    h.update(dev.to_host(r.elm).tobytes())
    h.update(dev.to_host(r.blm).tobytes())
print(h.hexdigest())
