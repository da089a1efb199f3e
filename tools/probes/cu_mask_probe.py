"""Probe for DESIGN.md section 5 ("one launch per nside-128 operator on one XCD"): what do the kernels of a coarse-level CG operator cost when
they only get 1/8 of the chip?  Runs opfilt_tt's one-call operator at (nside, lmax) = (128, 256) and (256, 512) on an ordinary stream and on a
stream created with a CU mask of 32 CUs (hipExtStreamCreateWithCUMask), HIP-event timing over many applications.
usage (GPU box): python3 tools/probes/cu_mask_probe.py"""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from plancklens_amd import hp, shts  # noqa: E402

hip = ctypes.CDLL('libamdhip64.so')


def masked_stream(ncu):
    """a stream restricted to `ncu` CUs: one bit per CU, the first ncu bits of the 256-bit mask set"""
    words = (ctypes.c_uint32 * 8)(*[0] * 8)
    for i in range(ncu):
        words[i // 32] |= (1 << (i % 32))
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def time_op(stream, nside, lmax, n=200):
    rng = np.random.default_rng(1)
    npix = 12 * nside ** 2
    ninv = torch.from_numpy(rng.uniform(0.5, 1.5, npix)).cuda()
    nalm = hp.Alm.getsize(lmax)
    x = torch.from_numpy(rng.standard_normal(nalm) + 1j * rng.standard_normal(nalm)).cuda()
    bl = np.ones(lmax + 1)
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        for _ in range(10):
            shts.cg_fwd_tt(x, nside, lmax, ninv, fl_in=bl, fl_out=bl)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(n):
            shts.cg_fwd_tt(x, nside, lmax, ninv, fl_in=bl, fl_out=bl)
        e1.record(stream)
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


if __name__ == '__main__':
    full = torch.cuda.Stream()
    for ncu in (32, 64, 128):
        m = masked_stream(ncu)
        for nside, lmax in ((128, 256), (256, 512)):
            a = time_op(full, nside, lmax)
            b = time_op(m, nside, lmax)
            print('operator (prep, Legendre synthesis, ring round trip, Legendre analysis, post) at nside %d lmax %d: %.1f us on the whole chip, %.1f us on %d CUs'
                  % (nside, lmax, a, b, ncu))
