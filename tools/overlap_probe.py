"""Can independent transforms overlap (Legendre of one under the ring FFTs of another)?  Two full plans, two streams."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from plancklens_amd import shts, _lib
import ctypes
nside = lmax = 2048
n = (lmax + 1) * (lmax + 2) // 2
L = _lib.lib()
plans = [shts.Plan(nside, lmax) for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(3)]
alms = [torch.randn((2, n), dtype=torch.complex128, device='cuda') for _ in range(3)]
maps = [torch.empty((2, 12 * nside ** 2), dtype=torch.float64, device='cuda') for _ in range(3)]


def run(i, spin, st):
    _lib.check(L.pl_alm2map(plans[i].h, spin, alms[i].data_ptr(), maps[i].data_ptr(), None, 1, ctypes.c_void_p(st.cuda_stream)))


for i in range(3):
    run(i, 2, streams[i])
torch.cuda.synchronize()
for nconc in (1, 2, 3):
    torch.cuda.synchronize(); t0 = time.time()
    for rep in range(4):
        for i in range(nconc):
            run(i, 2, streams[i])
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 4
    print('%d concurrent spin-2 syntheses: %.2f ms per batch, %.2f ms per transform' % (nconc, 1e3 * dt, 1e3 * dt / nconc))
