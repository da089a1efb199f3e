"""kernel durations of the CG vector primitives at a coarse-level size (run under rocprofv3 --kernel-trace --stats)"""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
from plancklens_amd import dev, hp
lmax = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = hp.Alm.getsize(lmax)
a = torch.randn(n, dtype=torch.complex128, device='cuda'); b = torch.randn(n, dtype=torch.complex128, device='cuda')
fl = np.ones(lmax + 1)
for _ in range(200):
    d = dev.alm_dot([(a, b)])
    dev.axpy_dev(a, b, d, d, 1e-9)
    dev.almxfl_add(a, b, fl, out=a)
    c = dev.alm_copy(a, lmax // 2)
    dev.alm_splice(c, a, lmax // 2)
torch.cuda.synchronize()
