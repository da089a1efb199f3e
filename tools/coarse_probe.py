"""Latency of the Legendre kernels on the coarse grids of the CG multigrid chain, by rings per lane (PLSHTS_R0 / R0A / RS / RSA are
read at every launch).  Run under rocprofv3 --kernel-trace --stats: the kernel names carry R.  usage: coarse_probe.py nside lmax"""
import os
import sys

import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from plancklens_amd import hp, shts

nside, lmax = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(0)
n = hp.Alm.getsize(lmax)
a = torch.tensor(rng.standard_normal(n) + 1j * rng.standard_normal(n), device='cuda')
eb = torch.stack([a, a.flip(0)])
rs = (0,) if len(sys.argv) > 3 else (1, 2, 3, 4, 6)  # third argument: default rings per lane only
for r in rs:
    os.environ['PLSHTS_R0'] = os.environ['PLSHTS_R0A'] = str(r)
    os.environ['PLSHTS_RS'] = os.environ['PLSHTS_RSA'] = str(min(r, 4))
    for _ in range(30):
        m = shts.alm2map(a, nside, lmax=lmax)
        shts.map2alm(m, lmax=lmax, iter=0)
        if r <= 4:
            qu = shts.alm2map_spin(eb, nside, 2, lmax)
            shts.map2alm_spin(qu, 2, lmax)
    torch.cuda.synchronize()
