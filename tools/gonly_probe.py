"""times the gradient-only spin synthesis against the general one (development aid)"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from plancklens_amd import shts
nside = lmax = 2048
n = (lmax + 1) * (lmax + 2) // 2
g = torch.randn(n, dtype=torch.complex128, device='cuda')
z = torch.zeros_like(g)
plan = shts.get_plan(nside, lmax)
for name, arg in (('general', [g, z]), ('grad-only', [g, None])):
    for spin in (1, 3):
        shts.alm2map_spin(arg, nside, spin, lmax)
        plan.profile(True); plan.profile_read()
        for _ in range(3):
            shts.alm2map_spin(arg, nside, spin, lmax)
        torch.cuda.synchronize()
        pr = plan.profile_read(); plan.profile(False)
        print(name, 'spin', spin, 'leg_synths %.3f ms' % (pr['leg_synths'][0] / pr['leg_synths'][1]))
