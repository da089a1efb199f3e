"""times the gradient-only spin synthesis against the general one, interleaved after a long warm-up (development aid)"""
import sys
import torch
sys.path.insert(0, '.')
from plancklens_amd import shts
nside = lmax = 2048
n = (lmax + 1) * (lmax + 2) // 2
g = torch.randn(n, dtype=torch.complex128, device='cuda')
z = torch.zeros_like(g)
plan = shts.get_plan(nside, lmax)
for _ in range(30):
    shts.alm2map_spin([g, z], nside, 1, lmax)
for rep in range(3):
    for name, arg in (('general', [g, z]), ('grad-only', [g, None])):
        for spin in (1, 3):
            plan.profile(True); plan.profile_read()
            for _ in range(10):
                shts.alm2map_spin(arg, nside, spin, lmax)
            torch.cuda.synchronize()
            pr = plan.profile_read(); plan.profile(False)
            k = 'leg_synths' if pr['leg_synths'][1] else 'leg_synths_grad'
            print(name, 'spin', spin, k, '%.3f ms' % (pr[k][0] / pr[k][1]))
