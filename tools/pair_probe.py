"""times the paired spin-1 synthesis (pl_alm2map_pair) against the general + gradient-only transforms it replaces (development aid)"""
import sys
import torch
sys.path.insert(0, '.')
from plancklens_amd import shts
nside = lmax = 2048
n = (lmax + 1) * (lmax + 2) // 2
g, c, g2 = (torch.randn(n, dtype=torch.complex128, device='cuda') for _ in range(3))
plan = shts.get_plan(nside, lmax)
for _ in range(20):
    shts.alm2map_spin([g, c], nside, 1, lmax)
for rep in range(3):
    plan.profile(True); plan.profile_read()
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    for _ in range(10):
        shts.alm2map_spin([g, c], nside, 1, lmax); shts.alm2map_spin([g2, None], nside, 1, lmax)
    e1.record()
    for _ in range(10):
        shts.alm2map_spin_pair([g, c], g2, nside, 1, lmax)
    e2.record()
    torch.cuda.synchronize()
    pr = plan.profile_read(); plan.profile(False)
    print('separate %.3f ms  paired %.3f ms per pair of transforms;' % (e0.elapsed_time(e1) / 10, e1.elapsed_time(e2) / 10),
          {k: round(v[0] / max(v[1], 1), 3) for k, v in pr.items() if v[1]})
