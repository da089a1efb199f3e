"""Batched synthesis (two simulations on one recursion) against two single transforms: time per pair of maps, full transforms
and Legendre stage only (HIP-event profile of the plan)."""
import sys, time
import torch
sys.path.insert(0, '.')
from plancklens_amd import shts
nside = lmax = 2048
plan = shts.get_plan(nside, lmax)
n = plan.nalm
for spin in (2, 3):
    a1 = torch.randn((2, n), dtype=torch.complex128, device='cuda')
    a2 = torch.randn((2, n), dtype=torch.complex128, device='cuda')
    def single():
        shts.alm2map_spin([a1[0], a1[1]], nside, spin, lmax); shts.alm2map_spin([a2[0], a2[1]], nside, spin, lmax)
    def batch():
        shts.alm2map_spin_batch2([a1[0], a1[1]], [a2[0], a2[1]], nside, spin, lmax)
    for name, fn in (('two single transforms', single), ('batched', batch)):
        fn(); torch.cuda.synchronize()
        plan.profile(True); plan.profile_read()
        t0 = time.time()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 5
        pr = plan.profile_read(); plan.profile(False)
        leg = sum(v[0] for k, v in pr.items() if k.startswith('leg_')) / 5
        fft = pr['fft_synth'][0] / 5
        print('spin %d %-22s %.2f ms per pair of maps (Legendre %.2f ms, ring FFTs %.2f ms)' % (spin, name, 1e3 * dt, leg, fft))
