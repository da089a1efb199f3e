"""Two scalar syntheses on one Legendre recursion (k_leg_synth0<R, true>, round 6) against two single launches: Legendre-stage time from the
plan's HIP-event profile, whole transforms from stream events.  usage (GPU box): python3 tools/s0_pair_ab.py [nside] [lmax] [reps]"""
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from plancklens_amd import hp, shts

nside = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
lmax = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
rng = np.random.default_rng(0)
n = hp.Alm.getsize(lmax)
a = torch.from_numpy(rng.standard_normal((2, n)) + 1j * rng.standard_normal((2, n))).cuda()
plan = shts.get_plan(nside, lmax)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    plan.profile(True)
    plan.profile_read()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    prof = plan.profile_read()
    plan.profile(False)
    return e0.elapsed_time(e1) / reps, {k: (v[0] / max(v[1], 1), v[1]) for k, v in prof.items() if v[1]}


for rep in range(2):
    t1, p1 = timed(lambda: (shts.alm2map(a[0], nside, lmax=lmax), shts.alm2map(a[1], nside, lmax=lmax)))
    t2, p2 = timed(lambda: shts.alm2map_batch2(a[0], a[1], nside, lmax=lmax))
    l1 = 2 * p1['leg_synth0'][0]
    l2 = p2.get('leg_synth0_pair', p2.get('leg_synth0'))[0]
    print('nside %d lmax %d: two single transforms %.3f ms (Legendre 2 x %.3f), pair %.3f ms (Legendre %.3f): Legendre ratio %.3f, transform ratio %.3f'
          % (nside, lmax, t1, l1 / 2, t2, l2, l2 / l1, t2 / t1), flush=True)
