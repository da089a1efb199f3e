"""Stage-level timing of the SHT kernels at one (nside, lmax) -- development aid for profiling runs.
usage: python3 tools/kernel_bench.py [nside] [lmax] [reps] [stages: ls,la,ps,pa] [spins: 0,2] [plan options: fft_legacy=1,...]"""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from plancklens_amd import _lib, shts

nside = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
lmax = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
stages = sys.argv[4].split(',') if len(sys.argv) > 4 else ['ls', 'la', 'ps', 'pa']
spins = [int(s) for s in sys.argv[5].split(',')] if len(sys.argv) > 5 else [0, 2]
opts = {kv.split('=')[0]: int(kv.split('=')[1]) for kv in sys.argv[6].split(',')} if len(sys.argv) > 6 else {}  # pl_plan_opts, e.g. fft_legacy=1
L = _lib.lib()
with shts.plan_options(**opts):
    plan = shts.get_plan(nside, lmax)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(1)
nalm = (lmax + 1) * (lmax + 2) // 2
steps = nalm * 2 * nside
for spin in spins:
    nc = 1 if spin == 0 else 2
    alm = torch.from_numpy(rng.standard_normal((nc, nalm)) + 1j * rng.standard_normal((nc, nalm))).cuda()
    mp = torch.from_numpy(rng.standard_normal((nc, 12 * nside ** 2))).cuda()
    ph = torch.zeros(plan.phase_doubles(spin), dtype=torch.float64, device='cuda')
    a2 = torch.zeros_like(alm)
    fns = {'ls': lambda: L.pl_legendre_synth(plan.h, spin, alm.data_ptr(), None, ph.data_ptr(), st),
           'ps': lambda: L.pl_phase2map(plan.h, spin, ph.data_ptr(), mp.data_ptr(), st),
           'pa': lambda: L.pl_map2phase(plan.h, spin, mp.data_ptr(), ph.data_ptr(), st),
           'la': lambda: L.pl_legendre_anal(plan.h, spin, ph.data_ptr(), a2.data_ptr(), None, st)}
    for name in ['ls', 'ps', 'pa', 'la']:
        if name not in stages:
            fns[name]()
            continue
        fn = fns[name]
        _lib.check(fn())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = (8 if spin == 0 else 24) * steps
        print('nside %d lmax %d spin %d %s: %.3f ms  (%.1f alg TF/s)' % (nside, lmax, spin, name, ms, fl / ms / 1e9), flush=True)
