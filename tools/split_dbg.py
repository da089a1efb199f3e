"""Debug aid: ring FFT stages with split Bluestein rings (PLSHTS_FFT_SPLIT) against the unsplit classes, per ring pair."""
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nside, lmax = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3:  # child: dump
    import ctypes
    import torch
    from plancklens_amd import _lib, shts
    L = _lib.lib()
    plan = shts.get_plan(nside, lmax)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(3)
    nph = plan.phase_doubles(0)
    ph = torch.from_numpy(rng.standard_normal(nph)).cuda()
    # m = 0 phases are real
    mp = torch.zeros(12 * nside ** 2, dtype=torch.float64, device='cuda')
    _lib.check(L.pl_phase2map(plan.h, 0, ph.data_ptr(), mp.data_ptr(), st))
    torch.cuda.synchronize()
    np.save(sys.argv[3] + '_map.npy', mp.cpu().numpy())
    m2 = torch.from_numpy(rng.standard_normal(12 * nside ** 2)).cuda()
    ph2 = torch.zeros(nph, dtype=torch.float64, device='cuda')
    _lib.check(L.pl_map2phase(plan.h, 0, m2.data_ptr(), ph2.data_ptr(), st))
    torch.cuda.synchronize()
    np.save(sys.argv[3] + '_ph.npy', ph2.cpu().numpy())
    sys.exit(0)
for s in ('0', '1024'):
    subprocess.check_call([sys.executable, __file__, str(nside), str(lmax), '/tmp/splitdbg_' + s], env=dict(os.environ, PLSHTS_FFT_SPLIT=s))
a, b = np.load('/tmp/splitdbg_0_map.npy'), np.load('/tmp/splitdbg_1024_map.npy')
# per ring errors (north cap rings i: 4 i pixels starting at 2 i (i - 1))
print('map: max abs diff %.3e of scale %.3e' % (np.abs(a - b).max(), np.abs(a).max()))
bad = [i for i in range(1, nside) if np.abs(a[2 * i * (i - 1):2 * i * (i - 1) + 4 * i] - b[2 * i * (i - 1):2 * i * (i - 1) + 4 * i]).max() > 1e-9]
print('bad north cap rings:', bad[:10], '...', bad[-10:], len(bad))
for i in list(range(1410, 1425)) + [1500, 1800, 2000, 2040, 2047]:
    sl = slice(2 * i * (i - 1), 2 * i * (i - 1) + 4 * i)
    d = np.abs(a[sl] - b[sl])
    print('ring q=%d: max diff %.3e at pixel %d (first half max %.3e, second half %.3e)' % (i, d.max(), d.argmax(), d[:4 * i].reshape(4, i)[:, :1024].max(), d.reshape(4, i)[:, 1024:].max() if i > 1024 else 0))
a, b = np.load('/tmp/splitdbg_0_ph.npy'), np.load('/tmp/splitdbg_1024_ph.npy')
print('phase: max abs diff %.3e of scale %.3e' % (np.abs(a - b).max(), np.abs(a).max()))
d_ = np.abs(a - b).reshape(-1, (lmax + 1 + 3) // 4 * 4 * 4).max(axis=1)
badp = np.where(d_ > 1e-15)[0]
print('bad pairs (phase):', badp[:10], '...', badp[-10:], len(badp))
mstride = (lmax + 1 + 3) // 4 * 4
a, b = a.reshape(-1, mstride, 4), b.reshape(-1, mstride, 4)
for ip in [0, 1, 10, 100, 500, 620, 629, 630, 640, 700, 1000]:
    d = np.abs(a[ip] - b[ip])
    print('pair %d: max diff %.3e at m=%d' % (ip, d.max(), d.max(axis=1).argmax()))
