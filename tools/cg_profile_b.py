"""One block solve of the temperature or polarization CG at nside = lmax = 2048 for kernel traces (tools/prof_cg_batch.sh):
    python3 tools/cg_profile_b.py [iters] [B] [t|p]
3 solves after a warm-up one (graph capture of the nested stages)."""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from plancklens_amd import dev, hp, shts, utils
from plancklens_amd.filt import filt_cinv
sys.path.insert(0, 'tools')
import cg_bench

nside, lmax = 2048, 2048
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kind = sys.argv[3] if len(sys.argv) > 3 else 't'
rng = np.random.default_rng(7)
npix = hp.nside2npix(nside)
cl = utils.camb_clfile(os.path.join('plancklens_amd', 'data', 'cls', 'FFP10_wdipole_lensedCls.dat'), lmax=lmax)
transf = hp.gauss_beam(5. / 60. / 180. * np.pi, lmax=lmax)
mask = cg_bench.make_mask(nside, rng)
tmp = tempfile.mkdtemp(prefix='cgprofb_')
pcf = os.path.join(tmp, 'dense.pk')
gen = torch.Generator(device='cuda')
gen.manual_seed(3)
if kind == 't':
    f = filt_cinv.cinv_t(os.path.join(tmp, 'cinv'), lmax, nside, cl, transf, [np.array([3. / 35. ** 2]) * mask], chain_descr=cg_bench.chain('t', iters, lmax, nside, pcf))
    dmaps = [torch.randn(npix, generator=gen, dtype=torch.float64, device='cuda') * 50. for _ in range(B)]
else:
    f = filt_cinv.cinv_p(os.path.join(tmp, 'cinv'), lmax, nside, cl, transf, [[np.array([3. / 55. ** 2]) * mask]], chain_descr=cg_bench.chain('p', iters, lmax, nside, pcf))
    dmaps = [[torch.randn(npix, generator=gen, dtype=torch.float64, device='cuda') * 5. for _ in range(2)] for _ in range(B)]
f.chain.plogdepth = -1
sys.stdout = open(os.devnull, 'w')
run = (lambda: f.apply_ivf(dmaps[0])) if B == 1 else (lambda: f.apply_ivf_batch(dmaps))
run()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(3):
    run()
torch.cuda.synchronize()
sys.stdout = sys.__stdout__
print('un-instrumented: %.2f ms per top-level iteration (B = %d, %s)' % (1e3 * (time.time() - t0) / 3 / iters, B, kind))
