"""Two independent mean-field loops of the MV estimator at the same time on two streams of one process (two plan contexts: own workspaces, own captured graphs) against
the two loops one after the other: does the chip have room for a second reconstruction pair beside the first (ring-FFT stages and latency-bound kernels of one under the
FMA-bound Legendre kernels of the other)?  The qcinv solves gain 1.06-1.13x this way (filt_cinv.run_tp); the estimator's single-simulation lanes gained nothing.
usage (GPU box): python3 tools/qe_two_streams.py [nside] [pairs per loop]"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from plancklens_amd import dev, hp, qest, shts, utils  # noqa: E402
from plancklens_amd.filt import filt_cinv, filt_simple  # noqa: E402

nside = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
lmax = nside
cl_len = utils.camb_clfile(os.path.join(ROOT, 'plancklens_amd', 'data', 'cls', 'FFP10_wdipole_lensedCls.dat'), lmax=lmax)
transf = hp.gauss_beam(5. / 60. / 180. * np.pi, lmax=lmax)
arcmin = np.pi / 180. / 60.
ftl = utils.cli(cl_len['tt'] + (35. * arcmin) ** 2 * utils.cli(transf ** 2))
fel = utils.cli(cl_len['ee'] + (55. * arcmin) ** 2 * utils.cli(transf ** 2))
fbl = utils.cli(cl_len['bb'] + (55. * arcmin) ** 2 * utils.cli(transf ** 2))
for f in (ftl, fel, fbl):
    f[:100] = 0.
sims = bench.resident_sims(nside, lmax, cl_len, transf, 35., 55., seed=1, nsets=2)
tmp = tempfile.mkdtemp(prefix='qe2s_')
libs = []
for tag in 'ab':
    ivfs = filt_simple.library_fullsky_sepTP(os.path.join(tmp, 'ivfs' + tag), sims, nside, transf, cl_len, ftl, fel, fbl, cache=False)
    libs.append(qest.library_sepTP(os.path.join(tmp, 'q' + tag), ivfs, ivfs, cl_len['te'], nside, lmax_qlm=lmax, cache=False))
qa, qb = libs


def loop(q, first):
    def job():
        q._mem.clear()
        mf = q.get_sim_qlm_mf('p', np.arange(first, first + 2 * npairs))
        for f_ in list(dev.host_future._in_flight):
            f_.result()
        return mf
    return job


# warm-up: both libraries capture their pair graphs, qb inside the second plan context (run_tp(warm=False) runs the two jobs one after the other in the contexts of the concurrent form)
for rep in range(4):
    filt_cinv.run_tp(loop(qa, 1000 + 20 * rep), loop(qb, 2000 + 20 * rep), warm=False)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    ma, mb = filt_cinv.run_tp(loop(qa, 0), loop(qb, 100), warm=False)
    torch.cuda.synchronize()
    t_seq = time.perf_counter() - t0
    t0 = time.perf_counter()
    ma2, mb2 = filt_cinv.run_tp(loop(qa, 0), loop(qb, 100), warm=True)
    torch.cuda.synchronize()
    t_con = time.perf_counter() - t0
    same = bool(np.array_equal(ma, ma2) and np.array_equal(mb, mb2))
    n = 4 * npairs
    print('nside %d, 2 x %d reconstructions: one loop after the other %.1f ms (%.2f ms each), both at once on two streams %.1f ms (%.2f ms each): x %.3f; same mean fields: %s'
          % (nside, 2 * npairs, 1e3 * t_seq, 1e3 * t_seq / n, 1e3 * t_con, 1e3 * t_con / n, t_seq / t_con, same), flush=True)
