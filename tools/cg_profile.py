"""Where a top-level iteration of the temperature CG goes: SHT calls per resolution with synchronised timings, against the
un-instrumented wall time (development aid).   usage: python3 tools/cg_profile.py [iters]"""
import collections
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from plancklens_amd import dev, hp, shts, utils
from plancklens_amd.filt import filt_cinv
from plancklens_amd.qcinv import cd_solve, opfilt_tt

nside, lmax = 2048, 2048
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(7)
npix = hp.nside2npix(nside)
cl = utils.camb_clfile(os.path.join('plancklens_amd', 'data', 'cls', 'FFP10_wdipole_lensedCls.dat'), lmax=lmax)
transf = hp.gauss_beam(5. / 60. / 180. * np.pi, lmax=lmax)
x, y, z = hp.pix2vec(nside)
mask = (np.abs(z) > np.sin(np.radians(20.))).astype(float)
vamin = np.sqrt(hp.nside2pixarea(nside, degrees=True)) * 60
tmap = shts.alm2map(hp.almxfl(hp.synalm(cl['tt'], lmax, rng), transf), nside) + 35. / vamin * rng.standard_normal(npix)
tmp = tempfile.mkdtemp(prefix='cgprof_')
pcf = os.path.join(tmp, 'dense_t.pk')
chain = [[3, ["split(dense(" + pcf + "), 64, diag_cl)"], 256, 128, 3, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()],
         [2, ["split(stage(3),  256, diag_cl)"], 512, 256, 3, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()],
         [1, ["split(stage(2),  512, diag_cl)"], 1024, 512, 3, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()],
         [0, ["split(stage(1), 1024, diag_cl)"], lmax, nside, iters, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()]]
f = filt_cinv.cinv_t(os.path.join(tmp, 'cinv_t'), lmax, nside, cl, transf, [np.array([3. / 35. ** 2]) * mask], chain_descr=chain)
f.chain.plogdepth = -1
dmap = dev.to_dev(tmap)
f.apply_ivf(dmap)
torch.cuda.synchronize()
t0 = time.time(); f.apply_ivf(dmap); torch.cuda.synchronize(); wall = time.time() - t0
print('un-instrumented: %.1f ms per top-level iteration' % (1e3 * wall / iters))

stats = collections.defaultdict(lambda: [0, 0.])
a2m, m2a = opfilt_tt.alm2map, opfilt_tt.map2alm


def timed(fn, tag):
    def w(*args, **kw):
        torch.cuda.synchronize(); t = time.time()
        r = fn(*args, **kw)
        torch.cuda.synchronize()
        ns = args[1] if tag == 'synth' else hp.npix2nside(args[0].numel())
        s = stats[(tag, int(ns))]; s[0] += 1; s[1] += time.time() - t
        return r
    return w


opfilt_tt.alm2map, opfilt_tt.map2alm = timed(a2m, 'synth'), timed(m2a, 'anal')
t0 = time.time(); f.apply_ivf(dmap); torch.cuda.synchronize(); wall2 = time.time() - t0
print('instrumented: %.1f ms per iteration' % (1e3 * wall2 / iters))
tot = 0.
for k in sorted(stats):
    n, t = stats[k]
    tot += t
    print('%-6s nside %4d: %5d calls (%.1f / iter), %7.2f ms each, %7.1f ms / iter' % (k[0], k[1], n, n / iters, 1e3 * t / n, 1e3 * t / iters))
print('SHT total %.1f ms / iter; everything else %.1f ms / iter' % (1e3 * tot / iters, 1e3 * (wall2 - tot) / iters))
