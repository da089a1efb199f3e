"""BASELINE config 4: qcinv CG Wiener filter T + P at nside = lmax = 2048 on a masked sky, fixed number of top-level
iterations; prints CG iterations/s.   usage: python tools/cg_bench.py [nside] [lmax] [iters]"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from plancklens_amd import dev, hp, shts, utils
from plancklens_amd.filt import filt_cinv
from plancklens_amd.qcinv import cd_solve

nside = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
lmax = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rng = np.random.default_rng(7)
npix = hp.nside2npix(nside)
cl = utils.camb_clfile(os.path.join('plancklens_amd', 'data', 'cls', 'FFP10_wdipole_lensedCls.dat'), lmax=lmax)
transf = hp.gauss_beam(5. / 60. / 180. * np.pi, lmax=lmax)
nlev_t, nlev_p = 35., 55.
# mask: |b| < 20 deg band + 200 random 1-degree discs (fsky ~ 0.6)
x, y, z = hp.pix2vec(nside)
mask = (np.abs(z) > np.sin(np.radians(20.))).astype(float)
cen = rng.standard_normal((200, 3)); cen /= np.linalg.norm(cen, axis=1)[:, None]
vec = np.stack([x, y, z])
for c in cen:
    mask[(c @ vec) > np.cos(np.radians(1.))] = 0.
print('fsky', mask.mean(), flush=True)
vamin = np.sqrt(hp.nside2pixarea(nside, degrees=True)) * 60
tmap = shts.alm2map(hp.almxfl(hp.synalm(cl['tt'], lmax, rng), transf), nside) + nlev_t / vamin * rng.standard_normal(npix)
q, u = shts.alm2map_spin([hp.almxfl(hp.synalm(cl['ee'], lmax, rng), transf), hp.almxfl(hp.synalm(cl['bb'], lmax, rng), transf)], nside, 2, lmax)
q += nlev_p / vamin * rng.standard_normal(npix); u += nlev_p / vamin * rng.standard_normal(npix)
tmp = tempfile.mkdtemp(prefix='cgbench_')


def chain(kind, n):
    pcf = os.path.join(tmp, 'dense_%s.pk' % kind)
    if kind == 't':
        return [[3, ["split(dense(" + pcf + "), 64, diag_cl)"], 256, 128, 3, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()],
                [2, ["split(stage(3),  256, diag_cl)"], 512, 256, 3, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()],
                [1, ["split(stage(2),  512, diag_cl)"], 1024, 512, 3, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()],
                [0, ["split(stage(1), 1024, diag_cl)"], lmax, nside, n, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()]]
    return [[2, ["split(dense(" + pcf + "), 32, diag_cl)"], 512, 256, 3, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()],
            [1, ["split(stage(2),  512, diag_cl)"], 1024, 512, 3, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()],
            [0, ["split(stage(1), 1024, diag_cl)"], lmax, nside, n, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()]]


ninv_t = [np.array([3. / nlev_t ** 2]) * mask]
ninv_p = [[np.array([3. / nlev_p ** 2]) * mask]]
res = {}
for kind in [k for k in ('t', 'p') if k in os.environ.get('CG_BENCH_ONLY', 'tp')]:
    t0 = time.time()
    if kind == 't':
        f = filt_cinv.cinv_t(os.path.join(tmp, 'cinv_t'), lmax, nside, cl, transf, ninv_t, chain_descr=chain('t', iters))
        f.chain.plogdepth = -1
        dmap = dev.to_dev(tmap)
        f.apply_ivf(dmap)  # builds the dense preconditioner (cached) and warms everything
    else:
        f = filt_cinv.cinv_p(os.path.join(tmp, 'cinv_p'), lmax, nside, cl, transf, ninv_p, chain_descr=chain('p', iters))
        f.chain.plogdepth = -1
        dmap = [dev.to_dev(q), dev.to_dev(u)]
        f.apply_ivf(dmap)
    setup = time.time() - t0
    trace = []
    log0 = f.chain.log
    f.chain.log = lambda stage, it, eps, **kw: (trace.append((stage.depth, it, eps)) if stage.depth == 0 else None, log0(stage, it, eps, **kw))
    torch.cuda.synchronize()
    t0 = time.time()
    f.apply_ivf(dmap)
    torch.cuda.synchronize()
    dt = time.time() - t0
    res[kind] = {'iters': iters, 'seconds': dt, 'iters_per_s': iters / dt, 'first_call_incl_dense_setup_s': setup,
                 'eps_trace': [float(t[2]) for t in trace][:iters + 1]}
    print(kind, json.dumps(res[kind]), flush=True)
if os.environ.get('CG_BENCH_JOINT', '0') == '1':
    # joint T+P filter (cinv_tp, default 4-stage chain; the dense block is 12 675 coarse fwd_ops to build, cached afterwards)
    pcf = os.path.join(tmp, 'dense_tp.pk')
    chain_tp = [[3, ["split(dense(" + pcf + "), 64, diag_cl)"], 256, 128, 3, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()],
                [2, ["split(stage(3),  256, diag_cl)"], 512, 256, 3, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()],
                [1, ["split(stage(2),  512, diag_cl)"], 1024, 512, 3, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()],
                [0, ["split(stage(1), 1024, diag_cl)"], lmax, nside, iters, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()]]
    t0 = time.time()
    cl_tp = {k: cl[k] for k in ['tt', 'ee', 'bb', 'te']}
    f = filt_cinv.cinv_tp(os.path.join(tmp, 'cinv_tp'), lmax, nside, cl_tp, transf, [ninv_t[0], ninv_p[0][0]], marge_monopole=True,
                          marge_dipole=True, chain_descr=chain_tp)
    f.chain.plogdepth = -1
    dmaps = [dev.to_dev(tmap), dev.to_dev(q), dev.to_dev(u)]
    f.apply_ivf(dmaps)
    setup = time.time() - t0
    torch.cuda.synchronize()
    t0 = time.time()
    f.apply_ivf(dmaps)
    torch.cuda.synchronize()
    dt = time.time() - t0
    res['tp_joint'] = {'iters': iters, 'seconds': dt, 'iters_per_s': iters / dt, 'first_call_incl_dense_setup_s': setup}
    print('tp_joint', json.dumps(res['tp_joint']), flush=True)
if 't' not in res or 'p' not in res:  # CG_BENCH_ONLY: a single filter (profiling runs)
    sys.exit(0)
tp = iters / (res['t']['seconds'] + res['p']['seconds'])
print(json.dumps({'metric': 'CG-iter/sec (cinv_t + cinv_p, nside=%d lmax=%d, masked fsky=%.2f)' % (nside, lmax, mask.mean()),
                  'T_iters_per_s': res['t']['iters_per_s'], 'P_iters_per_s': res['p']['iters_per_s'], 'TP_iters_per_s': tp,
                  'TP_joint_iters_per_s': res.get('tp_joint', {}).get('iters_per_s')}))
