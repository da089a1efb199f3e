"""BASELINE config 4: qcinv CG Wiener filter T + P at nside = lmax = 2048 on a masked sky (|b| < 20 deg band + 200 random
1-degree discs, fsky ~ 0.65), fixed number of top-level iterations (eps_min = 0); CG iterations/s.
The multigrid chains are the defaults of filt_cinv.py:112-116 (T) and :236-239 (P) with the stage-0 iteration count set;
the dense coarse preconditioner is built (and cached) by a first apply_ivf outside the timed region.

    python tools/cg_bench.py [nside] [lmax] [iters]        (CG_BENCH_ONLY=t|p, CG_BENCH_JOINT=1 for cinv_tp)
bench.py imports run() for the `cg` block of its JSON line."""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# FP64 floor of one top-level iteration of the default chains at nside = lmax = 2048 (78.6 TFLOP/s):
# T 1.83e11 flop, P 5.68e11 flop, T + P 7.51e11 flop = 9.6 ms
FLOP_PER_ITER_2048 = {'t': 1.83e11, 'p': 5.68e11}
REPS = int(os.environ.get('CG_BENCH_REPS', '3'))  # timed solves per filter at B = 1 (median reported)


def make_mask(nside, rng):
    from plancklens_amd import hp
    x, y, z = hp.pix2vec(nside)
    mask = (np.abs(z) > np.sin(np.radians(20.))).astype(float)
    cen = rng.standard_normal((200, 3))
    cen /= np.linalg.norm(cen, axis=1)[:, None]
    vec = np.stack([x, y, z])
    for c in cen:
        mask[(c @ vec) > np.cos(np.radians(1.))] = 0.
    return mask


def inputs(nside, lmax, alm2map, alm2map_spin):
    """The workload of BASELINE config 4 as a recipe with the transforms as arguments: this benchmark makes it with the product's SHTs, the
    golden generator (tests/golden/make_golden.py cinv2048, the reference's own cinv_t / cinv_p at this size) with the oracle's.  One seeded
    stream: mask discs, T sky, T noise, E sky, B sky, Q noise, U noise."""
    from plancklens_amd import hp, utils
    rng = np.random.default_rng(7)
    npix = hp.nside2npix(nside)
    cl = utils.camb_clfile(os.path.join(ROOT, 'plancklens_amd', 'data', 'cls', 'FFP10_wdipole_lensedCls.dat'), lmax=lmax)
    transf = hp.gauss_beam(5. / 60. / 180. * np.pi, lmax=lmax)
    nlev_t, nlev_p = 35., 55.
    mask = make_mask(nside, rng)
    vamin = np.sqrt(hp.nside2pixarea(nside, degrees=True)) * 60
    tmap = np.asarray(alm2map(hp.almxfl(hp.synalm(cl['tt'], lmax, rng), transf), nside)) + nlev_t / vamin * rng.standard_normal(npix)
    elm = hp.almxfl(hp.synalm(cl['ee'], lmax, rng), transf)
    blm = hp.almxfl(hp.synalm(cl['bb'], lmax, rng), transf)
    q, u = (np.asarray(x) for x in alm2map_spin([elm, blm], nside, 2, lmax))
    q = q + nlev_p / vamin * rng.standard_normal(npix)
    u = u + nlev_p / vamin * rng.standard_normal(npix)
    return {'cl': cl, 'transf': transf, 'mask': mask, 'tmap': tmap, 'qmap': q, 'umap': u, 'nlev_t': nlev_t, 'nlev_p': nlev_p, 'vamin': vamin,
            'ninv_t': [np.array([3. / nlev_t ** 2]) * mask], 'ninv_p': [[np.array([3. / nlev_p ** 2]) * mask]]}


def chain(kind, n, lmax, nside, pcf, cd_solve=None):
    """default chains of filt_cinv.py:112-116 (T) / :236-239 (P) with the stage-0 iteration count n and eps_min = 0; cd_solve: the module whose
    tr_cg / cache_mem the entries carry (the golden generator passes the reference's)"""
    if cd_solve is None:
        from plancklens_amd.qcinv import cd_solve
    if kind in ('t', 'tp'):
        return [[3, ["split(dense(" + pcf + "), 64, diag_cl)"], 256, 128, 3, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()],
                [2, ["split(stage(3),  256, diag_cl)"], 512, 256, 3, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()],
                [1, ["split(stage(2),  512, diag_cl)"], 1024, 512, 3, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()],
                [0, ["split(stage(1), 1024, diag_cl)"], lmax, nside, n, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()]]
    return [[2, ["split(dense(" + pcf + "), 32, diag_cl)"], 512, 256, 3, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()],
            [1, ["split(stage(2),  512, diag_cl)"], 1024, 512, 3, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()],
            [0, ["split(stage(1), 1024, diag_cl)"], lmax, nside, n, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()]]


def run(nside=2048, lmax=2048, iters=100, kinds=('t', 'p'), joint=False, verbose=False, peak_tflops=78.6, batches=()):
    """verbose=False: nothing reaches stdout (the filter classes print their set-up like the reference does; bench.py must
    print exactly one JSON line)."""
    if not verbose:
        stdout = sys.stdout
        sys.stdout = open(os.devnull, 'w')
        try:
            return _run(nside, lmax, iters, kinds, joint, False, peak_tflops, batches)
        finally:
            sys.stdout.close()
            sys.stdout = stdout
    return _run(nside, lmax, iters, kinds, joint, True, peak_tflops, batches)


def _run(nside, lmax, iters, kinds, joint, verbose, peak_tflops, batches=()):
    """batches: block sizes B > 1 to time as well -- B simulations (different data maps, same noise model) filtered in ONE block
    solve (cinv_*.apply_ivf_batch); reported per simulation: iterations/s of one solve x B."""
    import torch
    from plancklens_amd import dev, hp, shts, utils
    from plancklens_amd.filt import filt_cinv
    if os.environ.get('CG_BENCH_PLAN_OPTS'):  # development aid: pl_plan_opts for every plan of the run, e.g. fft_generic_nside=0
        shts.plan_options(**{kv.split('=')[0]: int(kv.split('=')[1]) for kv in os.environ['CG_BENCH_PLAN_OPTS'].split(',')}).__enter__()
    d = inputs(nside, lmax, shts.alm2map, shts.alm2map_spin)
    cl, transf, mask, tmap, q, u = d['cl'], d['transf'], d['mask'], d['tmap'], d['qmap'], d['umap']
    nlev_t, nlev_p, vamin = d['nlev_t'], d['nlev_p'], d['vamin']
    fsky = float(mask.mean())
    tmp = tempfile.mkdtemp(prefix='cgbench_')
    ninv_t, ninv_p = d['ninv_t'], d['ninv_p']
    res = {'fsky': fsky, 'iters': iters}

    def timed(f, dmap):
        if True:
            t0 = time.time()
            f.apply_ivf(dmap)  # builds the dense preconditioner (cached afterwards) and warms everything, graph capture included
            torch.cuda.synchronize()
            setup = time.time() - t0
            trace = []
            log0 = f.chain.log
            f.chain.log = lambda stage, it, eps, **kw: (trace.append(float(eps)) if stage.depth == 0 else None, log0(stage, it, eps, **kw))
            torch.cuda.synchronize()
            bdir = os.environ.get('CG_BENCH_BARRIER_DIR')  # concurrency probe: wait until the other process is here too
            if bdir:
                open(os.path.join(bdir, 'ready_%d' % os.getpid()), 'w').close()
                while len([x for x in os.listdir(bdir) if x.startswith('ready_')]) < int(os.environ.get('CG_BENCH_BARRIER_N', '2')):
                    time.sleep(0.001)
            dts = []
            for rep in range(REPS):  # REPS solves of `iters` iterations each, timed one by one: the median is reported (solves differ by +-3 %)
                t0 = time.time()
                f.apply_ivf(dmap)
                torch.cuda.synchronize()
                dts.append(time.time() - t0)
                if rep == 0:
                    trace = trace[:]  # the residual trace of the first timed solve
                    f.chain.log = log0
            dt = float(np.median(dts))
        return dt, setup, trace, dts

    filters, dmaps_of = {}, {}
    for kind in kinds:
        pcf = os.path.join(tmp, 'dense_%s.pk' % kind)
        if kind == 't':
            f = filt_cinv.cinv_t(os.path.join(tmp, 'cinv_t'), lmax, nside, cl, transf, ninv_t, chain_descr=chain('t', iters, lmax, nside, pcf))
            dmap = dev.to_dev(tmap)
        else:
            f = filt_cinv.cinv_p(os.path.join(tmp, 'cinv_p'), lmax, nside, cl, transf, ninv_p, chain_descr=chain('p', iters, lmax, nside, pcf))
            dmap = [dev.to_dev(q), dev.to_dev(u)]
        filters[kind], dmaps_of[kind] = f, dmap
        torch.cuda.synchronize()
        t0 = time.time()
        f.chain.instantiate()  # parses the chain: builds (or loads) the dense coarse preconditioner
        torch.cuda.synchronize()
        t_chain = time.time() - t0
        f.chain.plogdepth = -1
        dt, setup, trace, dts = timed(f, dmap)
        res[kind] = {'seconds': dt, 'seconds_each_solve': dts, 'iters_per_s': iters / dt, 'ms_per_iter': 1e3 * dt / iters,
                     'first_call_incl_dense_setup_s': setup + t_chain,
                     'chain_setup_s': t_chain,
                     'eps_first_last': [trace[0], trace[-1]] if trace else None}
        if nside == 2048 and lmax == 2048:
            res[kind]['frac_of_fp64_floor'] = FLOP_PER_ITER_2048[kind] / peak_tflops / 1e12 / (dt / iters)
        if verbose:
            print(kind, json.dumps(res[kind]), flush=True)
        for B in batches:
            if B <= 1:
                continue
            # B different data maps: the resident map scaled and point-reflected copies of it with other noise draws would do as well;
            # what matters to the timing is that every entry is a full right-hand side of its own
            gen = torch.Generator(device='cuda')
            gen.manual_seed(100 + B)
            if kind == 't':
                dmaps = [dmap] + [dmap + torch.randn(dmap.shape, generator=gen, dtype=torch.float64, device='cuda') * (nlev_t / vamin) for _ in range(B - 1)]
            else:
                dmaps = [dmap] + [[c + torch.randn(c.shape, generator=gen, dtype=torch.float64, device='cuda') * (nlev_p / vamin) for c in dmap] for _ in range(B - 1)]
            f.apply_ivf_batch(dmaps)  # warm-up: graph capture of the nested stages for this block size
            torch.cuda.synchronize()
            t0 = time.time()
            out = f.apply_ivf_batch(dmaps)
            torch.cuda.synchronize()
            dtb = time.time() - t0
            first = out[0] if kind == 't' else out[0][0]
            res.setdefault('batched', {}).setdefault(kind, {})[str(B)] = {
                'seconds': dtb, 'ms_per_block_iter': 1e3 * dtb / iters, 'iters_per_s_per_sim': B * iters / dtb,
                'speedup_per_sim_vs_B1': (B * iters / dtb) / (iters / dt)}
            del out, dmaps, first
            if verbose:
                print(kind, 'B =', B, json.dumps(res['batched'][kind][str(B)]), flush=True)
    if joint:
        pcf = os.path.join(tmp, 'dense_tp.pk')
        cl_tp = {k: cl[k] for k in ['tt', 'ee', 'bb', 'te']}
        f = filt_cinv.cinv_tp(os.path.join(tmp, 'cinv_tp'), lmax, nside, cl_tp, transf, [ninv_t[0], ninv_p[0][0]], marge_monopole=True,
                              marge_dipole=True, chain_descr=chain('tp', iters, lmax, nside, pcf))
        f.chain.plogdepth = -1
        dt, setup, _, _ = timed(f, [dev.to_dev(tmap), dev.to_dev(q), dev.to_dev(u)])
        res['tp_joint'] = {'seconds': dt, 'iters_per_s': iters / dt, 'first_call_incl_dense_setup_s': setup}
        if verbose:
            print('tp_joint', json.dumps(res['tp_joint']), flush=True)
    if 't' in res and 'p' in res and os.environ.get('CG_BENCH_CONCURRENT', '1') != '0':
        # cinv_t and cinv_p of the simulation at the same time on two streams of this process (filt_cinv.apply_ivf_tp): the form
        # library_cinv_sepTP.filter_sims runs.  One "T + P iteration" = one top-level iteration of each solve.
        try:
            ft, fp = filters['t'], filters['p']
            filt_cinv.apply_ivf_tp(ft, dmaps_of['t'], fp, dmaps_of['p'])  # first call: one after the other, in the contexts of the concurrent form
            torch.cuda.synchronize()
            dts = []
            for rep in range(REPS):
                t0 = time.time()
                filt_cinv.apply_ivf_tp(ft, dmaps_of['t'], fp, dmaps_of['p'])
                torch.cuda.synchronize()
                dts.append(time.time() - t0)
            dtc = float(np.median(dts))
            res['tp_concurrent'] = {'seconds': dtc, 'seconds_each_solve': dts, 'iters_per_s': iters / dtc, 'ms_per_iter': 1e3 * dtc / iters,
                                    'speedup_vs_one_after_the_other': (res['t']['seconds'] + res['p']['seconds']) / dtc}
            if nside == 2048 and lmax == 2048:
                res['tp_concurrent']['frac_of_fp64_floor'] = (FLOP_PER_ITER_2048['t'] + FLOP_PER_ITER_2048['p']) / peak_tflops / 1e12 / (dtc / iters)
            # the two solves' streams on different hardware queues (measured: dev.streams_overlap)?  On one queue they run one after the other
            res['tp_concurrent']['streams_overlap'] = bool(dev.streams_overlap(torch.cuda.current_stream(), filt_cinv._tp_side_stream()))
            if os.environ.get('CG_BENCH_QUEUE_DIAG'):  # which streams run beside which, here and now (stderr)
                names = ['current', 'tp_side'] + ['new%d' % i for i in range(4)]
                strs = [torch.cuda.current_stream(), filt_cinv._tp_side_stream()] + [torch.cuda.Stream() for _ in range(4)]
                from plancklens_amd import _lib as _l
                for ctx, tag in ((0, 'T'), (filt_cinv.P_CONTEXT, 'P')):  # the ring-FFT side streams of the fine-level plans of the two solves
                    with shts.plan_context(ctx):
                        ph = shts.get_plan(nside, lmax).h
                    for i in range(3):
                        ptr = _l.lib().pl_plan_side_stream(ph, i)
                        if ptr:
                            names.append('%s.s%d' % (tag, i))
                            strs.append(torch.cuda.ExternalStream(ptr))
                for i in range(len(strs)):
                    sys.stderr.write('%-8s %s\n' % (names[i], ' '.join('.' if j == i else ('1' if dev.streams_overlap(strs[i], strs[j]) else '0') for j in range(len(strs)))))
            # block solves of B simulations, T block and P block at the same time (what filter_sims runs with its default batch)
            for B in batches:
                if B < 4:  # (B = 2 adds little to the picture and 8 s to the run)
                    continue
                gen = torch.Generator(device='cuda')
                gen.manual_seed(200 + B)
                tm = [dmaps_of['t']] + [dmaps_of['t'] + torch.randn(dmaps_of['t'].shape, generator=gen, dtype=torch.float64, device='cuda') * (nlev_t / vamin)
                                        for _ in range(B - 1)]
                pm = [dmaps_of['p']] + [[c + torch.randn(c.shape, generator=gen, dtype=torch.float64, device='cuda') * (nlev_p / vamin) for c in dmaps_of['p']]
                                        for _ in range(B - 1)]
                filt_cinv.apply_ivf_tp(ft, tm, fp, pm)  # contexts / graphs of this block shape
                torch.cuda.synchronize()
                t0 = time.time()
                filt_cinv.apply_ivf_tp(ft, tm, fp, pm)
                torch.cuda.synchronize()
                dtb = time.time() - t0
                e = {'seconds': dtb, 'iters_per_s_per_sim': B * iters / dtb}
                if nside == 2048 and lmax == 2048:
                    e['frac_of_fp64_floor'] = (FLOP_PER_ITER_2048['t'] + FLOP_PER_ITER_2048['p']) / peak_tflops / 1e12 / (dtb / iters / B)
                res['tp_concurrent'].setdefault('block_solves', {})[str(B)] = e
                del tm, pm
        except Exception as e:  # the sequential figures stand on their own
            res['tp_concurrent'] = dict(res.get('tp_concurrent', {}), error=repr(e))
        if verbose:
            print('tp_concurrent', json.dumps(res['tp_concurrent']), flush=True)
    if 't' in res and 'p' in res:
        tot = res['t']['seconds'] + res['p']['seconds']
        res['tp'] = {'iters_per_s': iters / tot, 'ms_per_iter': 1e3 * tot / iters}
        if nside == 2048 and lmax == 2048:
            floor_ms = 1e3 * (FLOP_PER_ITER_2048['t'] + FLOP_PER_ITER_2048['p']) / peak_tflops / 1e12
            res['tp']['fp64_floor_ms_per_iter'] = floor_ms
            res['tp']['frac_of_fp64_floor'] = floor_ms / res['tp']['ms_per_iter']
        for B in batches:
            bt, bp = res.get('batched', {}).get('t', {}).get(str(B)), res.get('batched', {}).get('p', {}).get(str(B))
            if bt and bp:
                tot_b = bt['seconds'] + bp['seconds']
                e = {'iters_per_s_per_sim': B * iters / tot_b, 'speedup_per_sim_vs_B1': (B * iters / tot_b) / (iters / tot)}
                if nside == 2048 and lmax == 2048:
                    e['frac_of_fp64_floor'] = floor_ms / (1e3 * tot_b / iters / B)
                res['batched'].setdefault('tp', {})[str(B)] = e
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return res


if __name__ == '__main__':
    nside = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    lmax = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    only = os.environ.get('CG_BENCH_ONLY', 'tp')
    batches = [int(b) for b in os.environ.get('CG_BENCH_BATCHES', '2,4,8').split(',') if b.strip()]
    r = run(nside, lmax, iters, kinds=[k for k in ('t', 'p') if k in only], joint=os.environ.get('CG_BENCH_JOINT', '0') == '1', verbose=True,
            batches=batches)
    print(json.dumps({'metric': 'CG-iter/sec (cinv_t + cinv_p, nside=%d lmax=%d, masked fsky=%.2f, %d iterations)' % (nside, lmax, r['fsky'], iters),
                      'T_iters_per_s': r.get('t', {}).get('iters_per_s'), 'P_iters_per_s': r.get('p', {}).get('iters_per_s'),
                      'TP_iters_per_s': r.get('tp', {}).get('iters_per_s'), 'TP_frac_of_fp64_floor': r.get('tp', {}).get('frac_of_fp64_floor'),
                      'TP_concurrent': r.get('tp_concurrent'),
                      'TP_joint_iters_per_s': r.get('tp_joint', {}).get('iters_per_s'), 'batched': r.get('batched')}))
