"""Headline benchmark: minimum-variance ('p') lensing quadratic-estimator reconstructions per second at
nside = lmax = lmax_qlm = 2048 on MI355X (BASELINE.json metric).

One step = one reconstruction = T, Q, U maps (resident in HBM) -> isotropic inverse-variance filter
(filt_simple.py:397-407) -> qest.library_sepTP.get_sim_qlm('p') -> gradient + curl alm copied to host memory.
9 spherical harmonic transforms per step (2 scalar + 7 spin-weighted pairs), all FP64.

    python bench.py --gpus N --steps K --warmup W
For N > 1 it is launched by torch.distributed.run, one rank per GPU: every rank reconstructs its own
simulations (weak scaling: jobs[rank::size] of run_qlms.py:72) and the ranks meet in one RCCL all-reduce of the
mean-field sum and one all-gather of the last gradient alm inside the timed region.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (spin-weighted Legendre synthesis):
achieved = algorithmic flops per launch (24 flop per (l, m, ring pair): SURVEY.md 8(d)) / mean launch
duration measured with HIP events on the launch stream over the timed region.  The binding ceiling of that
kernel is FP64 vector-FMA issue, reported under the "mfma" (TFLOP/s) arm of the schema: gfx950's FP64 MFMA peak
equals its FP64 vector peak and no MFMA is used (a recurrence, not a contraction).  `cpu_baseline` times the
CPU oracle (oracle/, "port") on a bounded ring sample of the same workload on the host cores of this box.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X datasheet FP64 vector (= matrix) peak; SURVEY.md 8(d)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec
# k_leg_synths, spin 2, nside = lmax = 2048: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), bytes per launch
# (profiles/round1_h_pmc_traffic.csv; refreshed whenever the kernel changes materially)
SYNTHS_TRAFFIC_BYTES = (430280 + 373984) * 1024


def fma_ceilings():
    """Sustained v_fma_f64 rates of this GPU measured live (pl_fma64_rate_tflops: 16 independent chains per lane, no memory):
    the datasheet peak is not reachable in steady state, and the rate depends on how many vector sources an FMA reads."""
    from plancklens_amd import _lib
    L = _lib.lib()
    names = {1: 'two_scalar_sources (synthesis mix)', 0: 'one_vector_one_scalar', 2: 'three_vector_sources (analysis mix)'}
    return {names[m]: float(L.pl_fma64_rate_tflops(m, 20000, None)) for m in (1, 0, 2)}


def executed_flops(nside, lmax, spin):
    """Flops of the (l, m, ring pair) recursion steps a Legendre kernel actually runs: rings with m > mlim(theta) are
    pruned (same rule as the kernels, csrc/tables.cpp mlim_ring).  spin >= 1: 12 FMA per step; spin 0: the two-step
    recursion does 6 FMA per pair of l."""
    from plancklens_amd import hp
    cth, sth, _, _, _ = hp.ring_info(nside)
    cth, sth = cth[:2 * nside], sth[:2 * nside]           # north member of every ring pair (equator included)
    ofs = max(100., 0.01 * lmax)
    b = -2. * spin * np.abs(cth)
    t1 = lmax * sth + ofs
    disc = b * b - 4. * (spin * spin - t1 * t1)
    ml = np.where(disc <= 0, lmax, np.minimum((-b + np.sqrt(np.maximum(disc, 0.))) / 2., lmax))
    ml = np.minimum(np.floor(ml + 0.5).astype(np.int64), lmax)
    m = np.arange(lmax + 1)
    nl = np.cumsum(lmax - np.maximum(m, spin) + 1)        # l-steps for all m <= M
    steps = float(np.sum(nl[ml]))
    return steps * (24. if spin else 6.)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--nside', type=int, default=2048)
    ap.add_argument('--lmax', type=int, default=2048)
    ap.add_argument('--key', type=str, default='p')
    ap.add_argument('--lmax-qlm', type=int, default=None, help='band-limit of the output qlm (default: lmax)')
    ap.add_argument('--qe-only', action='store_true',
                    help='time the estimator from filtered alms that are already resident (the cost of a further key in the reference, qest.py:184-185)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=20.0)
    return ap.parse_args()


class resident_sims(object):
    """Synthetic T, Q, U maps held in HBM: Gaussian T/E/B sky with TE correlation (FFP10 lensed spectra) x 5'
    beam + white noise (35 / 55 muK-arcmin), SURVEY.md 8(d).  The same maps serve every simulation index."""

    def __init__(self, nside, lmax, cls, transf, nlev_t, nlev_p, seed):
        import torch
        from plancklens_amd import dev, hp, shts
        rng = np.random.default_rng(seed)
        n = hp.Alm.getsize(lmax)

        def unit():
            a = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) / np.sqrt(2.)
            a[:lmax + 1] = np.sqrt(2.) * a[:lmax + 1].real
            return a
        u1, u2, u3 = unit(), unit(), unit()
        tt, ee, bb, te = (cls[k][:lmax + 1] for k in ['tt', 'ee', 'bb', 'te'])
        r = te * np.where(tt > 0, 1. / np.sqrt(np.where(tt > 0, tt, 1.)), 0.)
        tlm = hp.almxfl(u1, np.sqrt(tt))
        elm = hp.almxfl(u1, r) + hp.almxfl(u2, np.sqrt(np.maximum(ee - r ** 2, 0.)))
        blm = hp.almxfl(u3, np.sqrt(bb))
        gen = torch.Generator(device='cuda')
        gen.manual_seed(seed)
        vamin = np.sqrt(hp.nside2pixarea(nside, degrees=True)) * 60
        npix = hp.nside2npix(nside)
        self.tmap = shts.alm2map(dev.to_dev(tlm), nside, fl=transf) + nlev_t / vamin * torch.randn(npix, generator=gen, dtype=torch.float64, device='cuda')
        q, u = shts.alm2map_spin([dev.to_dev(elm), dev.to_dev(blm)], nside, 2, lmax, fl=transf)
        self.qmap = q + nlev_p / vamin * torch.randn(npix, generator=gen, dtype=torch.float64, device='cuda')
        self.umap = u + nlev_p / vamin * torch.randn(npix, generator=gen, dtype=torch.float64, device='cuda')
        self.seed = seed

    def hashdict(self):
        return {'resident_sims': self.seed}

    def get_sim_tmap(self, idx):
        return self.tmap

    def get_sim_pmap(self, idx):
        return self.qmap, self.umap


def cpu_baseline(nside, lmax, target_seconds):
    """The oracle's Legendre stage (C, OpenMP, all host cores) + numpy ring FFTs on every `stride`-th ring
    pair of each of the 9 SHTs of one 'p' reconstruction; time x stride = seconds per reconstruction."""
    from oracle import sht_oracle as so
    ncores = os.cpu_count() or 1
    c, s, pair, slots = so._pair_geometry(nside, True)
    rng = np.random.default_rng(5)
    nalm = so.alm_size(lmax)
    alm2 = (rng.standard_normal((2, nalm)) + 1j * rng.standard_normal((2, nalm)))
    synth = [0, 2, 3, 1, 1]   # Tb map, (Qb, Ub), spin-3 leg, spin-1 leg (P), spin-1 gradient leg (T)
    anal = [0, 2, 1, 1]       # T filter, P filter, two final spin-1 analyses (qest.py:318-322)

    def run(stride):
        sel = np.arange(0, 2 * nside, stride)
        cs, ss, ps = c[sel], s[sel], pair[sel]
        sl = np.full(2 * sel.size, -1, dtype=np.int64)
        sl[0::2] = slots[0::2][sel]
        sl[1::2] = slots[1::2][sel]
        maps = rng.standard_normal((2, 12 * nside ** 2))
        t0 = time.perf_counter()
        for spin in synth:
            nc = 1 if spin == 0 else 2
            ph = so.legendre(0, 1, spin, lmax, lmax, cs, ss, ps, alm=alm2[:nc], nthreads=ncores)
            for i in range(nc):
                so._phase2map(ph[i], nside, lmax, sl)
        for spin in anal:
            nc = 1 if spin == 0 else 2
            ph = np.stack([so._map2phase(maps[i], nside, lmax, sl) for i in range(nc)])
            so.legendre(1, 1, spin, lmax, lmax, cs, ss, ps, phase=ph, nthreads=ncores)
        return time.perf_counter() - t0

    stride = 64
    t = run(stride)  # calibration pass (also warms the library)
    est_full = t * stride
    stride = 1
    while est_full / stride > target_seconds and stride < 64:
        stride *= 2
    t = run(stride)
    sec_per_rec = t * stride
    return {'value': 1.0 / sec_per_rec, 'unit': 'reconstructions/s', 'cores': ncores, 'kind': 'port',
            'sample': "every %d-th ring pair (of %d) of each of the 9 SHTs (2 scalar + 7 spin pairs) of one 'p' reconstruction at nside=%d lmax=%d, "
                      "measured %.1f s x %d" % (stride, 2 * nside, nside, lmax, t, stride)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    assert torch.cuda.is_available(), 'bench.py needs the MI355X (no CPU path)'
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='nccl')
    from plancklens_amd.helpers import mpi
    mpi.rank, mpi.size = rank, world

    from plancklens_amd import dev, hp, qest, shts, utils
    from plancklens_amd.filt import filt_simple

    nside, lmax, key = args.nside, args.lmax, args.key
    lmax_qlm = lmax if args.lmax_qlm is None else args.lmax_qlm
    cl_len = utils.camb_clfile(os.path.join(ROOT, 'plancklens_amd', 'data', 'cls', 'FFP10_wdipole_lensedCls.dat'), lmax=lmax)
    transf = hp.gauss_beam(5. / 60. / 180. * np.pi, lmax=lmax)
    nlev_t, nlev_p, lmin_ivf = 35., 55., 100
    arcmin = np.pi / 180. / 60.
    ftl = utils.cli(cl_len['tt'] + (nlev_t * arcmin) ** 2 * utils.cli(transf ** 2))
    fel = utils.cli(cl_len['ee'] + (nlev_p * arcmin) ** 2 * utils.cli(transf ** 2))
    fbl = utils.cli(cl_len['bb'] + (nlev_p * arcmin) ** 2 * utils.cli(transf ** 2))
    for f in (ftl, fel, fbl):
        f[:min(lmin_ivf, lmax // 4)] = 0.

    sims = resident_sims(nside, lmax, cl_len, transf, nlev_t, nlev_p, seed=1000 + rank)
    tmp = tempfile.mkdtemp(prefix='plbench_r%d_' % rank)
    mpi_rank_saved = mpi.rank
    mpi.rank = 0  # every rank owns a private scratch directory: all of them create their hash files
    ivfs = filt_simple.library_fullsky_sepTP(os.path.join(tmp, 'ivfs'), sims, nside, transf, cl_len, ftl, fel, fbl, cache=False)
    qlms = qest.library_sepTP(os.path.join(tmp, 'qlms'), ivfs, ivfs, cl_len['te'], nside, lmax_qlm=lmax_qlm, cache=False)
    mpi.rank = mpi_rank_saved
    plan = shts.get_plan(nside, lmax)

    mf_sum = torch.zeros(hp.Alm.getsize(lmax_qlm), dtype=torch.complex128, device='cuda')
    state = {'idx': rank, 'last': None}

    def step():
        idx = state['idx']
        if not args.qe_only:
            state['idx'] += world                  # jobs[rank::size]
        G = qlms.get_sim_qlm(key, idx)             # host array: filter + QE + device-to-host copy
        C = qlms.get_sim_qlm('x' + key[1:], idx)   # curl comes out of the same evaluation
        qlms._mem.clear()
        if not args.qe_only:
            ivfs._dev_cache.clear()
        state['last'] = (G, C)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    plan.profile(True)
    plan.profile_read()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        if qlms._last_dev is not None:
            mf_sum += qlms._last_dev[0]            # running mean-field sum stays on the device
        else:
            mf_sum += dev.to_dev(state['last'][0])
    if world > 1:
        buf = torch.view_as_real(mf_sum)
        dist.all_reduce(buf)                                     # mean-field sum over ranks (RCCL)
        last = torch.view_as_real(dev.to_dev(state['last'][0]))
        gathered = [torch.empty_like(last) for _ in range(world)]
        dist.all_gather(gathered, last)                          # output qlm all-gather over xGMI
    sync_all()
    dt = time.perf_counter() - t0
    prof = plan.profile_read()
    plan.profile(False)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        nalm = hp.Alm.getsize(lmax)
        steps_leg = nalm * 2 * nside                      # (l, m, ring pair) recursion steps per transform
        flops_spin, flops_scal = 24.0 * steps_leg, 8.0 * steps_leg
        exec_spin, exec_scal = executed_flops(nside, lmax, 2), executed_flops(nside, lmax, 0)
        npix = hp.nside2npix(nside)
        ms, cnt = prof['leg_synths']
        res = {
            'metric': "QE reconstructions/sec at nside=%d lmax=%d ('%s' MV)" % (nside, lmax, key),
            'value': world * args.steps / dt, 'unit': 'reconstructions/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': "'%s' MV quadratic estimator from T,Q,U maps: isotropic filter + qest.library_sepTP, "
                                   "nside=%d lmax_ivf=%d lmax_qlm=%d, 9 SHTs (2 scalar + 7 spin pairs) per reconstruction (BASELINE.json headline config)"
                                   % (key, nside, lmax, lmax_qlm) +
                                   (' -- QE-ONLY variant: filtered alms already resident, the filter transforms are not timed' if args.qe_only else ''),
                       'nside': nside, 'lmax': lmax, 'lmax_qlm': lmax_qlm, 'key': key, 'sims_per_gpu': args.steps,
                       'parallelism': 'sim-sharded x%d' % world},
        }
        if cnt > 0:
            avg_ms = ms / cnt
            ach = flops_spin / (avg_ms * 1e-3) / 1e12
            res['roofline'] = {'bound': 'mfma', 'achieved': ach, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': ach / FP64_PEAK_TFLOPS, 'traffic': SYNTHS_TRAFFIC_BYTES, 'kernel': 'k_leg_synths (spin-weighted Legendre synthesis)',
                               'avg_launch_ms': avg_ms, 'launches': cnt,
                               'executed_tflops': exec_spin / (avg_ms * 1e-3) / 1e12,
                               'fma_issue_ceiling_measured_tflops': fma_ceilings(),
                               'frac_executed': exec_spin / (avg_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                               'note': 'FP64 vector-FMA issue bound (v_fma_f64); gfx950 FP64 MFMA peak = FP64 vector peak; no MFMA used. '
                                       'achieved = SURVEY 8(d) fixed-denominator count (24 flop x nalm x 2 nside, polar pruning not counted); '
                                       'executed_tflops counts only the (l, m, ring pair) steps the kernel runs after libsharp-style polar pruning; '
                                       'fma_issue_ceiling_measured_tflops is what a pure FMA loop sustains on this GPU. '
                                       'traffic = FETCH_SIZE + WRITE_SIZE bytes per launch from profiles/ (PMC passes of an earlier run of the same kernel; '
                                       'coefficient-table re-reads by the ring groups are served by L2 / Infinity Cache)'}
        per_kernel = {}
        # gradient-only synthesis (curl alm = 0, shts.alm2map_spin([G, None])): 8 recurrence + 8 accumulation flop per step;
        # paired synthesis (general + gradient-only input on one recursion, shts.alm2map_spin_pair): 8 recurrence + 16 + 8
        # accumulation flop per step (SURVEY's fixed count for the two transforms it replaces would be 48)
        alg = {'leg_synth0': flops_scal, 'leg_synths': flops_spin, 'leg_anal0': flops_scal, 'leg_anals': flops_spin,
               'leg_synths_grad': flops_spin * 16. / 24., 'leg_synths_pair': flops_spin * 32. / 24.}
        exe = {'leg_synth0': exec_scal, 'leg_synths': exec_spin, 'leg_anal0': exec_scal, 'leg_anals': exec_spin,
               'leg_synths_grad': exec_spin * 16. / 24., 'leg_synths_pair': exec_spin * 32. / 24.}
        for k, (m_, c_) in prof.items():
            if c_ == 0:
                continue
            ent = {'avg_ms': m_ / c_, 'launches': c_, 'share_of_step': m_ / (1e3 * dt)}
            if k in alg:
                ent['alg_tflops'] = alg[k] / (m_ / c_ * 1e-3) / 1e12
                ent['executed_tflops'] = exe[k] / (m_ / c_ * 1e-3) / 1e12
            else:  # ring FFT stage: algorithmic bytes = 8 npix + 32 nrings_pairs (mmax+1) per component
                ent['alg_gbs'] = (8.0 * npix + 32.0 * 2 * nside * (lmax + 1)) / (m_ / c_ * 1e-3) / 1e9
            per_kernel[k] = ent
        res['kernels'] = per_kernel
        # whole-reconstruction algorithmic traffic (counting rule of SURVEY.md 8(d))
        b_scal = 8.0 * npix + 16.0 * nalm
        b_spin = 2 * b_scal
        bytes_rec = 2 * b_scal + 7 * b_spin   # each map (8 npix B) and alm (16 nalm B) component touched once per transform
        res['hbm'] = {'algorithmic_GB_per_reconstruction': bytes_rec / 1e9,
                      'achieved_GBs': bytes_rec / 1e9 / (dt / args.steps), 'peak_GBs': HBM_PEAK_GBS,
                      'frac': bytes_rec / 1e9 / (dt / args.steps) / HBM_PEAK_GBS,
                      'note': 'path is FP64-FMA bound (arithmetic intensity ~190 flop/B): a low HBM fraction is a property of the algorithm'}
        if world == 1 and not args.no_cpu_baseline:
            try:
                res['cpu_baseline'] = cpu_baseline(nside, lmax, args.cpu_seconds)
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                res['cpu_baseline'] = {'value': None, 'unit': 'reconstructions/s', 'cores': os.cpu_count(), 'kind': 'port',
                                       'sample': 'failed: %r' % (e,)}
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()
