"""Headline benchmark: minimum-variance ('p') lensing quadratic-estimator reconstructions per second at
nside = lmax = lmax_qlm = 2048 on MI355X, and CG iterations per second of the qcinv Wiener filter (BASELINE.json metric;
SURVEY.md 8(d)).

One step = one reconstruction = T, Q, U maps (resident in HBM: `--resident-sets` distinct realisations, default 2, consecutive pairs of
simulations are served different ones) -> isotropic inverse-variance filter
(filt_simple.py:397-407) -> qest.library_sepTP.get_sim_qlm('p') -> gradient + curl alm copied to host memory.
9 spherical harmonic transforms per step (2 scalar + 7 spin-weighted pairs), all FP64.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher: this process touches no GPU and starts N rank processes of itself (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in their environment), one per GPU.  Under torch.distributed.run the ranks come from the launcher
and WORLD_SIZE must equal --gpus.  The timed region is the PRODUCT's mean-field evaluation over world x K simulations,
qest.library.get_sim_qlm_mf: every rank reconstructs its share jobs[rank::size] (K reconstructions per rank, weak
scaling, run_qlms.py:72), the running sum stays on the device and one RCCL all-reduce completes it
(plancklens_amd/parallel.py); then the last gradient alm of every rank is all-gathered over xGMI.  The library serves
its share two simulations at a time (the spin-2 and spin-3 leg syntheses of a pair share one Legendre recursion each,
pl_alm2map_batch2; PLENS_OPTIONS=batch2=0 evaluates them one by one): a step is still one reconstruction.

Prints ONE JSON line (rank 0).
 * `roofline`: the Legendre kernel with the largest summed time inside the timed region (HIP events on the launch
   stream, pl_profile_*).  The binding ceiling is FP64 vector-FMA issue, reported under the "mfma" (TFLOP/s) arm of the
   schema: gfx950's FP64 MFMA peak equals its FP64 vector peak and no MFMA is used (a recurrence, not a contraction).
   `achieved` / `frac` follow SURVEY.md 8(d): the ALGORITHMIC flop count of a launch (no credit for pruned rings or skipped steps) / its mean
   duration; `achieved_executed` / `frac_executed` count the flops the kernel really issues (pl_plan_executed_steps), `frac_of_measured_issue_ceiling`
   relates those to the FMA rate a pure loop of the same operand mix sustains on this GPU in this process.
 * `kernels`: every timed stage (per launch; ring-FFT GB/s per component).
 * `cg`: BASELINE config 4 (cinv_t + cinv_p, masked sky, 100 top-level iterations, dense preconditioner cached outside
   the timed region), rank 0 at N = 1 only.
 * `cpu_baseline`: the CPU oracle (oracle/, "port") on the host cores of this box, rank 0 at N = 1 only.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X datasheet FP64 vector (= matrix) peak; SURVEY.md 8(d)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6300.0  # MI355X_MICROARCH.md chip table: measured streaming rate
# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), bytes per launch at nside = lmax = 2048, by profile kind: read from
# the NEWEST profiles/round*_pmc_traffic.csv (tools/prof_round.sh writes one per evidence set; refreshed whenever a kernel changes
# materially).  FETCH_SIZE on gfx950 counts half of the bytes of 16-B-per-lane streaming reads and is uncalibrated for the scalar table
# streams of these kernels.
PMC_KERNEL_OF_KIND = {'leg_anals': 'k_leg_anals<', 'leg_synths': 'k_leg_synths<', 'leg_anal0': 'k_leg_anal0<', 'leg_synth0': 'k_leg_synth0<'}


def pmc_traffic():
    """({profile kind: FETCH_SIZE + WRITE_SIZE bytes per launch}, source file) from the newest PMC summary under profiles/ (by round and
    letter in the file name); ({}, None) when there is none."""
    import csv
    import glob
    import re
    def order(f):  # round number, then the letter of the evidence set
        m = re.match(r'round(\d+)_([a-z]+)_', os.path.basename(f))
        return (int(m.group(1)), m.group(2)) if m else (0, '')
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'round*_pmc_traffic.csv')), key=order)
    if not files:
        return {}, None
    out = {}
    with open(files[-1]) as fh:
        rows = list(csv.DictReader(line for line in fh if not line.startswith('#')))
    for kind, pat in PMC_KERNEL_OF_KIND.items():
        hit = [r for r in rows if pat in r['kernel'] and 'pair' not in r['kernel']]
        if hit:  # (the first match is the variant with the largest traffic: the file is sorted that way)
            out[kind] = int((float(hit[0]['FETCH_SIZE_KB']) + float(hit[0]['WRITE_SIZE_KB'])) * 1024)
    return out, os.path.relpath(files[-1], ROOT)


KERNEL_NAMES = {'leg_synth0': 'k_leg_synth0 (scalar Legendre synthesis)', 'leg_synths': 'k_leg_synths (spin-weighted Legendre synthesis)',
                'leg_anal0': 'k_leg_anal0 (scalar Legendre analysis)', 'leg_anals': 'k_leg_anals (spin-weighted Legendre analysis)',
                'leg_synths_grad': 'k_leg_synths<GONLY> (gradient-only spin synthesis)',
                'leg_synths_pair': 'k_leg_synths<IN2=1> (general + gradient-only spin-1 synthesis on one recursion)',
                'leg_synths_batch2': 'k_leg_synths<IN2=2> (the same spin synthesis of two simulations on one recursion)',
                'leg_synth0_pair': 'k_leg_synth0<IN2> (the scalar synthesis of two simulations on one recursion)'}


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


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None, help='timed reconstructions per GPU (default 10; 32 with --config 5: 256 simulations over 8 GPUs)')
    ap.add_argument('--config', type=int, default=None, choices=[1, 2, 3, 4, 5],
                    help="a BASELINE.json configuration by number: 1 = 'ptt' nside=lmax=512; 2 = 'ptt' 2048; 3 = 'p_p' 2048; 4 = the default line (MV 'p' at 2048 "
                         "with its CG block: config 4 is the `cg` object); 5 = MV 'p' nside=lmax=4096, 32 simulations per GPU (256 over 8 GPUs, sim-sharded + "
                         "RCCL all-reduce / all-gather), no CG / CPU legs.  Sets --nside / --lmax / --key (and --steps unless given)")
    ap.add_argument('--warmup', type=int, default=4, help='untimed reconstructions (the GPU needs ~0.1 s of work to reach its sustained clocks)')
    ap.add_argument('--nside', type=int, default=2048)
    ap.add_argument('--lmax', type=int, default=2048)
    ap.add_argument('--key', type=str, default='p')
    ap.add_argument('--lmax-qlm', type=int, default=None, help='band-limit of the output qlm (default: lmax)')
    ap.add_argument('--qe-only', action='store_true',
                    help='time the estimator from filtered alms that are already resident (the cost of a further key in the reference, qest.py:184-185)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-from-sims', action='store_true', help="skip the `from_sims` leg (reconstructions timed including the device-side generation of their input maps)")
    ap.add_argument('--cpu-seconds', type=float, default=75.0, help='budget of the CPU baseline (warm-up and all repetitions together)')
    ap.add_argument('--no-cg', action='store_true', help='skip the CG block (BASELINE config 4)')
    ap.add_argument('--cg-iters', type=int, default=100)
    ap.add_argument('--cg-batches', type=str, default='2,4,8', help='block sizes B > 1 of the CG block (simulations filtered together); empty: none')
    ap.add_argument('--sims-seed', type=int, default=None, help='seed of the resident input maps (default 1000 + rank: every rank its own sky)')
    ap.add_argument('--resident-sets', type=int, default=2, help='distinct T, Q, U realisations resident in HBM; consecutive pairs of simulations are served different ones')
    ap.add_argument('--plan-opt', action='append', default=[], metavar='NAME=VALUE',
                    help='pl_plan_opts field for every plan of the run (shts.plan_options), e.g. fft_legacy=1: a development aid, not a number to quote')
    ap.add_argument('--no-plan-stats', action='store_true', help='skip the nside-4096 plan-creation measurement (time and host memory of the table build)')
    args = ap.parse_args(argv)
    if args.config is not None:
        nside, key = {1: (512, 'ptt'), 2: (2048, 'ptt'), 3: (2048, 'p_p'), 4: (2048, 'p'), 5: (4096, 'p')}[args.config]
        args.nside = args.lmax = nside
        args.key = key
        if args.config == 5:
            args.no_cg = args.no_cpu_baseline = args.no_plan_stats = True
            if args.steps is None:
                args.steps = 32
    if args.steps is None:
        args.steps = 10
    return args


# ------------------------------------------------------------------------------------------------------------------
# launcher: N rank processes of this script, started by a parent that never touches a GPU
# ------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv):
    """Start n fresh rank processes of this script (never exec from a process that has initialised HIP: the parent imports
    neither torch nor the HIP library).  Rank 0 inherits stdout (the one JSON line); the other ranks' stdout goes to stderr."""
    port = os.environ.get('MASTER_PORT') or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else sys.stderr))
    # poll all ranks: one that dies leaves the others waiting in the rendezvous or a collective (up to 10 minutes) -- end them at once
    rc = 0
    live = list(procs)
    while live and not rc:
        time.sleep(0.2)
        for p in list(live):
            if p.poll() is not None:
                live.remove(p)
                rc = rc or p.returncode
    if rc:
        time.sleep(2.0)  # let the failing rank's siblings print their own errors, if any
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    return rc


class resident_sims(object):
    """Synthetic T, Q, U maps held in HBM: Gaussian T/E/B sky with TE correlation (FFP10 lensed spectra) x 5'
    beam + white noise (35 / 55 muK-arcmin), SURVEY.md 8(d).  `nsets` distinct realisations (seed, seed + 7919, ...) are resident; simulation
    idx is served set (idx // 2) % nsets, so that consecutive PAIRS of a mean-field loop -- the unit the estimator evaluates -- see different
    maps (round 6: a replayed pair reads its inputs through a table of device addresses, so this costs 8 bytes per map, not a copy)."""

    def __init__(self, nside, lmax, cls, transf, nlev_t, nlev_p, seed, nsets=2):
        self.seed, self.nsets = seed, max(1, int(nsets))
        self.sets = [self._make(nside, lmax, cls, transf, nlev_t, nlev_p, seed + 7919 * k) for k in range(self.nsets)]
        self.tmap, self.qmap, self.umap = self.sets[0]

    @staticmethod
    def _make(nside, lmax, cls, transf, nlev_t, nlev_p, seed):
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
        tmap = shts.alm2map(dev.to_dev(tlm), nside, fl=transf) + nlev_t / vamin * torch.randn(npix, generator=gen, dtype=torch.float64, device='cuda')
        q, u = shts.alm2map_spin([dev.to_dev(elm), dev.to_dev(blm)], nside, 2, lmax, fl=transf)
        qu = torch.empty((2, npix), dtype=torch.float64, device='cuda')  # (Q, U) as the two rows of one array: what map2alm_spin takes
        qu[0] = q + nlev_p / vamin * torch.randn(npix, generator=gen, dtype=torch.float64, device='cuda')
        qu[1] = u + nlev_p / vamin * torch.randn(npix, generator=gen, dtype=torch.float64, device='cuda')
        return tmap, qu[0], qu[1]

    # the tensors handed out are never modified: a consumer that keeps a copy of one (the static input slots of the estimator's replayed
    # graph on its slot route, options.opts.qe_indirect off) need not copy it again while storage, shape and version are unchanged
    stable_maps = True

    def hashdict(self):
        return {'resident_sims': self.seed, 'nsets': self.nsets}

    def _set(self, idx):
        return self.sets[(int(idx) // 2) % self.nsets]

    def get_sim_tmap(self, idx):
        return self._set(idx)[0]

    def get_sim_pmap(self, idx):
        s_ = self._set(idx)
        return s_[1], s_[2]


def usable_cpus():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU slot of a shared box sees all
    256 hardware threads in os.cpu_count() but is throttled to its quota: 256 OpenMP threads then run 10x slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(round(float(quota) / float(period)))))
    except (OSError, ValueError):
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, int(round(q / per))))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_baseline(nside, lmax, budget_seconds, reps=3):
    """One 'p' reconstruction's 9 SHTs with the CPU oracle, both stages in C with OpenMP over all host cores (Legendre stage
    threaded over m, ring FFTs over rings), maps preallocated.  Every ring pair of every transform (no extrapolation)
    whenever 1 warm-up + 1 repetition fit the budget -- then as many repetitions up to `reps` as fit, median reported;
    otherwise every stride-th ring pair with the time extrapolated through the measured fixed + per-ring cost model (stated
    in the result)."""
    from oracle import sht_oracle as so
    ncores = usable_cpus()
    c, s, pair, slots = so._pair_geometry(nside, True)
    rng = np.random.default_rng(5)
    nalm = so.alm_size(lmax)
    alm2 = (rng.standard_normal((2, nalm)) + 1j * rng.standard_normal((2, nalm)))
    maps = rng.standard_normal((2, 12 * nside ** 2))
    outm = np.zeros((2, 12 * nside ** 2))
    synth = [0, 2, 3, 1, 1]   # Tb map, (Qb, Ub), spin-3 leg, spin-1 leg (P), spin-1 gradient leg (T)
    anal = [0, 2, 1, 1]       # T filter, P filter, the two final spin-1 analyses of the reference (qest.py:318-322)

    bufs = {}

    def buf(name, shape):
        """result arrays of the C stages, allocated (and zeroed) once per shape: a fresh 270 MB array per call costs its page faults
        inside the threaded C loops -- the GPU path works in preallocated workspaces too"""
        if bufs.get(name) is None or bufs[name].shape != shape:
            bufs[name] = np.zeros(shape, dtype=np.complex128)
        return bufs[name]

    def run(stride):
        sel = np.arange(0, 2 * nside, stride)
        cs, ss, ps = c[sel], s[sel], pair[sel]
        sl = np.full(2 * sel.size, -1, dtype=np.int64)
        sl[0::2] = slots[0::2][sel]
        sl[1::2] = slots[1::2][sel]
        t0 = time.perf_counter()
        for spin in synth:
            nc = 1 if spin == 0 else 2
            ph = so.legendre(0, 1, spin, lmax, lmax, cs, ss, ps, alm=alm2[:nc], nthreads=ncores, out=buf('ph%d' % nc, (nc, sl.size, lmax + 1)))
            for i in range(nc):
                so.ring_fft_c(0, nside, lmax, sl, phase=ph[i], out=outm[i], nthreads=ncores)
        for spin in anal:
            nc = 1 if spin == 0 else 2
            ph = buf('ph%d' % nc, (nc, sl.size, lmax + 1))
            for i in range(nc):
                so.ring_fft_c(1, nside, lmax, sl, m=maps[i], out=ph[i], nthreads=ncores)
            so.legendre(1, 1, spin, lmax, lmax, cs, ss, ps, phase=ph, nthreads=ncores, out=buf('alm%d' % nc, (nc, nalm)))
        return time.perf_counter() - t0

    # cost model t(nrings) = fixed + per_ring * nrings from two sparse passes (the per-m table set-up of the Legendre stage
    # does not shrink with the ring sample)
    run(64)
    t64, t16 = run(64), run(16)
    n64, n16, nfull = len(range(0, 2 * nside, 64)), len(range(0, 2 * nside, 16)), 2 * nside
    per_ring = max((t16 - t64) / (n16 - n64), 0.)
    fixed = max(t64 - per_ring * n64, 0.)
    est_full = fixed + per_ring * nfull
    if 2 * est_full <= budget_seconds:
        stride, nrep = 1, int(max(1, min(reps, budget_seconds // est_full - 1)))
        run(1)  # warm-up at full size
    else:
        stride, nrep = 2, reps
        while (fixed + per_ring * nfull / stride) * (nrep + 1) > budget_seconds and stride < 64:
            stride *= 2
        run(stride)
    ts = sorted(run(stride) for _ in range(nrep))
    t = ts[len(ts) // 2]
    sec_per_rec = t if stride == 1 else fixed + max(t - fixed, 0.05 * t) * stride
    flop_rec = (2 * 8 + 7 * 24) * float(nalm) * 2 * nside  # 2 scalar + 7 spin-weighted transforms, SURVEY 8(d) fixed denominator
    sample = ("every ring pair (all %d), no extrapolation" % nfull if stride == 1 else
              "every %d-th ring pair (of %d), EXTRAPOLATED as fixed + (t - fixed) x %d with fixed = %.2f s measured from two sparse passes"
              % (stride, nfull, stride, fixed))
    return {'value': 1.0 / sec_per_rec, 'unit': 'reconstructions/s', 'cores': ncores, 'kind': 'port',
            'extrapolated_from_ring_stride': stride, 'repetitions': nrep, 'seconds_per_reconstruction': sec_per_rec,
            'gflops_per_core': flop_rec / sec_per_rec / ncores / 1e9,
            'sample': "%s, of each of the 9 SHTs of one 'p' reconstruction as the reference runs it (2 scalar + 7 spin-weighted pairs, "
                      "qest.py:318-322) at nside=%d lmax=%d; oracle Legendre stage and ring FFTs in C with OpenMP on %d threads (= usable CPUs: affinity mask capped by the cgroup quota; "
                      "os.cpu_count() = %d); "
                      "1 warm-up + %d repetition(s), median %.2f s (min %.2f, max %.2f) = %.1f GFLOP/s per core by the SURVEY 8(d) count "
                      "(8 / 24 flop per (l, m, ring pair): %.3g flop per reconstruction); Legendre accumulation vectorised over 8 ring pairs "
                      "(omp simd, -march=native), ring FFTs radix-2 / Bluestein with a ring and its mirror in one complex transform, result "
                      "arrays preallocated; a long-double-checked restatement, still a few times below a hand-tuned libsharp"
                      % (sample, nside, lmax, ncores, os.cpu_count() or 1, nrep, t, ts[0], ts[-1], flop_rec / sec_per_rec / ncores / 1e9, flop_rec)}


def _healpy_sequence(hp, nside, lmax):
    """(seconds, (G, C)) of one 'p' reconstruction as the reference runs it, on the module `hp` (healpy, or a stand-in with its signatures): the
    hp.* calls in the reference's order with numpy between them -- isotropic filter (filt_simple.py:397-407), the polarization and temperature
    estimators (qest.py:248-285 over lib_filt2map_sepTP.get_irespmap / get_gpmap(3) / get_gpmap(1) / get_irestmap / get_gtmap, :506-530,
    :566-638) and their sum (:318-322).  Inputs: seeded random maps, unit filters above l = 1, C^TE = 0.1 (the cost does not depend on the values)."""
    rng = np.random.default_rng(5)
    npix = 12 * nside ** 2
    tmap, qmap, umap = rng.standard_normal((3, npix))
    fl = np.ones(lmax + 1)
    fl[:2] = 0.
    clte = 0.1 * fl
    lw = -np.sqrt(np.arange(lmax + 1, dtype=float) * np.arange(1, lmax + 2))  # qest.py:260,282

    def gp_fl(spin):  # qest.py:494-501
        f = (np.arange(2, lmax + 3, dtype=float) * np.arange(-1, lmax)) if spin == 1 else (np.arange(-2, lmax - 1, dtype=float) * np.arange(3, lmax + 4))
        f[:spin] *= 0.
        return np.sqrt(f)

    t0 = time.perf_counter()
    tlm = hp.almxfl(hp.map2alm(tmap, lmax=lmax, iter=0), fl)                                  # filt_simple.py:397-401
    elm, blm = hp.map2alm_spin([qmap, umap], 2, lmax=lmax)                                    # :403-407
    elm, blm = hp.almxfl(elm, fl), hp.almxfl(blm, fl)
    # polarization estimator (qest.py:265-285)
    repmap, impmap = hp.alm2map_spin([elm * 0.5, blm * 0.5], nside, 2, lmax)                      # get_irespmap, qest.py:521-530
    ewf, bwf = hp.almxfl(elm, fl) + hp.almxfl(tlm, clte), hp.almxfl(blm, fl)                  # Wiener-filtered E (with C^TE T), B
    Gs, Cs = hp.alm2map_spin([hp.almxfl(ewf, gp_fl(3)), hp.almxfl(bwf, gp_fl(3))], nside, 3, lmax)  # get_gpmap(idx, 3), :597-638
    GC = (repmap - 1j * impmap) * (Gs + 1j * Cs)
    Gs, Cs = hp.alm2map_spin([hp.almxfl(ewf, gp_fl(1)), hp.almxfl(bwf, gp_fl(1))], nside, 1, lmax)  # get_gpmap(idx, 1)
    GC -= (repmap + 1j * impmap) * (Gs - 1j * Cs)
    del repmap, impmap, Gs, Cs
    GP, CP = hp.map2alm_spin([GC.real, GC.imag], 1, lmax=lmax)
    del GC
    hp.almxfl(GP, lw, inplace=True)
    hp.almxfl(CP, lw, inplace=True)
    # temperature estimator (qest.py:248-263)
    tb = hp.alm2map(tlm, nside, lmax=lmax)                                                    # get_irestmap, :506-514
    twf = hp.almxfl(tlm, fl) + hp.almxfl(elm, clte)
    G, C = hp.alm2map_spin([hp.almxfl(twf, lw), np.zeros_like(twf)], nside, 1, lmax)        # get_gtmap, :566-593
    G *= tb
    C *= tb
    GT, CT = hp.map2alm_spin([G, C], 1, lmax=lmax)
    hp.almxfl(GT, lw, inplace=True)
    hp.almxfl(CT, lw, inplace=True)
    out = GP + GT, CP + CT                                                                    # :318-322
    return time.perf_counter() - t0, out


def _cpu_baseline_healpy_estimate(hp, nside, lmax):
    return _healpy_sequence(hp, nside, lmax)[1]


def cpu_baseline_healpy(hp, nside, lmax, budget_seconds, reps=3):
    """BASELINE.md section 2 / SURVEY.md 8(d), first choice: if `import healpy` succeeds on the box, the reference path itself (_healpy_sequence)
    timed on the host cores.  OpenMP threads of healpy's libsharp: OMP_NUM_THREADS as found, reported."""
    ncores = usable_cpus()

    def one():
        return _healpy_sequence(hp, nside, lmax)

    t_first, _ = one()  # warm-up (also the cost estimate)
    nrep = int(max(1, min(reps, budget_seconds // max(t_first, 1e-9) - 1)))
    ts = sorted(one()[0] for _ in range(nrep))
    t = ts[len(ts) // 2]
    nalm = (lmax + 1) * (lmax + 2) // 2
    flop_rec = (2 * 8 + 7 * 24) * float(nalm) * 2 * nside
    return {'value': 1.0 / t, 'unit': 'reconstructions/s', 'cores': ncores, 'kind': 'healpy', 'repetitions': nrep, 'seconds_per_reconstruction': t,
            'gflops_per_core': flop_rec / t / ncores / 1e9, 'healpy_version': getattr(hp, '__version__', None),
            'omp_num_threads': os.environ.get('OMP_NUM_THREADS'),
            'sample': "the reference's own call sequence for one 'p' reconstruction on healpy %s (filt_simple.py:397-407 + qest.py:248-285,318-322: 2 scalar + "
                      "7 spin-weighted transforms, every ring, numpy between them) at nside=%d lmax=%d; 1 warm-up + %d repetition(s), median %.2f s "
                      "(min %.2f, max %.2f); %d usable CPUs, OMP_NUM_THREADS=%s" % (getattr(hp, '__version__', '?'), nside, lmax, nrep, t, ts[0], ts[-1], ncores,
                                                                                  os.environ.get('OMP_NUM_THREADS'))}


def cpu_baseline_any(nside, lmax, budget_seconds):
    """healpy if it can be imported on this box (the reference path itself, kind 'healpy'), the in-repo restatement otherwise (kind 'port')"""
    try:
        import healpy
    except Exception:
        healpy = None
    if healpy is not None:
        try:
            return cpu_baseline_healpy(healpy, nside, lmax, budget_seconds)
        except Exception as e:  # a healpy that imports but cannot run the sequence: say so, fall back
            res = cpu_baseline(nside, lmax, budget_seconds)
            res['healpy_error'] = repr(e)
            return res
    return cpu_baseline(nside, lmax, budget_seconds)


def stub_rank(args, rank, world):
    """Launcher self-test (PLBENCH_STUB=1, tests/test_bench_launcher.py): gloo on CPU, no GPU work, no number."""
    import torch
    import torch.distributed as dist
    if os.environ.get('PLBENCH_STUB_FAIL_RANK') == str(rank):  # a rank that dies before the rendezvous (launcher test)
        sys.stderr.write('bench.py stub: rank %d fails on request\n' % rank)
        return 3
    if world > 1:
        dist.init_process_group(backend='gloo')
    for _ in range(args.warmup + args.steps):
        time.sleep(0.001)
    seen = torch.ones(1)
    if world > 1:
        dist.all_reduce(seen)
        dist.barrier()
    if rank == 0:
        print(json.dumps({'metric': 'launcher self-test', 'value': None, 'unit': 'reconstructions/s', 'n_gpus': world, 'steps': args.steps,
                          'warmup': args.warmup, 'ranks_seen': int(seen.item()), 'data': 'STUB (launcher self-test, no GPU work)'}))
    if world > 1:
        dist.destroy_process_group()
    return 0


def run_rank(args):
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        sys.stderr.write('bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node equal to --gpus\n' % (args.gpus, world))
        return 2
    if os.environ.get('PLBENCH_STUB', '0') == '1':
        return stub_rank(args, rank, world)
    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available(), 'bench.py needs the MI355X (no CPU path)'
    # PLENS_DIST_BACKEND=gloo (tests): the ranks may share a GPU, collectives are staged through the host (plancklens_amd.parallel);
    # the default -- and the only form a number is quoted from -- is nccl = RCCL with one GPU per rank
    backend = os.environ.get('PLENS_DIST_BACKEND') or 'nccl'
    ndev = torch.cuda.device_count()  # (device_count does not initialise the GPU)
    if local_rank >= ndev and backend == 'nccl':
        sys.stderr.write('bench.py: rank %d of %d has no GPU of its own (%d visible): one rank per GPU\n' % (rank, world, ndev))
        return 2
    torch.cuda.set_device(local_rank % max(1, ndev))
    use_dist = world > 1 or ('RANK' in os.environ and os.environ.get('PLENS_DIST_FORCE', '0') == '1')
    if use_dist:  # (PLENS_DIST_FORCE=1 under a launcher: RCCL collectives with a single rank, tests/test_gpu_bench.py)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend=backend)
    from plancklens_amd.helpers import mpi
    mpi.rank, mpi.size = rank, world

    from plancklens_amd import dev, hp, options, parallel, qest, shts, utils
    from plancklens_amd.filt import filt_simple

    nside, lmax, key = args.nside, args.lmax, args.key
    lmax_qlm = lmax if args.lmax_qlm is None else args.lmax_qlm
    if args.plan_opt:
        shts.plan_options(**{kv.split('=')[0]: int(kv.split('=')[1]) for kv in args.plan_opt}).__enter__()  # (for the life of the process)
    cl_len = utils.camb_clfile(os.path.join(ROOT, 'plancklens_amd', 'data', 'cls', 'FFP10_wdipole_lensedCls.dat'), lmax=lmax)
    transf = hp.gauss_beam(5. / 60. / 180. * np.pi, lmax=lmax)
    nlev_t, nlev_p, lmin_ivf = 35., 55., 100
    arcmin = np.pi / 180. / 60.
    ftl = utils.cli(cl_len['tt'] + (nlev_t * arcmin) ** 2 * utils.cli(transf ** 2))
    fel = utils.cli(cl_len['ee'] + (nlev_p * arcmin) ** 2 * utils.cli(transf ** 2))
    fbl = utils.cli(cl_len['bb'] + (nlev_p * arcmin) ** 2 * utils.cli(transf ** 2))
    for f in (ftl, fel, fbl):
        f[:min(lmin_ivf, lmax // 4)] = 0.

    t_plan0 = time.perf_counter()
    shts.get_plan(nside, lmax)  # geometry, recursion and ring-FFT tables of the benchmarked grid (built on the host, uploaded once)
    plan_create_s = time.perf_counter() - t_plan0
    sims = resident_sims(nside, lmax, cl_len, transf, nlev_t, nlev_p, seed=(1000 + rank) if args.sims_seed is None else args.sims_seed, nsets=args.resident_sets)
    tmp = tempfile.mkdtemp(prefix='plbench_r%d_' % rank)
    mpi.rank = 0  # every rank owns a private scratch directory: all of them create their hash files
    mpi.size = 1
    try:
        ivfs = filt_simple.library_fullsky_sepTP(os.path.join(tmp, 'ivfs'), sims, nside, transf, cl_len, ftl, fel, fbl, cache=False)
        qlms = qest.library_sepTP(os.path.join(tmp, 'qlms'), ivfs, ivfs, cl_len['te'], nside, lmax_qlm=lmax_qlm, cache=False)
    finally:
        mpi.rank, mpi.size = rank, world
    plan = shts.get_plan(nside, lmax)

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def reduce_max(vals):
        """max over ranks of a few floats (device tensors on nccl, host tensors on a CPU backend)"""
        tt = torch.tensor(list(vals), dtype=torch.float64, device='cuda' if backend == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return [float(x) for x in tt.tolist()]

    K = args.steps
    # set-up, not a step: the paired-simulation transforms are run once on zeros so that their workspaces (four-component phase
    # array, second coefficient array) exist and their code objects are loaded before anything is timed
    if key == 'p' and options.opts.batch2:
        z = torch.zeros((2, hp.Alm.getsize(lmax)), dtype=torch.complex128, device='cuda')
        for spin in (2, 3):
            shts.alm2map_spin_batch2([z[0], z[1]], [z[0], z[1]], nside, spin, lmax)
        del z
    # warm-up: W reconstructions per rank on indices outside the timed set, through the same (paired) route as the timed ones
    qlms.get_sim_qlms(key, [10 ** 6 + world * w + rank for w in range(args.warmup)])
    # set-up, not steps: the library captures a pair of reconstructions into a HIP graph after `graph_after` eager pairs (qest.library.
    # _pair_graph); whatever the warm-up count, the capture and a first replay happen here, before anything is timed
    graphed = False
    if not args.qe_only and qlms._pair_getter(key, lmax_qlm) is not None and options.opts.qe_graph:
        for rep in range(qlms.graph_after + 3):
            if any(isinstance(g.get('graph'), torch.cuda.CUDAGraph) and not g.get('first', False) for g in getattr(qlms, '_pair_graphs', {}).values()):
                graphed = True
                break
            qlms.get_sim_qlms(key, [2 * 10 ** 6 + 2 * rep, 2 * 10 ** 6 + 2 * rep + 1])
    if args.qe_only:  # the filtered alms of the timed indices are made resident beforehand
        for idx in range(rank, world * K, world):
            for name in ('tlm', 'elm', 'blm'):
                ivfs.get_sim_alm_dev(name, idx)
        assert ivfs._dev_slots >= K, 'increase the resident-alm slots for --qe-only with this many steps'
    qlms._mem.clear()
    for f_ in list(dev.host_future._in_flight):
        f_.result()
    sync_all()
    if not graphed:  # eager timed region: the per-stage HIP events ride in it
        plan.profile(True)
        plan.profile_read()
    t0 = time.perf_counter()
    # K reconstructions on this rank (jobs[rank::size] of world x K simulations), device-resident sum, RCCL all-reduce
    mf = qlms.get_sim_qlm_mf(key, np.arange(world * K), collective=True)
    assert qlms._last_dev is not None, 'the estimator library kept no device result for key %s' % key
    gathered = parallel.allgather(qlms._last_dev[0])  # output qlm all-gather over xGMI
    for f_ in list(dev.host_future._in_flight):  # every gradient / curl alm of the timed reconstructions is in host memory (numpy) at dt
        f_.result()
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0  # this rank's own work (collectives included), before the closing barrier
    sync_all()
    dt = time.perf_counter() - t0
    assert len(gathered) == world and mf.size == hp.Alm.getsize(lmax_qlm)
    nrec = sum(1 for (k_, i_) in qlms._mem if k_ == key and isinstance(i_, (int, np.integer)))
    assert nrec == K, 'timed region ran %d reconstructions on rank %d, expected %d' % (nrec, rank, K)
    last_dev, last_dev_key = qlms._last_dev, qlms._last_dev_key
    eager_pass = None
    if graphed:
        # Per-kernel durations: HIP events cannot be recorded inside a replayed graph, so the SAME K reconstructions per rank run once
        # more right here, launched eagerly (use_graph off, same paired kernels in the same order) with the per-stage events on the launch
        # stream.  `value` is the replayed region above; `kernels` / `roofline` come from this pass, whose own rate is reported beside it.
        g_timed = last_dev[0].clone()
        qlms._mem.clear()
        qlms.use_graph = False
        sync_all()
        plan.profile(True)
        plan.profile_read()
        t0e = time.perf_counter()
        qlms.get_sim_qlm_mf(key, np.arange(world * K, 2 * world * K), collective=True)
        for f_ in list(dev.host_future._in_flight):
            f_.result()
        torch.cuda.synchronize()
        sync_all()
        dte = time.perf_counter() - t0e
        qlms.use_graph = True
        eager_pass = {'ms_per_step': 1e3 * dte / K, 'value': world * K / dte,
                      'note': 'the same K reconstructions per rank launched eagerly (no graph replay) with per-stage HIP events: where `kernels` and '
                              '`roofline.avg_launch_ms` are measured; shares are of this pass'}
        last_dev = (g_timed, None)
    prof = plan.profile_read()
    plan.profile(False)
    dt_prof = dte if graphed else dt
    # self-check, outside the timed region: the last gradient alm of the timed (paired / replayed) route against a fresh evaluation of the
    # same simulation through the SINGLE-simulation eager route -- the one tests/test_gpu_fullsize.py compares with the oracle at this
    # size.  The two routes form every sum in the same order: the difference must be exactly zero.
    selfcheck = None
    single = {'p': lambda i: qlms._get_sim_MVgclm(i, 'p'), 'p_p': lambda i: qlms._get_sim_Pgclm(i, 'p_p'), 'ptt': lambda i: qlms._get_sim_Tgclm(i, 'ptt')}
    if key in single and last_dev_key is not None and not args.qe_only:
        g_timed = dev.to_host(last_dev[0])
        idx_last = last_dev_key[1]
        g_single = np.asarray(dev.resolve(single[key](idx_last)[0]))
        selfcheck = float(np.max(np.abs(g_single - g_timed)))
    # device memory: the plan of the benchmarked grid with the tables of every spin the key used (geometry, recursion coefficients, ring-FFT tables,
    # seed tables per kernel family: pl_plan_bytes; hipMalloc, not the framework's allocator), the framework allocator's peak, and what the
    # driver reports as used (everything, this process's share of the GPU)
    free_b, total_b = torch.cuda.mem_get_info()
    mem_stats = {'plan_device_mb': plan.bytes() / 2. ** 20, 'torch_peak_allocated_mb': torch.cuda.max_memory_allocated() / 2. ** 20,
                 'torch_peak_reserved_mb': torch.cuda.max_memory_reserved() / 2. ** 20, 'device_used_mb': (total_b - free_b) / 2. ** 20,
                 'device_total_mb': total_b / 2. ** 20,
                 'note': 'after the timed region; plan_device_mb = pl_plan_bytes of the (nside, lmax) plan incl. the seed tables of every spin used so far; '
                         'device_used_mb = total - free of hipMemGetInfo (all plans, workspaces, the caching allocator, captured graphs)'}
    ranks_seen = world
    dt_ranks = [dt]
    if use_dist:
        # per-rank times of the timed region (before the closing barrier a slow rank shows up as a long time of its own, after it as
        # everybody's): gathered so that imbalance between GPUs is visible in the line; `value` uses the maximum
        mine = torch.zeros(world, dtype=torch.float64, device='cuda' if backend == 'nccl' else 'cpu')
        mine[rank] = dt_local
        dist.all_reduce(mine)
        dt_ranks = [float(x) for x in mine.tolist()]
        dt, = reduce_max([dt])
        one = torch.ones(1, device='cuda' if backend == 'nccl' else 'cpu')
        dist.all_reduce(one)
        ranks_seen = int(one.item())

    # ---- the same reconstructions INCLUDING the device-side generation of their inputs (simulation synthesis must not
    # be what limits an 8-GPU run): phases drawn on the GPU (Philox), sky alms coloured, T / Q / U maps synthesised, white noise added
    # (sims.maps.cmb_maps_nlev on sims.phas.*_dev, as params/idealized_example.py does with PLENS_DEVICE_SIMS=1) -> filter -> QE.
    from_sims = None
    if not args.no_from_sims and not args.qe_only:
        try:
            from plancklens_amd.sims import cmbs, maps as sim_maps, phas
            pix_ph = phas.pix_lib_phas_dev(os.path.join(tmp, 'pix_phas'), 3, (hp.nside2npix(nside),), seed=11 + rank)
            sky_ph = phas.lib_phas_dev(os.path.join(tmp, 'sky_phas'), 3, lmax, seed=12 + rank)
            skies = cmbs.sims_cmb_unl({k: cl_len[k] for k in ['tt', 'ee', 'bb', 'te']}, sky_ph)
            gsims = sim_maps.cmb_maps_nlev(skies, transf, nlev_t, nlev_p, nside, pix_lib_phas=pix_ph, device_maps=True)
            mpi.rank, mpi.size = 0, 1
            try:
                ivfs_g = filt_simple.library_fullsky_sepTP(os.path.join(tmp, 'ivfs_g'), gsims, nside, transf, cl_len, ftl, fel, fbl, cache=False)
                qlms_g = qest.library_sepTP(os.path.join(tmp, 'qlms_g'), ivfs_g, ivfs_g, cl_len['te'], nside, lmax_qlm=lmax_qlm, cache=False)
            finally:  # (a constructor that raises must not leave every rank believing it is alone for the rest of the run)
                mpi.rank, mpi.size = rank, world
            qlms_g.get_sim_qlms(key, [10 ** 6 + world * w + rank for w in range(min(args.warmup, 2))])
            if graphed:  # set-up, as for the resident-input leg: the pair graph of this library is captured and replayed once before the timed region
                for rep in range(qlms_g.graph_after + 3):
                    if any(isinstance(g_.get('graph'), torch.cuda.CUDAGraph) for g_ in getattr(qlms_g, '_pair_graphs', {}).values()):
                        break
                    qlms_g.get_sim_qlms(key, [3 * 10 ** 6 + 2 * rep, 3 * 10 ** 6 + 2 * rep + 1])
                qlms_g.get_sim_qlms(key, [4 * 10 ** 6, 4 * 10 ** 6 + 1])  # (one replay through the simulation library's *_into route)
            qlms_g._mem.clear()
            sync_all()
            t0 = time.perf_counter()
            qlms_g.get_sim_qlm_mf(key, np.arange(world * K), collective=True)
            for f_ in list(dev.host_future._in_flight):
                f_.result()
            sync_all()
            dtg = time.perf_counter() - t0
            # generation alone (maps of K further simulations made and dropped)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for idx in range(0, K - K % 2, 2):  # in pairs, as the estimator's loop asks for them (the two polarization skies on one recursion)
                i0, i1 = 2 * 10 ** 6 + idx, 2 * 10 ** 6 + idx + 1
                gsims.hint_pair(i0, i1)
                for i_ in (i0, i1):
                    gsims.get_sim_tmap(i_)
                    gsims.get_sim_pmap(i_)
            for idx in range(K - K % 2, K):
                gsims.get_sim_tmap(2 * 10 ** 6 + idx)
                gsims.get_sim_pmap(2 * 10 ** 6 + idx)
            torch.cuda.synchronize()
            dgen = time.perf_counter() - t0
            if use_dist:
                dtg, dgen = reduce_max([dtg, dgen])
            from_sims = {'value': world * K / dtg, 'unit': 'reconstructions/s', 'ms_per_step': 1e3 * dtg / K,
                         'generation_ms_per_simulation': 1e3 * dgen / K, 'generation_share_of_step': dgen / dtg,
                         'note': "same estimator, inputs NOT resident: every simulation's T, Q, U maps are generated on the device inside the "
                                 "timed region (Philox phases -> coloured alms -> alm2map + alm2map_spin + white noise: sims.maps.cmb_maps_nlev, "
                                 "maps.py:46-77,136-173 of the reference); `value` of the line stays the resident-input rate"}
            del gsims, ivfs_g, qlms_g, skies
        except Exception as e:  # a report, never a reason to lose the headline number
            from_sims = {'error': repr(e)}

    if rank == 0:
        nalm = hp.Alm.getsize(lmax)
        steps_leg = nalm * 2 * nside                      # (l, m, ring pair) recursion steps per transform
        flops_spin, flops_scal = 24.0 * steps_leg, 8.0 * steps_leg
        exec_spin, exec_scal = executed_flops(nside, lmax, 2), executed_flops(nside, lmax, 0)
        # what the kernels run since round 5: the plan's seed tables start every wave where its recursion-only phase would have ended
        # (pl_plan_executed_steps: steps from there on, every ring-pair slot of a running wave counted), by kernel family
        exec_by_family, exe_useful = {}, {}
        try:
            from plancklens_amd import _lib as _plib
            hpl = shts.get_plan(nside, lmax).h
            for nm, sp, fam, fl_ in (('leg_synth0', 0, 0, 12.), ('leg_anal0', 0, 1, 12.), ('leg_synths', 2, 0, 24.), ('leg_anals', 2, 1, 24.)):
                st_ = int(_plib.lib().pl_plan_executed_steps(hpl, sp, fam))
                if st_ >= 0:
                    exec_by_family[nm] = st_ * fl_
                su_ = int(_plib.lib().pl_plan_useful_steps(hpl, sp, fam))
                if su_ >= 0:
                    exe_useful[nm] = su_ * fl_
        except Exception:
            exec_by_family = {}
        npix = hp.nside2npix(nside)
        res = {
            'metric': "QE reconstructions/sec at nside=%d lmax=%d ('%s' MV); CG-iter/sec" % (nside, lmax, key),
            'value': world * K / dt, 'unit': 'reconstructions/s', 'n_gpus': world, 'steps': K,
            'warmup': args.warmup, 'ms_per_step': 1e3 * dt / K, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic', 'ranks_seen': ranks_seen,
            'ms_per_step_by_rank': [1e3 * x / K for x in dt_ranks],
            'selfcheck_max_abs_diff': selfcheck,  # timed route vs single-simulation eager route, same simulation (rank 0); must be 0.0
            'graph_replay': graphed,  # timed region = replayed HIP graphs of reconstruction pairs (qest.library._pair_graph)
            'options': options.opts.as_dict(),  # the host layer's run-time options (plancklens_amd/options.py; PLENS_OPTIONS)
            # sum |mf_lm|^2 of the mean field the timed region produced (all ranks' simulations: after the all-reduce every rank holds it)
            'mean_field_checksum': float(np.sum(np.abs(mf) ** 2)),
            'plan_create': {'nside': nside, 'lmax': lmax, 'seconds': plan_create_s},
            'memory': mem_stats,
            'config': {'workload': "'%s' MV quadratic estimator from T,Q,U maps: isotropic filter + qest.library_sepTP, "
                                   "nside=%d lmax_ivf=%d lmax_qlm=%d, 9 SHTs (2 scalar + 7 spin pairs) per reconstruction (BASELINE.json headline config); "
                                   "timed region = qest.library.get_sim_qlm_mf over %d simulations (%d per GPU) + all-gather of the last qlm"
                                   % (key, nside, lmax, lmax_qlm, world * K, K) +
                                   (' -- QE-ONLY variant: filtered alms already resident, the filter transforms are not timed' if args.qe_only else ''),
                       'nside': nside, 'lmax': lmax, 'lmax_qlm': lmax_qlm, 'key': key, 'sims_per_gpu': K,
                       'resident_map_sets': sims.nsets,  # distinct T, Q, U realisations in HBM: consecutive pairs of simulations read different ones
                       'parallelism': 'sim-sharded x%d' % world},
        }
        # gradient-only synthesis (curl alm = 0, shts.alm2map_spin([G, None])): 8 recurrence + 8 accumulation flop per step;
        # paired synthesis (general + gradient-only input on one recursion, shts.alm2map_spin_pair): 8 recurrence + 16 + 8
        # accumulation flop per step (SURVEY's fixed count for the two transforms it replaces would be 48)
        # batched synthesis (two simulations on one recursion, shts.alm2map_spin_batch2): 8 recurrence + 2 x 16 accumulation flop per
        # step for TWO maps (the fixed count of the two transforms it replaces is 48)
        # two scalar syntheses on one recursion (shts.alm2map_batch2): 10 instead of 2 x 6 FMAs per two-l step; the fixed count of the two transforms is 16 flop per (l, m, ring pair)
        alg = {'leg_synth0': flops_scal, 'leg_synths': flops_spin, 'leg_anal0': flops_scal, 'leg_anals': flops_spin,
               'leg_synths_grad': flops_spin * 16. / 24., 'leg_synths_pair': flops_spin * 32. / 24., 'leg_synths_batch2': flops_spin * 40. / 24.,
               'leg_synth0_pair': 2. * flops_scal}
        e_s0, e_a0 = exec_by_family.get('leg_synth0', exec_scal), exec_by_family.get('leg_anal0', exec_scal)
        e_ss, e_as = exec_by_family.get('leg_synths', exec_spin), exec_by_family.get('leg_anals', exec_spin)
        exe = {'leg_synth0': e_s0, 'leg_synths': e_ss, 'leg_anal0': e_a0, 'leg_anals': e_as,
               'leg_synths_grad': e_ss * 16. / 24., 'leg_synths_pair': e_ss * 32. / 24., 'leg_synths_batch2': e_ss * 40. / 24.,
               'leg_synth0_pair': e_s0 * 20. / 12.}
        # components per ring-FFT stage launch are not recorded by the profile; both directions move, per component,
        # 8 npix + 32 npairs (mmax + 1) algorithmic bytes.  Launch mix of one 'p' reconstruction (qest._get_sim_MVgclm):
        # synthesis stages of 1 + 2 + 2 + 4 components in 4 launches, analysis stages of 1 + 2 + 2 in 3 launches.
        # With paired simulations (leg_synths_batch2 present) two reconstructions issue 6 synthesis stages of 1 + 1 + 4 + 4 + 4 + 4.
        comps_per_launch = {'fft_synth': 9. / 4., 'fft_anal': 5. / 3.} if (key == 'p' and not args.qe_only) else {}
        if comps_per_launch and prof.get('leg_synths_batch2', (0., 0))[1] > 0:
            nsyn = prof['fft_synth'][1]
            comps_per_launch['fft_synth'] = 9. * K / nsyn if nsyn else 3.
        fft_bytes_comp = 8.0 * npix + 32.0 * 2 * nside * (lmax + 1)
        per_kernel = {}
        for k, (m_, c_) in prof.items():
            if c_ == 0:
                continue
            ent = {'avg_ms': m_ / c_, 'launches': c_, 'share_of_step': m_ / (1e3 * dt_prof)}
            if k in alg:
                ent['executed_tflops'] = exe[k] / (m_ / c_ * 1e-3) / 1e12
                if k in exe_useful:
                    ent['executed_useful_tflops'] = exe_useful[k] / (m_ / c_ * 1e-3) / 1e12
                ent['fixed_denominator_tflops'] = alg[k] / (m_ / c_ * 1e-3) / 1e12
            elif k in comps_per_launch:
                ent['components_per_launch'] = comps_per_launch[k]
                ent['ms_per_component'] = m_ / c_ / comps_per_launch[k]
                ent['alg_gbs_per_component'] = fft_bytes_comp / (ent['ms_per_component'] * 1e-3) / 1e9
                ent['frac_of_achievable_hbm'] = ent['alg_gbs_per_component'] / HBM_ACHIEVABLE_GBS
            per_kernel[k] = ent
        res['kernels'] = per_kernel
        if eager_pass is not None:
            res['eager_pass'] = eager_pass
        leg = [k for k in per_kernel if k in alg]
        if leg:
            dom = max(leg, key=lambda k: prof[k][0])  # largest summed time in the timed region
            ms, cnt = prof[dom]
            avg_ms = ms / cnt
            ach = exe[dom] / (avg_ms * 1e-3) / 1e12
            fixed = alg[dom] / (avg_ms * 1e-3) / 1e12
            useful = exe_useful.get(dom)
            pmc_bytes, pmc_source = pmc_traffic()
            ceil = fma_ceilings()
            # the FMA mix of the dominant kernel: analysis accumulates with three vector sources, synthesis with a scalar coefficient operand
            mix = 'three_vector_sources (analysis mix)' if 'anal' in dom else 'two_scalar_sources (synthesis mix)'
            res['roofline'] = {'bound': 'mfma', 'achieved': fixed, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': fixed / FP64_PEAK_TFLOPS,
                               'traffic': pmc_bytes.get(dom) if (nside, lmax) == (2048, 2048) else None, 'traffic_source': pmc_source,
                               'kernel': KERNEL_NAMES.get(dom, dom), 'avg_launch_ms': avg_ms, 'launches': cnt,
                               'share_of_step': ms / (1e3 * dt_prof),
                               'achieved_executed': ach, 'frac_executed': ach / FP64_PEAK_TFLOPS,
                               'achieved_executed_useful': None if useful is None else useful / (avg_ms * 1e-3) / 1e12,
                               'frac_of_measured_issue_ceiling': ach / ceil[mix] if ceil.get(mix) else None,
                               'issue_ceiling_mix': mix,
                               'fma_issue_ceiling_measured_tflops': ceil,
                               'note': 'dominant kernel = largest summed time in the timed region. FP64 vector-FMA issue bound (v_fma_f64), reported under the "mfma" '
                                       '(TFLOP/s) arm of the schema: gfx950 FP64 MFMA peak = FP64 vector peak, and NO MFMA instruction is used (a recurrence, not a '
                                       'contraction). achieved / frac = SURVEY 8(d) ALGORITHMIC count (24 or 8 flop x nalm x 2 nside per launch, no credit for pruned '
                                       'rings or for the steps the seed tables skip -- so it is no ceiling: it can exceed what is executed) / mean launch time (HIP '
                                       'events on the launch stream) against the datasheet 78.6 TF. achieved_executed / frac_executed = the flops the kernel really '
                                       'issues (every ring-pair slot of every running wavefront from the step its seed table starts it at, pl_plan_executed_steps); '
                                       'achieved_executed_useful counts only slots holding an unpruned ring (pl_plan_useful_steps). frac_of_measured_issue_ceiling = '
                                       'achieved_executed / the rate a pure FMA loop of the same operand mix sustains on this GPU in this process '
                                       '(fma_issue_ceiling_measured_tflops). traffic = FETCH_SIZE + WRITE_SIZE bytes per launch (%s: PMC passes of the same kernel)'
                                       % pmc_source}
        # whole-reconstruction algorithmic traffic (counting rule of SURVEY.md 8(d))
        b_scal = 8.0 * npix + 16.0 * nalm
        b_spin = 2 * b_scal
        bytes_rec = 2 * b_scal + 7 * b_spin   # each map (8 npix B) and alm (16 nalm B) component touched once per transform
        res['hbm'] = {'algorithmic_GB_per_reconstruction': bytes_rec / 1e9,
                      'achieved_GBs': bytes_rec / 1e9 / (dt / K), 'peak_GBs': HBM_PEAK_GBS,
                      'frac': bytes_rec / 1e9 / (dt / K) / HBM_PEAK_GBS,
                      'note': 'path is FP64-FMA bound (arithmetic intensity ~190 flop/B): a low HBM fraction is a property of the algorithm'}
    if rank == 0 and from_sims is not None:
        res['from_sims'] = from_sims
    # release the QE working set before the CG block
    del sims, ivfs, qlms, gathered
    if world == 1 and rank == 0 and not args.no_plan_stats and nside == 2048:
        # BASELINE config 5 starts 8 ranks at once, each building the tables of an nside-4096 plan on the host (long double, ~1 GB of
        # vectors) before uploading them: the first thing that can time out or run a host out of memory.  Measured here, outside every
        # timed region: wall time of pl_plan_create, growth of the process's peak resident set, device bytes of the plan.
        try:
            import resource
            rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
            t0p = time.perf_counter()
            p4 = shts.Plan(4096, 4096)
            t4 = time.perf_counter() - t0p
            mb0 = p4.bytes() / 2. ** 20
            t0s = time.perf_counter()
            from plancklens_amd import _lib as _plib4
            by_spin = {}
            for sp in (1, 2, 3):  # first use of a spin builds its recursion tables (host) and seed tables (device): the MV estimator uses all three
                b_before = p4.bytes()
                assert _plib4.lib().pl_plan_executed_steps(p4.h, sp, 0) != 0
                by_spin[str(sp)] = (p4.bytes() - b_before) / 2. ** 20
            t4s = time.perf_counter() - t0s
            rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
            res['plan_create']['nside_4096'] = {'seconds': t4, 'spin_tables_seconds': t4s, 'host_peak_rss_growth_mb': (rss1 - rss0) / 1024.,
                                                'host_peak_rss_mb': rss1 / 1024., 'device_mb_spin0': mb0, 'device_mb_by_spin': by_spin,
                                                'device_mb': p4.bytes() / 2. ** 20,
                                                'note': 'pl_plan_create(4096, 4096) in this process after the 2048 run, then the tables of spins 1, 2, 3 (built on '
                                                        "first use; the MV estimator of BASELINE config 5 uses all of them): device_mb = the plan's whole footprint "
                                                        'incl. the seed tables of both kernel families of every spin'}
            del p4
        except Exception as e:
            res['plan_create']['nside_4096'] = {'error': repr(e)}
    if world == 1 and rank == 0:
        if not args.no_cg:
            try:
                sys.path.insert(0, os.path.join(ROOT, 'tools'))
                import cg_bench
                cg = cg_bench.run(nside, lmax, args.cg_iters, kinds=('t', 'p'), peak_tflops=FP64_PEAK_TFLOPS,
                                  batches=[int(b) for b in args.cg_batches.split(',') if b.strip()])
                res['cg'] = {'metric': 'CG-iter/sec: qcinv multigrid Wiener filter, cinv_t + cinv_p, nside=%d lmax=%d, masked sky fsky=%.2f, '
                                       '%d top-level iterations each (eps_min=0), default chains, dense preconditioner cached outside the timed region; '
                                       'median of %d solves per filter'
                                       % (nside, lmax, cg['fsky'], args.cg_iters, len(cg['t']['seconds_each_solve'])),
                             'T_iters_per_s': cg['t']['iters_per_s'], 'P_iters_per_s': cg['p']['iters_per_s'],
                             'TP_iters_per_s': cg['tp']['iters_per_s'], 'TP_ms_per_iter': cg['tp']['ms_per_iter'],
                             'fp64_floor_ms_per_iter': cg['tp'].get('fp64_floor_ms_per_iter'),
                             'frac_of_fp64_floor': cg['tp'].get('frac_of_fp64_floor'),
                             'floor_note': "fixed denominator: SURVEY 8(d)'s flop count of one top-level iteration as the reference schedules it (T 1.83e11 + P 5.68e11 flop, "
                                           "incl. its 63 coarse temperature operators, of which this code executes 39: the directions of an iteration that is "
                                           "not going to happen are not formed) at 78.6 TFLOP/s",
                             'seconds_each_solve': {'t': cg['t']['seconds_each_solve'], 'p': cg['p']['seconds_each_solve']},
                             'dense_setup_s': {'t': cg['t']['first_call_incl_dense_setup_s'], 'p': cg['p']['first_call_incl_dense_setup_s']},
                             'residual_first_last': {'t': cg['t']['eps_first_last'], 'p': cg['p']['eps_first_last']},
                             # cinv_t and cinv_p of one simulation at the same time on two streams of this process (filt_cinv.apply_ivf_tp,
                             # what library_cinv_sepTP.filter_sims runs); the T / P / TP figures above are one solve after the other
                             'TP_concurrent': cg.get('tp_concurrent'),
                             # B simulations filtered in ONE block solve (cinv_*.apply_ivf_batch): every launch carries all B, each with
                             # its own step lengths; iterations/s per simulation = B x iterations/s of the block solve
                             'block_solves': {'note': 'B right-hand sides (simulations sharing the noise model) per solve; per-simulation rates; '
                                                      'B = 1 is the figures above', 'per_B': cg.get('batched')}}
            except Exception as e:  # a report, never a reason to lose the QE number
                res['cg'] = {'error': repr(e)}
        if not args.no_cpu_baseline:
            try:
                res['cpu_baseline'] = cpu_baseline_any(nside, lmax, args.cpu_seconds)
            except Exception as e:
                res['cpu_baseline'] = {'value': None, 'unit': 'reconstructions/s', 'cores': usable_cpus(), 'kind': 'port',
                                       'sample': 'failed: %r' % (e,)}
    if rank == 0:
        res['graphs'] = dict(options.stats)  # HIP-graph captures of this process and the ones that failed and fell back to eager launches
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return 0


def main():
    args = parse()
    if args.gpus < 1:
        sys.stderr.write('bench.py: --gpus must be >= 1\n')
        return 2
    if 'RANK' not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus, sys.argv[1:])
    return run_rank(args)


if __name__ == '__main__':
    sys.exit(main())
