"""bench.py --gpus N must itself start N ranks (the driver also launches it under torch.distributed.run, where the ranks come
from the launcher).  Run here on CPU with the launcher's self-test rank body (PLBENCH_STUB=1: gloo, no GPU work, no number):
the parent starts the ranks, rank 0 prints the one JSON line and reports how many ranks took part."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra):
    env = dict(os.environ, PLBENCH_STUB='1', **env_extra)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        if k not in env_extra:
            env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


def test_gpus_flag_starts_that_many_ranks():
    rc, out, err = _run(['--gpus', '2', '--steps', '3', '--warmup', '1'], {})
    assert rc == 0, err
    lines = [l for l in out.splitlines() if l.startswith('{')]  # (gloo prints its own connection notes on stdout)
    assert len(lines) == 1, out  # ONE JSON line, from rank 0
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['ranks_seen'] == 2 and res['steps'] == 3 and res['warmup'] == 1
    assert res['value'] is None and 'STUB' in res['data']  # the self-test can never be mistaken for a measurement


def test_single_rank_and_launcher_mismatch():
    rc, out, err = _run(['--gpus', '1', '--steps', '2', '--warmup', '0'], {})
    assert rc == 0, err
    assert json.loads([l for l in out.splitlines() if l.startswith('{')][0])['ranks_seen'] == 1
    # under an external launcher WORLD_SIZE must agree with --gpus: fail loudly instead of silently running on one GPU
    rc, out, err = _run(['--gpus', '4'], {'RANK': '0', 'WORLD_SIZE': '1', 'LOCAL_RANK': '0'})
    assert rc != 0 and 'WORLD_SIZE' in err and out.strip() == ''


def test_launcher_ends_the_job_when_a_rank_dies(tmp_path):
    """A rank that exits with an error must end the whole job promptly (its siblings would wait in the rendezvous for minutes)."""
    import time
    env = dict(os.environ, PLBENCH_STUB='1', PLBENCH_STUB_FAIL_RANK='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode != 0
    assert time.time() - t0 < 60


def test_eight_stub_ranks():
    """the driver's 8-GPU line through the launcher's self-test body: eight ranks rendezvous, one JSON line"""
    rc, out, err = _run(['--gpus', '8', '--steps', '2', '--warmup', '1'], {})
    assert rc == 0, err
    res = json.loads([l for l in out.splitlines() if l.startswith('{')][0])
    assert res['n_gpus'] == 8 and res['ranks_seen'] == 8


def test_config_switch():
    """`--config 5` = BASELINE config 5 (MV 'p', nside = lmax = 4096, 32 simulations per GPU: 256 over 8, no CG / CPU legs): the driver need not
    guess flags"""
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse(['--config', '5', '--gpus', '8'])
    assert (a.nside, a.lmax, a.key, a.steps, a.no_cg, a.no_cpu_baseline) == (4096, 4096, 'p', 32, True, True)
    assert bench.parse(['--config', '5', '--steps', '4']).steps == 4
    a = bench.parse(['--config', '1'])
    assert (a.nside, a.lmax, a.key, a.steps) == (512, 512, 'ptt', 10)
    a = bench.parse([])
    assert (a.nside, a.lmax, a.key, a.steps, a.gpus) == (2048, 2048, 'p', 10, 1)


def test_cpu_baseline_healpy_branch_runs_the_reference_sequence():
    """bench.py's `cpu_baseline`: when `import healpy` succeeds the reference's own hp.* call sequence is what is timed (kind 'healpy').  healpy
    is absent from this image, so the branch is exercised with a stand-in module built from the CPU oracle's transforms at a tiny size, and
    its estimate is compared with the oracle's restatement of the same estimator (oracle/qe_oracle.py) -- a branch that only runs on somebody
    else's box must at least be the right sequence."""
    import types
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    from oracle import hp_oracle as oh, sht_oracle as so, qe_oracle as qo
    so.build()
    m = types.ModuleType('healpy')
    m.__version__ = 'stand-in'
    m.almxfl = lambda alm, fl, inplace=False: (alm.__imul__(oh.almxfl(np.ones_like(alm), fl)) if inplace else oh.almxfl(alm, fl))
    m.map2alm = lambda mp, lmax=None, iter=0: so.map2alm(np.asarray(mp), lmax=lmax, iter=iter)
    m.alm2map = lambda alm, nside, lmax=None: so.alm2map(np.asarray(alm), nside, lmax=lmax)
    m.alm2map_spin = lambda gc, nside, spin, lmax: so.alm2map_spin([np.asarray(gc[0]), np.asarray(gc[1])], nside, spin, lmax)
    m.map2alm_spin = lambda maps, spin, lmax=None: so.map2alm_spin([np.asarray(maps[0]), np.asarray(maps[1])], spin, lmax)
    nside, lmax = 16, 32
    res = bench.cpu_baseline_healpy(m, nside, lmax, budget_seconds=5., reps=1)
    assert res['kind'] == 'healpy' and res['value'] > 0 and res['cores'] >= 1 and 'qest.py:248-285' in res['sample']
    # the sequence itself against the oracle's estimator on the same inputs (random maps of seed 5, unit filters above l = 1, C^TE = 0.1)
    def run_once():
        rng = np.random.default_rng(5)
        tmap, qmap, umap = rng.standard_normal((3, 12 * nside ** 2))
        fl = np.ones(lmax + 1)
        fl[:2] = 0.
        cls = {'tt': fl, 'ee': fl, 'bb': fl, 'te': 0.1 * fl}
        t, e, b = qo.filter_maps(tmap, qmap, umap, lmax, fl, fl, fl, np.ones(lmax + 1))
        return qo.qe_sepTP('p', (t, e, b), (t, e, b), cls, nside, lmax)
    Go, Co = run_once()
    G, C = bench._cpu_baseline_healpy_estimate(m, nside, lmax)
    assert np.sqrt(np.sum(np.abs(G - Go) ** 2) / np.sum(np.abs(Go) ** 2)) < 1e-10
    assert np.sqrt(np.sum(np.abs(C - Co) ** 2) / np.sum(np.abs(Co) ** 2)) < 1e-10
