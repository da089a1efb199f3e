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
