"""The bench driver's one-line JSON contract, on a small configuration (the default run is nside = lmax = 2048)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--nside', '64', '--lmax', '64', '--steps', '2', '--warmup', '1',
                          '--cpu-seconds', '1', '--no-cg'], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout  # exactly one JSON line on stdout
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1 and d['higher_is_better'] is True
    assert d['vs_baseline'] is None and d['dtype'] == 'f64' and d['data'] == 'synthetic' and d['scaling'] == 'weak'
    assert d['unit'] == 'reconstructions/s' and d['value'] > 0 and abs(d['value'] * d['ms_per_step'] / 1e3 - 1.) < 1e-6
    assert 'workload' in d['config'] and 'model' not in d['config']
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s') and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    c = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] in ('reference', 'port') and c['value'] > 0 and c['cores'] >= 1
    assert d['ranks_seen'] == 1 and 'kernels' in d and 'leg_anals' in d['kernels']
    assert len(d['ms_per_step_by_rank']) == 1 and 0 < d['ms_per_step_by_rank'][0] <= d['ms_per_step'] * 1.0001
    f = d['from_sims']  # the same reconstructions with their inputs generated on the device inside the timed region
    assert 'error' not in f, f
    assert f['value'] > 0 and f['generation_ms_per_simulation'] > 0 and 0 < f['generation_share_of_step'] < 1


def test_bench_under_launcher_with_rccl_collectives():
    """The driver's launch line (torch.distributed.run, here with one rank -- the box has one GPU): RANK / WORLD_SIZE from the
    launcher, process group on nccl (= RCCL), and with PLENS_DIST_FORCE=1 the mean-field all-reduce and the qlm all-gather of
    the timed region really go through RCCL."""
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--nside', '64', '--lmax', '64', '--steps', '2',
           '--warmup', '1', '--no-cg', '--no-cpu-baseline']
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=dict(os.environ, PLENS_DIST_FORCE='1'))
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][0])
    assert d['n_gpus'] == 1 and d['ranks_seen'] == 1 and d['value'] > 0 and len(d['ms_per_step_by_rank']) == 1 and 'error' not in d['from_sims']


def _bench_line(args, env=None, nranks=1, timeout=900):
    import socket
    if nranks == 1:
        cmd = [sys.executable, os.path.join(ROOT, 'bench.py')] + args
    else:
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nranks), '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', str(nranks)] + args
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][0])


def test_bench_two_ranks_on_one_gpu_give_the_one_rank_mean_field():
    """The N > 1 path of bench.py for real (no stub) on the one GPU of the box: two rank processes under the driver's launcher sharing
    device 0, process group on gloo (PLENS_DIST_BACKEND=gloo: collectives staged through the host).  Both ranks are seen, each
    reports its own time, and the mean field over world x K simulations -- sharded jobs[rank::size], summed by the all-reduce --
    equals the one computed by a single rank over the same simulations (all ranks on the same input maps: --sims-seed)."""
    common = ['--nside', '64', '--lmax', '64', '--warmup', '2', '--no-cg', '--no-cpu-baseline', '--no-from-sims', '--sims-seed', '7']
    one = _bench_line(common + ['--steps', '4'])
    two = _bench_line(common + ['--steps', '2'], env={'PLENS_DIST_BACKEND': 'gloo'}, nranks=2)
    assert one['ranks_seen'] == 1 and two['ranks_seen'] == 2 and two['n_gpus'] == 2
    assert len(two['ms_per_step_by_rank']) == 2 and all(t > 0 for t in two['ms_per_step_by_rank'])
    assert two['steps'] == 2 and abs(two['value'] * two['ms_per_step'] / 1e3 - 2.) < 1e-6  # whole-job rate: world x K reconstructions / time
    assert one['mean_field_checksum'] > 0
    assert abs(two['mean_field_checksum'] / one['mean_field_checksum'] - 1.) < 1e-12
    assert one['selfcheck_max_abs_diff'] == 0.0 and two['selfcheck_max_abs_diff'] == 0.0
    assert one['plan_create']['seconds'] > 0
