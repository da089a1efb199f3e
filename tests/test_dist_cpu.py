"""Multi-process path on CPU: two ranks over gloo (the GPU run uses the same code over RCCL): rank / size / barrier of
helpers.mpi, the reference's jobs[rank::size] sharding, and the mean-field all-reduce / qlm all-gather of
plancklens_amd.parallel."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch
sys.path.insert(0, %(root)r)
from plancklens_amd.helpers import mpi
from plancklens_amd import parallel
rank, size = mpi.init(backend='gloo')
assert (rank, size) == (int(os.environ['RANK']), 2) and mpi.rank == rank and mpi.size == 2
jobs = [(idx, k) for k in ('p', 'ptt') for idx in range(7)]
mine = parallel.shard(jobs)
assert mine == jobs[rank::2]
mpi.barrier()
# mean field over 7 "simulations": every rank evaluates its shard only
like = torch.zeros(11, dtype=torch.complex128)
calls = []
def get_qlm(idx):
    calls.append(idx)
    return torch.full((11,), complex(idx, -2 * idx), dtype=torch.complex128)
mf = parallel.mean_field(get_qlm, np.arange(7), like)
assert calls == list(range(7))[rank::2]
assert torch.allclose(mf, torch.full((11,), complex(3., -6.), dtype=torch.complex128))
# numpy route of the all-reduce, and the all-gather
x = np.full(5, rank + 1.0)
parallel.allreduce_sum(x)
assert np.all(x == 3.0)
g = parallel.allgather(torch.full((4,), complex(rank, 1.0), dtype=torch.complex128))
assert len(g) == 2 and g[0][0] == complex(0, 1) and g[1][0] == complex(1, 1)
assert mpi.bcast({'a': rank} if rank == 0 else None) == {'a': 0}
mpi.barrier()
mpi.finalize()
print('worker %%d ok' %% rank)
'''


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_gloo(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % {'root': ROOT})
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert 'worker %d ok' % r in o


def test_serial_fallbacks():
    """Without a launcher the reference's serial values apply (helpers/mpi.py:34-53)."""
    from plancklens_amd.helpers import mpi
    from plancklens_amd import parallel
    assert mpi.rank == 0 and mpi.size == 1 and mpi.barrier() == -1
    assert parallel.shard(range(5)) == [0, 1, 2, 3, 4]
    x = np.ones(3)
    assert parallel.allreduce_sum(x) is x
