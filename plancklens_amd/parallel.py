"""One-process-per-GPU job sharding and the two collectives the hot path needs.

The reference fans (simulation, key) jobs over MPI ranks with jobs[rank::size] and exchanges results through files
(examples/run_qlms.py:57,72,92,106; mean field = file-based average, qest.py:238-244).  Here the same static
round-robin is kept and the only cross-rank data movement -- the mean-field sum and the gather of the output qlm --
is done with RCCL collectives over xGMI (torch.distributed backend "nccl"; "gloo" on CPU for the tests).
No data-path collective exists inside a reconstruction: simulations are independent.
"""
import numpy as np
import torch

from .helpers import mpi


def shard(jobs):
    """This rank's jobs: the reference's static round-robin jobs[rank::size]."""
    return list(jobs)[mpi.rank::mpi.size]


def _dist():
    """torch.distributed when a process group of more than one rank is up (PLENS_DIST_FORCE=1: also with a single rank, so that
    the collectives can be exercised on a one-GPU box)"""
    import os
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    return dist if (dist.get_world_size() > 1 or os.environ.get('PLENS_DIST_FORCE', '0') == '1') else None


def _as_real(t):
    return torch.view_as_real(t) if t.is_complex() else t


def allreduce_sum(x):
    """In-place sum over ranks of a (complex or real) tensor or numpy array; returns it."""
    dist = _dist()
    if dist is None:
        return x
    if isinstance(x, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(x))
        if dist.get_backend() == 'nccl':
            t = t.cuda()
        dist.all_reduce(_as_real(t))
        x[...] = t.cpu().numpy()
        return x
    if x.is_cuda and dist.get_backend() != 'nccl':  # device tensors over a CPU backend (gloo in the tests): staged through the host
        h = _as_real(x).cpu()
        dist.all_reduce(h)
        _as_real(x).copy_(h)
        return x
    dist.all_reduce(_as_real(x))
    return x


def allgather(x):
    """List (one entry per rank) of tensors equal to every rank's x (same shape on all ranks)."""
    dist = _dist()
    if dist is None:
        return [x]
    r = _as_real(x.contiguous())
    staged = r.is_cuda and dist.get_backend() != 'nccl'
    if staged:
        r = r.cpu()
    out = [torch.empty_like(r) for _ in range(dist.get_world_size())]
    dist.all_gather(out, r)
    if staged:
        out = [o.to(x.device) for o in out]
    return [torch.view_as_complex(o) if x.is_complex() else o for o in out]


def mean_field(get_qlm, idxs, like, get_pair=None, collective=True):
    """Mean of get_qlm(idx) over ALL idxs.  collective (a COLLECTIVE call: every rank of the job must make it, with the same
    idxs): each rank evaluates only its shard, sums locally, one all-reduce, divide.  collective=False: the calling rank evaluates
    every simulation by itself and no other rank is involved (the reference's loop, qest.py:238-244).
    `like` is a zero tensor giving shape / dtype / device of the accumulator.  get_pair(idx0, idx1) -> (q0, q1), when given, serves
    the shard two simulations at a time (their transforms share Legendre recursions); the terms are added in index order either way."""
    idxs = list(np.unique(np.asarray(idxs)))
    acc = torch.zeros_like(like)
    mine = shard(idxs) if collective else idxs

    def add(q):
        acc.add_(q if isinstance(q, torch.Tensor) else torch.as_tensor(q).to(acc.device))
    n2 = len(mine) - len(mine) % 2 if get_pair is not None else 0
    for i in range(0, n2, 2):
        q0, q1 = get_pair(mine[i], mine[i + 1])
        add(q0)
        add(q1)
    for idx in mine[n2:]:
        add(get_qlm(idx))
    if collective:
        allreduce_sum(acc)
    if len(idxs) > 0:
        acc /= len(idxs)
    return acc
