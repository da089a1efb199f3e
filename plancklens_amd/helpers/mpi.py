"""rank / size / barrier of plancklens/helpers/mpi.py:19-53, re-backed by one process per GPU.

The reference only ever uses `rank`, `size` and `barrier` (job slicing jobs[rank::size] and rank-0-writes
guards; SURVEY.md section 2 "Parallelism strategies").  Here they come from torch.distributed (backend "nccl"
= RCCL on ROCm, "gloo" on CPU) when the process was launched under torchrun (RANK / WORLD_SIZE in the
environment), and fall back to the reference's serial values otherwise.
"""
import os

rank = 0
size = 1
_initialised_here = False


def _dist():
    try:
        import torch.distributed as dist
        return dist
    except ImportError:  # pragma: no cover
        return None


def init(backend=None):
    """Join the torch.distributed world if launched by torchrun; idempotent."""
    global rank, size, _initialised_here
    dist = _dist()
    if dist is None or 'RANK' not in os.environ or 'WORLD_SIZE' not in os.environ:
        return rank, size
    if not dist.is_initialized():
        import torch
        if backend is None:  # PLENS_DIST_BACKEND=gloo: several ranks sharing one GPU (tests), collectives staged through the host
            backend = os.environ.get('PLENS_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if torch.cuda.is_available():
            local, ndev = int(os.environ.get('LOCAL_RANK', 0)), max(1, torch.cuda.device_count())
            if backend == 'nccl' and local >= ndev:  # RCCL needs one GPU per rank: two ranks on one device hang at the first collective
                raise RuntimeError('LOCAL_RANK %d but only %d GPU(s) visible: the nccl (RCCL) backend needs one GPU per rank '
                                   '(PLENS_DIST_BACKEND=gloo lets ranks share a GPU)' % (local, ndev))
            torch.cuda.set_device(local % ndev)
        dist.init_process_group(backend=backend)
        _initialised_here = True
    rank, size = dist.get_rank(), dist.get_world_size()
    return rank, size


def barrier():
    dist = _dist()
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    return -1


def bcast(obj, root=0):
    dist = _dist()
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        lst = [obj]
        dist.broadcast_object_list(lst, src=root)
        return lst[0]
    return obj


def finalize():
    dist = _dist()
    if dist is not None and dist.is_initialized() and _initialised_here:
        dist.destroy_process_group()
    return -1


if os.environ.get('USE_PLANCKLENS_MPI', '1') not in ('0', 'False', 'false'):
    dist_ = _dist()
    if dist_ is not None and dist_.is_initialized():
        rank, size = dist_.get_rank(), dist_.get_world_size()
