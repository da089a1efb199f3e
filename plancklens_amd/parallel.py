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
        q = q if isinstance(q, torch.Tensor) else torch.as_tensor(q).to(acc.device)
        if acc.is_cuda and q.is_cuda and q.dtype == acc.dtype and q.shape == acc.shape and q.is_contiguous() and acc.dtype in (torch.float64, torch.complex128):
            from . import dev
            dev.add_(acc, q)  # pl_axpy: one streaming pass
        else:
            acc.add_(q)
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


# ---- one transform over several GPUs: Legendre stage sharded by m-group, ring FFTs by ring pair ("m-blocks") ------------
def _all_to_all(send):
    """send: (nranks, n) float64 tensor, row s for rank s; returns the (nranks, n) tensor of rows received (row s from rank s).
    RCCL on device tensors; a CPU backend (gloo in the tests) is staged through the host."""
    dist = _dist()
    if dist is None:
        return send
    staged = send.is_cuda and dist.get_backend() != 'nccl'
    src = send.cpu() if staged else send
    out = torch.empty_like(src)
    dist.all_to_all_single(out, src) if dist.get_backend() == 'nccl' else _all_to_all_gloo(dist, out, src)
    return out.to(send.device) if staged else out


def _all_gather(send, out):
    """send: (n,) float64 tensor; out: (nranks, n), row s = rank s's send.  RCCL all_gather_into_tensor on device tensors; a CPU backend
    (gloo in the tests) is staged through the host."""
    dist = _dist()
    if dist is None:
        out[0].copy_(send)
        return out
    if send.is_cuda and dist.get_backend() != 'nccl':
        rows = [torch.empty(send.shape, dtype=send.dtype) for _ in range(dist.get_world_size())]
        dist.all_gather(rows, send.cpu())
        for s, row in enumerate(rows):
            out[s].copy_(row)
        return out
    dist.all_gather_into_tensor(out, send)
    return out


def _all_to_all_gloo(dist, out, src):
    """gloo has no all_to_all: one scatter per root (test backend only)"""
    rank, size = dist.get_rank(), dist.get_world_size()
    for root in range(size):
        dist.scatter(out[root], [src[s].contiguous() for s in range(size)] if rank == root else None, src=root)


class sharded_sht(object):
    """alm2map / map2alm of ONE map over the ranks of the job: every rank runs the Legendre stage for its m-groups (4 orders each,
    interleaved) and the ring FFTs for its ring pairs (interleaved), with one all-to-all of phase slices between them -- per rank
    and transform (1 - 1 / R) of 32 B x npairs x (mmax + 1) x ncomp / R (30 MB at nside = lmax = 2048, spin 2, R = 8), against
    8 npix x ncomp for an all-reduce of partial maps.  alm inputs are full arrays on every rank; map2alm returns the full alm on
    every rank (sum over ranks of shares that are zero outside the rank's m-groups: exact); alm2map returns the rank's own rings
    (`gather=False`: the other pixels are left as they are -- pixel-local products need no more) or the full map on every rank.
    Collective calls: every rank of the job makes them in the same order.  rank / size default to the job's (helpers.mpi)."""

    def __init__(self, nside, lmax, rank=None, size=None):
        from . import shts
        self.rank = mpi.rank if rank is None else rank
        self.size = mpi.size if size is None else size
        # (without a process group _all_to_all / _all_gather would hand back the rank's own buffers: wrong maps, silently)
        assert self.size == 1 or _dist() is not None, 'sharded_sht over %d ranks needs torch.distributed to be initialised' % self.size
        self.nside, self.lmax = nside, lmax
        self.plan = shts.get_shard_plan(nside, lmax, self.rank, self.size)
        self._mask = None
        self._bufs = {}

    def _buf(self, name, shape, device):
        """exchange buffers, kept per shape (sizes depend on geometry, R and the number of components only)"""
        b = self._bufs.get(name)
        if b is None or tuple(b.shape) != tuple(shape) or b.device != device:
            b = self._bufs[name] = torch.empty(shape, dtype=torch.float64, device=device)
        return b

    def _exchange(self, phase, ncomp, synth):
        """synthesis: my m-groups of everybody's ring pairs go out, everybody's m-groups of my ring pairs come in; analysis: the reverse"""
        from . import _lib, dev
        L, h, R, r = _lib.lib(), self.plan.h, self.size, self.rank
        sel_out = (lambda s: (s, R, r, R)) if synth else (lambda s: (r, R, s, R))   # (pair0, pair_stride, mg0, mg_stride) sent to rank s
        sel_in = (lambda s: (r, R, s, R)) if synth else (lambda s: (s, R, r, R))    # ... received from rank s
        # equal splits: the largest slice of any pair of ranks is the one of ring pairs 0, R, ... and m-groups 0, R, ... (the counts
        # ceil((n - start) / R) do not grow with the start) -- the same number on every rank, no collective needed to agree on it
        n_all = int(L.pl_phase_pack_doubles(h, ncomp, 0, R, 0, R))
        send = self._buf('send%d' % ncomp, (R, n_all), phase.device)  # (the tails past a slice are never read by the receiver)
        for s in range(R):
            _lib.check(L.pl_phase_pack(h, ncomp, phase.data_ptr(), send[s].data_ptr(), *sel_out(s), dev.stream_ptr()))
        recv = _all_to_all(send)
        for s in range(R):
            _lib.check(L.pl_phase_unpack(h, ncomp, phase.data_ptr(), recv[s].data_ptr(), *sel_in(s), dev.stream_ptr()))

    def _gather_rings(self, m, ncomp):
        """every rank's own ring pairs into every rank's map: one all-gather of 1 / R of the map per rank (pl_map_pack_rings), instead
        of an all-reduce of R zero-padded maps (8 npix ncomp bytes per rank: 3.2 GB at nside 4096, spin 2)"""
        from . import _lib, dev
        L, h, R, r = _lib.lib(), self.plan.h, self.size, self.rank
        if R == 1:
            return m
        per = [int(L.pl_map_pack_doubles(h, s, R)) for s in range(R)]
        n_all = ncomp * max(per)
        send = self._buf('gsend%d' % ncomp, (n_all,), m.device)
        _lib.check(L.pl_map_pack_rings(h, ncomp, m.data_ptr(), send.data_ptr(), r, R, dev.stream_ptr()))
        recv = _all_gather(send, self._buf('grecv%d' % ncomp, (R, n_all), m.device))
        for s in range(R):
            if s != r:
                _lib.check(L.pl_map_unpack_rings(h, ncomp, m.data_ptr(), recv[s].data_ptr(), s, R, dev.stream_ptr()))
        return m

    def own_pixels(self):
        """bool device tensor: the pixels of this rank's ring pairs"""
        if self._mask is None:
            from . import hp
            ring = hp.pix2ring(self.nside)  # 1 ... 4 nside - 1
            pair = np.minimum(ring, 4 * self.nside - ring) - 1
            self._mask = torch.from_numpy(pair % self.size == self.rank).cuda()
        return self._mask

    def alm2map(self, alm, spin=0, fl=None, gather=True):
        """alm: device tensor [nalm] (spin 0) or [2, nalm] (spin s, gradient and curl)"""
        from . import _lib, dev, shts
        L, h = _lib.lib(), self.plan.h
        ncomp = 1 if spin == 0 else 2
        a = alm.to(torch.complex128).contiguous()
        assert a.numel() == ncomp * self.plan.nalm
        f = shts._fl_arg(fl, self.lmax, True)
        phase = torch.empty(self.plan.phase_doubles(spin), dtype=torch.float64, device=a.device)
        _lib.check(L.pl_legendre_synth(h, int(spin), a.data_ptr(), shts._ptr(f), phase.data_ptr(), dev.stream_ptr()))
        self._exchange(phase, ncomp, synth=True)
        shape = (ncomp, self.plan.npix) if ncomp == 2 else (self.plan.npix,)
        # gather: every pixel is written, by this rank's ring FFTs or by the unpack of another rank's rings; else the others stay zero
        m = (torch.empty if gather else torch.zeros)(shape, dtype=torch.float64, device=a.device)
        _lib.check(L.pl_phase2map(h, int(spin), phase.data_ptr(), m.data_ptr(), dev.stream_ptr()))
        if gather:
            self._gather_rings(m, ncomp)
        return m

    def map2alm(self, m, spin=0, fl=None):
        """m: device tensor [npix] or [2, npix]; only the pixels of this rank's rings are read"""
        from . import _lib, dev, shts
        L, h = _lib.lib(), self.plan.h
        ncomp = 1 if spin == 0 else 2
        mm = m.to(torch.float64).contiguous()
        assert mm.numel() == ncomp * self.plan.npix
        f = shts._fl_arg(fl, self.lmax, True)
        phase = torch.empty(self.plan.phase_doubles(spin), dtype=torch.float64, device=mm.device)
        _lib.check(L.pl_map2phase(h, int(spin), mm.data_ptr(), phase.data_ptr(), dev.stream_ptr()))
        self._exchange(phase, ncomp, synth=False)
        alm = torch.empty((ncomp, self.plan.nalm) if ncomp == 2 else self.plan.nalm, dtype=torch.complex128, device=mm.device)
        _lib.check(L.pl_legendre_anal(h, int(spin), phase.data_ptr(), alm.data_ptr(), shts._ptr(f), dev.stream_ptr()))
        _lib.check(L.pl_alm_keep_mgroups(self.lmax, ncomp, alm.data_ptr(), self.rank, self.size, dev.stream_ptr()))
        return allreduce_sum(alm)


def allreduce_max(n):
    dist = _dist()
    if dist is None:
        return int(n)
    t = torch.tensor([int(n)], dtype=torch.int64)
    if dist.get_backend() == 'nccl':
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())
