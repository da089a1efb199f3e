"""Device-array plumbing for the QE / CG layers: torch CUDA tensors hold maps (float64) and alms
(complex128) in HBM; the arithmetic is done by the HIP kernels of plancklens_amd/csrc (through shts /
_lib), torch only allocates, copies and launches trivial element-wise glue on its current stream."""
import ctypes

import numpy as np
import torch

from . import _lib
from .hp import Alm

_LIDX = {}


def device():
    if not torch.cuda.is_available():
        raise RuntimeError('no GPU visible: the QE / CG layers of plancklens_amd run on the MI355X only')
    return torch.device('cuda', torch.cuda.current_device())


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def to_dev(x, dtype=None):
    if isinstance(x, torch.Tensor):
        t = x.to(device())
    else:
        t = torch.from_numpy(np.ascontiguousarray(x)).to(device(), non_blocking=False)
    return t if dtype is None else t.to(dtype)


_PINNED_FREE = {}
_COPY_STREAMS = {}
_SPIN = {}


def streams_overlap(a, b):
    """True when kernels launched on streams a and b run at the same time.  The HIP runtime spreads its streams over a few hardware
    queues (four by default); two streams that fell onto one queue run their kernels one after the other however independent they are.
    Measured, not assumed: one spin kernel alone on a, then one on each stream started together -- the pair takes as long as one
    kernel (different queues) or as two (same queue).  Synchronises the device: set-up time only, never during a graph capture."""
    if a.cuda_stream == b.cuda_stream:
        return False
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    done = torch.cuda.Event()
    d = torch.cuda.current_device()

    def alone(cycles):
        torch.cuda.synchronize()
        with torch.cuda.stream(a):
            ev[0].record()
            torch.cuda._sleep(cycles)
            ev[1].record()
        torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1])
    if d not in _SPIN:  # a spin of ~1 ms: long against launch latencies and event granularity
        cycles = 1 << 18
        alone(cycles)  # (first launch of the spin kernel: not timed)
        for _ in range(10):
            if alone(cycles) >= 0.8:
                break
            cycles *= 2
        _SPIN[d] = cycles
    cycles = _SPIN[d]
    t1 = min(alone(cycles), alone(cycles))
    with torch.cuda.stream(a):
        ev[2].record()
    b.wait_event(ev[2])
    with torch.cuda.stream(b):
        torch.cuda._sleep(cycles)
        done.record()
    with torch.cuda.stream(a):
        torch.cuda._sleep(cycles)
        a.wait_event(done)
        ev[3].record()
    torch.cuda.synchronize()
    return ev[2].elapsed_time(ev[3]) < 1.5 * t1


def concurrent_stream(beside=(), tries=8):
    """A torch stream whose kernels overlap those of every stream in `beside` (default: the current stream): up to `tries` streams of
    torch's pool are tried (streams_overlap); the first one if none qualifies -- a stream that does not overlap is still correct."""
    beside = list(beside) or [torch.cuda.current_stream()]
    first = None
    from . import options
    dbg = options.opts.debug
    for i in range(tries):
        s = torch.cuda.Stream()
        first = first or s
        try:
            if all(streams_overlap(o, s) for o in beside):
                if dbg:
                    print('concurrent_stream: candidate %d overlaps' % i, flush=True)
                return s
        except RuntimeError as e:  # (e.g. called while a capture is under way: no probing then)
            if dbg:
                print('concurrent_stream: probe failed: %r' % (e,), flush=True)
            break
    if dbg:
        print('concurrent_stream: none of %d candidates overlaps' % tries, flush=True)
    return first


def _copy_stream():
    d = torch.cuda.current_device()
    if d not in _COPY_STREAMS:
        _COPY_STREAMS[d] = concurrent_stream() if not torch.cuda.is_current_stream_capturing() else torch.cuda.Stream()
    return _COPY_STREAMS[d]


class host_future(object):
    """A device -> host copy in flight into a pinned staging buffer from a small recycled pool (hipHostMalloc of a 33 MB alm costs
    milliseconds and synchronises: never on the per-result path), so the caller's stream goes on with the next reconstruction while
    the result crosses PCIe.  result() returns an ordinary numpy array that owns the data.  Two forms:
      threaded (large results): the copy runs on a copy stream; a helper thread waits for it, moves the data out of the staging
        buffer and hands the buffer back -- the 33 MB host copies stay off the launching thread;
      lazy (threaded=False; small results, where a thread hand-off costs more than the copy): the copy is issued on the caller's
        stream into a pinned tensor of its own and nobody waits for it until result() is called, which returns that memory."""
    MAX_IN_FLIGHT = 8  # (two results per reconstruction: four reconstructions of slack for the launching thread when the helpers are slow -- a busy host)
    _in_flight = []
    _pool = None

    def __init__(self, t, threaded=True):
        from . import shts
        shts.join_lanes()  # results of transforms still running on side lanes
        t = t.detach().contiguous()
        host_future._in_flight[:] = [f for f in host_future._in_flight if not f.done()]
        if not threaded:
            # small result: a pinned tensor of its own from torch's caching host allocator (after warm-up the block of a result the caller has
            # dropped: no hipHostMalloc), filled by an asynchronous copy on the caller's stream; result() hands out that tensor's memory as the
            # numpy array -- no second pass over the data on the launching thread, no staging buffer to give back
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._arr = None
            self._job = None
            self._lazy = (ev, h, t)  # (t: the source stays alive until the copy has been waited for)
            host_future._in_flight.append(self)
            return
        while len([f for f in host_future._in_flight if f._job is not None]) >= host_future.MAX_IN_FLIGHT:  # staging buffers all busy: wait for the oldest copy
            next(f for f in host_future._in_flight if f._job is not None).result()
            host_future._in_flight[:] = [f for f in host_future._in_flight if not f.done()]
        key = (tuple(t.shape), t.dtype)
        if key not in _PINNED_FREE:  # the whole staging pool of this shape at once, on first use (i.e. during warm-up)
            _PINNED_FREE[key] = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for _ in range(host_future.MAX_IN_FLIGHT)]
        free = _PINNED_FREE[key]
        h = free.pop() if free else torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        self._arr = None
        self._job = None
        cs = _copy_stream()
        cs.wait_stream(torch.cuda.current_stream())
        from . import options
        nblk = int(options.opts.d2h_blocks)
        with torch.cuda.stream(cs):
            if nblk > 0 and t.element_size() * t.numel() % 8 == 0 and t.data_ptr() % 16 == 0:
                # pl_copy_slim: a few workgroups writing straight into the pinned (device-mapped) buffer
                _lib.check(_lib.lib().pl_copy_slim(t.data_ptr(), h.data_ptr(), t.element_size() * t.numel() // 8, nblk,
                                                   ctypes.c_void_p(cs.cuda_stream)))
            else:
                h.copy_(t, non_blocking=True)
            ev = torch.cuda.Event(blocking=True)  # the helper thread sleeps on it: a spinning waiter takes a core from the launching thread on a busy host
            ev.record(cs)
        t.record_stream(cs)
        if host_future._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            host_future._pool = ThreadPoolExecutor(max_workers=2, thread_name_prefix='plens_d2h')

        def finish():
            ev.synchronize()
            arr = h.numpy().copy()
            free.append(h)
            return arr
        self._job = host_future._pool.submit(finish)
        host_future._in_flight.append(self)

    def done(self):
        if self._arr is not None:
            return True
        if self._job is not None:
            return self._job.done()
        return self._lazy is not None and self._lazy[0].query()  # lazy: the copy has landed (result() then costs nothing)

    def result(self):
        if self._arr is None:
            if self._job is not None:
                self._arr = self._job.result()
            else:
                ev, h, _ = self._lazy
                ev.synchronize()
                self._arr = h.numpy()  # (a view: the array keeps the pinned tensor alive and torch recycles its block afterwards)
                self._lazy = None
        return self._arr


def _release_staging():
    """pending copies, pinned buffers and copy streams go before the interpreter (and the HIP runtime) shut down"""
    try:
        for f in list(host_future._in_flight):
            f.result()
        del host_future._in_flight[:]
        if host_future._pool is not None:
            host_future._pool.shutdown(wait=True)
            host_future._pool = None
        _PINNED_FREE.clear()
        _COPY_STREAMS.clear()
    except Exception:
        pass


import atexit  # noqa: E402
atexit.register(_release_staging)


def to_host(t):
    """Device tensor -> numpy array (blocking).  The array is a view of a pinned tensor of its own (torch's host allocator
    recycles the pinned blocks of arrays that have been dropped): one PCIe copy, no second pass over the data."""
    if not isinstance(t, torch.Tensor):
        return np.asarray(t)
    if not t.is_cuda:
        return t.detach().numpy()
    from . import shts
    shts.join_lanes()  # results of transforms still running on side lanes
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t.detach(), non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return h.numpy()


def resolve(x):
    """numpy array of a result that may still be a host_future"""
    return x.result() if isinstance(x, host_future) else x


def lidx(lmax):
    """Device int64 tensor: l of every entry of a healpy alm array."""
    key = (lmax, torch.cuda.current_device())
    if key not in _LIDX:
        ls = np.concatenate([np.arange(m, lmax + 1, dtype=np.int64) for m in range(lmax + 1)])
        _LIDX[key] = torch.from_numpy(ls).to(device())
    return _LIDX[key]


_FL_CACHE = {}


def fl_dev(fl, lmax):
    """l-filter as a device float64 tensor of length lmax + 1 (zero-extended / truncated: hp.almxfl semantics).
    Host arrays are uploaded once per content: the same few filters (beams, C_l, 1 / C_l, l-weights) are applied
    thousands of times, and every pageable upload is a blocking copy that idles the GPU.  The cached tensors are
    read-only by convention (callers never modify a filter in place)."""
    if isinstance(fl, torch.Tensor):
        f = torch.zeros(lmax + 1, dtype=torch.float64, device=device())
        n = min(lmax + 1, fl.numel())
        f[:n] = fl[:n].to(device())
        return f
    a = np.zeros(lmax + 1, dtype=np.float64)
    src = np.asarray(fl, dtype=np.float64)
    n = min(lmax + 1, src.size)
    a[:n] = src[:n]
    key = (lmax, torch.cuda.current_device(), hash(a.tobytes()))
    hit = _FL_CACHE.get(key)
    if hit is not None and hit[0].shape == a.shape and np.array_equal(hit[0], a):
        return hit[1]
    # no eviction: a captured HIP graph (qcinv.multigrid) holds the raw pointers of the filters it was recorded with, and a
    # filter is 8 (lmax + 1) bytes -- thousands of distinct ones are still only tens of MB
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError('l-filter upload requested while a HIP graph is being captured (filter not seen during warm-up)')
    t = torch.from_numpy(a).to(device())
    _FL_CACHE[key] = (a, t)
    return t


def bshape(a):
    """(nb, n) of a CG vector: a 1-D tensor is one array of n entries, a 2-D tensor [nb, n] a block of nb arrays (several
    right-hand sides solved together, the `_b` entry points of include/plshts.h)"""
    assert a.dim() in (1, 2), a.shape
    return (1, a.shape[0]) if a.dim() == 1 else (a.shape[0], a.shape[1])


def _same_block(*ts):
    nb = bshape(ts[0])[0]
    for t in ts:
        assert t.is_contiguous() and t.dim() == ts[0].dim() and bshape(t)[0] == nb, [tuple(x.shape) for x in ts]
    return nb


def almxfl(alm, fl):
    """hp.almxfl on a device alm (or block of alms) through the C ABI (pl_almxfl / pl_almxfl_b); returns a new tensor."""
    nb, n = bshape(alm)
    lmax = Alm.getlmax(n)
    assert lmax >= 0
    f = fl_dev(fl, lmax)
    alm = alm.contiguous()
    out = torch.empty_like(alm)
    if alm.dim() == 1:
        _lib.check(_lib.lib().pl_almxfl(lmax, alm.data_ptr(), f.data_ptr(), lmax + 1, out.data_ptr(), stream_ptr()))
    else:
        _lib.check(_lib.lib().pl_almxfl_b(lmax, nb, alm.data_ptr(), f.data_ptr(), lmax + 1, out.data_ptr(), stream_ptr()))
    return out


def alm_copy(alm, lmax_out):
    nb, n = bshape(alm)
    lmax_in = Alm.getlmax(n)
    if lmax_out == lmax_in:
        return alm.clone()
    alm = alm.contiguous()
    if alm.dim() == 1:
        out = torch.empty(Alm.getsize(lmax_out), dtype=torch.complex128, device=alm.device)
        _lib.check(_lib.lib().pl_alm_copy(lmax_in, alm.data_ptr(), lmax_out, out.data_ptr(), stream_ptr()))
    else:
        out = torch.empty((nb, Alm.getsize(lmax_out)), dtype=torch.complex128, device=alm.device)
        _lib.check(_lib.lib().pl_alm_copy_b(lmax_in, nb, alm.data_ptr(), lmax_out, out.data_ptr(), stream_ptr()))
    return out


def alm_splice(alm_lo, alm_hi, lsplit):
    """alm_lo for l <= lsplit, alm_hi above; band-limit of alm_hi (pl_alm_splice)."""
    return alm_splice_fl(alm_lo, alm_hi, None, lsplit)


def alm_splice_dot_count(lmax_hi):
    """partial sums per batch entry that alm_splice_fl(dot=...) leaves"""
    return int(_lib.lib().pl_alm_splice_dot_count(int(lmax_hi)))


def alm_splice_fl(alm_lo, alm_hi, fl_hi, lsplit, dot=None):
    """alm_lo for l <= lsplit, fl_hi[l] * alm_hi above (fl_hi None: alm_hi); band-limit of alm_hi (pl_alm_splice / _fl / _b).
    dot = (q, lmin, pre): the kernel also leaves the partial sums of <result, q> in the float64 device tensor `pre`
    ([alm_splice_dot_count] or [nb, ...]; pl_alm_splice_dot_b) -- what cg_axpy_pre takes."""
    nb = _same_block(alm_lo, alm_hi)
    lmax_lo, lmax_hi = Alm.getlmax(bshape(alm_lo)[1]), Alm.getlmax(bshape(alm_hi)[1])
    assert lmax_lo >= lsplit and lmax_hi >= lsplit, (lmax_lo, lmax_hi, lsplit)
    out = torch.empty_like(alm_hi)
    f = None if fl_hi is None else fl_dev(fl_hi, lmax_hi).data_ptr()
    L = _lib.lib()
    if dot is not None:
        q, lmin, pre = dot
        assert q.shape == alm_hi.shape and q.dtype == torch.complex128 and q.is_contiguous() and alm_hi.dtype == torch.complex128
        assert pre.dtype == torch.float64 and pre.is_contiguous() and pre.numel() == nb * alm_splice_dot_count(lmax_hi)
        _lib.check(L.pl_alm_splice_dot_b(lmax_lo, nb, alm_lo.data_ptr(), lmax_hi, alm_hi.data_ptr(), f, int(lsplit), out.data_ptr(), q.data_ptr(), int(lmin),
                                         pre.data_ptr(), stream_ptr()))
    elif alm_hi.dim() == 2:
        _lib.check(L.pl_alm_splice_b(lmax_lo, nb, alm_lo.data_ptr(), lmax_hi, alm_hi.data_ptr(), f, int(lsplit), out.data_ptr(), stream_ptr()))
    elif f is None:
        _lib.check(L.pl_alm_splice(lmax_lo, alm_lo.data_ptr(), lmax_hi, alm_hi.data_ptr(), int(lsplit), out.data_ptr(), stream_ptr()))
    else:
        _lib.check(L.pl_alm_splice_fl(lmax_lo, alm_lo.data_ptr(), lmax_hi, alm_hi.data_ptr(), f, int(lsplit), out.data_ptr(), stream_ptr()))
    return out


def almxfl_add(a, b, fl, out=None):
    """a + f_l b in one pass (pl_almxfl_add); out may be a."""
    nb = _same_block(a, b)
    lmax = Alm.getlmax(bshape(a)[1])
    assert b.shape == a.shape
    f = fl_dev(fl, lmax)
    out = torch.empty_like(a) if out is None else out
    if a.dim() == 1:
        _lib.check(_lib.lib().pl_almxfl_add(lmax, a.data_ptr(), b.data_ptr(), f.data_ptr(), lmax + 1, out.data_ptr(), stream_ptr()))
    else:
        _lib.check(_lib.lib().pl_almxfl_add_b(lmax, nb, a.data_ptr(), b.data_ptr(), f.data_ptr(), lmax + 1, out.data_ptr(), stream_ptr()))
    return out


def alm_lincomb(outputs):
    """[sum_t fl_t alm_t for the (alm, fl) terms of an output] for one or two outputs of one or two terms each, in ONE launch
    (pl_alm_lincomb); returns the rows of one (nout, nalm) tensor -- (G, C) of a spin transform as its synthesis takes them, no stacking
    copy.  Rounded exactly as almxfl of the first term followed by almxfl_add of the second."""
    nout = len(outputs)
    assert 1 <= nout <= 2 and all(1 <= len(o) <= 2 for o in outputs)
    a0 = outputs[0][0][0]
    n = a0.numel()
    lmax = Alm.getlmax(n)
    out = torch.empty((nout, n), dtype=torch.complex128, device=a0.device)
    alms, fls, keep = [None] * (2 * nout), [None] * (2 * nout), []
    for k, terms in enumerate(outputs):
        for t, (alm, fl) in enumerate(terms):
            assert alm.dim() == 1 and alm.numel() == n and alm.dtype == torch.complex128 and alm.is_cuda
            alm = alm.contiguous()
            f = fl_dev(fl, lmax)  # a tensor `fl` yields a fresh temporary: it must outlive the launch like the alms do
            keep += [alm, f]
            alms[2 * k + t], fls[2 * k + t] = alm.data_ptr(), f.data_ptr()
    nterm = (ctypes.c_int * nout)(*[len(o) for o in outputs])
    pa, pf = (ctypes.c_void_p * (2 * nout))(*alms), (ctypes.c_void_p * (2 * nout))(*fls)
    po = (ctypes.c_void_p * nout)(*[out[k].data_ptr() for k in range(nout)])
    _lib.check(_lib.lib().pl_alm_lincomb(lmax, nout, nterm, pa, pf, po, stream_ptr()))
    del keep  # (alive until here: the caching allocator may hand a dropped block to the next fl_dev of the loop above)
    return [out[k] for k in range(nout)]


DOT_PARTS = 64  # PL_DOT_PARTS of include/plshts.h


def alm_dot(pairs, lmin=0):
    """sum over the (a, b) pairs of sum_{l >= lmin} (2l + 1) C_l^{ab}, left on the device as DOT_PARTS partial sums (the
    value is their sum in index order: `float(dot.sum())` on the host, or axpy_dev on the device).  One deterministic
    launch per pair (pl_alm_dot), nothing comes back to the host.  Blocks [nb, nalm]: (nb, DOT_PARTS) partial sums, one scalar
    product per entry (value: .sum(-1))."""
    nb = bshape(pairs[0][0])[0]
    blk = pairs[0][0].dim() == 2
    out = torch.empty((nb, DOT_PARTS) if blk else DOT_PARTS, dtype=torch.float64, device=device())
    for i, (a, b) in enumerate(pairs):
        assert a.shape == b.shape and a.dtype == torch.complex128 and b.dtype == torch.complex128 and _same_block(a, b) == nb
        lmax = Alm.getlmax(bshape(a)[1])
        if blk:
            _lib.check(_lib.lib().pl_alm_dot_b(lmax, int(lmin), nb, a.data_ptr(), b.data_ptr(), int(i > 0), out.data_ptr(), stream_ptr()))
        else:
            _lib.check(_lib.lib().pl_alm_dot(lmax, int(lmin), a.data_ptr(), b.data_ptr(), int(i > 0), out.data_ptr(), stream_ptr()))
    return out


def axpy_dev(y, x, num, den=None, sign=1.0):
    """y += sign * num / den * x in place; num, den: scalar products as returned by alm_dot (pl_axpy_dev); blocks: per entry."""
    assert y.shape == x.shape and y.dtype == x.dtype
    nb = _same_block(y, x)
    assert num.numel() == nb * DOT_PARTS and (den is None or den.numel() == nb * DOT_PARTS)
    n = bshape(y)[1] * (2 if y.is_complex() else 1)
    if y.dim() == 1:
        _lib.check(_lib.lib().pl_axpy_dev(n, num.data_ptr(), None if den is None else den.data_ptr(), float(sign), x.data_ptr(), y.data_ptr(),
                                         stream_ptr()))
    else:
        _lib.check(_lib.lib().pl_axpy_dev_b(n, nb, num.data_ptr(), None if den is None else den.data_ptr(), float(sign), x.data_ptr(),
                                           y.data_ptr(), stream_ptr()))
    return y


_CG_BARRIER = {}


def _ctx_key():
    """(device, plan context of the calling thread): scratch buffers are per solver context (shts.plan_context), so that two solvers
    running at the same time on different streams never share one"""
    from . import shts
    return (torch.cuda.current_device(), shts.context())


def cg_barrier():
    """the per-device, per-context grid-barrier words of pl_cg_dot_axpy (a solver launches on one stream at a time)"""
    d = _ctx_key()
    if d not in _CG_BARRIER:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('CG barrier words requested while a HIP graph is being captured')
        _CG_BARRIER[d] = torch.zeros(4, dtype=torch.int32, device=device())
    return _CG_BARRIER[d]


def cg_barrier_timed_out(reset=False):
    """True if a grid barrier of pl_cg_dot_axpy gave up on this device (results after that are invalid); synchronises.
    reset: the barrier words are zeroed afterwards (a barrier that gave up leaves its arrival counter non-zero)."""
    d = _ctx_key()
    if d not in _CG_BARRIER or torch.cuda.is_current_stream_capturing():
        return False
    out = int(_CG_BARRIER[d][2]) != 0
    if reset and out:
        _CG_BARRIER[d].zero_()
    return out


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def cg_dot_axpy(a, b1, y1, x1, sign1, b2=None, den=None, y2=None, x2=None, sign2=-1.0, lmin=0, one_launch=False, active=None):
    """Scalar products and the updates they scale (pl_cg_dot_axpy: two launches for all fields, or one with a grid barrier inside
    when `one_launch`) over the fields of the lists a, b1, ...:
    parts1 = <a, b1>, parts2 = <a, b2>; c = parts2 / parts1 (b2 given) or parts1 / den; y1 += sign1 c x1, y2 += sign2 c x2.
    Returns (parts1, parts2), device tensors of DOT_PARTS partial sums as alm_dot returns them.
    Blocks ([nb, nalm] fields, pl_cg_dot_axpy_b): everything per entry -- nb scalar products, nb step lengths; `active` (nb float64
    zeros / ones on the device, optional) multiplies the step lengths: an entry with 0 keeps its vectors."""
    nf = len(a)
    for group in (b1, y1, x1, b2, y2, x2):
        assert group is None or len(group) == nf
    nb, blk = bshape(a[0])[0], a[0].dim() == 2
    for k in range(nf):
        for group in (a, b1, y1, x1, b2, y2, x2):
            assert group is None or (group[k].dtype == torch.complex128 and group[k].is_contiguous() and group[k].shape == a[k].shape)
        assert bshape(a[k])[0] == nb and (a[k].dim() == 2) == blk
    lmax = (ctypes.c_int * nf)(*[Alm.getlmax(bshape(t)[1]) for t in a])
    pshape = (nb, DOT_PARTS) if blk else DOT_PARTS
    parts1 = torch.empty(pshape, dtype=torch.float64, device=device())
    parts2 = torch.empty(pshape, dtype=torch.float64, device=device()) if b2 is not None else None
    assert (b2 is None) != (den is None) and (den is None or den.numel() == nb * DOT_PARTS)
    args = (nf, lmax, int(lmin), _ptr_array(a), _ptr_array(b1), None if b2 is None else _ptr_array(b2),
            parts1.data_ptr(), None if parts2 is None else parts2.data_ptr(),
            None if den is None else den.data_ptr(), _ptr_array(y1), _ptr_array(x1), float(sign1),
            None if y2 is None else _ptr_array(y2), None if x2 is None else _ptr_array(x2), float(sign2))
    if blk:
        assert active is None or (active.numel() == nb and active.dtype == torch.float64 and active.is_cuda)
        _lib.check(_lib.lib().pl_cg_dot_axpy_b(nb, *args, None if active is None else active.data_ptr(), stream_ptr()))
    else:
        assert active is None, 'per-entry stopping is a block-vector feature'
        _lib.check(_lib.lib().pl_cg_dot_axpy(*args, cg_barrier().data_ptr() if one_launch else None, stream_ptr()))
    return parts1, parts2


def cg_axpy_pre(pre, y1, x1, sign1, den=None, y2=None, x2=None, sign2=-1.0, active=None, assign_y1=False):
    """The updates of cg_dot_axpy from scalar products that the kernel producing the vector left as partial sums (pl_cg_axpy_pre_b):
    pre = (pre1, pre2) or (pre1, None), [npre] or [nb, npre] each.  den given: c = sum(pre1) / sum(den); else c = sum(pre2) / sum(pre1);
    y1 += sign1 c x1 (assign_y1: y1 = sign1 c x1, y1 is not read), y2 += sign2 c x2.  Returns (parts1, parts2): the totals of pre1 / pre2 as
    DOT_PARTS partial sums (alm_dot's form)."""
    pre1, pre2 = pre
    nf = len(y1)
    for group in (x1, y2, x2):
        assert group is None or len(group) == nf
    nb, blk = bshape(y1[0])[0], y1[0].dim() == 2
    for k in range(nf):
        for group in (y1, x1, y2, x2):
            assert group is None or (group[k].dtype == torch.complex128 and group[k].is_contiguous() and group[k].shape == y1[k].shape)
    assert (pre2 is None) != (den is None) and (den is None or den.numel() == nb * DOT_PARTS)
    npre = pre1.shape[-1]
    for t in (pre1, pre2):
        assert t is None or (t.dtype == torch.float64 and t.is_contiguous() and t.numel() == nb * npre)
    lmax = (ctypes.c_int * nf)(*[Alm.getlmax(bshape(t)[1]) for t in y1])
    pshape = (nb, DOT_PARTS) if blk else DOT_PARTS
    parts1 = torch.empty(pshape, dtype=torch.float64, device=device())
    parts2 = torch.empty(pshape, dtype=torch.float64, device=device()) if pre2 is not None else None
    assert active is None or (blk and active.numel() == nb and active.dtype == torch.float64 and active.is_cuda)
    _lib.check(_lib.lib().pl_cg_axpy_pre_b(nb, nf, lmax, int(npre), pre1.data_ptr(), None if pre2 is None else pre2.data_ptr(),
                                           None if den is None else den.data_ptr(), parts1.data_ptr(), None if parts2 is None else parts2.data_ptr(),
                                           _ptr_array(y1), _ptr_array(x1), float(sign1), None if y2 is None else _ptr_array(y2),
                                           None if x2 is None else _ptr_array(x2), float(sign2), None if active is None else active.data_ptr(),
                                           int(bool(assign_y1)), stream_ptr()))
    return parts1, parts2


TEMPLATE_MAX_MODES = 16  # PL_TEMPLATE_MAX_MODES of include/plshts.h
_TPROJ_SCRATCH = {}


def tproj_scratch(nb=1):
    """the per-device scratch of pl_template_project / pl_cg_fwd_tt (nb entries of a batch: nb times the size; it only grows, and an
    outgrown one is kept alive -- a captured HIP graph may hold its address)"""
    d = _ctx_key()
    cur = _TPROJ_SCRATCH.get(d)
    if cur is None or cur[-1].numel() < nb * TEMPLATE_MAX_MODES * 256:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('template-projection scratch requested while a HIP graph is being captured')
        _TPROJ_SCRATCH.setdefault(d, []).append(torch.empty(nb * TEMPLATE_MAX_MODES * 256, dtype=torch.float64, device=device()))
    return _TPROJ_SCRATCH[d][-1]


def template_project(tmap, n_inv, pmat, rmat):
    """tmap <- n_inv tmap - rmat^t (pmat (n_inv tmap)) in place, two launches (pl_template_project); pmat, rmat: (nmodes, npix);
    tmap [npix] or a block [nb, npix]."""
    nmodes, npix = pmat.shape
    nb, n = bshape(tmap)
    assert rmat.shape == pmat.shape and n == npix and n_inv.numel() == npix and pmat.is_contiguous() and rmat.is_contiguous() and tmap.is_contiguous()
    if tmap.dim() == 1:
        _lib.check(_lib.lib().pl_template_project(npix, nmodes, tmap.data_ptr(), n_inv.data_ptr(), pmat.data_ptr(), rmat.data_ptr(),
                                                 tproj_scratch().data_ptr(), stream_ptr()))
    else:
        _lib.check(_lib.lib().pl_template_project_b(npix, nmodes, nb, tmap.data_ptr(), n_inv.data_ptr(), pmat.data_ptr(), rmat.data_ptr(),
                                                   tproj_scratch(nb).data_ptr(), stream_ptr()))
    return tmap


def lowrank_update(y, x, pmat, rmat):
    """y <- y - rmat^t (pmat x) in place (pl_lowrank_update_b): x, y alm (complex128) or real vectors, or blocks [nb, .] of them, read as
    real vectors of n doubles; pmat, rmat: real (nmodes, n) device matrices.  The harmonic-space form of the template projection."""
    nmodes, n = pmat.shape
    nb, nx = bshape(x)
    if x.is_complex():
        nx *= 2
    assert rmat.shape == pmat.shape and nx == n and y.shape == x.shape and y.dtype == x.dtype and x.is_contiguous() and y.is_contiguous()
    assert pmat.is_contiguous() and rmat.is_contiguous() and pmat.dtype == torch.float64 and 1 <= nmodes <= TEMPLATE_MAX_MODES
    _lib.check(_lib.lib().pl_lowrank_update_b(n, nmodes, nb, x.data_ptr(), y.data_ptr(), pmat.data_ptr(), rmat.data_ptr(),
                                             tproj_scratch(nb).data_ptr(), stream_ptr()))
    return y


_TPROJ_MD_SCRATCH = {}


def tproj_md_scratch(ndoubles):
    """the per-device scratch of pl_template_project_md_b / pl_cg_fwd_tt_md_b (grows; outgrown buffers stay alive for captured graphs)"""
    d = _ctx_key()
    cur = _TPROJ_MD_SCRATCH.get(d)
    if cur is None or cur[-1].numel() < ndoubles:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('template-projection scratch requested while a HIP graph is being captured')
        _TPROJ_MD_SCRATCH.setdefault(d, []).append(torch.empty(int(ndoubles), dtype=torch.float64, device=device()))
    return _TPROJ_MD_SCRATCH[d][-1]


def template_project_md(tmap, n_inv, nside, lmax, pinv):
    """tmap <- n_inv tmap with monopole and dipole projected out, the templates evaluated from the ring geometry (pl_template_project_md_b);
    tmap [npix] or a block [nb, npix]; pinv: (P^t N^-1 P)^-1 as a (4, 4) device tensor.  Any plan of this nside serves (geometry only)."""
    from . import shts
    plan = shts.geometry_plan(nside, lmax)
    nb, n = bshape(tmap)
    assert n == plan.npix and n_inv.numel() == n and tmap.is_contiguous() and n_inv.is_contiguous() and pinv.numel() == 16 and pinv.is_contiguous()
    L = _lib.lib()
    scratch = tproj_md_scratch(L.pl_template_md_scratch_doubles(plan.h, nb))
    _lib.check(L.pl_template_project_md_b(plan.h, nb, tmap.data_ptr(), n_inv.data_ptr(), pinv.data_ptr(), scratch.data_ptr(), stream_ptr()))
    return tmap


def gemv(amat, x, out=None):
    """y = A x on the device (pl_gemv): A a contiguous (nrows, ncols) float64 tensor, x float64 of ncols entries -- or a block
    [nb, ncols] -> [nb, nrows] (pl_gemv_b: the matrix is read once for all right-hand sides)."""
    assert amat.dim() == 2 and amat.is_contiguous() and amat.dtype == torch.float64 and x.dtype == torch.float64
    x = x.contiguous()
    nb, n = bshape(x)
    assert n == amat.shape[1], (x.shape, amat.shape)
    if out is None:
        out = torch.empty(amat.shape[0] if x.dim() == 1 else (nb, amat.shape[0]), dtype=torch.float64, device=amat.device)
    if x.dim() == 1:
        _lib.check(_lib.lib().pl_gemv(amat.shape[0], amat.shape[1], amat.shape[1], amat.data_ptr(), x.data_ptr(), out.data_ptr(), stream_ptr()))
    else:
        _lib.check(_lib.lib().pl_gemv_b(amat.shape[0], amat.shape[1], amat.shape[1], amat.data_ptr(), nb, x.data_ptr(), out.data_ptr(), stream_ptr()))
    return out


_SPLIT_MAP = {}


def gemv_split(amat, alms_hi, lmax_lo, fls_hi, dot=None):
    """pre_op_split in one launch (pl_gemv_split) for the fields alms_hi (one tensor, or two: E and B) with their high-l filters fls_hi:
    per field [rows of amat ([fields] truncated to lmax_lo) for l <= lmax_lo | fl_hi alm_hi above], band-limit of the inputs.
    amat: pre_op_dense's flat matrix for these fields at lmax_lo.  Returns the list of output tensors.
    dot = (qs, lmin): also the partial sums of sum_f <out[f], qs[f]> (pl_gemv_split_dot); returns (outs, pre)."""
    nf = len(alms_hi)
    lmax_hi = Alm.getlmax(alms_hi[0].shape[0])
    n = nf * (lmax_lo + 1) * (lmax_lo + 2)
    for a in alms_hi:
        assert a.dim() == 1 and a.is_contiguous() and a.dtype == torch.complex128 and a.shape == alms_hi[0].shape and lmax_hi > lmax_lo
    assert nf in (1, 2) and len(fls_hi) == nf and amat.shape == (n, n) and amat.is_contiguous() and amat.dtype == torch.float64
    key = (lmax_lo, lmax_hi, torch.cuda.current_device())
    if key not in _SPLIT_MAP:  # positions of the lmax_lo entries in the lmax_hi layout
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('index map requested while a HIP graph is being captured')
        m = np.concatenate([np.full(lmax_lo + 1 - mm, mm) for mm in range(lmax_lo + 1)])
        l = np.concatenate([np.arange(mm, lmax_lo + 1) for mm in range(lmax_lo + 1)])
        _SPLIT_MAP[key] = torch.from_numpy((m * (2 * lmax_hi + 1 - m) // 2 + l).astype(np.int32)).to(device())
    outs = [torch.empty_like(a) for a in alms_hi]
    fls = [fl_dev(f, lmax_hi) for f in fls_hi]
    if dot is not None:
        qs, lmin = dot
        assert len(qs) == nf and all(q.shape == alms_hi[0].shape and q.dtype == torch.complex128 and q.is_contiguous() for q in qs)
        pre = torch.empty(_lib.lib().pl_gemv_split_dot_count(nf, int(lmax_lo), int(lmax_hi)), dtype=torch.float64, device=device())
        _lib.check(_lib.lib().pl_gemv_split_dot(nf, int(lmax_lo), int(lmax_hi), n, amat.data_ptr(), _ptr_array(alms_hi), _SPLIT_MAP[key].data_ptr(),
                                                _ptr_array(fls), _ptr_array(outs), _ptr_array(qs), int(lmin), pre.data_ptr(), stream_ptr()))
        return outs, pre
    _lib.check(_lib.lib().pl_gemv_split(nf, int(lmax_lo), int(lmax_hi), n, amat.data_ptr(), _ptr_array(alms_hi), _SPLIT_MAP[key].data_ptr(),
                                        _ptr_array(fls), _ptr_array(outs), stream_ptr()))
    return outs


def add_(y, x):
    """y += x in place for two same-shape float64 / complex128 device tensors, one launch of pl_axpy (y = 1.0 x + y: exact); the running
    sums of the mean-field loop -- the framework's element-wise complex128 add runs at a quarter of the streaming rate"""
    assert y.shape == x.shape and y.dtype == x.dtype and y.is_contiguous() and x.is_contiguous() and y.dtype in (torch.float64, torch.complex128)
    n = y.numel() * (2 if y.is_complex() else 1)
    _lib.check(_lib.lib().pl_axpy(n, 1.0, x.data_ptr(), y.data_ptr(), y.data_ptr(), stream_ptr()))
    return y


def alm2cl(a, b=None):
    lmax = Alm.getlmax(a.numel())
    out = torch.empty(lmax + 1, dtype=torch.float64, device=a.device)
    _lib.check(_lib.lib().pl_alm2cl(lmax, a.data_ptr(), (a if b is None else b).data_ptr(), out.data_ptr(), stream_ptr()))
    return out


def map_mul(a, b, out=None):
    out = torch.empty_like(a) if out is None else out
    _lib.check(_lib.lib().pl_map_mul(a.numel(), a.data_ptr(), b.data_ptr(), out.data_ptr(), stream_ptr()))
    return out


def map_qu_weight(qmap, umap, nqq, nqu, nuu):
    """(Q, U) <- (nqq Q + nqu U, nqu Q + nuu U) in place, one launch (pl_map_qu_weight)"""
    for t in (qmap, umap, nqq, nqu, nuu):
        assert t.is_contiguous() and t.dtype == torch.float64 and t.numel() == qmap.numel()
    _lib.check(_lib.lib().pl_map_qu_weight(qmap.numel(), qmap.data_ptr(), umap.data_ptr(), nqq.data_ptr(), nqu.data_ptr(), nuu.data_ptr(),
                                          stream_ptr()))


def map_cmul(ar, ai, s1, br, bi, s2, sign, outr, outi, accumulate):
    """(outr + i outi) (+)= sign (ar + i s1 ai)(br + i s2 bi)"""
    _lib.check(_lib.lib().pl_map_cmul(ar.numel(), ar.data_ptr(), ai.data_ptr(), float(s1), br.data_ptr(), bi.data_ptr(),
                                     float(s2), float(sign), outr.data_ptr(), outi.data_ptr(), int(accumulate), stream_ptr()))


def qe_lens_product(tpart, ppart):
    """(re, im) of (rep - i imp)(g3 + i c3) - (rep + i imp)(g1 - i c1) + tmap (gt + i ct) in one pass (pl_qe_lens_product).
    tpart = (tmap, gt, ct) or None; ppart = (rep, imp, g3, c3, g1, c1) or None."""
    ref = tpart[0] if tpart is not None else ppart[0]
    out = torch.empty((2, ref.numel()), dtype=torch.float64, device=ref.device)
    ptr = lambda t: t.data_ptr()
    targs = [ptr(t) for t in tpart] if tpart is not None else [None] * 3
    pargs = [ptr(t) for t in ppart] if ppart is not None else [None] * 6
    _lib.check(_lib.lib().pl_qe_lens_product(ref.numel(), *targs, *pargs, out[0].data_ptr(), out[1].data_ptr(), stream_ptr()))
    return out[0], out[1]
