"""MI355X spherical harmonic transforms behind the reference's SHT seam.

Mirrors plancklens/shts.py:12-35 (alm2map, map2alm, alm2map_spin, map2alm_spin with healpy's signatures and
semantics) and the direct hp.* SHT call shapes used by the reference (qest.py:259,514, opfilt_tp.py:276,281).
The arithmetic runs in the hand-written HIP kernels of plancklens_amd/csrc through the C ABI of
include/plshts.h; there is no CPU fallback (a missing library or GPU raises).

numpy in -> numpy out (host pointers, synchronous: the reference's blocking call semantics, inputs are never
modified).  torch CUDA tensors in -> torch CUDA tensors out (device pointers, asynchronous on torch's current
stream): used by the device-resident QE / CG layers.
"""
import ctypes
import threading

import numpy as np

from . import _lib
from .hp import Alm, npix2nside

try:  # torch is only plumbing for device memory / streams
    import torch
except ImportError:  # pragma: no cover
    torch = None

_PLANS = {}


class Plan(object):
    """One (nside, lmax) plan of the HIP engine (ring geometry, recursion and FFT tables, workspaces)."""

    def __init__(self, nside, lmax, shard=None, opts=()):
        """shard = (rank, nranks): the plan of one rank of a transform sharded by m-group / ring pair (pl_plan_create_shard).
        opts: ((name, value), ...) of pl_plan_opts fields (see `plan_options`); empty = the library's defaults."""
        L = _lib.lib()
        if _lib.device_count() < 1:
            raise RuntimeError('no HIP device visible: plancklens_amd.shts has no CPU path')
        h = ctypes.c_void_p()
        o = _lib.PlanOpts(0, -1, -1, -1, -1, -1)
        for k, v in dict(opts).items():
            assert hasattr(o, k), 'unknown plan option %s' % k
            setattr(o, k, int(v))
        rank, nranks = (0, 1) if shard is None else (int(shard[0]), int(shard[1]))
        _lib.check(L.pl_plan_create_opts(int(nside), int(lmax), rank, nranks, ctypes.byref(o), ctypes.byref(h)))
        self.shard = shard
        self.opts = tuple(sorted(dict(opts).items()))
        self.h = h
        self.nside, self.lmax = int(nside), int(lmax)
        self.npix = int(L.pl_plan_npix(h))
        self.nalm = int(L.pl_plan_nalm(h))

    def fork(self, i):
        """Execution context i on the same tables (own workspaces and side streams, pl_plan_fork): transforms issued on
        different forks and streams may overlap.  Fork 0 is the plan itself."""
        if i == 0:
            return self
        if getattr(self, '_parent', None) is not None:  # a fork of a fork is another fork of the original plan
            return self._parent.fork((getattr(self, '_fork_key', None), i))
        forks = self.__dict__.setdefault('_forks', {})
        if i not in forks:
            h = ctypes.c_void_p()
            _lib.check(_lib.lib().pl_plan_fork(self.h, ctypes.byref(h)))
            f = Plan.__new__(Plan)
            f.h, f.nside, f.lmax, f.npix, f.nalm, f._parent = h, self.nside, self.lmax, self.npix, self.nalm, self
            f.shard, f._fork_key = self.shard, i
            if getattr(self, '_profiling', False):
                f.profile(True)
            forks[i] = f
        return forks[i]

    def __del__(self):
        try:
            for f in list(self.__dict__.get('_forks', {}).values()):  # forks go before the tables they use
                f.__del__()
            self.__dict__.get('_forks', {}).clear()
            if getattr(self, 'h', None):
                _lib.lib().pl_plan_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def bytes(self):
        return int(_lib.lib().pl_plan_bytes(self.h))

    def phase_doubles(self, spin):
        return int(_lib.lib().pl_plan_phase_doubles(self.h, int(spin)))

    PROFILE_KINDS = ('leg_synth0', 'leg_synths', 'leg_anal0', 'leg_anals', 'fft_synth', 'fft_anal', 'leg_synths_grad', 'leg_synths_pair',
                     'leg_synths_batch2', 'leg_synth0_pair')

    def profile(self, on=True):
        self._profiling = bool(on)
        _lib.check(_lib.lib().pl_profile_enable(self.h, int(on)))
        for f in self.__dict__.get('_forks', {}).values():
            f.profile(on)

    def profile_read(self):
        """{kind: (total ms, launches)} of the HIP-event timings recorded since the last read (forks included)."""
        ms = (ctypes.c_double * len(self.PROFILE_KINDS))()
        cnt = (ctypes.c_int64 * len(self.PROFILE_KINDS))()
        _lib.check(_lib.lib().pl_profile_read(self.h, ms, cnt))
        ret = {k: (ms[i], int(cnt[i])) for i, k in enumerate(self.PROFILE_KINDS)}
        for f in self.__dict__.get('_forks', {}).values():
            for k, (m_, c_) in f.profile_read().items():
                ret[k] = (ret[k][0] + m_, ret[k][1] + c_)
        return ret


def _plan_key(nside, lmax):
    """A plan lives on the device that was current when it was created (tables, workspaces, side streams)."""
    dev_id = torch.cuda.current_device() if (torch is not None and torch.cuda.is_available()) else 0
    return (int(nside), int(lmax), dev_id)


# ---- plan options: the ring-FFT routing of the plans created inside a `plan_options` block --------------------------------
_OPTS = ()


class plan_options(object):
    """`with shts.plan_options(fft_legacy=1):` -- transforms issued inside use plans created with these pl_plan_opts fields
    (fft_legacy, fft_split_min, fft_nyq_min, fft_min_fast, fft_generic_nside, seed_tables; include/plshts.h).  Plans are cached per option set, so the default plans of
    the process are untouched: the fast-vs-generic tests compare two plans of one grid this way.  Options are explicit arguments of
    pl_plan_create_opts -- the library reads no environment variable at plan creation."""

    def __init__(self, **opts):
        self.opts = tuple(sorted((k, int(v)) for k, v in opts.items()))

    def __enter__(self):
        global _OPTS
        self.prev = _OPTS
        _OPTS = self.opts
        return self

    def __exit__(self, *exc):
        global _OPTS
        _OPTS = self.prev
        return False


# ---- plan contexts: independent solvers of one process on different streams ---------------------------------------------
_CTX = threading.local()
_PLANS_LOCK = threading.RLock()


def context():
    """The plan context of the calling thread (0 unless inside `plan_context`)."""
    return getattr(_CTX, 'i', 0)


class plan_context(object):
    """`with shts.plan_context(i):` -- every plan handed out inside (get_plan, and with it every transform and one-call CG operator)
    is fork i of the shared plan: same geometry / recursion / FFT tables, own workspaces, phase buffer and ring-FFT side streams.
    Two solvers that run at the same time on different streams of one process -- cinv_t and cinv_p of a simulation
    (filt_cinv.library_cinv_sepTP.filter_sims) -- each work inside a context of their own, so that neither the eager launches nor
    the HIP graphs captured there (which hold workspace addresses) share a buffer.  The context is a property of the thread;
    the per-device scratch buffers of plancklens_amd.dev are keyed by it as well.  Context 0 is the plain plan."""

    def __init__(self, i):
        self.i = int(i)

    def __enter__(self):
        self.prev = context()
        _CTX.i = self.i
        return self

    def __exit__(self, *exc):
        _CTX.i = self.prev
        return False


def get_plan(nside, lmax):
    key = _plan_key(nside, lmax)
    if _OPTS:
        key = key + (('opts',) + _OPTS,)
    with _PLANS_LOCK:
        if key not in _PLANS:
            _PLANS[key] = Plan(key[0], key[1], opts=_OPTS)
        c = context()
        return _PLANS[key] if c == 0 else _PLANS[key].fork(('ctx', c))


def geometry_plan(nside, lmax):
    """A plan of this nside for a caller that only needs the ring geometry (pl_template_project_md_b): an existing plan of any
    band-limit serves -- the one with the largest lmax already built in this context is returned, so that a geometry-only need never
    builds (or, inside a captured region, allocates) a second set of recursion tables and workspaces; (nside, lmax) is created only
    when the grid has no plan yet."""
    dev_id = _plan_key(nside, lmax)[2]
    with _PLANS_LOCK:
        have = [k for k in _PLANS if len(k) == 3 and k[0] == int(nside) and k[2] == dev_id] if not _OPTS else []
    if have:
        return get_plan(nside, max(k[1] for k in have))
    return get_plan(nside, lmax)


def get_shard_plan(nside, lmax, rank, nranks):
    key = _plan_key(nside, lmax) + ('shard', int(rank), int(nranks))
    if key not in _PLANS:
        _PLANS[key] = Plan(nside, lmax, shard=(int(rank), int(nranks)))
    return _PLANS[key]


def clear_plans():
    _PLANS.clear()


# ---- lanes: the ring-FFT stage of independent transforms on side streams ------------------------------------------------
_LANE = None
_LANE_STREAMS = {}
_LANE_PHASE = {}
_LANE_USED = set()


class lane(object):
    """`with shts.lane(i):` -- device-tensor transforms issued inside are split into their two stages: the FMA-bound
    Legendre stage stays on the current stream (so the Legendre kernels of successive transforms run back to back), the
    latency-bound ring-FFT stage goes to side stream i with fork i of the plan and a phase buffer of its own.  The FFT
    stage of one transform then overlaps with the Legendre stage of the next.  The maps returned by syntheses issued in
    a lane are complete only after `shts.join_lanes()`.  Analyses are not split here (their FFT stage comes first: to
    overlap it with another transform both FFT stages must be issued before either Legendre stage, see map2alm_many).
    Lane 0 = no splitting."""

    def __init__(self, i):
        self.i = int(i)

    def __enter__(self):
        global _LANE
        self.prev = _LANE
        _LANE = self.i
        return self

    def __exit__(self, *exc):
        global _LANE
        _LANE = self.prev
        return False


def _lane_stream(i):
    key = (i, torch.cuda.current_device())
    if key not in _LANE_STREAMS:
        _LANE_STREAMS[key] = torch.cuda.Stream()
    return _LANE_STREAMS[key]


def _lane_phase(plan, i, spin):
    """phase buffer of lane i (kept for the life of the process: a lane is used for the same few transform shapes)"""
    n = plan.phase_doubles(spin)
    key = (i, plan.nside, plan.lmax, torch.cuda.current_device())
    buf = _LANE_PHASE.get(key)
    if buf is None or buf.numel() < n:
        buf = torch.empty(n, dtype=torch.float64, device='cuda')
        _LANE_PHASE[key] = buf
    return buf


def _lane_active():
    return _LANE is not None and _LANE != 0 and not torch.cuda.is_current_stream_capturing()


def join_lanes():
    """the current stream waits for the FFT stages still running on the side lanes"""
    if not _LANE_USED:
        return
    main = torch.cuda.current_stream()
    for i in sorted(_LANE_USED):
        main.wait_stream(_lane_stream(i))
    _LANE_USED.clear()


def _is_dev(x):
    return torch is not None and isinstance(x, torch.Tensor)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(a):
    if a is None:
        return None
    if _is_dev(a):
        return ctypes.c_void_p(a.data_ptr())
    return ctypes.c_void_p(a.ctypes.data)


def _fl_arg(fl, lmax, dev):
    """l-filter zero-extended / truncated to lmax + 1 entries (hp.almxfl semantics)."""
    if fl is None:
        return None
    if dev:
        from . import dev as _dev
        return _dev.fl_dev(fl, lmax)  # cached upload of host filters
    f = np.zeros(lmax + 1, dtype=np.float64)
    fl = np.asarray(fl, dtype=np.float64)
    n = min(lmax + 1, fl.size)
    f[:n] = fl[:n]
    return f


def _synth(spin, alm, nside, lmax, fl=None, grad_only=False):
    """alm: (nalm,) or (2, nalm) complex128 -> map (npix,) or (2, npix) float64."""
    dev = _is_dev(alm)
    plan = get_plan(nside, lmax)
    ncomp = 1 if spin == 0 else 2
    f = _fl_arg(fl, lmax, dev)
    L = _lib.lib()
    fn = L.pl_alm2map_grad if grad_only else L.pl_alm2map
    nin = 1 if grad_only else ncomp
    if dev:
        a = alm.to(torch.complex128).contiguous()
        assert a.numel() == nin * plan.nalm, (a.shape, plan.nalm)
        out = torch.empty((ncomp, plan.npix) if ncomp == 2 else (plan.npix,), dtype=torch.float64, device=a.device)
        if _lane_active():  # Legendre stage here, ring FFTs on the side lane
            i, ls = _LANE, _lane_stream(_LANE)
            ph = _lane_phase(plan, i, spin)
            # the lane's previous FFT stage must have consumed the phase buffer before it is overwritten
            torch.cuda.current_stream().wait_stream(ls)
            leg = L.pl_legendre_synth_grad if grad_only else L.pl_legendre_synth
            _lib.check(leg(plan.h, spin, _ptr(a), _ptr(f), _ptr(ph), _stream()))
            ls.wait_stream(torch.cuda.current_stream())
            out.record_stream(ls)
            _lib.check(L.pl_phase2map(plan.fork(i).h, spin, _ptr(ph), _ptr(out), ctypes.c_void_p(ls.cuda_stream)))
            _LANE_USED.add(i)
            return out
        _lib.check(fn(plan.h, spin, _ptr(a), _ptr(out), _ptr(f), _lib.PL_DEVICE, _stream()))
        return out
    a = np.ascontiguousarray(alm, dtype=np.complex128)
    assert a.size == nin * plan.nalm, (a.shape, plan.nalm)
    out = np.empty((ncomp, plan.npix) if ncomp == 2 else (plan.npix,), dtype=np.float64)
    _lib.check(fn(plan.h, spin, _ptr(a), _ptr(out), _ptr(f), _lib.PL_HOST, None))
    return out


def lane_active():
    """True inside a `lane` context: transforms are split over two streams there, and the one-call operators are not used."""
    return _lane_active()


def plan_all_generic(nside, lmax):
    """True when every ring of this grid runs in the generic ring-FFT kernel (the coarse levels of the CG chains): the CG operator
    then folds the stored-map template projection into its two FFT launches"""
    return bool(_lib.lib().pl_plan_fft_all_generic(get_plan(nside, lmax).h))


class post_dots(object):
    """Request for the scalar products <d, q> and <d, r> of the CG step (cd_solve.py:66-84), handed to cg_fwd_tt / cg_fwd_pp (`dots=`): d is the
    operator's input, q its result, r the listed fields (laid out as the result).  The kernel that writes q forms them on its way out
    (pl_plan_arm_post_dots); afterwards `pre` = (pre1, pre2), per batch entry the partial sums of <d, q> and <d, r> that dev.cg_axpy_pre takes."""

    def __init__(self, r, lmin=0):
        self.r, self.lmin, self.pre = list(r), int(lmin), None


def _arm_post_dots(plan, dots, d, nb):
    assert len(d) == len(dots.r)
    for t in list(d) + dots.r:
        assert t.dtype == torch.complex128 and t.is_contiguous() and t.shape == d[0].shape and t.is_cuda, 'post_dots: fields laid out as the operator result'
    L = _lib.lib()
    npre = L.pl_post_dots_count(plan.h)
    pre = torch.empty((2, nb, npre) if d[0].dim() == 2 else (2, npre), dtype=torch.float64, device=d[0].device)
    nf = len(d)
    pd = (ctypes.c_void_p * nf)(*[t.data_ptr() for t in d])
    pr = (ctypes.c_void_p * nf)(*[t.data_ptr() for t in dots.r])
    _lib.check(L.pl_plan_arm_post_dots(plan.h, nf, pd, pr, dots.lmin, _ptr(pre[0]), _ptr(pre[1])))
    dots.pre = (pre[0], pre[1])


def cg_fwd_tt(alm, nside, lmax, n_inv, fl_in=None, fl_out=None, pmat=None, rmat=None, scratch=None, alm_add=None, fl_add=None, pinv_md=None, lowrank=None, dots=None):
    """fl_out Y^t [N^-1 - N^-1 P (P^t N^-1 P)^-1 P^t N^-1] Y (fl_in alm) + fl_add alm_add on the device, one call (pl_cg_fwd_tt):
    fwd_op.calc of plancklens/qcinv/opfilt_tt.py:67-73.  pmat, rmat: (nmodes, npix) device matrices or None.
    lowrank = (hpm, hrm): plain weighting in pixel space and the template projection as the rank-nmodes update result -= hrm^t (hpm alm)
    in the same call (pl_cg_fwd_tt_lr_b; hpm, hrm: (nmodes, 2 nalm) real device matrices)."""
    plan = get_plan(nside, lmax)
    a = alm.to(torch.complex128).contiguous()
    nb = a.shape[0] if a.dim() == 2 else 1  # a block [nb, nalm]: nb right-hand sides through every launch (pl_cg_fwd_tt_b)
    assert a.shape[-1] == plan.nalm and a.numel() == nb * plan.nalm and n_inv.numel() == plan.npix and n_inv.is_contiguous(), (a.shape, plan.nalm)
    nmodes = 0 if pmat is None else pmat.shape[0]
    if nmodes:
        assert pmat.shape == (nmodes, plan.npix) and rmat.shape == pmat.shape and pmat.is_contiguous() and rmat.is_contiguous()
        assert scratch is not None and scratch.numel() >= nb * 16 * 256
    if pinv_md is not None:  # monopole + dipole from the ring geometry (pl_cg_fwd_tt_md_b): no template matrices
        assert pmat is None and pinv_md.numel() == 16 and pinv_md.is_contiguous() and pinv_md.dtype == torch.float64
        from . import dev as _dev
        scratch = _dev.tproj_md_scratch(_lib.lib().pl_template_md_scratch_doubles(plan.h, nb))
    out = torch.empty_like(a)
    fi, fo = _fl_arg(fl_in, lmax, True), _fl_arg(fl_out, lmax, True)
    fa = _fl_arg(fl_add, lmax, True) if alm_add is not None else None
    if alm_add is not None:
        alm_add = alm_add.contiguous()
        assert alm_add.shape == a.shape and alm_add.dtype == torch.complex128
    lr_scratch = None
    if lowrank is not None:
        hpm, hrm = lowrank
        assert pmat is None and pinv_md is None and hpm.shape == hrm.shape and hpm.shape[1] == 2 * plan.nalm and hpm.is_contiguous() and hrm.is_contiguous()
        assert hpm.dtype == torch.float64 and hrm.dtype == torch.float64
        from . import dev as _dev
        lr_scratch = _dev.tproj_scratch(nb)  # (raises inside a graph capture when it would have to allocate)
    # <alm, result> and <alm, dots.r> from the kernel that writes the result.  One shot: armed only now that every check above has
    # passed, and consumed by the call below at its very top (the library disarms before any of its own early returns)
    if dots is not None:
        _arm_post_dots(plan, dots, [a], nb)
    if lowrank is not None:
        _lib.check(_lib.lib().pl_cg_fwd_tt_lr_b(plan.h, nb, _ptr(a), _ptr(fi), _ptr(n_inv), int(hpm.shape[0]), _ptr(hpm), _ptr(hrm), _ptr(lr_scratch),
                                                _ptr(alm_add), _ptr(fa), _ptr(out), _ptr(fo), _stream()))
    elif pinv_md is not None:
        _lib.check(_lib.lib().pl_cg_fwd_tt_md_b(plan.h, nb, _ptr(a), _ptr(fi), _ptr(n_inv), _ptr(pinv_md), _ptr(scratch), _ptr(alm_add), _ptr(fa),
                                                _ptr(out), _ptr(fo), _stream()))
    elif a.dim() == 2:
        _lib.check(_lib.lib().pl_cg_fwd_tt_b(plan.h, nb, _ptr(a), _ptr(fi), _ptr(n_inv), nmodes, _ptr(pmat), _ptr(rmat), _ptr(scratch),
                                             _ptr(alm_add), _ptr(fa), _ptr(out), _ptr(fo), _stream()))
    else:
        _lib.check(_lib.lib().pl_cg_fwd_tt(plan.h, _ptr(a), _ptr(fi), _ptr(n_inv), nmodes, _ptr(pmat), _ptr(rmat), _ptr(scratch), _ptr(alm_add),
                                           _ptr(fa), _ptr(out), _ptr(fo), _stream()))
    return out


def cg_fwd_pp(elm, blm, nside, lmax, n_inv, fl_in=None, fl_out=None, add=None, fl_add_e=None, fl_add_b=None, n_qu=None, n_uu=None, dots=None):
    """fl_out Y2^t [n_inv Y2 (fl_in (E, B))] + (fl_add_e E_add, fl_add_b B_add) on the device, one call (pl_cg_fwd_pp): fwd_op.calc of
    plancklens/qcinv/opfilt_pp.py:69-78 for a single inverse-noise map.  Returns (elm, blm), two views of one (2, nalm) tensor."""
    plan = get_plan(nside, lmax)
    e, b = elm.contiguous(), blm.contiguous()
    nb = e.shape[0] if e.dim() == 2 else 1  # blocks [nb, nalm]: nb right-hand sides through every launch (pl_cg_fwd_pp_b)
    assert e.shape == b.shape and e.shape[-1] == plan.nalm and e.numel() == nb * plan.nalm and e.dtype == torch.complex128 and b.dtype == torch.complex128
    assert n_inv.numel() == plan.npix and n_inv.is_contiguous() and n_inv.dtype == torch.float64
    for n_ in (n_qu, n_uu):  # (QQ, QU, UU) noise: n_inv is the QQ map (pl_cg_fwd_pp_qu_b)
        assert (n_qu is None) == (n_uu is None) and (n_ is None or (n_.numel() == plan.npix and n_.is_contiguous() and n_.dtype == torch.float64))
    out = torch.empty((2,) + tuple(e.shape), dtype=torch.complex128, device=e.device)
    fi, fo = _fl_arg(fl_in, lmax, True), _fl_arg(fl_out, lmax, True)
    ae = ab = fe = fb = None
    if add is not None:
        ae, ab = add[0].contiguous(), add[1].contiguous()
        assert ae.shape == e.shape and ab.shape == e.shape and ae.dtype == torch.complex128 and ab.dtype == torch.complex128
        fe, fb = _fl_arg(fl_add_e, lmax, True), _fl_arg(fl_add_b, lmax, True)
    if dots is not None:  # as in cg_fwd_tt, over both fields
        _arm_post_dots(plan, dots, [e, b], nb)
    if n_qu is not None:
        _lib.check(_lib.lib().pl_cg_fwd_pp_qu_b(plan.h, nb, _ptr(e), _ptr(b), _ptr(fi), _ptr(n_inv), _ptr(n_qu), _ptr(n_uu), _ptr(ae), _ptr(ab),
                                                _ptr(fe), _ptr(fb), _ptr(out[0]), _ptr(out[1]), _ptr(fo), _stream()))
    elif e.dim() == 2:
        _lib.check(_lib.lib().pl_cg_fwd_pp_b(plan.h, nb, _ptr(e), _ptr(b), _ptr(fi), _ptr(n_inv), _ptr(ae), _ptr(ab), _ptr(fe), _ptr(fb),
                                             _ptr(out[0]), _ptr(out[1]), _ptr(fo), _stream()))
    else:
        _lib.check(_lib.lib().pl_cg_fwd_pp(plan.h, _ptr(e), _ptr(b), _ptr(fi), _ptr(n_inv), _ptr(ae), _ptr(ab), _ptr(fe), _ptr(fb),
                                           _ptr(out[0]), _ptr(out[1]), _ptr(fo), _stream()))
    return out[0], out[1]


def _anal(spin, maps, lmax, fl=None):
    dev = _is_dev(maps)
    ncomp = 1 if spin == 0 else 2
    L = _lib.lib()
    if dev:
        m = maps.to(torch.float64).contiguous()
        npix = m.numel() // ncomp
        plan = get_plan(npix2nside(npix), lmax)
        f = _fl_arg(fl, lmax, True)
        out = torch.empty((ncomp, plan.nalm) if ncomp == 2 else (plan.nalm,), dtype=torch.complex128, device=m.device)
        _lib.check(L.pl_map2alm(plan.h, spin, _ptr(m), _ptr(out), _ptr(f), _lib.PL_DEVICE, _stream()))
        return out
    m = np.ascontiguousarray(maps, dtype=np.float64)
    npix = m.size // ncomp
    plan = get_plan(npix2nside(npix), lmax)
    f = _fl_arg(fl, lmax, False)
    out = np.empty((ncomp, plan.nalm) if ncomp == 2 else (plan.nalm,), dtype=np.complex128)
    _lib.check(L.pl_map2alm(plan.h, spin, _ptr(m), _ptr(out), _ptr(f), _lib.PL_HOST, None))
    return out


class MapRef(object):
    """An input map of an analysis that is read THROUGH a table of device addresses (pl_map2alm_ind): entry `k` of `table` (a device int64
    tensor) holds the address of an npix float64 map when the kernels run.  What a captured HIP graph takes as its input so that a replay can
    serve other maps without copying them into fixed slots (qest.library._pair_graph); map2alm / map2alm_spin accept it in place of a tensor
    ((Q, U) as two consecutive entries of one table)."""

    def __init__(self, table, k, npix):
        assert isinstance(table, torch.Tensor) and table.is_cuda and table.dtype == torch.int64 and table.is_contiguous() and 0 <= k < table.numel()
        self.table, self.k, self.npix = table, int(k), int(npix)

    def __len__(self):
        return self.npix

    def numel(self):
        return self.npix

    def ptr(self):
        """address of the table entry (what pl_map2alm_ind takes)"""
        return ctypes.c_void_p(self.table.data_ptr() + 8 * self.k)


def store_addresses(addrs, table):
    """device addresses (python ints, at most 8) -> the first entries of `table` (device int64) on the current stream (pl_store_addresses)"""
    n = len(addrs)
    assert 1 <= n <= 8 and table.numel() >= n and table.dtype == torch.int64 and table.is_cuda
    arr = (ctypes.c_uint64 * n)(*[int(a) for a in addrs])
    _lib.check(_lib.lib().pl_store_addresses(n, arr, ctypes.c_void_p(table.data_ptr()), _stream()))


def _anal_ind(spin, refs, lmax, fl=None):
    ncomp = 1 if spin == 0 else 2
    assert len(refs) == ncomp and all(isinstance(r, MapRef) for r in refs)
    assert ncomp == 1 or (refs[1].table is refs[0].table and refs[1].k == refs[0].k + 1 and refs[1].npix == refs[0].npix), 'the components of a spin transform are consecutive entries of one table'
    plan = get_plan(npix2nside(refs[0].npix), lmax)
    f = _fl_arg(fl, lmax, True)
    out = torch.empty((ncomp, plan.nalm) if ncomp == 2 else (plan.nalm,), dtype=torch.complex128, device=refs[0].table.device)
    _lib.check(_lib.lib().pl_map2alm_ind(plan.h, int(spin), refs[0].ptr(), _ptr(out), _ptr(f), _stream()))
    return out


def _stack(pair):
    if _is_dev(pair):
        return pair
    if _is_dev(pair[0]):
        a, b = pair[0], pair[1]
        base = a._base
        if (base is not None and base is b._base and base.dim() == 2 and base.shape[0] == 2 and base.is_contiguous()
                and a.data_ptr() == base.data_ptr() and b.data_ptr() == base[1].data_ptr() and a.numel() == base.shape[1]):
            return base  # the two halves of one (2, n) tensor (what alm2map_spin / map2alm_spin return): no copy
        return torch.stack([a, b])
    return np.stack([np.asarray(pair[0]), np.asarray(pair[1])])


# ---- the four functions of plancklens/shts.py --------------------------------------------------------
def alm2map(alm, nside, lmax=None, mmax=None, pol=False, verbose=False, fl=None, **kwargs):
    """hp.alm2map (shts.py:12-15): T(p) = sum a_lm Y_lm(p).  pol=True takes (t, e, b) (opfilt_tp.py:276)."""
    if pol or (not _is_dev(alm) and np.ndim(alm) == 2 and len(alm) == 3) or (_is_dev(alm) and alm.dim() == 2 and alm.shape[0] == 3):
        t, e, b = alm[0], alm[1], alm[2]
        lm = Alm.getlmax(t.numel() if _is_dev(t) else np.size(t)) if lmax is None else lmax
        tm = alm2map(t, nside, lmax=lm)
        qu = alm2map_spin([e, b], nside, 2, lm)
        return [tm, qu[0], qu[1]]
    size = alm.numel() if _is_dev(alm) else np.size(alm)
    if lmax is None:
        lmax = Alm.getlmax(size)
    assert lmax >= 0 and Alm.getsize(lmax) == size, 'alm size does not match lmax (mmax = lmax only)'
    assert mmax is None or mmax == lmax
    return _synth(0, alm, nside, lmax, fl=fl)


def map2alm(m, lmax=None, mmax=None, iter=0, pol=False, use_weights=False, fl=None, **kwargs):
    """hp.map2alm(iter=0) (shts.py:16-20): uniform-weight quadrature, no Jacobi iterations.
    pol=True takes [T, Q, U] (opfilt_tp.py:281)."""
    assert iter == 0, 'only iter=0 is implemented: every reference call passes iter=0 (SURVEY.md Appendix A.2)'
    assert not use_weights
    if isinstance(m, MapRef):
        lmax = 3 * npix2nside(m.npix) - 1 if lmax is None else lmax
        assert mmax is None or mmax == lmax
        return _anal_ind(0, [m], lmax, fl=fl)
    if pol or (not _is_dev(m) and np.ndim(m) == 2 and len(m) == 3) or (_is_dev(m) and m.dim() == 2 and m.shape[0] == 3):
        t = map2alm(m[0], lmax=lmax, iter=0)
        lm = Alm.getlmax(t.numel() if _is_dev(t) else t.size)
        e, b = map2alm_spin([m[1], m[2]], 2, lm)
        return [t, e, b]
    npix = m.numel() if _is_dev(m) else np.size(m)
    nside = npix2nside(npix)
    if lmax is None:
        lmax = 3 * nside - 1
    assert mmax is None or mmax == lmax
    return _anal(0, m, lmax, fl=fl)


def alm2map_spin(gclm, nside, spin, lmax, mmax=None, fl=None):
    """hp.alm2map_spin (shts.py:22-24): (Re, Im) of sum -(G + iC) _sY_lm, spin = 1, 2, 3."""
    assert spin > 0, spin
    assert len(gclm) == 2, len(gclm)
    assert mmax is None or mmax == lmax
    if gclm[1] is None:  # extension: no curl component (C = 0), e.g. the gradient legs of the temperature estimators
        out = _synth(int(spin), gclm[0], nside, lmax, fl=fl, grad_only=True)
    else:
        out = _synth(int(spin), _stack(gclm), nside, lmax, fl=fl)
    return [out[0], out[1]]


def alm2map_spin_pair(gclm, glm2, nside, spin, lmax, fl=None, fl2=None):
    """Two spin-s syntheses on one Legendre recursion (pl_alm2map_pair, device arrays only): alm2map_spin(gclm, ...) with
    filter fl and alm2map_spin([glm2, 0], ...) with filter fl2.  Returns ([Q, U], [Q2, U2]); equal to the two separate
    calls to rounding (the sums are formed in the same order)."""
    assert spin > 0 and len(gclm) == 2
    a = _stack(gclm)
    assert _is_dev(a) and _is_dev(glm2), 'alm2map_spin_pair works on device arrays'
    plan = get_plan(nside, lmax)
    a = a.to(torch.complex128).contiguous()
    g2 = glm2.to(torch.complex128).contiguous()
    assert a.numel() == 2 * plan.nalm and g2.numel() == plan.nalm, (a.shape, g2.shape, plan.nalm)
    f, f2 = _fl_arg(fl, lmax, True), _fl_arg(fl2, lmax, True)
    out = torch.empty((4, plan.npix), dtype=torch.float64, device=a.device)
    _lib.check(_lib.lib().pl_alm2map_pair(plan.h, int(spin), _ptr(a), _ptr(f), _ptr(g2), _ptr(f2), _ptr(out), _stream()))
    return [out[0], out[1]], [out[2], out[3]]


def alm2map_spin_grad_pair(glm1, glm2, nside, spin, lmax, fl=None, fl2=None):
    """Two gradient-only spin-s syntheses alm2map_spin([glm, 0], ...) on one Legendre recursion (pl_alm2map_grad_pair, device arrays
    only): returns ([Q1, U1], [Q2, U2]), bit-identical to the two separate calls at 3/4 of their Legendre work."""
    assert spin > 0 and _is_dev(glm1) and _is_dev(glm2), 'alm2map_spin_grad_pair works on device arrays'
    plan = get_plan(nside, lmax)
    g1, g2 = glm1.to(torch.complex128).contiguous(), glm2.to(torch.complex128).contiguous()
    assert g1.numel() == plan.nalm and g2.numel() == plan.nalm, (g1.shape, g2.shape, plan.nalm)
    f, f2 = _fl_arg(fl, lmax, True), _fl_arg(fl if fl2 is None else fl2, lmax, True)
    out = torch.empty((4, plan.npix), dtype=torch.float64, device=g1.device)
    _lib.check(_lib.lib().pl_alm2map_grad_pair(plan.h, int(spin), _ptr(g1), _ptr(f), _ptr(g2), _ptr(f2), _ptr(out), _stream()))
    return [out[0], out[1]], [out[2], out[3]]


def alm2map_spin_batch2(gclm1, gclm2, nside, spin, lmax, fl=None):
    """The same spin-s synthesis of two inputs (two simulations) on one Legendre recursion (pl_alm2map_batch2, device arrays
    only): returns ([Q1, U1], [Q2, U2]), bit-identical to two alm2map_spin calls at 5/6 of their Legendre work."""
    assert spin > 0 and len(gclm1) == 2 and len(gclm2) == 2
    a1, a2 = _stack(gclm1), _stack(gclm2)
    assert _is_dev(a1) and _is_dev(a2), 'alm2map_spin_batch2 works on device arrays'
    plan = get_plan(nside, lmax)
    a1, a2 = a1.to(torch.complex128).contiguous(), a2.to(torch.complex128).contiguous()
    assert a1.numel() == 2 * plan.nalm and a2.numel() == 2 * plan.nalm, (a1.shape, a2.shape, plan.nalm)
    f = _fl_arg(fl, lmax, True)
    out = torch.empty((4, plan.npix), dtype=torch.float64, device=a1.device)
    _lib.check(_lib.lib().pl_alm2map_batch2(plan.h, int(spin), _ptr(a1), _ptr(a2), _ptr(f), _ptr(out), _stream()))
    return [out[0], out[1]], [out[2], out[3]]


def alm2map_batch2(alm1, alm2, nside, lmax=None, fl=None):
    """The scalar synthesis of two inputs (two simulations) in one call (pl_alm2map_batch2 with spin 0, device arrays only): returns the two
    maps as the rows of one (2, npix) tensor, bit-identical to two alm2map calls.  On grids of nside >= 1024 the two inputs share one Legendre
    recursion (k_leg_synth0<R, true>: 10 instead of 12 FMAs per two-l step and ring pair for the two maps)."""
    assert _is_dev(alm1) and _is_dev(alm2), 'alm2map_batch2 works on device arrays'
    a1, a2 = alm1.to(torch.complex128).contiguous(), alm2.to(torch.complex128).contiguous()
    if lmax is None:
        lmax = Alm.getlmax(a1.numel())
    plan = get_plan(nside, lmax)
    assert a1.numel() == plan.nalm and a2.numel() == plan.nalm, (a1.shape, a2.shape, plan.nalm)
    f = _fl_arg(fl, lmax, True)
    out = torch.empty((2, plan.npix), dtype=torch.float64, device=a1.device)
    _lib.check(_lib.lib().pl_alm2map_batch2(plan.h, 0, _ptr(a1), _ptr(a2), _ptr(f), _ptr(out), _stream()))
    return out


def map2alm_spin(maps, spin, lmax=None, mmax=None, fl=None):
    """hp.map2alm_spin (shts.py:26-30): the 4 pi / npix weighted adjoint of alm2map_spin; returns [G, C]."""
    assert spin > 0, spin
    assert len(maps) == 2
    if isinstance(maps[0], MapRef):
        lmax = 3 * npix2nside(maps[0].npix) - 1 if lmax is None else lmax
        assert mmax is None or mmax == lmax
        out = _anal_ind(int(spin), list(maps), lmax, fl=fl)
        return [out[0], out[1]]
    m = _stack(maps)
    npix = m.shape[1]
    if lmax is None:
        lmax = 3 * npix2nside(npix) - 1
    assert mmax is None or mmax == lmax
    out = _anal(int(spin), m, lmax, fl=fl)
    return [out[0], out[1]]
