"""Multigrid-preconditioned CG chains, API of plancklens/qcinv/multigrid.py (`multigrid_chain` :25-111,
`parse_pre_op_descr` :113-160, `pre_op_split` :163-182, `pre_op_multigrid` :185-215).

A chain is a list of stages [id, pre_ops_descr, lmax, nside, iter_max, eps_min, tr, cache]; the descriptors are the
reference's mini-language: "split(<low>, lsplit, <high>)", "diag_cl", "dense(<cache file>)", "stage(<id>)".
All vectors handed between the stages stay on the device."""
from __future__ import print_function

import copy
import gc
import re
import sys

import numpy as np
import torch

from .. import options
from . import cd_monitors, cd_solve, util, util_alm


class multigrid_stage(object):
    def __init__(self, ids, pre_ops_descr, lmax, nside, iter_max, eps_min, tr, cache):
        self.depth = ids
        self.pre_ops_descr = pre_ops_descr
        self.lmax = lmax
        self.nside = nside
        self.iter_max = iter_max
        self.eps_min = eps_min
        self.tr = tr
        self.cache = cache
        self.pre_ops = []


class multigrid_chain(object):
    def __init__(self, opfilt, chain_descr, s_cls, n_inv_filt, debug_log_prefix=None, plogdepth=0):
        self.debug_log_prefix = debug_log_prefix
        self.plogdepth = plogdepth
        self.opfilt = opfilt
        self.chain_descr = chain_descr
        self.s_cls = s_cls
        self.n_inv_filt = n_inv_filt
        stages = {}
        for [sid, pre_ops_descr, lmax, nside, iter_max, eps_min, tr, cache] in self.chain_descr:
            stages[sid] = multigrid_stage(sid, pre_ops_descr, lmax, nside, iter_max, eps_min, tr, cache)
            for descr in pre_ops_descr:  # coarser stages were parsed before and are reachable through `stages`
                stages[sid].pre_ops.append(parse_pre_op_descr(descr, opfilt=self.opfilt, s_cls=self.s_cls,
                                                              n_inv_filt=self.n_inv_filt, stages=stages, lmax=lmax,
                                                              nside=nside, chain=self))
        self.bstage = stages[0]
        self.iter_tot = 0
        self.watch = util.stopwatch()
        self.prev_eps = None

    def solve(self, soltn, tpn_map, apply_fini='', dot_op=None):
        """Solves in place for `soltn` given the data map(s); finishes with opfilt.apply_fini (S^-1 x).

        Block solves (several simulations filtered together, every launch carrying all of them): `soltn` holds [nb, nalm] device
        tensors and `tpn_map` is the list of the nb data maps (nb (Q, U) pairs / (T, Q, U) triplets).  Scalar products, step
        lengths and the stopping rule are per entry (the block solve ends when every entry has met it; an entry that has is frozen
        where its own solve would have returned): results equal nb separate solves."""
        assert hasattr(self.opfilt, 'apply_fini%s' % apply_fini)
        finifunc = getattr(self.opfilt, 'apply_fini%s' % apply_fini)
        self.watch = util.stopwatch()
        self.iter_tot = 0
        self.prev_eps = None
        if dot_op is None:
            dot_op = self.opfilt.dot_op()
        logger = (lambda iter, eps, stage=self.bstage, **kwargs: self.log(stage, iter, eps, **kwargs))
        if _is_block(soltn):
            assert len(tpn_map) == _parts(soltn)[0].shape[0], 'one data map (set) per entry of the block'
            tpn_alm = self.opfilt.calc_prep_batch(tpn_map, self.s_cls, self.n_inv_filt)
        else:
            tpn_alm = self.opfilt.calc_prep(tpn_map, self.s_cls, self.n_inv_filt)
        monitor = cd_monitors.monitor_basic(dot_op, logger=logger, iter_max=self.bstage.iter_max,
                                            eps_min=self.bstage.eps_min, d0=dot_op(tpn_alm, tpn_alm))
        fwd_op = self.opfilt.fwd_op(self.s_cls, self.n_inv_filt)
        pre_ops, pace = self.bstage.pre_ops, getattr(self, 'pace', None)
        if pace is not None:
            # two solves of this process running at the same time (filt_cinv.run_tp): each waits for the other before it launches
            # its preconditioner.  The stopping criterion just before has drained this solve's stream, so the two chains of small
            # coarse-level kernels start together and overlap each other instead of hiding behind the other solve's chip-filling
            # fine-level transforms.
            pre_ops = [_paced(op, pace) for op in pre_ops]
        try:
            self.last_iters = cd_solve.cd_solve(soltn, tpn_alm, fwd_op, pre_ops, dot_op, monitor, tr=self.bstage.tr, cache=self.bstage.cache)
        finally:
            if pace is not None:
                pace.leave()
        finifunc(soltn, self.s_cls, self.n_inv_filt)

    def log(self, stage, iter, eps, **kwargs):
        self.iter_tot += 1
        elapsed = self.watch.elapsed()
        if stage.depth > self.plogdepth:
            return
        log_str = '   ' * stage.depth + '(%4d, %04d) [%s] (%d, %.8f)' % (stage.nside, stage.lmax, str(elapsed), iter, eps) + '\n'
        sys.stdout.write(log_str)
        if self.debug_log_prefix is not None:
            with open(self.debug_log_prefix + 'stage_all.dat', 'a') as f:
                f.write(log_str)
            if stage.depth == 0:
                from .. import dev
                s = kwargs['soltn']
                np.save(self.debug_log_prefix + 'stage_soltn_%d_%04d.npy' % (stage.depth, iter),
                        dev.to_host(s) if not hasattr(s, 'elm') else np.array([dev.to_host(s.elm), dev.to_host(s.blm)]))
            with open(self.debug_log_prefix + 'stage_%d.dat' % stage.depth, 'a') as f:
                f.write('%05d %05d %10.6e %05d %s\n' % (self.iter_tot, int(elapsed), eps, iter, str(elapsed)))


class pace(object):
    """Rendezvous of the top-level iterations of solves that run at the same time on different streams (one thread each): `wait`
    returns once every party still in the game has arrived (or after `timeout` seconds: pacing is an optimisation, never a
    condition for progress); a solve that ends calls `leave`."""

    def __init__(self, parties, timeout=0.25):
        import threading
        self.cond = threading.Condition()
        self.parties, self.timeout = parties, timeout
        self.waiting, self.generation = 0, 0

    def wait(self):
        with self.cond:
            if self.parties <= 1:
                return
            self.waiting += 1
            if self.waiting >= self.parties:
                self.waiting = 0
                self.generation += 1
                self.cond.notify_all()
                return
            gen = self.generation
            if not self.cond.wait_for(lambda: self.generation != gen, timeout=self.timeout):
                self.waiting = max(0, self.waiting - 1)  # gave up: go on alone

    def leave(self):
        with self.cond:
            self.parties -= 1
            if self.waiting >= self.parties > 0:
                self.waiting = 0
                self.generation += 1
            self.cond.notify_all()


class _paced(object):
    """a preconditioner that meets the other solve at `pc` before it runs"""

    def __init__(self, op, pc):
        self.op, self.pc = op, pc
        if hasattr(op, 'with_dot'):
            self.with_dot = self._with_dot

    def __call__(self, v):
        self.pc.wait()
        return self.op(v)

    def _with_dot(self, v, q, lmin):
        self.pc.wait()
        return self.op.with_dot(v, q, lmin)


def parse_pre_op_descr(pre_op_descr, **kwargs):
    m = re.match(r"split\((.*),\s*(.*),\s*(.*)\)\Z", pre_op_descr)
    if m:
        low_descr, lsplit, hgh_descr = m.groups()
        lsplit = int(lsplit)
        kwargs_low = copy.copy(kwargs)
        kwargs_low['lmax'] = lsplit
        kwargs_hgh = copy.copy(kwargs)
        kwargs_hgh['lmin'] = lsplit + 1
        return pre_op_split(lsplit, kwargs['lmax'], parse_pre_op_descr(low_descr, **kwargs_low),
                            parse_pre_op_descr(hgh_descr, **kwargs_hgh))
    if re.match(r"diag_cl\Z", pre_op_descr):
        return kwargs['opfilt'].pre_op_diag(kwargs['s_cls'], kwargs['n_inv_filt'])
    m = re.match(r"dense(\((.*)\))?\Z", pre_op_descr)
    if m:
        cache_fname = m.group(2)
        if cache_fname in ('', 'None'):
            cache_fname = None
        print('creating dense preconditioner. (nside = %d, lmax = %d, cache = %s)' % (kwargs['nside'], kwargs['lmax'], cache_fname))
        fwd_op = kwargs['opfilt'].fwd_op(kwargs['s_cls'], kwargs['n_inv_filt'].degrade(kwargs['nside']))
        return kwargs['opfilt'].pre_op_dense(kwargs['lmax'], fwd_op, cache_fname=cache_fname)
    m = re.match(r"stage\((.*)\)\Z", pre_op_descr)
    if m:
        stage = kwargs['stages'][int(m.group(1))]
        logger = (lambda iter, eps, stage=stage, chain=kwargs['chain'], **kw: chain.log(stage, iter, eps, **kw))
        assert stage.lmax == kwargs['lmax']
        chain = kwargs['chain']
        quiet = (lambda stage=stage, chain=chain: stage.depth > chain.plogdepth and chain.debug_log_prefix is None)
        return pre_op_multigrid(kwargs['opfilt'], stage.lmax, stage.nside, kwargs['s_cls'],
                                kwargs['n_inv_filt'].degrade(stage.nside), stage.pre_ops, logger, stage.tr, stage.cache,
                                stage.iter_max, stage.eps_min, quiet=quiet)
    assert 0, 'pre_op_descr ' + pre_op_descr + ' is unrecognized!'


class pre_op_split(object):
    """Low multipoles (l <= lsplit) through one preconditioner, the rest through another."""

    def __init__(self, lsplit, lmax, pre_op_low, pre_op_hgh):
        self.lsplit = lsplit
        self.lmax = lmax
        self.pre_op_low = pre_op_low
        self.pre_op_hgh = pre_op_hgh
        self.iter = 0

    def __call__(self, talm):
        return self.calc(talm)

    def _low(self, tmp):
        """the low-l preconditioner on a temporary it may overwrite"""
        return self.pre_op_low.calc_owned(tmp) if hasattr(self.pre_op_low, 'calc_owned') else self.pre_op_low(tmp)

    def with_dot(self, talm, q, lmin):
        """(calc(talm), pre): pre = the partial sums of <result, q> left by the kernel that writes the result (the one-launch split around
        the dense block, or the splice with the diagonal high-l part), for dot_op.ortho(pre=...); None where neither form applies"""
        if _lmax_of(talm) != self.lmax or _lmax_of(q) != self.lmax or _is_block(talm) != _is_block(q):
            return self.calc(talm), None
        if hasattr(self.pre_op_low, 'split_apply'):
            self.iter += 1
            ret = self.pre_op_low.split_apply(talm, self.lsplit, self.pre_op_hgh, dot=(q, lmin))
            if ret is not None:
                return ret
            self.iter -= 1
            return self.calc(talm), None
        if hasattr(self.pre_op_hgh, 'splice_above') and all(isinstance(p, torch.Tensor) and p.is_cuda and p.dtype == torch.complex128 and p.is_contiguous()
                                                            for p in _parts(talm) + _parts(q)):
            self.iter += 1
            talm_low = self._low(util_alm.alm_copy(talm, lmax=self.lsplit))
            ret = self.pre_op_hgh.splice_above(talm_low, talm, self.lsplit, dot=(q, lmin))
            if ret is not None:
                return ret
            talm_hgh = self.pre_op_hgh(talm)
            return util_alm.alm_splice(talm_low, talm_hgh, self.lsplit), None
        return self.calc(talm), None

    def calc(self, talm):
        self.iter += 1
        if hasattr(self.pre_op_low, 'split_apply') and _lmax_of(talm) == self.lmax:
            ret = self.pre_op_low.split_apply(talm, self.lsplit, self.pre_op_hgh)  # dense block + diagonal + splice in one launch
            if ret is not None:
                return ret
        talm_low = self._low(util_alm.alm_copy(talm, lmax=self.lsplit))
        if hasattr(self.pre_op_hgh, 'splice_above') and _lmax_of(talm) == self.lmax:
            ret = self.pre_op_hgh.splice_above(talm_low, talm, self.lsplit)  # diagonal high-l part applied inside the splice
            if ret is not None:
                return ret
        # preconditioners do not modify their argument (cd_solve's contract): no copy when the band-limit already matches
        talm_hgh = self.pre_op_hgh(talm if _lmax_of(talm) == self.lmax else util_alm.alm_copy(talm, lmax=self.lmax))
        return util_alm.alm_splice(talm_low, talm_hgh, self.lsplit)


class pre_op_multigrid(object):
    """A few CG iterations at a coarser (nside, lmax) as preconditioner."""

    def __init__(self, opfilt, lmax, nside, s_cls, n_inv_filt, pre_ops, logger, tr, cache, iter_max, eps_min, quiet=None):
        self.quiet = quiet  # callable: True when this stage's iterations are not logged anywhere
        self._graphs = {}
        self.opfilt = opfilt
        self.fwd_op = opfilt.fwd_op(s_cls, n_inv_filt)
        self.lmax = lmax
        self.nside = nside
        self.s_cls = s_cls
        self.pre_ops = pre_ops
        self.logger = logger
        self.tr = tr
        self.cache = cache
        self.iter_max = iter_max
        self.eps_min = eps_min

    def __call__(self, talm):
        return self.calc(talm)

    def calc_owned(self, talm):
        """calc for a caller that hands over a temporary: talm may be overwritten (it becomes the residual of the nested solve)"""
        return self.calc(talm, owned=True)

    # A nested solve with a fixed iteration count and nobody reading its log has no host-side data dependence: the same
    # few thousand small kernels (coarse SHTs, dense mat-vec, alm arithmetic) in the same order every time, and at the
    # coarse resolutions they cost less to run than to launch from Python.  After `graph_after` eager calls the outermost
    # such preconditioner is captured into a HIP graph (torch.cuda.CUDAGraph; the kernels of libplshts are launched on
    # torch's current stream, so they are captured with everything else) and replayed from then on.
    graph_after = 2
    graph_fallbacks = 0  # captures of this stage that failed (it then stays eager)

    def _capturable(self, talm):
        if not options.opts.cg_graph or self.iter_max == np.inf or self.eps_min != 0.:
            return False
        if self.quiet is None or not self.quiet():
            return False
        parts = _parts(talm)
        return all(isinstance(p, torch.Tensor) and p.is_cuda for p in parts) and not torch.cuda.is_current_stream_capturing()

    def calc(self, talm, owned=False):
        if not self._capturable(talm):
            return self._calc_eager(talm, owned)
        from .. import shts
        # (a graph holds the workspace addresses of the plan context it was recorded in: one graph per context)
        key = tuple((tuple(p.shape), p.dtype) for p in _parts(talm)) + (shts.context(),)
        st = self._graphs.setdefault(key, {'calls': 0, 'graph': None})
        if st['graph'] is None:
            st['calls'] += 1
            if st['calls'] <= self.graph_after:
                return self._calc_eager(talm, owned)
            try:
                st['in'] = [torch.empty_like(p) for p in _parts(talm)]
                for d, p in zip(st['in'], _parts(talm)):
                    d.copy_(p)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                # The cyclic garbage collector must not run while the stream is capturing: finalisers of unrelated garbage
                # (pinned host buffers, plans, older graphs) make HIP calls that are illegal during capture and abort the
                # process.  torch.cuda.graph collects once on entry; keep the collector off until the capture has ended.
                gc_was_on = gc.isenabled()
                gc.disable()
                try:
                    # thread_local: another solver of this process may be launching (and allocating) on its own stream meanwhile
                    with torch.cuda.graph(g, capture_error_mode='thread_local'):
                        out = self._calc_eager(_like(talm, st['in']), True)  # (the graph's input buffers are refilled before every replay)
                finally:
                    if gc_was_on:
                        gc.enable()
                st['out'], st['graph'] = _parts(out), g
                options.count('cg_graph_captures')
            except Exception as e:  # capture is an optimisation: fall back to the eager path for good -- counted (options.stats), not printed
                self.graph_fallbacks += 1
                options.count('cg_graph_fallbacks', 'pre_op_multigrid depth %s: %s' % (getattr(self, 'depth', '?'), str(e).split('\n')[0]))
                if options.opts.debug:
                    import traceback
                    traceback.print_exc()
                torch.cuda.synchronize()
                st['graph'] = False
                return self._calc_eager(talm, owned)
        if st['graph'] is False:
            return self._calc_eager(talm, owned)
        for d, p in zip(st['in'], _parts(talm)):
            d.copy_(p)
        st['graph'].replay()
        return _like(talm, [o.clone() for o in st['out']])

    def _calc_eager(self, talm, owned=False):
        quiet = bool(self.quiet is not None and self.quiet())
        monitor = cd_monitors.monitor_basic(self.opfilt.dot_op(), iter_max=self.iter_max, eps_min=self.eps_min, logger=self.logger, quiet=quiet)
        dev_vec = all(isinstance(p, torch.Tensor) for p in _parts(talm))
        # nobody looks at the solution before the first step (quiet, fixed iteration count): it need not be zero-filled -- the first step
        # writes it (cd_solve x_uninit)
        uninit = dev_vec and quiet and self.eps_min == 0. and self.iter_max >= 1
        if dev_vec:
            soltn = _like(talm, [torch.empty_like(p) if uninit else torch.zeros_like(p) for p in _parts(talm)])
        else:
            soltn = talm * 0.0
        # the right-hand side is a private copy: with fewer iterations than the residual refresh period cd_solve may use it as
        # its residual (one copy less); a caller that hands over a temporary of this band-limit (`owned`) has made that copy already;
        # a solution at the band-limit of the input needs no splice
        scratch = self.iter_max < cd_solve.ROUNDOFF
        b = talm if (owned and _lmax_of(talm) == self.lmax) else util_alm.alm_copy(talm, lmax=self.lmax)
        cd_solve.cd_solve(soltn, b, self.fwd_op, self.pre_ops, self.opfilt.dot_op(), monitor, tr=self.tr, cache=self.cache, x_is_zero=True,
                          b_scratch=scratch, x_uninit=uninit)
        if _lmax_of(talm) == self.lmax:
            return soltn
        return util_alm.alm_splice(soltn, talm, self.lmax)


def _lmax_of(v):
    return v.lmax if hasattr(v, 'lmax') else util_alm.Alm.getlmax(util_alm._size(v))


def _is_block(v):
    """True for a block vector: its component tensors are [nb, nalm] (nb right-hand sides solved together)"""
    p = _parts(v)[0]
    return isinstance(p, torch.Tensor) and p.dim() == 2


def _parts(v):
    """component tensors of a CG vector (plain alm, eblm, teblm)"""
    if hasattr(v, 'tlm'):
        return [v.tlm, v.elm, v.blm]
    if hasattr(v, 'elm'):
        return [v.elm, v.blm]
    return [v]


def _like(v, parts):
    if hasattr(v, 'tlm'):
        return util_alm.teblm(parts)
    if hasattr(v, 'elm'):
        return util_alm.eblm(parts)
    return parts[0]
