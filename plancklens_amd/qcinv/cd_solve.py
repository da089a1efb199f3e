"""Flexible conjugate-directions solver, API and algorithm of plancklens/qcinv/cd_solve.py (`cd_solve` :35-107,
`cache_mem` :15-32, `tr_cg` / `tr_cd` / `PTR` :7-12).  The vectors are opaque objects supporting + - * += -=
(numpy arrays, device tensors, eblm / teblm); all arithmetic on them happens wherever they live."""
from .. import options

import numpy as np


def PTR(p, t, r):
    return lambda i: max(0, i - max(p, int(min(t, np.mod(i, r)))))


tr_cg = (lambda i: i - 1)
tr_cd = (lambda i: 0)


class cache_mem(dict):
    """In-memory store of the search directions of past iterations."""

    def __init__(self):
        dict.__init__(self)

    def store(self, key, data):
        dTAd_inv, searchdirs, searchfwds = data
        self[key] = [dTAd_inv, searchdirs, searchfwds]

    def restore(self, key):
        return self[key]

    def remove(self, key):
        del self[key]

    def trim(self, keys):
        assert set(keys).issubset(self.keys())
        for key in set(self.keys()) - set(keys):
            del self[key]


ROUNDOFF = 25  # iterations between residual refreshes (cd_solve.py:35)


def cd_solve(x, b, fwd_op, pre_ops, dot_op, criterion, tr, cache=None, roundoff=ROUNDOFF, x_is_zero=False, b_scratch=False, x_uninit=False):
    """Solves fwd_op(x) = b in place on x by preconditioned conjugate directions; returns the iteration count.

        fwd_op, the pre_ops and dot_op must not modify their arguments.  `tr` selects how many past search
        directions each new one is orthogonalised against (tr_cg: the last one).  The residual is recomputed
        from scratch every `roundoff` iterations.  x_is_zero: the caller guarantees x = 0 on entry (the nested
        multigrid solves), which saves the first fwd_op.  b_scratch (with x_is_zero): b may be overwritten and the solve ends
        before the first residual refresh -- b itself becomes the residual.  x_uninit (with x_is_zero): x stands for zero but its memory
        holds anything; the first step writes it (dot_op.step(x_init=True)) where it can, else it is zero-filled first -- for callers whose
        criterion does not look at x before the first step.

        With a single preconditioner and a dot_op that offers `dev(a, b)` (a 0-dim device tensor) the step lengths
        stay on the device: no host synchronisation inside an iteration, same arithmetic.  If the dot_op also offers
        `parts(a, b)` (the scalar product in whatever device-resident form `axpy` accepts) and `axpy(y, x, num, den, sign)`
        (y += sign num / den x in place) every scalar product and vector update is one launch and the cache holds
        d^t A d instead of its inverse; with `step` and `ortho` as well the two scalar products of an iteration are one launch for
        all fields and its two updates another (pl_cg_dot_axpy; options.opts.cg_merged = False keeps them apart), likewise the
        re-orthogonalisation.  (Joining each pair into one launch with a grid-wide barrier was measured on MI355X to cost what the
        kernel boundary costs -- DESIGN.md sections 5 and 9 -- and is no longer a solver option; the C entry keeps the form.)
    """
    if cache is None:
        cache = cache_mem()
    n_pre = len(pre_ops)
    on_dev = n_pre == 1 and hasattr(dot_op, 'dev')
    fused = on_dev and hasattr(dot_op, 'axpy') and hasattr(dot_op, 'parts')
    return _cd_solve(x, b, fwd_op, pre_ops, dot_op, criterion, tr, cache, roundoff, x_is_zero, b_scratch, x_uninit)


def _zero(x):
    for name in ('tlm', 'elm', 'blm'):
        if hasattr(x, name):
            getattr(x, name).zero_()
    if not hasattr(x, 'elm'):
        x.zero_() if hasattr(x, 'zero_') else x.fill(0.)


def _cd_solve(x, b, fwd_op, pre_ops, dot_op, criterion, tr, cache, roundoff, x_is_zero, b_scratch, x_uninit=False):
    assert x_is_zero or not x_uninit
    n_pre = len(pre_ops)
    on_dev = n_pre == 1 and hasattr(dot_op, 'dev')
    fused = on_dev and hasattr(dot_op, 'axpy') and hasattr(dot_op, 'parts')
    merged = fused and hasattr(dot_op, 'step') and hasattr(dot_op, 'ortho') and options.opts.cg_merged
    if x_is_zero and b_scratch:
        residual, b = b, None
    else:
        residual = b * 1.0 if x_is_zero else b - fwd_op(x)
    it = 0
    # The criterion is asked before the preconditioners run: the search directions of an iteration that is not going to
    # happen are never formed (the reference forms and drops them, cd_solve.py:59,93 -- with nested multigrid stages of three
    # iterations that is a quarter of all coarse work).  Same calls of criterion with the same arguments, same iterates.
    if criterion(it, x, residual):
        if x_uninit:
            _zero(x)
        return it
    searchdirs = [op(residual) for op in pre_ops]
    # the two scalar products of a step from the kernel that writes fwd_op's result (fwd_op.with_dots, dot_op.step(pre=...)): one launch
    # less per iteration (options.opts.cg_post_dots = False: their own launch)
    post_dots = merged and hasattr(fwd_op, 'with_dots') and options.opts.cg_post_dots
    while True:
        pre = None
        if post_dots:
            q, pre = fwd_op.with_dots(searchdirs[0], residual)
            searchfwds = [q]
        else:
            searchfwds = [fwd_op(d) for d in searchdirs]
        if fused:
            fresh_residual = np.mod(it + 1, roundoff) == 0
            kw0 = {}
            if x_uninit and it == 0:  # the first step writes x (or x is zero-filled now)
                if pre is not None:
                    kw0 = {'x_init': True}
                else:
                    _zero(x)
            active = getattr(criterion, 'active', None)  # block vectors: 0 / 1 per entry, entries that have converged stand still
            if active is not None:
                assert merged, 'per-entry stopping of a block solve needs dot_op.step'
                # (a residual refresh recomputes the frozen entries' residuals too: harmless, their solutions no longer move and
                # the monitor's verdict on them is final)
                kw = {} if pre is None else {'pre': pre}
                dTAd, delta = dot_op.step(x, searchdirs[0], residual, searchfwds[0], update_r=not fresh_residual, active=active, **kw, **kw0)
            elif pre is not None:
                dTAd, delta = dot_op.step(x, searchdirs[0], residual, searchfwds[0], update_r=not fresh_residual, pre=pre, **kw0)
            elif merged:  # both scalar products in one launch, both updates of all fields in another (or all in one)
                dTAd, delta = dot_op.step(x, searchdirs[0], residual, searchfwds[0], update_r=not fresh_residual)
            else:
                dTAd = dot_op.parts(searchdirs[0], searchfwds[0])
                delta = dot_op.parts(searchdirs[0], residual)
                dot_op.axpy(x, searchdirs[0], delta, dTAd, 1.0)
                if not fresh_residual:
                    dot_op.axpy(residual, searchfwds[0], delta, dTAd, -1.0)
            cache.store(it, [dTAd, searchdirs, searchfwds])
            it += 1
            if fresh_residual:
                assert b is not None, 'b_scratch: the solve ran into a residual refresh'
                residual = b - fwd_op(x)
            if criterion(it, x, residual):
                cache.trim(range(tr(it + 1), it))
                return it
            # one past direction to orthogonalise against (tr_cg) and a preconditioner whose last kernel can form <s, q'> on its way out:
            # the re-orthogonalisation is its update alone
            if post_dots and tr(it) == it - 1 and hasattr(pre_ops[0], 'with_dot') and hasattr(dot_op, 'lmin'):
                prev_dTAd, prev_dirs, prev_fwds = cache.restore(it - 1)
                s, opre = pre_ops[0].with_dot(residual, prev_fwds[0], dot_op.lmin)
                searchdirs = [s]
                if opre is not None:
                    dot_op.ortho(s, prev_fwds[0], prev_dirs[0], prev_dTAd, pre=opre)
                    cache.trim(range(tr(it + 1), it))
                    continue
            else:
                searchdirs = [op(residual) for op in pre_ops]
            for titer in range(tr(it), it):
                prev_dTAd, prev_dirs, prev_fwds = cache.restore(titer)
                if merged:
                    dot_op.ortho(searchdirs[0], prev_fwds[0], prev_dirs[0], prev_dTAd)
                else:
                    dot_op.axpy(searchdirs[0], prev_dirs[0], dot_op.parts(searchdirs[0], prev_fwds[0]), prev_dTAd, -1.0)
            cache.trim(range(tr(it + 1), it))
            continue
        if x_uninit and it == 0:
            _zero(x)
        if on_dev:
            dTAd_inv = 1.0 / dot_op.dev(searchdirs[0], searchfwds[0])
            alphas = [dot_op.dev(searchdirs[0], residual) * dTAd_inv]
        else:
            deltas = [dot_op(d, residual) for d in searchdirs]
            dTAd = np.zeros((n_pre, n_pre))
            for i1 in range(n_pre):
                for i2 in range(i1 + 1):
                    dTAd[i1, i2] = dTAd[i2, i1] = dot_op(searchdirs[i1], searchfwds[i2])
            dTAd_inv = np.linalg.inv(dTAd)
            alphas = np.dot(dTAd_inv, deltas)
        for d, alpha in zip(searchdirs, alphas):
            x += d * alpha
        cache.store(it, [dTAd_inv, searchdirs, searchfwds])
        it += 1
        if np.mod(it, roundoff) == 0:
            assert b is not None, 'b_scratch: the solve ran into a residual refresh'
            residual = b - fwd_op(x)
        else:
            for q, alpha in zip(searchfwds, alphas):
                residual -= q * alpha
        if criterion(it, x, residual):
            cache.trim(range(tr(it + 1), it))
            return it
        searchdirs = [op(residual) for op in pre_ops]
        for titer in range(tr(it), it):
            prev_dTAd_inv, prev_dirs, prev_fwds = cache.restore(titer)
            for d in searchdirs:
                if on_dev:
                    d -= prev_dirs[0] * (dot_op.dev(d, prev_fwds[0]) * prev_dTAd_inv)
                else:
                    proj = [dot_op(d, pq) for pq in prev_fwds]
                    betas = np.dot(prev_dTAd_inv, proj)
                    for beta, pd in zip(betas, prev_dirs):
                        d -= pd * beta
        cache.trim(range(tr(it + 1), it))
