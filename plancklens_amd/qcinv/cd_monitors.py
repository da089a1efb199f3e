"""Convergence monitor of the CG solver, API of plancklens/qcinv/cd_monitors.py (`monitor_basic` :12-41)."""
import sys

import numpy as np

from . import util

logger_basic = (lambda iter, eps, watch=None, **kwargs: sys.stdout.write('[' + str(watch.elapsed()) + '] ' + str((iter, eps)) + '\n'))
logger_none = (lambda iter, eps, watch=None, **kwargs: 0)


class monitor_basic(object):
    """Stops at iter_max or once |residual|^2 <= eps_min^2 d0 (d0: the first |residual|^2 unless given)."""

    def __init__(self, dot_op, iter_max=1000, eps_min=1.0e-10, logger=logger_basic, d0=None, quiet=False):
        """quiet: nobody reads this monitor's log.  With eps_min = 0 as well (fixed iteration count, the nested
        multigrid stages) the residual norm is not needed at all and is not computed: no host synchronisation."""
        self.quiet = quiet
        self.dot_op = dot_op
        self.iter_max = iter_max
        self.eps_min = eps_min
        self.logger = logger
        self.d0 = d0
        self.watch = util.stopwatch()
        # block vectors (several right-hand sides in one solve): entries that have met the stopping rule, and the 0 / 1 step-length
        # mask cd_solve hands to dot_op.step (None while every entry is still iterating)
        self.done = None
        self.active = None

    def criterion(self, iter, soltn, resid):
        if self.quiet and self.eps_min == 0.:
            return iter >= self.iter_max
        delta = self.dot_op(resid, resid)
        if iter == 0 and self.d0 is None:
            self.d0 = delta
        if np.ndim(delta) > 0:
            return self._criterion_block(iter, soltn, resid, np.asarray(delta, dtype=float))
        if self.logger is not None:
            with np.errstate(invalid='ignore', divide='ignore'):  # (0 / 0 for an all-zero right-hand side: nan in the log, as in the reference)
                eps = np.sqrt(np.float64(delta) / np.float64(self.d0))
            self.logger(iter, eps, watch=self.watch, soltn=soltn, resid=resid)
        return (iter >= self.iter_max) or (delta <= self.eps_min ** 2 * self.d0)

    def _criterion_block(self, iter, soltn, resid, delta):
        """The stopping rule applied to every entry of a block vector by itself: an entry that meets it is frozen (its step length
        is zero from then on, `active`), exactly where its own solve would have returned; the block solve ends when all have."""
        d0 = np.asarray(self.d0, dtype=float)
        if self.done is None:
            self.done = np.zeros(delta.shape, dtype=bool)
        if self.logger is not None:
            live = ~self.done if not np.all(self.done) else np.ones_like(self.done)
            with np.errstate(invalid='ignore', divide='ignore'):  # (an all-zero right-hand side has d0 = 0)
                eps = np.sqrt(delta[live] / d0[live])
            self.logger(iter, float(np.nanmax(eps)) if np.any(np.isfinite(eps)) else 0., watch=self.watch, soltn=soltn, resid=resid)
        newly = (delta <= self.eps_min ** 2 * d0) & ~self.done
        if np.any(newly):
            self.done |= newly
            if not np.all(self.done):
                import torch
                self.active = torch.from_numpy((~self.done).astype(np.float64)).to(resid.device if hasattr(resid, 'device') else _first_device(resid))
        return (iter >= self.iter_max) or bool(np.all(self.done))

    def __call__(self, *args):
        return self.criterion(*args)


def _first_device(v):
    for name in ('tlm', 'elm'):
        if hasattr(v, name):
            return getattr(v, name).device
    return v.device
