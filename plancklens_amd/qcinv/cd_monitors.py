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

    def criterion(self, iter, soltn, resid):
        if self.quiet and self.eps_min == 0.:
            return iter >= self.iter_max
        delta = self.dot_op(resid, resid)
        if iter == 0 and self.d0 is None:
            self.d0 = delta
        if self.logger is not None:
            self.logger(iter, np.sqrt(delta / self.d0), watch=self.watch, soltn=soltn, resid=resid)
        return (iter >= self.iter_max) or (delta <= self.eps_min ** 2 * self.d0)

    def __call__(self, *args):
        return self.criterion(*args)
