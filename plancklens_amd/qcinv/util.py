"""Small helpers of the CG filter behind the names of plancklens/qcinv/util.py (`dt`, `stopwatch` :8-36, `jit` :39-61,
`read_map` / `load_map` :63-79, `mask_hash` :81-95): wall-clock bookkeeping for the iteration log, deferred construction of
the (expensive) filter and chain objects, and the "map given as array | path | path,field | callable | list of those"
convention of the inverse-noise arguments."""
import time

import numpy as np

from .. import hp, utils


class dt(object):
    """a duration in seconds that prints as hh:mm:ss"""

    def __init__(self, seconds):
        self.dt = seconds

    def __int__(self):
        return int(self.dt)

    def __str__(self):
        m, s = divmod(int(self.dt), 60)
        h, m = divmod(m, 60)
        return '%02d:%02d:%02d' % (h, m, s)


class stopwatch(object):
    """lap() -> (since start, since previous lap or elapsed()); elapsed() -> since start; both as `dt`"""

    def __init__(self):
        self.st = self.lt = time.time()

    def _mark(self):
        now, prev = time.time(), self.lt
        self.lt = now
        return now - self.st, now - prev

    def lap(self):
        total, split = self._mark()
        return dt(total), dt(split)

    def elapsed(self):
        return dt(self._mark()[0])


class jit(object):
    """Stand-in for an object that is only built (ctype(*args, **kwargs)) when something is asked of it: the filter
    libraries construct their inverse-noise filters and multigrid chains this way, so that importing a parameter file does
    not read maps or touch the GPU.  Attribute reads and writes go to the real object."""
    _SLOTS = ('_jit_recipe', '_jit_target')

    def __init__(self, ctype, *cargs, **ckwds):
        object.__setattr__(self, '_jit_recipe', (ctype, cargs, ckwds))
        object.__setattr__(self, '_jit_target', None)

    def instantiate(self):
        recipe = object.__getattribute__(self, '_jit_recipe')
        if recipe is not None:
            ctype, cargs, ckwds = recipe
            object.__setattr__(self, '_jit_target', ctype(*cargs, **ckwds))
            object.__setattr__(self, '_jit_recipe', None)
        return object.__getattribute__(self, '_jit_target')

    def __getattr__(self, name):  # only reached for names the proxy itself does not have
        return getattr(self.instantiate(), name)

    def __setattr__(self, name, value):
        setattr(self.instantiate(), name, value)


def unjit(obj):
    """the object behind a `jit` stand-in (built now if it was not yet); anything else is returned as it is"""
    return obj.instantiate() if isinstance(obj, jit) else obj


def read_map(m):
    """The map behind `m`: an array is returned as is, a callable is called, 'file.fits' and 'file.fits,3' are read with
    hp.read_map (field 3 in the second form), a list stands for the product of its members."""
    if isinstance(m, list):
        out = read_map(m[0])
        for other in m[1:]:
            out = out * read_map(other)
        return out
    if callable(m):
        return m()
    if isinstance(m, str):
        path, _, field = m.partition(',')
        return hp.read_map(path, field=int(field)) if field else hp.read_map(path)
    return m


load_map = read_map


def mask_hash(m, dtype=bool):
    """short, stable label of a map argument for the hash dictionaries (paths by name, arrays by content)"""
    if m is None:
        return "none"
    if isinstance(m, list):
        return ''.join(mask_hash(x, dtype=dtype) for x in m)
    if isinstance(m, str):
        return m.replace('/', '_sl_').replace('.', '_')
    if isinstance(m, np.ndarray):
        return utils.clhash(m, dtype=dtype)
    if callable(m):
        return 'callable'
    raise AssertionError('mask_hash: %s not supported' % type(m))
