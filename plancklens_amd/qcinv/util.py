"""Small helpers of the CG filter, API of plancklens/qcinv/util.py (`stopwatch` :21-36, `jit` :39-61,
`read_map` :63-79, `mask_hash` :81-95)."""
import time

import numpy as np

from .. import hp, utils


class dt(object):
    def __init__(self, _dt):
        self.dt = _dt

    def __str__(self):
        return '%02d:%02d:%02d' % (self.dt // 3600, (self.dt % 3600) // 60, self.dt % 60)

    def __int__(self):
        return int(self.dt)


class stopwatch(object):
    def __init__(self):
        self.st = time.time()
        self.lt = self.st

    def lap(self):
        lt = time.time()
        ret = (dt(lt - self.st), dt(lt - self.lt))
        self.lt = lt
        return ret

    def elapsed(self):
        lt = time.time()
        self.lt = lt
        return dt(lt - self.st)


class jit(object):
    """Just-in-time instantiation proxy: the wrapped object is built on first attribute access."""

    def __init__(self, ctype, *cargs, **ckwds):
        self.__dict__['__jit_args'] = [ctype, cargs, ckwds]
        self.__dict__['__jit_obj'] = None

    def instantiate(self):
        ctype, cargs, ckwds = self.__dict__['__jit_args']
        self.__dict__['__jit_obj'] = ctype(*cargs, **ckwds)
        del self.__dict__['__jit_args']

    def __getattr__(self, attr):
        if self.__dict__['__jit_obj'] is None:
            self.instantiate()
        return getattr(self.__dict__['__jit_obj'], attr)

    def __setattr__(self, attr, val):
        if self.__dict__['__jit_obj'] is None:
            self.instantiate()
        setattr(self.__dict__['__jit_obj'], attr, val)


def read_map(m):
    """Map given as array, callable, path ('file.fits' or 'file.fits,field') or list of those (multiplied)."""
    if callable(m):
        return m()
    if isinstance(m, list):
        ma = read_map(m[0])
        for m2 in m[1:]:
            ma = ma * read_map(m2)
        return ma
    if not isinstance(m, str):
        return m
    if ',' not in m:
        return hp.read_map(m)
    fn, field = m.split(',')
    return hp.read_map(fn, field=int(field))


load_map = read_map


def mask_hash(m, dtype=bool):
    if m is None:
        return "none"
    if isinstance(m, list):
        return ''.join(mask_hash(mi, dtype=dtype) for mi in m)
    if isinstance(m, str):
        return m.replace('/', '_sl_').replace('.', '_')
    if isinstance(m, np.ndarray):
        return utils.clhash(m, dtype=dtype)
    if callable(m):
        return 'callable'
    assert 0, 'not implemented'
