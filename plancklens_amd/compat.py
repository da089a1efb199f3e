"""Import contract of the reference's parameter files (second table; params/idealized_example.py:24-33).

Those files say `import plancklens`, `from plancklens.filt import filt_simple, filt_util`, `from plancklens.n1 import n1`,
`from plancklens.sims import planck2018_sims, phas, maps, utils`, `import healpy as hp`.  `install()` registers this
package and its modules in sys.modules under the reference's names, and `plancklens_amd.hp` under the name `healpy` when
the real healpy is not importable, so that such a file loads unchanged and instantiates the MI355X libraries:

    import plancklens_amd.compat; plancklens_amd.compat.install()
    par = SourceFileLoader('par', '.../params/idealized_example.py').load_module()

What a parameter file can still trip over is data, not code: `hp.pixwin` needs healpy's window-function tables
(see hp.pixwin) and `planck2018_sims` reads the FFP10 files under $CFS.
"""
import importlib
import importlib.util
import sys

_MODULES = ['utils', 'utils_qe', 'utils_spin', 'qest', 'qecl', 'qresp', 'nhl', 'shts', 'wigners',
            'filt', 'filt.filt_simple', 'filt.filt_util', 'filt.filt_cinv',
            'qcinv', 'qcinv.cd_solve', 'qcinv.cd_monitors', 'qcinv.multigrid', 'qcinv.opfilt_tt', 'qcinv.opfilt_pp', 'qcinv.opfilt_tp',
            'qcinv.dense', 'qcinv.template_removal', 'qcinv.util_alm', 'qcinv.util',
            'sims', 'sims.maps', 'sims.cmbs', 'sims.phas', 'sims.utils', 'sims.planck2018_sims',
            'helpers', 'helpers.mpi', 'helpers.sql', 'n1', 'n1.n1']


def install(healpy='auto', verbose=False):
    """healpy: 'auto' = alias plancklens_amd.hp as `healpy` only if the real package cannot be imported; True / False force it.
    Returns the list of names registered.  Idempotent; never replaces a module that something else already registered
    under one of the names (a real `plancklens` installation wins -- remove it from sys.path to use this package)."""
    import plancklens_amd
    done = []

    def reg(name, mod):
        if name in sys.modules and sys.modules[name] is not mod:
            if verbose:
                print('plancklens_amd.compat: %s is already registered, left alone' % name)
            return
        sys.modules[name] = mod
        done.append(name)

    try:
        spec = importlib.util.find_spec('plancklens')
    except (ImportError, ValueError):
        spec = None
    if spec is not None and 'plancklens' not in sys.modules:
        raise ImportError('a real plancklens package is importable (%s): plancklens_amd.compat.install() would shadow it' % spec.origin)
    reg('plancklens', plancklens_amd)
    for m in _MODULES:
        mod = importlib.import_module('plancklens_amd.' + m)
        reg('plancklens.' + m, mod)
        parent, _, leaf = m.rpartition('.')
        par = plancklens_amd if not parent else importlib.import_module('plancklens_amd.' + parent)
        if not hasattr(par, leaf):
            setattr(par, leaf, mod)
    if healpy == 'auto':
        try:
            healpy = importlib.util.find_spec('healpy') is None and 'healpy' not in sys.modules
        except (ImportError, ValueError):
            healpy = 'healpy' not in sys.modules
    if healpy:
        from . import hp
        reg('healpy', hp)
    return done
