"""Run-time options of the host layer in ONE object (the C library takes its own as `pl_plan_opts` arguments, shts.plan_options).

Every entry used to be an environment variable of its own, read at the point of use; two libraries of one process could not differ in
them and a typo was silently the default.  Now:

    from plancklens_amd import options
    options.opts.cg_batch = 8                       # for the process
    with options.override(qe_graph=False): ...      # for a block (tests, A/B measurements)
    PLENS_OPTIONS="cg_graph=0,batch2=0" python ...  # from a shell (tools/*.sh): parsed once at import, unknown names raise

Where a class has the natural place for one, the option is also a constructor argument / attribute that defaults to the entry here
(`qest.library.use_graph`, `filt_cinv.library_cinv_sepTP(batch=...)`).  Switches whose A/B measurement lost are gone, not defaulted
off (`PLENS_CG_ONE_LAUNCH`: a grid barrier inside `pl_cg_dot_axpy`, measured equal to the kernel boundary it replaced -- DESIGN.md
sections 5 and 9).

`stats` counts the events a caller may want to ask about afterwards instead of reading stdout: graph captures that failed and fell
back to eager launches (`qe_graph_fallbacks`, `cg_graph_fallbacks`), captures that succeeded.
"""
import contextlib
import os
import threading


class Options(object):
    __slots__ = ('batch2', 'qe_graph', 'qe_graph_min_nside', 'qe_indirect', 'async_d2h', 'd2h_blocks', 'cg_graph', 'cg_merged', 'cg_post_dots', 'tp_concurrent', 'tp_pace', 'cg_batch',
                 'tproj_harm', 'tproj_md', 'dense_block', 'debug')

    def __init__(self):
        self.batch2 = True        # estimator: the same spin synthesis of two simulations on one recursion (pl_alm2map_batch2 / _grad_pair)
        self.qe_graph = True      # estimator: a pair of reconstructions as one replayed HIP graph (qest.library._pair_graph)
        self.qe_graph_min_nside = 0   # ... on grids of at least this nside (0: every grid; at nside 512 eager launches and the replay are equal within the noise of a shared host)
        self.qe_indirect = True   # ... its input maps read through a table of device addresses (pl_map2alm_ind) instead of copied into static slots
        self.async_d2h = True     # estimator: results cross PCIe on a copy stream, handed out as futures
        self.d2h_blocks = 64      # workgroups of the device -> pinned-host copy kernel
        self.cg_graph = True      # qcinv: fixed-count nested solves captured into HIP graphs (multigrid.pre_op_multigrid)
        self.cg_merged = True     # qcinv: the two scalar products / two updates of a step in one launch each (pl_cg_dot_axpy)
        self.cg_post_dots = True  # qcinv: the scalar products of a step from the kernel that writes the operator's result (pl_plan_arm_post_dots)
        self.tp_concurrent = True  # filt_cinv: cinv_t and cinv_p of a simulation at the same time on two streams
        self.tp_pace = True       # ... their top-level iterations started together (multigrid.pace)
        self.cg_batch = 4         # filt_cinv.library_cinv_sepTP.filter_sims: simulations per block solve
        self.tproj_harm = True    # opfilt_*: template projection as a low-rank update in harmonic space
        self.tproj_md = True      # opfilt_*: monopole + dipole projection from ring geometry (pl_template_project_md_b)
        self.dense_block = 32     # dense preconditioner build: columns per batched operator application
        self.debug = False        # tracebacks of failed graph captures, stream-assignment log

    def set(self, **kw):
        for k, v in kw.items():
            if k not in self.__slots__:
                raise KeyError('unknown option %r (known: %s)' % (k, ', '.join(self.__slots__)))
            cur = getattr(self, k)
            if isinstance(cur, bool):
                v = v if isinstance(v, bool) else str(v).strip().lower() not in ('0', 'false', 'no', 'off', '')
            else:
                v = type(cur)(v)
            setattr(self, k, v)
        return self

    def as_dict(self):
        return {k: getattr(self, k) for k in self.__slots__}


opts = Options()
if os.environ.get('PLENS_OPTIONS'):
    opts.set(**dict(kv.split('=', 1) for kv in os.environ['PLENS_OPTIONS'].split(',') if kv.strip()))


@contextlib.contextmanager
def override(**kw):
    """the named options changed inside the block, restored on exit (process-wide: not for blocks that run beside other solver threads)"""
    old = {k: getattr(opts, k) for k in kw}
    opts.set(**kw)
    try:
        yield opts
    finally:
        for k, v in old.items():
            setattr(opts, k, v)


_stats_lock = threading.Lock()
stats = {'qe_graph_captures': 0, 'qe_graph_fallbacks': 0, 'cg_graph_captures': 0, 'cg_graph_fallbacks': 0, 'last_fallback': None}


def count(name, detail=None):
    with _stats_lock:
        stats[name] = stats.get(name, 0) + 1
        if name.endswith('_fallbacks'):
            stats['last_fallback'] = detail
