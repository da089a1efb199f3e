"""Conjugate-gradient (anisotropic noise / masked sky) CMB filtering on the MI355X, API of
plancklens/filt/filt_cinv.py (`cinv` :22-53, `cinv_t` :56-203, `cinv_p` :206-338, `cinv_tp` :341-512,
`library_cinv_sepTP` :515-580, `library_cinv_jTP` :582-620).

Same multigrid chains, rescalings, side files (ftl.dat, fel.dat, fbl.dat, tal.dat, fmask.fits.gz, filt_hash.pk) and
stopping rule as the reference; the solve itself (plancklens_amd.qcinv) runs with every vector, map and SHT in HBM."""
from __future__ import print_function

import os
import pickle as pk

import numpy as np
import torch

from .. import dev, hp, options, utils
from ..helpers import mpi
from ..qcinv import cd_solve, multigrid, opfilt_pp, opfilt_tp, opfilt_tt, util, util_alm
from . import filt_simple


# The reference's default multigrid chains as data (filt_cinv.py:112-116 temperature, :236-239 polarization, :398-407 joint): per
# stage (band-limit, nside, dense band-limit or None); every coarse stage runs three unmonitored iterations, the stage it
# preconditions splits at its band-limit; the top stage (the filter's own lmax and nside) iterates to |residual| <= 1e-5 |rhs|.
_COARSE_STAGES = {'t': [(256, 128, 64), (512, 256, None), (1024, 512, None)],
                  'p': [(512, 256, 32), (1024, 512, None)],
                  'tp': [(256, 128, 64), (512, 256, None), (1024, 512, None)]}


def default_chain_descr(kind, lmax, nside, pcf):
    """chain descriptor in the reference's mini-language for cinv_t ('t'), cinv_p ('p') or cinv_tp ('tp'); pcf: cache file of the
    dense coarse preconditioner ('' / None: none)"""
    coarse = _COARSE_STAGES[kind]
    chain, below = [], None
    for depth, (st_lmax, st_nside, dense_lmax) in zip(range(len(coarse), 0, -1), coarse):
        if dense_lmax is not None:
            descr = "split(dense(%s), %d, diag_cl)" % (pcf, dense_lmax)
        else:
            descr = "split(stage(%d),  %d, diag_cl)" % (depth + 1, below)
        chain.append([depth, [descr], st_lmax, st_nside, 3, 0.0, cd_solve.tr_cg, cd_solve.cache_mem()])
        below = st_lmax
    chain.append([0, ["split(stage(1), %d, diag_cl)" % below], lmax, nside, np.inf, 1.0e-5, cd_solve.tr_cg, cd_solve.cache_mem()])
    return chain


# ---- cinv_t and cinv_p of a simulation at the same time -------------------------------------------------------------------
P_CONTEXT = 1  # the plan context (shts.plan_context) polarization solves run in when they may overlap a temperature solve
_TP_SIDE_STREAMS = {}


def _tp_side_stream():
    """the stream of the polarization solve, one per device: one whose kernels do run beside those of the caller's stream
    (dev.concurrent_stream: two streams that the runtime put on one hardware queue would run the two solves one after the other)"""
    d = torch.cuda.current_device()
    if d not in _TP_SIDE_STREAMS:
        _TP_SIDE_STREAMS[d] = dev.concurrent_stream()
    return _TP_SIDE_STREAMS[d]


def _p_context():
    """the plan context of this library's polarization solves: P_CONTEXT unless options.opts.tp_concurrent is off"""
    from .. import shts
    return shts.plan_context(P_CONTEXT if options.opts.tp_concurrent else 0)


def run_tp(job_t, job_p, warm=True):
    """(job_t(), job_p()) with the two jobs -- a temperature and a polarization solve, independent in the reference
    (filt_cinv.py:515-580 `_apply_ivf_t` / `_apply_ivf_p`, called one after the other from run_qlms.py:57-62) -- running at the same
    time in this process: job_t on the calling thread and torch's current stream, job_p on a helper thread and a side stream, inside
    plan context P_CONTEXT (own plan workspaces, scratch buffers and captured graphs; shts.plan_context).  Most of an iteration of
    either solve is the latency-bound chain of small kernels of its coarse multigrid levels, which leaves the other solve room on
    the chip.  The GIL is released inside every library call and every wait on the device, so the two launch threads interleave.
    warm=False: the same two jobs in the same contexts, one after the other on the calling thread -- used for the first solve of a
    shape, which allocates workspaces and captures the nested stages into HIP graphs.  Results are identical either way."""
    from .. import shts
    if not warm or not options.opts.tp_concurrent:
        rt = job_t()
        with _p_context():
            rp = job_p()
        return rt, rp
    import threading
    main = torch.cuda.current_stream()
    side = _tp_side_stream()
    side.wait_stream(main)  # inputs made on the caller's stream
    devid = torch.cuda.current_device()
    box = {}

    def work():
        try:
            torch.cuda.set_device(devid)  # (a new thread starts on device 0)
            with torch.cuda.stream(side), shts.plan_context(P_CONTEXT):
                box['r'] = job_p()
        except BaseException as e:  # re-raised on the calling thread
            box['e'] = e
    th = threading.Thread(target=work, name='plens_cinv_p')
    th.start()
    try:
        rt = job_t()
    finally:
        th.join()
        main.wait_stream(side)
    if 'e' in box:
        raise box['e']
    return rt, box['r']


def _dev_tensors(x):
    if isinstance(x, torch.Tensor):
        return [x]
    if isinstance(x, (list, tuple)):
        return [t for y in x for t in _dev_tensors(y)]
    return []


def apply_ivf_tp(cinv_t, tmap, cinv_p, pmap, soltn_t=None, soltn_p=None):
    """(cinv_t.apply_ivf(tmap), cinv_p.apply_ivf(pmap)) with the two solves overlapped on two streams of this process (run_tp).
    tmap / pmap may be lists of maps / of (Q, U) pairs: block solves (apply_ivf_batch).  The first call of a shape runs the solves one
    after the other (set-up, graph capture)."""
    blk_t = isinstance(tmap, (list, tuple))
    blk_p = isinstance(pmap[0], (list, tuple))
    key = ('tp', len(tmap) if blk_t else 0, len(pmap) if blk_p else 0)
    warm_t, warm_p = cinv_t.__dict__.setdefault('_tp_warm', set()), cinv_p.__dict__.setdefault('_tp_warm', set())
    job_t = (lambda: cinv_t.apply_ivf_batch(tmap, soltns=soltn_t)) if blk_t else (lambda: cinv_t.apply_ivf(tmap, soltn=soltn_t))
    job_p = (lambda: cinv_p.apply_ivf_batch(pmap, soltns=soltn_p)) if blk_p else (lambda: cinv_p.apply_ivf(pmap, soltn=soltn_p))
    warm = key in warm_t and key in warm_p
    paced = warm and options.opts.tp_concurrent and options.opts.tp_pace
    if paced:  # the two solves meet before every top-level preconditioner call (multigrid.pace)
        pc = multigrid.pace(2)
        util.unjit(cinv_t.chain).pace = util.unjit(cinv_p.chain).pace = pc
    try:
        rt, rp = run_tp(job_t, job_p, warm=warm)
    finally:
        if paced:
            util.unjit(cinv_t.chain).pace = util.unjit(cinv_p.chain).pace = None
    if warm:  # results made on the side stream are handed to the caller's stream (their memory returns to the side stream's pool)
        for t in _dev_tensors(rp):
            t.record_stream(torch.cuda.current_stream())
    warm_t.add(key)
    warm_p.add(key)
    return rt, rp


class cinv(object):
    def __init__(self, lib_dir, lmax):
        self.lib_dir = lib_dir
        self.lmax = lmax

    def _load(self, name, lmax):
        if lmax is None:
            lmax = self.lmax
        ret = np.loadtxt(os.path.join(self.lib_dir, name))
        assert len(ret) > lmax, (len(ret), lmax)
        return ret[:lmax + 1]

    def get_tal(self, a, lmax=None):
        assert a.lower() in ['t', 'e', 'b'], a
        return self._load("tal.dat", lmax)

    def get_fmask(self):
        return hp.read_map(os.path.join(self.lib_dir, "fmask.fits.gz"))

    def get_ftl(self, lmax=None):
        return self._load("ftl.dat", lmax)

    def get_fel(self, lmax=None):
        return self._load("fel.dat", lmax)

    def get_fbl(self, lmax=None):
        return self._load("fbl.dat", lmax)

    def _setup_dir(self, files):
        """rank 0 writes the side files, everybody waits, then the hash is checked (filt_cinv.py:121-139)."""
        if mpi.rank == 0:
            if not os.path.exists(self.lib_dir):
                os.makedirs(self.lib_dir)
            fn_hash = os.path.join(self.lib_dir, "filt_hash.pk")
            if not os.path.exists(fn_hash):
                pk.dump(self.hashdict(), open(fn_hash, 'wb'), protocol=2)
            for name, make in files:
                if not os.path.exists(os.path.join(self.lib_dir, name)):
                    make(os.path.join(self.lib_dir, name))
        mpi.barrier()
        fn_hash = os.path.join(self.lib_dir, "filt_hash.pk")
        utils.hash_check(pk.load(open(fn_hash, 'rb')), self.hashdict(), fn=fn_hash)


def _nlev_uKamin(ninv):
    """Effective white-noise level of an inverse-variance map over its unmasked pixels."""
    npix = ninv.numel()
    nz = int((ninv != 0.0).sum())
    return np.sqrt(4. * np.pi / npix / float(ninv.sum()) * nz) * 180. * 60. / np.pi


class cinv_t(cinv):
    r"""Temperature-only inverse-variance filter  :math:`\bar T = S^{-1}(S^{-1} + B^tY^tN^{-1}YB)^{-1}B^tY^tN^{-1} d`.

        Args as the reference: lib_dir, lmax, nside, cl (dict with 'tt'), transf, ninv (list of maps / paths whose
        product is the inverse pixel variance), rescal_cl ('default': D_l modes are solved for), marge_monopole,
        marge_dipole, marge_maps, pcf (dense preconditioner cache), chain_descr.
    """

    def __init__(self, lib_dir, lmax, nside, cl, transf, ninv, rescal_cl='default', marge_monopole=True, marge_dipole=True,
                 marge_maps=(), pcf='default', chain_descr=None):
        assert lib_dir is not None and lmax >= 1024 and nside >= 512, (lib_dir, lmax, nside)
        assert isinstance(ninv, list)
        super(cinv_t, self).__init__(lib_dir, lmax)
        if rescal_cl in ['default', None]:
            default_rescal = True
            rescal_cl = np.sqrt(np.arange(lmax + 1, dtype=float) * np.arange(1, lmax + 2, dtype=float) / 2. / np.pi)
        else:
            default_rescal = False
            assert len(rescal_cl) >= lmax + 1, [rescal_cl.shape, lmax]
        dl = {k: rescal_cl[:lmax + 1] ** 2 * cl[k][:lmax + 1] for k in cl.keys()}
        transf_dl = transf[:lmax + 1] * utils.cli(rescal_cl)
        self.nside = nside
        self.cl = cl
        self.dl = dl
        self.transf = transf[:lmax + 1]
        self.rescaled_transf = transf_dl
        self.rescal_cl = rescal_cl
        self.default_rescal = default_rescal
        self.ninv = ninv
        self.marge_monopole = marge_monopole
        self.marge_dipole = marge_dipole
        self.marge_maps = marge_maps
        pcf = os.path.join(lib_dir, "dense.pk") if pcf == 'default' else ''
        if chain_descr is None:
            chain_descr = default_chain_descr('t', lmax, nside, pcf)
        n_inv_filt = util.jit(opfilt_tt.alm_filter_ninv, ninv, transf_dl, marge_monopole=marge_monopole,
                              marge_dipole=marge_dipole, marge_maps=marge_maps)
        self.chain_descr = chain_descr
        self.chain = util.jit(multigrid.multigrid_chain, opfilt_tt, self.chain_descr, dl, n_inv_filt)
        self._setup_dir([("ftl.dat", lambda fn: np.savetxt(fn, self._calc_ftl())),
                         ("tal.dat", lambda fn: np.savetxt(fn, self._calc_tal())),
                         ("fmask.fits.gz", lambda fn: hp.write_map(fn, self._calc_mask()))])

    def _ninv_hash(self):
        return [utils.clhash(c) if (isinstance(c, np.ndarray) and c.size > 1) else c for c in self.ninv]

    def _calc_ftl(self):
        ninv = self.chain.n_inv_filt.n_inv
        nlev = _nlev_uKamin(ninv)
        print("cinv_t::noiseT_uk_arcmin = %.3f" % nlev)
        s_cls, b_transf = self.cl, self.transf
        if s_cls['tt'][0] == 0.:
            assert self.chain.n_inv_filt.marge_monopole
        if s_cls['tt'][1] == 0.:
            assert self.chain.n_inv_filt.marge_dipole
        ftl = utils.cli(s_cls['tt'][0:self.lmax + 1] + (nlev * np.pi / 180. / 60.) ** 2 * utils.cli(b_transf[0:self.lmax + 1] ** 2))
        if self.chain.n_inv_filt.marge_monopole:
            ftl[0] = 0.0
        if self.chain.n_inv_filt.marge_dipole:
            ftl[1] = 0.0
        return ftl

    def _calc_tal(self):
        return utils.cli(self.transf)

    def _calc_mask(self):
        ninv = self.chain.n_inv_filt.n_inv
        assert hp.npix2nside(ninv.numel()) == self.nside
        return dev.to_host((ninv > 0).to(torch.float64))

    def hashdict(self):
        hd = {'lmax': self.lmax, 'nside': self.nside, 'cltt': utils.clhash(self.cl['tt'][:self.lmax + 1]),
              'transf': utils.clhash(self.transf[:self.lmax + 1]), 'ninv': self._ninv_hash(),
              'marge_monopole': self.marge_monopole, 'marge_dipole': self.marge_dipole, 'marge_maps': self.marge_maps}
        if self.default_rescal is False:
            hd['rescal_cl'] = utils.clhash(self.rescal_cl)
        return hd

    def apply_ivf(self, tmap, soltn=None):
        """Inverse-variance filtered temperature alm of the map (numpy in -> numpy out, device tensor -> device tensor)."""
        on_dev = isinstance(tmap, torch.Tensor)
        if soltn is None:
            talm = torch.zeros(hp.Alm.getsize(self.lmax), dtype=torch.complex128, device=dev.device())
        else:
            talm = dev.to_dev(soltn, torch.complex128).clone()
        self.chain.solve(talm, tmap)
        talm = dev.almxfl(talm, self.rescal_cl)
        return talm if on_dev else dev.to_host(talm)

    def apply_ivf_batch(self, tmaps, soltns=None):
        """apply_ivf of several maps in ONE block solve: every launch of the multigrid CG carries all of them, each with its own
        scalar products, step lengths and stopping point (multigrid_chain.solve).  The reference filters its simulations one at a
        time (filt_cinv.py:196-203 under run_qlms.py:57-62); on the GPU the coarse levels of a solve are chains of dependent
        microsecond launches that take the extra right-hand sides almost for free.  Returns the list of filtered alms, equal to
        [apply_ivf(m) for m in tmaps]."""
        tmaps = list(tmaps)
        on_dev = isinstance(tmaps[0], torch.Tensor)
        nb, n = len(tmaps), hp.Alm.getsize(self.lmax)
        if soltns is None:
            talm = torch.zeros((nb, n), dtype=torch.complex128, device=dev.device())
        else:
            talm = torch.stack([dev.to_dev(s, torch.complex128) for s in soltns]).contiguous()
        self.chain.solve(talm, tmaps)
        talm = dev.almxfl(talm, self.rescal_cl)
        return [talm[i] if on_dev else dev.to_host(talm[i]) for i in range(nb)]


class cinv_p(cinv):
    r"""Polarization-only inverse-variance filter (E, B); ninv is a list of 1 (QQ = UU) or 3 (QQ, QU, UU) entries,
    each itself a list of maps / paths to multiply."""

    def __init__(self, lib_dir, lmax, nside, cl, transf, ninv, pcf='default', chain_descr=None, transf_blm=None,
                 marge_qmaps=(), marge_umaps=()):
        assert lib_dir is not None and lmax >= 1024 and nside >= 512, (lib_dir, lmax, nside)
        super(cinv_p, self).__init__(lib_dir, lmax)
        self.nside = nside
        self.cl = cl
        self.transf_e = transf
        self.transf_b = transf if transf_blm is None else transf_blm
        self.transf = transf if transf_blm is None else 0.5 * self.transf_e + 0.5 * self.transf_b
        self.ninv = ninv
        pcf = os.path.join(lib_dir, "dense.pk") if pcf == 'default' else None
        if chain_descr is None:
            chain_descr = default_chain_descr('p', lmax, nside, pcf)
        n_inv_filt = util.jit(opfilt_pp.alm_filter_ninv, ninv, transf[0:lmax + 1], b_transf_b=transf_blm,
                              marge_umaps=marge_umaps, marge_qmaps=marge_qmaps)
        self.chain_descr = chain_descr
        self.chain = util.jit(multigrid.multigrid_chain, opfilt_pp, chain_descr, cl, n_inv_filt)

        def _write_febl(fn):
            fel, fbl = self._calc_febl()
            np.savetxt(os.path.join(self.lib_dir, "fel.dat"), fel)
            np.savetxt(fn, fbl)
        self._setup_dir([("fbl.dat", _write_febl), ("tal.dat", lambda fn: np.savetxt(fn, self._calc_tal())),
                         ("fmask.fits.gz", lambda fn: hp.write_map(fn, self._calc_mask()))])

    def hashdict(self):
        return {'lmax': self.lmax, 'nside': self.nside, 'clee': utils.clhash(self.cl.get('ee', np.array([0.]))),
                'cleb': utils.clhash(self.cl.get('eb', np.array([0.]))), 'clbb': utils.clhash(self.cl.get('bb', np.array([0.]))),
                'transf': utils.clhash(self.transf), 'ninv': self._ninv_hash()}

    def _ninv_hash(self):
        return [[utils.clhash(c) if (isinstance(c, np.ndarray) and c.size > 1) else c for c in self.ninv[0]]]

    def apply_ivf(self, tmap, soltn=None):
        assert len(tmap) == 2
        on_dev = isinstance(tmap[0], torch.Tensor)
        n = hp.Alm.getsize(self.lmax)
        if soltn is not None:
            assert len(soltn) == 2 and hp.Alm.getlmax(np.size(soltn[0])) == self.lmax
            talm = util_alm.eblm([dev.to_dev(soltn[0], torch.complex128).clone(), dev.to_dev(soltn[1], torch.complex128).clone()])
        else:
            talm = util_alm.eblm([torch.zeros(n, dtype=torch.complex128, device=dev.device()),
                                  torch.zeros(n, dtype=torch.complex128, device=dev.device())])
        self.chain.solve(talm, [tmap[0], tmap[1]])
        if on_dev:
            return talm.elm, talm.blm
        return dev.to_host(talm.elm), dev.to_host(talm.blm)

    def apply_ivf_batch(self, pmaps, soltns=None):
        """apply_ivf of several (Q, U) pairs in ONE block solve (see cinv_t.apply_ivf_batch); returns the list of (elm, blm)."""
        pmaps = [list(p) for p in pmaps]
        assert all(len(p) == 2 for p in pmaps)
        on_dev = isinstance(pmaps[0][0], torch.Tensor)
        nb, n = len(pmaps), hp.Alm.getsize(self.lmax)
        if soltns is not None:
            talm = util_alm.eblm([torch.stack([dev.to_dev(s[k], torch.complex128) for s in soltns]).contiguous() for k in (0, 1)])
        else:
            talm = util_alm.eblm([torch.zeros((nb, n), dtype=torch.complex128, device=dev.device()) for _ in (0, 1)])
        self.chain.solve(talm, pmaps)
        if on_dev:
            return [(talm.elm[i], talm.blm[i]) for i in range(nb)]
        return [(dev.to_host(talm.elm[i]), dev.to_host(talm.blm[i])) for i in range(nb)]

    def _calc_febl(self):
        assert 'eb' not in self.chain.s_cls.keys()
        ninv = self.chain.n_inv_filt.get_ninv()
        if len(ninv) == 1:
            nlev = _nlev_uKamin(ninv[0])
        else:
            assert len(ninv) == 3
            nlev = 0.5 * _nlev_uKamin(ninv[0]) + 0.5 * _nlev_uKamin(ninv[2])
        print("cinv_p::noiseP_uk_arcmin = %.3f" % nlev)
        s_cls = self.chain.s_cls
        b_e, b_b = self.chain.n_inv_filt.b_transf_e, self.chain.n_inv_filt.b_transf_b
        fel = utils.cli(s_cls['ee'][:self.lmax + 1] + (nlev * np.pi / 180. / 60.) ** 2 * utils.cli(b_e[0:self.lmax + 1] ** 2))
        fbl = utils.cli(s_cls['bb'][:self.lmax + 1] + (nlev * np.pi / 180. / 60.) ** 2 * utils.cli(b_b[0:self.lmax + 1] ** 2))
        fel[0:2] *= 0.0
        fbl[0:2] *= 0.0
        return fel, fbl

    def _calc_tal(self):
        return utils.cli(self.transf)

    def _calc_mask(self):
        mask = np.ones(hp.nside2npix(self.nside), dtype=float)
        for ninv in self.chain.n_inv_filt.get_ninv():
            assert hp.npix2nside(ninv.numel()) == self.nside
            mask *= dev.to_host((ninv > 0.).to(torch.float64))
        return mask


class cinv_tp(object):
    r"""Joint temperature-polarization inverse-variance filter (filt_cinv.py:341-512).  ninv: [TT, (QQ + UU) / 2] or
    [TT, QQ, QU, UU], each entry a list of maps / paths / scalars to multiply.  The CG runs on D_l-rescaled spectra
    (rescal_cl) exactly as the reference does; the solution is scaled back on return."""

    def __init__(self, lib_dir, lmax, nside, cl, transf, ninv, marge_maps_t=(), marge_monopole=False, marge_dipole=False,
                 pcf='default', rescal_cl='default', chain_descr=None, transf_p=None):
        assert lmax >= 1024 and nside >= 512, (lmax, nside)
        assert len(ninv) == 2 or len(ninv) == 4  # TT, (QQ + UU) / 2 or TT, QQ, QU, UU
        ls = np.arange(lmax + 1, dtype=float)
        dl_w = np.sqrt(ls * (ls + 1.) / 2. / np.pi)
        if rescal_cl == 'default':
            rescal_cl = {a: dl_w.copy() for a in ['t', 'e', 'b']}
        elif rescal_cl is None:
            rescal_cl = {a: np.ones(lmax + 1, dtype=float) for a in ['t', 'e', 'b']}
        elif rescal_cl == 'tonly':
            rescal_cl = {a: np.ones(lmax + 1, dtype=float) for a in ['e', 'b']}
            rescal_cl['t'] = dl_w.copy()
        else:
            assert 0
        for k in rescal_cl.keys():
            rescal_cl[k] /= np.mean(rescal_cl[k])  # keeps the relative TEB weights of the spectra
        dl = {k: rescal_cl[k[0]] * rescal_cl[k[1]] * cl[k][:lmax + 1] for k in cl.keys()}
        if transf_p is None:
            transf_p = transf
        transf_dls = {a: transf_p[:lmax + 1] * utils.cli(rescal_cl[a]) for a in ['e', 'b']}
        transf_dls['t'] = transf[:lmax + 1] * utils.cli(rescal_cl['t'])
        self.lmax = lmax
        self.nside = nside
        self.cl = cl
        self.transf_t = transf
        self.transf_p = transf_p
        self.ninv = ninv
        self.marge_maps_t = marge_maps_t
        self.marge_maps_p = []
        self.lib_dir = lib_dir
        self.rescal_cl = rescal_cl
        if chain_descr is None:
            pcf = os.path.join(lib_dir, "dense_tp.pk") if pcf == 'default' else None
            chain_descr = default_chain_descr('tp', lmax, nside, pcf)
        n_inv_filt = util.jit(opfilt_tp.alm_filter_ninv, ninv, transf_dls['t'], b_transf_e=transf_dls['e'], b_transf_b=transf_dls['b'],
                              marge_maps_t=marge_maps_t, marge_monopole=marge_monopole, marge_dipole=marge_dipole)
        self.chain_descr = chain_descr
        self.chain = util.jit(multigrid.multigrid_chain, opfilt_tp, chain_descr, dl, n_inv_filt)
        if mpi.rank == 0:
            if not os.path.exists(lib_dir):
                os.makedirs(lib_dir)
            if not os.path.exists(os.path.join(lib_dir, "filt_hash.pk")):
                pk.dump(self.hashdict(), open(os.path.join(lib_dir, "filt_hash.pk"), 'wb'), protocol=2)
            if not os.path.exists(os.path.join(lib_dir, "fal.pk")):
                pk.dump(self._calc_fal(), open(os.path.join(lib_dir, "fal.pk"), 'wb'), protocol=2)
            if not os.path.exists(os.path.join(lib_dir, "fmask.fits.gz")):
                hp.write_map(os.path.join(lib_dir, "fmask.fits.gz"), self.calc_mask())
        mpi.barrier()
        fn = os.path.join(lib_dir, "filt_hash.pk")
        utils.hash_check(pk.load(open(fn, 'rb')), self.hashdict(), fn=fn)

    def hashdict(self):
        ret = {'lmax': self.lmax, 'nside': self.nside,
               'rescal_cl': {k: utils.clhash(self.rescal_cl[k]) for k in self.rescal_cl.keys()},
               'cls': {k: utils.clhash(self.cl[k]) for k in self.cl.keys()},
               'transf': utils.clhash(self.transf_t), 'ninv': self._ninv_hash(),
               'marge_maps_t': self.marge_maps_t, 'marge_maps_p': self.marge_maps_p}
        if self.transf_p is not self.transf_t:
            ret['transf_p'] = utils.clhash(self.transf_p)
        return ret

    def get_fal(self, lmax=None):
        fal = pk.load(open(os.path.join(self.lib_dir, "fal.pk"), 'rb'))
        return fal if lmax is None else {k: v[:lmax + 1] for k, v in fal.items()}

    def _calc_fal(self):
        """Isotropic approximation to the filtering matrix (used by the response calculations)."""
        ninv = self.chain.n_inv_filt.n_inv
        assert len(ninv) == 2, 'implement this, easy'
        nlevt, nlevp = _nlev_uKamin(ninv[0]), _nlev_uKamin(ninv[1])
        print("cinv_tp::noiseT_uk_arcmin = %.3f" % nlevt)
        print("cinv_tp::noiseP_uk_arcmin = %.3f" % nlevp)
        fals = np.zeros((self.lmax + 1, 3, 3), dtype=float)
        for i, a in enumerate(['t', 'e', 'b']):
            for j, b in enumerate(['t', 'e', 'b']):
                fals[:, i, j] = self.cl.get(a + b, self.cl.get(b + a, np.zeros(self.lmax + 1)))[:self.lmax + 1]
        fals[1:, 0, 0] += ((nlevt / 180 / 60 * np.pi) / self.transf_t[1:self.lmax + 1]) ** 2
        fals[2:, 1, 1] += ((nlevp / 180 / 60 * np.pi) / self.transf_p[2:self.lmax + 1]) ** 2
        fals[2:, 2, 2] += ((nlevp / 180 / 60 * np.pi) / self.transf_p[2:self.lmax + 1]) ** 2
        fals = np.linalg.pinv(fals)
        fals_dict = {}
        for i, a in enumerate(['t', 'e', 'b']):
            for j, b in enumerate(['t', 'e', 'b'][i:]):
                if np.any(fals[:, i, i + j]):
                    fals_dict[a + b] = fals[:, i, i + j]
        return fals_dict

    def calc_mask(self):
        mask = np.ones(hp.nside2npix(self.nside), dtype=float)
        for ninv in self.chain.n_inv_filt.n_inv:
            assert hp.npix2nside(ninv.numel()) == self.nside
            mask *= dev.to_host((ninv > 0.).to(torch.float64))
        return mask

    def get_fmask(self):
        return hp.read_map(os.path.join(self.lib_dir, "fmask.fits.gz"))

    def apply_ivf(self, tqumap, soltn=None, apply_fini=''):
        assert len(tqumap) == 3
        on_dev = isinstance(tqumap[0], torch.Tensor)
        n = hp.Alm.getsize(self.lmax)
        if soltn is None:
            alms = [torch.zeros(n, dtype=torch.complex128, device=dev.device()) for _ in range(3)]
        else:
            alms = [dev.almxfl(dev.to_dev(a, torch.complex128), self.rescal_cl[f]) for a, f in zip(soltn, 'teb')]
        talm = util_alm.teblm(alms)
        self.chain.solve(talm, [tqumap[0], tqumap[1], tqumap[2]], apply_fini=apply_fini)
        ret = [dev.almxfl(a, self.rescal_cl[f]) for a, f in zip((talm.tlm, talm.elm, talm.blm), 'teb')]
        return tuple(ret) if on_dev else tuple(dev.to_host(a) for a in ret)

    def apply_ivf_batch(self, tqumaps, apply_fini=''):
        """apply_ivf of several (T, Q, U) triplets in ONE block solve (see cinv_t.apply_ivf_batch); returns the list of (tlm, elm, blm)."""
        tqumaps = [list(m) for m in tqumaps]
        assert all(len(m) == 3 for m in tqumaps)
        on_dev = isinstance(tqumaps[0][0], torch.Tensor)
        nb, n = len(tqumaps), hp.Alm.getsize(self.lmax)
        talm = util_alm.teblm([torch.zeros((nb, n), dtype=torch.complex128, device=dev.device()) for _ in range(3)])
        self.chain.solve(talm, tqumaps, apply_fini=apply_fini)
        ret = [dev.almxfl(a, self.rescal_cl[f]) for a, f in zip((talm.tlm, talm.elm, talm.blm), 'teb')]
        if on_dev:
            return [tuple(a[i] for a in ret) for i in range(nb)]
        return [tuple(dev.to_host(a[i]) for a in ret) for i in range(nb)]

    def _ninv_hash(self):
        def h(c):  # arrays by content, nested lists element-wise, paths and scalars as they are
            if isinstance(c, (list, tuple)):
                return [h(x) for x in c]
            return utils.clhash(c) if (isinstance(c, np.ndarray) and c.size > 1) else c
        return [[h(c) for c in self.ninv]]


class library_cinv_sepTP(filt_simple.library_sepTP):
    """Filters a simulation library with separate temperature and polarization CG filters (filt_cinv.py:515-580)."""

    def __init__(self, lib_dir, sim_lib, cinvt, cinvp, cl_weights, soltn_lib=None):
        self.cinv_t = cinvt
        self.cinv_p = cinvp
        super(library_cinv_sepTP, self).__init__(lib_dir, sim_lib, cl_weights, soltn_lib=soltn_lib)
        if mpi.rank == 0:
            fname_mask = os.path.join(self.lib_dir, "fmask.fits.gz")
            if not os.path.exists(fname_mask):
                fmask = self.cinv_t.get_fmask()
                assert np.all(fmask == self.cinv_p.get_fmask())
                hp.write_map(fname_mask, fmask)
        mpi.barrier()
        fn = os.path.join(lib_dir, "filt_hash.pk")
        utils.hash_check(pk.load(open(fn, 'rb')), self.hashdict(), fn=fn)

    def hashdict(self):
        return {'cinv_t': self.cinv_t.hashdict(), 'cinv_p': self.cinv_p.hashdict(), 'sim_lib': self.sim_lib.hashdict()}

    def get_fmask(self):
        return hp.read_map(os.path.join(self.lib_dir, "fmask.fits.gz"))

    def get_tal(self, a, lmax=None):
        assert a.lower() in ['t', 'e', 'b'], a
        return self.cinv_t.get_tal(a, lmax=lmax) if a.lower() == 't' else self.cinv_p.get_tal(a, lmax=lmax)

    def get_ftl(self, lmax=None):
        return self.cinv_t.get_ftl(lmax=lmax)

    def get_fel(self, lmax=None):
        return self.cinv_p.get_fel(lmax=lmax)

    def get_fbl(self, lmax=None):
        return self.cinv_p.get_fbl(lmax=lmax)

    def _apply_ivf_t(self, tmap, soltn=None):
        return self.cinv_t.apply_ivf(tmap, soltn=soltn)

    def _apply_ivf_p(self, pmap, soltn=None):
        with _p_context():  # (the context the polarization solves of filter_sims run in: one set of workspaces and graphs)
            return self.cinv_p.apply_ivf(pmap, soltn=soltn)

    def supports_block(self, a):
        """True when the 't' / 'p' filter takes block vectors (several simulations per solve): the one-call operators only -- Q / U
        templates, more than dev.TEMPLATE_MAX_MODES temperature modes, an EB spectrum or replaced transforms rule it out"""
        f = self.cinv_t if a == 't' else self.cinv_p
        ok = util.unjit(f.chain.n_inv_filt).supports_block()
        if a == 'p':
            ok = ok and 'eb' not in f.cl
        return bool(ok)

    def filter_sims(self, idxs, fields='tp', batch=None, concurrent=None):
        """Filters (and caches) the simulations `idxs` that are not cached yet, `batch` at a time in block solves of the CG
        (cinv_t / cinv_p.apply_ivf_batch: every launch of a solve carries the whole block) -- what the driver's filtering phase
        calls instead of looping over get_sim_tlm / get_sim_elm one simulation at a time (run_qlms.py:57-62).  Same cache files, same
        alms.  batch: block size (default options.opts.cg_batch = 4; 1 = one solve per simulation); a filter that cannot take block
        vectors (supports_block) is served one simulation at a time.  concurrent (default options.opts.tp_concurrent): when both 't'
        and 'p' are asked for, the temperature and polarization solves of a block run at the same time on two streams of this
        process (apply_ivf_tp) instead of one after the other.  With cache=False the results only live in the device cache of
        _dev_slots simulations, which grows to the block size; more indices than that per call are solved again when asked for.
        Returns True (the wrappers of filt_util forward the call and report whether the library underneath took it)."""
        if batch is None:
            batch = int(options.opts.cg_batch)
        batch = max(1, batch)
        if concurrent is None:
            concurrent = bool(options.opts.tp_concurrent)
        for a in fields:
            assert a in 'tp', a
        fields = [a for a in 'tp' if a in fields]
        names = {'t': ['t'], 'p': ['e', 'b']}

        def missing(a, i):
            return not (self.cache and all(os.path.exists(self._fn(n, i)) for n in names[a])) \
                and not all(n in self._dev_cache.get(i, {}) for n in names[a])
        todo = {a: [i for i in idxs if missing(a, i)] for a in fields}
        bsz = {a: (batch if (self.supports_block(a) and self.soltn_lib is None) else 1) for a in fields}  # (starting points come one by one)
        if not self.cache:
            n_all = len(set(i for a in fields for i in todo[a]))
            if n_all > self._dev_slots:
                self._dev_slots = min(n_all, max(self._dev_slots, batch))  # (instance attribute: this library only)
            if n_all > self._dev_slots:
                print('filter_sims: cache=False and %d simulations asked for: only the last %d stay resident' % (n_all, self._dev_slots))

        def data(a, i):
            if a == 't':
                return dev.to_dev(self.sim_lib.get_sim_tmap(i), torch.float64)
            return [dev.to_dev(m, torch.float64) for m in self.sim_lib.get_sim_pmap(i)]

        def store(a, blk, outs):
            for i, out in zip(blk, outs):
                ent = self._dev_entry(i)
                for n, x in zip(names[a], out):
                    ent[n] = dev.to_dev(x).clone()  # (rows of the block solution: own storage, so that the block can be released)
                    if self.cache:
                        hp.write_alm(self._fn(n, i), dev.to_host(ent[n]), overwrite=True)

        def solve(a, blk):
            f = self.cinv_t if a == 't' else self.cinv_p
            if len(blk) == 1:
                soltn = None if self.soltn_lib is None else (self._soltn_t(blk[0]) if a == 't' else self._soltn_p(blk[0]))
                out = f.apply_ivf(data(a, blk[0]), soltn=soltn)
                return [(out,)] if a == 't' else [tuple(out)]
            outs = f.apply_ivf_batch([data(a, i) for i in blk])
            return [(x,) for x in outs] if a == 't' else [tuple(x) for x in outs]

        if concurrent and len(fields) == 2:
            # blocks of simulations that need both filters: T block and P block at the same time
            both = [i for i in todo['t'] if i in set(todo['p'])]
            step = min(bsz['t'], bsz['p'])
            for k in range(0, len(both), step):
                blk = both[k:k + step]
                if len(blk) == 1:
                    st = None if self.soltn_lib is None else self._soltn_t(blk[0])
                    sp = None if self.soltn_lib is None else self._soltn_p(blk[0])
                    rt, rp = apply_ivf_tp(self.cinv_t, data('t', blk[0]), self.cinv_p, data('p', blk[0]), soltn_t=st, soltn_p=sp)
                    rt, rp = [(rt,)], [tuple(rp)]
                else:
                    rt, rp = apply_ivf_tp(self.cinv_t, [data('t', i) for i in blk], self.cinv_p, [data('p', i) for i in blk])
                    rt, rp = [(x,) for x in rt], [tuple(x) for x in rp]
                store('t', blk, rt)
                store('p', blk, rp)
            done = set(both)
            todo = {a: [i for i in todo[a] if i not in done] for a in fields}
        for a in fields:
            for k in range(0, len(todo[a]), bsz[a]):
                blk = todo[a][k:k + bsz[a]]
                if a == 'p':  # (polarization solves always in their own plan context, see run_tp)
                    with _p_context():
                        outs = solve(a, blk)
                else:
                    outs = solve(a, blk)
                store(a, blk, outs)
        return True

    def get_tmliklm(self, idx):
        return hp.almxfl(self.get_sim_tlm(idx), self.cinv_t.cl['tt'])

    def get_emliklm(self, idx):
        return hp.almxfl(self.get_sim_elm(idx), self.cinv_p.cl['ee'])

    def get_bmliklm(self, idx):
        return hp.almxfl(self.get_sim_blm(idx), self.cinv_p.cl['bb'])


class library_cinv_jTP(filt_simple.library_jTP):
    """Filters a simulation library with the joint temperature-polarization CG filter (filt_cinv.py:582-620)."""

    def __init__(self, lib_dir, sim_lib, cinv_jtp, cl_weights, soltn_lib=None):
        self.cinv_tp = cinv_jtp
        super(library_cinv_jTP, self).__init__(lib_dir, sim_lib, cl_weights, soltn_lib=soltn_lib)
        if mpi.rank == 0:
            fname_mask = os.path.join(self.lib_dir, "fmask.fits.gz")
            if not os.path.exists(fname_mask):
                hp.write_map(fname_mask, self.cinv_tp.get_fmask())
        mpi.barrier()
        fn = os.path.join(lib_dir, "filt_hash.pk")
        utils.hash_check(pk.load(open(fn, 'rb')), self.hashdict(), fn=fn)

    def hashdict(self):
        return {'cinv_tp': self.cinv_tp.hashdict(), 'clw': {k: utils.clhash(self.cl[k]) for k in self.cl.keys()},
                'sim_lib': self.sim_lib.hashdict()}

    def get_fmask(self):
        return hp.read_map(os.path.join(self.lib_dir, "fmask.fits.gz"))

    def get_fal(self, lmax=None):
        return self.cinv_tp.get_fal(lmax=lmax)

    def _apply_ivf(self, tqumap, soltn=None):
        return self.cinv_tp.apply_ivf(tqumap, soltn=soltn)
