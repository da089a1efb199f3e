"""Parameter file with the structure of the reference's params/anisofilt_example.py (:41-166): masked sky, separate temperature and
polarization conjugate-gradient filters (filt_cinv.cinv_t / cinv_p with the default multigrid chains, monopole and dipole
marginalised), an a-posteriori l-cut of the filtered alms (filt_util.library_ftl), the three QE libraries (dd / ds / ss) and
their spectra libraries -- on synthetic inputs.

What differs from the reference file, and why (hard part 6): the FFP10 simulations and the Planck lensing
mask live on NERSC, and hp.pixwin needs a data file packaged inside healpy; here the skies are seeded Gaussian realisations of the
fiducial lensed spectra, the transfer function is the 5' beam alone and the mask is synthetic (a |b| < 20 deg band plus 200
one-degree discs, fsky ~ 0.65: the mask of BASELINE config 4, tools/cg_bench.py).  The N1 library (n1f Fortran) is not provided.
The driver (examples/run_qlms.py -ivt -ivp) filters each rank's simulations several at a time in block solves of the CG
(library_cinv_sepTP.filter_sims, options.opts.cg_batch right-hand sides per solve, default 4; PLENS_OPTIONS=cg_batch=N).
Sizes can be reduced through the environment for quick runs: PLENS_NSIDE (>= 512), PLENS_LMAX (>= 1024), PLENS_NSIMS.
"""
import os

import numpy as np

import plancklens_amd
from plancklens_amd import hp, nhl, qecl, qest, qresp, utils
from plancklens_amd.filt import filt_cinv, filt_util
from plancklens_amd.sims import cmbs, maps, phas, utils as maps_utils

assert 'PLENS' in os.environ.keys(), 'Set env. variable PLENS to a writeable folder'
TEMP = os.path.join(os.environ['PLENS'], 'temp', 'anisofilt_example')
cls_path = os.path.join(os.path.dirname(os.path.abspath(plancklens_amd.__file__)), 'data', 'cls')

nside = int(os.environ.get('PLENS_NSIDE', 2048))
lmax_ivf = int(os.environ.get('PLENS_LMAX', 2048))
lmin_ivf = 100
lmax_qlm = min(4096, 2 * lmax_ivf)
nlev_t = 35.
nlev_p = 55.
nsims = int(os.environ.get('PLENS_NSIMS', 300))

transf = hp.gauss_beam(5. / 60. / 180. * np.pi, lmax=lmax_ivf)
cl_len = utils.camb_clfile(os.path.join(cls_path, 'FFP10_wdipole_lensedCls.dat'))
cl_weight = utils.camb_clfile(os.path.join(cls_path, 'FFP10_wdipole_lensedCls.dat'))
cl_weight['bb'] *= 0.

DEVICE_SIMS = os.environ.get('PLENS_DEVICE_SIMS', '0') == '1'
_pix, _sky = (phas.pix_lib_phas_dev, phas.lib_phas_dev) if DEVICE_SIMS else (phas.pix_lib_phas_seeded, phas.lib_phas_seeded)
pix_phas = _pix(os.path.join(TEMP, 'pix_phas_nside%s%s' % (nside, '_dev' * DEVICE_SIMS)), 3, (hp.nside2npix(nside),), seed=11)
sky_phas = _sky(os.path.join(TEMP, 'sky_phas_lmax%s%s' % (lmax_ivf, '_dev' * DEVICE_SIMS)), 3, lmax_ivf, seed=12)
skies = cmbs.sims_cmb_unl({k: cl_len[k] for k in ['tt', 'ee', 'bb', 'te']}, sky_phas)
sims = maps_utils.sim_lib_shuffle(maps.cmb_maps_nlev(skies, transf, nlev_t, nlev_p, nside, pix_lib_phas=pix_phas, device_maps=DEVICE_SIMS),
                                  {idx: nsims if idx == -1 else idx for idx in range(-1, nsims)})


def _mask():
    """|b| < 20 deg removed plus 200 seeded discs of one degree radius (made once, cached as a map file like the reference's mask path)"""
    x, y, z = hp.pix2vec(nside)
    mask = (np.abs(z) > np.sin(np.radians(20.))).astype(float)
    cen = np.random.default_rng(7).standard_normal((200, 3))
    cen /= np.linalg.norm(cen, axis=1)[:, None]
    vec = np.stack([x, y, z])
    for c in cen:
        mask[(c @ vec) > np.cos(np.radians(1.))] = 0.
    return mask


maskpath = os.path.join(TEMP, 'mask_nside%s.fits' % nside)
if not os.path.exists(maskpath):
    from plancklens_amd.helpers import mpi
    if mpi.rank == 0:
        os.makedirs(TEMP, exist_ok=True)
        hp.write_map(maskpath, _mask())
    mpi.barrier()
maskpaths = [maskpath]

libdir_cinvt = os.path.join(TEMP, 'cinv_t')
libdir_cinvp = os.path.join(TEMP, 'cinv_p')
libdir_ivfs = os.path.join(TEMP, 'ivfs')

# homogeneous noise outside the mask; the scalar is 1 / (noise variance per pixel): pixel area in arcmin^2 / nlev^2
pixarea = hp.nside2pixarea(nside, degrees=True) * 3600.
ninv_t = [np.array([pixarea / nlev_t ** 2])] + maskpaths
cinv_t = filt_cinv.cinv_t(libdir_cinvt, lmax_ivf, nside, cl_len, transf, ninv_t, marge_monopole=True, marge_dipole=True, marge_maps=[])

ninv_p = [[np.array([pixarea / nlev_p ** 2])] + maskpaths]
cinv_p = filt_cinv.cinv_p(libdir_cinvp, lmax_ivf, nside, cl_len, transf, ninv_p)

ivfs_raw = filt_cinv.library_cinv_sepTP(libdir_ivfs, sims, cinv_t, cinv_p, cl_len)
ftl = np.ones(lmax_ivf + 1, dtype=float) * (np.arange(lmax_ivf + 1) >= lmin_ivf)
fel = np.ones(lmax_ivf + 1, dtype=float) * (np.arange(lmax_ivf + 1) >= lmin_ivf)
fbl = np.ones(lmax_ivf + 1, dtype=float) * (np.arange(lmax_ivf + 1) >= lmin_ivf)
ivfs = filt_util.library_ftl(ivfs_raw, lmax_ivf, ftl, fel, fbl)

nblk = max(1, min(60, nsims // 5))
ss_dict = {k: v for k, v in zip(np.concatenate([range(i * nblk, (i + 1) * nblk) for i in range(0, nsims // nblk)]),
                                np.concatenate([np.roll(range(i * nblk, (i + 1) * nblk), -1) for i in range(0, nsims // nblk)]))}
ds_dict = {k: -1 for k in range(nsims)}
ivfs_d = filt_util.library_shuffle(ivfs, ds_dict)
ivfs_s = filt_util.library_shuffle(ivfs, ss_dict)

qlms_dd = qest.library_sepTP(os.path.join(TEMP, 'qlms_dd'), ivfs, ivfs, cl_len['te'], nside, lmax_qlm=lmax_qlm)
qlms_ds = qest.library_sepTP(os.path.join(TEMP, 'qlms_ds'), ivfs, ivfs_d, cl_len['te'], nside, lmax_qlm=lmax_qlm)
qlms_ss = qest.library_sepTP(os.path.join(TEMP, 'qlms_ss'), ivfs, ivfs_s, cl_len['te'], nside, lmax_qlm=lmax_qlm)

mc_sims_bias = np.arange(min(60, nsims // 5))
mc_sims_var = np.arange(min(60, nsims // 5), nsims)
qcls_dd = qecl.library(os.path.join(TEMP, 'qcls_dd'), qlms_dd, qlms_dd, mc_sims_bias)
qcls_ds = qecl.library(os.path.join(TEMP, 'qcls_ds'), qlms_ds, qlms_ds, np.array([]))
qcls_ss = qecl.library(os.path.join(TEMP, 'qcls_ss'), qlms_ss, qlms_ss, np.array([]))

nhl_dd = nhl.nhl_lib_simple(os.path.join(TEMP, 'nhl_dd'), ivfs, cl_weight, lmax_qlm)
qresp_dd = qresp.resp_lib_simple(os.path.join(TEMP, 'qresp'), lmax_ivf, cl_weight, cl_len,
                                 {'t': ivfs.get_ftl(), 'e': ivfs.get_fel(), 'b': ivfs.get_fbl()}, lmax_qlm)
