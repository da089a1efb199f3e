"""Parameter file with the structure of the reference's params/idealized_example.py (:36-131): full-sky isotropic
filtering, the three QE libraries (dd / ds / ss), their spectra libraries -- on synthetic inputs.

What differs from the reference file, and why (hard part 6): the FFP10 simulations live on NERSC
($CFS) and hp.pixwin needs a data file packaged inside healpy, neither of which exists here; the skies are seeded
Gaussian realisations of the fiducial lensed spectra (sims.cmbs.sims_cmb_unl) and the transfer function is the 5'
beam alone.  The analytic response and N0 libraries (qresp, nhl) are instantiated as in the reference; the N1 library
(n1, Fortran kernels outside the hot path) is not.
Sizes can be reduced through the environment for quick runs: PLENS_NSIDE, PLENS_LMAX, PLENS_NSIMS.
"""
import os

import numpy as np

import plancklens_amd
from plancklens_amd import hp, nhl, qecl, qest, qresp, utils
from plancklens_amd.filt import filt_simple, filt_util
from plancklens_amd.sims import cmbs, maps, phas, utils as maps_utils

assert 'PLENS' in os.environ.keys(), 'Set env. variable PLENS to a writeable folder'
TEMP = os.path.join(os.environ['PLENS'], 'temp', 'idealized_example')
cls_path = os.path.join(os.path.dirname(os.path.abspath(plancklens_amd.__file__)), 'data', 'cls')

nside = int(os.environ.get('PLENS_NSIDE', 2048))
lmax_ivf = int(os.environ.get('PLENS_LMAX', 2048))
lmin_ivf = min(100, lmax_ivf // 8)
lmax_qlm = min(4096, 2 * lmax_ivf)
nlev_t = 35.
nlev_p = 55.
nsims = int(os.environ.get('PLENS_NSIMS', 300))

transf = hp.gauss_beam(5. / 60. / 180. * np.pi, lmax=lmax_ivf)
cl_len = utils.camb_clfile(os.path.join(cls_path, 'FFP10_wdipole_lensedCls.dat'))
cl_weight = utils.camb_clfile(os.path.join(cls_path, 'FFP10_wdipole_lensedCls.dat'))
cl_weight['bb'] *= 0.

# PLENS_DEVICE_SIMS=1: phases, sky alms, maps and noise are generated on the GPU (torch Philox streams instead of numpy's:
# other realisations, same statistics, still a pure function of (seed, field, index)) and never pass through host memory
DEVICE_SIMS = os.environ.get('PLENS_DEVICE_SIMS', '0') == '1'
# (host phases: the counter-seeded libraries -- every rank of a sharded run draws the same simulation for the same index without a
# shared generator-state database; phas.pix_lib_phas / phas.lib_phas are the reference's state-recording ones)
_pix, _sky = (phas.pix_lib_phas_dev, phas.lib_phas_dev) if DEVICE_SIMS else (phas.pix_lib_phas_seeded, phas.lib_phas_seeded)
pix_phas = _pix(os.path.join(TEMP, 'pix_phas_nside%s%s' % (nside, '_dev' * DEVICE_SIMS)), 3, (hp.nside2npix(nside),), seed=11)
sky_phas = _sky(os.path.join(TEMP, 'sky_phas_lmax%s%s' % (lmax_ivf, '_dev' * DEVICE_SIMS)), 3, lmax_ivf, seed=12)
skies = cmbs.sims_cmb_unl({k: cl_len[k] for k in ['tt', 'ee', 'bb', 'te']}, sky_phas)
sims = maps_utils.sim_lib_shuffle(maps.cmb_maps_nlev(skies, transf, nlev_t, nlev_p, nside, pix_lib_phas=pix_phas, device_maps=DEVICE_SIMS),
                                  {idx: nsims if idx == -1 else idx for idx in range(-1, nsims)})

ftl = utils.cli(cl_len['tt'][:lmax_ivf + 1] + (nlev_t / 60. / 180. * np.pi / transf) ** 2)
fel = utils.cli(cl_len['ee'][:lmax_ivf + 1] + (nlev_p / 60. / 180. * np.pi / transf) ** 2)
fbl = utils.cli(cl_len['bb'][:lmax_ivf + 1] + (nlev_p / 60. / 180. * np.pi / transf) ** 2)
ftl[:lmin_ivf] *= 0.
fel[:lmin_ivf] *= 0.
fbl[:lmin_ivf] *= 0.

ivfs = filt_simple.library_fullsky_sepTP(os.path.join(TEMP, 'ivfs'), sims, nside, transf, cl_len, ftl, fel, fbl, cache=True)

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

#---- semi-analytical Gaussian lensing bias library (idealized_example.py:123):
nhl_dd = nhl.nhl_lib_simple(os.path.join(TEMP, 'nhl_dd'), ivfs, cl_weight, lmax_qlm)

#---- QE response calculation library (idealized_example.py:130-131); the N1 library (n1f Fortran) is not provided:
qresp_dd = qresp.resp_lib_simple(os.path.join(TEMP, 'qresp'), lmax_ivf, cl_weight, cl_len,
                                 {'t': ivfs.get_ftl(), 'e': ivfs.get_fel(), 'b': ivfs.get_fbl()}, lmax_qlm)
