"""End-to-end on the GPU: the run_qlms driver over the idealized parameter file at a small size (filter -> QE ->
mean field -> spectra with the reference's cache layout), then a sanity check of the spectrum against the expected
Gaussian reconstruction noise level order of magnitude."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_run_qlms_idealized_small(tmp_path):
    env = dict(os.environ, PLENS=str(tmp_path), PLENS_NSIDE='64', PLENS_LMAX='128', PLENS_NSIMS='10')
    cmd = [sys.executable, os.path.join(ROOT, 'examples', 'run_qlms.py'), os.path.join(ROOT, 'params', 'idealized_example.py'),
           '-imin', '0', '-imax', '5', '-k', 'p', 'ptt', '-kA', 'p', '-kB', 'p', '-ivt', '-ivp', '-dd', '-ds', '-ss', '-mfdd']
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert out.returncode == 0, out.stdout.decode()[-3000:]
    temp = os.path.join(str(tmp_path), 'temp', 'idealized_example')
    for f in ['ivfs/sim_0000_tlm.fits', 'ivfs/sim_0003_elm.fits', 'ivfs/dat_tlm.fits', 'qlms_dd/sim_p_0004.fits', 'qlms_dd/sim_x_0004.fits',
              'qlms_ds/sim_ptt_0001.fits', 'qlms_ss/sim_p_0002.fits', 'qcls_dd/cldb.db', 'qlms_dd/qe_sim_hash.pk']:
        assert os.path.exists(os.path.join(temp, f)), f
    # second invocation: everything is served from the cache, nothing is recomputed (restart-by-cache)
    t_before = os.path.getmtime(os.path.join(temp, 'qlms_dd', 'sim_p_0004.fits'))
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert out.returncode == 0
    assert os.path.getmtime(os.path.join(temp, 'qlms_dd', 'sim_p_0004.fits')) == t_before
    sys.path.insert(0, ROOT)
    from plancklens_amd import hp
    qlm = hp.read_alm(os.path.join(temp, 'qlms_dd', 'sim_p_0004.fits'))
    assert hp.Alm.getlmax(qlm.size) == 256 and np.all(np.isfinite(qlm.real)) and np.abs(qlm).max() > 0


def test_two_ranks_share_the_mean_field(tmp_path):
    """The product's distributed path on the GPU box: two ranks (gloo rendezvous, both on GPU 0, collectives staged through the
    host -- RCCL needs one GPU per rank) run the driver with -mfdd.  qest.library.get_sim_qlm_mf shards the simulations of
    each mean field over the ranks and all-reduces the device-resident sums; the cache files must equal those of a
    single-process run to rounding, and every simulation must have been reconstructed by exactly one rank."""
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    args = [os.path.join(ROOT, 'examples', 'run_qlms.py'), os.path.join(ROOT, 'params', 'idealized_example.py'),
            '-imin', '0', '-imax', '5', '-k', 'p', '-kA', 'p', '-kB', 'p', '-ivt', '-ivp', '-dd', '-mfdd']
    base = dict(os.environ, PLENS_NSIDE='32', PLENS_LMAX='64', PLENS_NSIMS='20')  # mean-field simulations 0..3, halves [0, 2] and [1, 3]
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        base.pop(k, None)
    one = subprocess.run([sys.executable] + args, env=dict(base, PLENS=str(tmp_path / 'one')), cwd=ROOT, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, timeout=900)
    assert one.returncode == 0, one.stdout.decode()[-3000:]
    procs = [subprocess.Popen([sys.executable] + args, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              env=dict(base, PLENS=str(tmp_path / 'two'), RANK=str(r), WORLD_SIZE='2', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1',
                                       MASTER_PORT=str(port), PLENS_DIST_BACKEND='gloo')) for r in range(2)]
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    # jobs[rank::size]: each rank reconstructed its own simulations only
    for r, o in enumerate(outs):
        line = [l for l in o.splitlines() if 'doing QE sims' in l][0]
        mine = [int(x) for x in line.split('QE sims [')[1].split(']')[0].split(',')]
        assert mine == list(range(0, 6))[r::2], (r, mine)
    sys.path.insert(0, ROOT)
    from plancklens_amd import hp
    d1 = os.path.join(str(tmp_path), 'one', 'temp', 'idealized_example', 'qlms_dd')
    d2 = os.path.join(str(tmp_path), 'two', 'temp', 'idealized_example', 'qlms_dd')
    mfs = sorted(f for f in os.listdir(d1) if f.startswith('simMF_'))
    assert len(mfs) == 2 and mfs == sorted(f for f in os.listdir(d2) if f.startswith('simMF_'))
    for f in mfs + ['sim_p_0003.fits']:
        a, b = hp.read_alm(os.path.join(d1, f)), hp.read_alm(os.path.join(d2, f))
        assert np.abs(a).max() > 0 and np.abs(a - b).max() < 1e-12 * np.abs(a).max(), f


def test_device_side_simulation_libraries(tmp_path):
    """phases, correlated sky alms, sky maps and noise generated on the GPU (no host arrays on the way):
    reproducible per (seed, field, index), unit variance, right spectra including the TE correlation."""
    import numpy as np
    import torch
    from plancklens_amd import dev, hp, utils
    from plancklens_amd.sims import cmbs, maps, phas
    lmax, nside = 256, 128
    cls = utils.camb_clfile(os.path.join(ROOT, 'plancklens_amd', 'data', 'cls', 'FFP10_wdipole_lensedCls.dat'), lmax=lmax)
    pha = phas.lib_phas_dev(str(tmp_path / 'pha'), 3, lmax, seed=11)
    a, b = pha.get_sim(3, idf=1), pha.get_sim(3, idf=1)
    assert a.is_cuda and a.dtype == torch.complex128 and bool((a == b).all())            # pure function of (seed, idf, idx)
    assert not bool((a == pha.get_sim(4, idf=1)).all()) and not bool((a == pha.get_sim(3, idf=2)).all())
    assert float(a[:lmax + 1].imag.abs().max()) == 0.                                      # real m = 0 column
    cl_unit = dev.to_host(dev.alm2cl(a))
    assert abs(np.mean(cl_unit[20:]) - 1.) < 0.03                                          # unit variance per mode
    sky = cmbs.sims_cmb_unl({k: cls[k] for k in ['tt', 'ee', 'bb', 'te']}, pha)
    acc = {k: 0. for k in ['tt', 'ee', 'te']}
    nsim = 6
    for i in range(nsim):
        t, e = sky.get_sim_tlm(i), sky.get_sim_elm(i)
        assert t.is_cuda
        acc['tt'] += dev.to_host(dev.alm2cl(t)); acc['ee'] += dev.to_host(dev.alm2cl(e)); acc['te'] += dev.to_host(dev.alm2cl(t, e))
    for k in acc:
        band = slice(50, 250)
        assert abs(np.sum(acc[k][band] / nsim) / np.sum(cls[k][band]) - 1.) < 0.05, k
    noise = phas.pix_lib_phas_dev(str(tmp_path / 'pix'), 3, (hp.nside2npix(nside),), seed=5)
    sims = maps.cmb_maps_nlev(sky, hp.gauss_beam(30. / 60 / 180 * np.pi, lmax=lmax), 20., 30., nside, pix_lib_phas=noise, device_maps=True)
    tmap = sims.get_sim_tmap(0)
    q, u = sims.get_sim_pmap(0)
    assert tmap.is_cuda and q.is_cuda and u.is_cuda and tmap.numel() == hp.nside2npix(nside)
    assert bool((tmap == sims.get_sim_tmap(0)).all())
    vamin = np.sqrt(hp.nside2pixarea(nside, degrees=True)) * 60
    n = noise.get_sim(0, idf=1)
    assert abs(float(n.std()) - 1.) < 0.01 and abs(float(n.mean())) < 0.01
    assert float((q - 30. / vamin * n).std()) < float(q.std())                            # the Q map contains that noise realisation


def test_run_qlms_with_device_side_sims(tmp_path):
    """the same driver with PLENS_DEVICE_SIMS=1: inputs generated on the GPU end to end"""
    env = dict(os.environ, PLENS=str(tmp_path), PLENS_NSIDE='64', PLENS_LMAX='128', PLENS_NSIMS='10', PLENS_DEVICE_SIMS='1')
    cmd = [sys.executable, os.path.join(ROOT, 'examples', 'run_qlms.py'), os.path.join(ROOT, 'params', 'idealized_example.py'),
           '-imin', '0', '-imax', '2', '-k', 'p', '-ivt', '-ivp', '-dd']
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert out.returncode == 0, out.stdout.decode()[-3000:]
    temp = os.path.join(str(tmp_path), 'temp', 'idealized_example')
    sys.path.insert(0, ROOT)
    from plancklens_amd import hp
    qlm = hp.read_alm(os.path.join(temp, 'qlms_dd', 'sim_p_0001.fits'))
    assert np.all(np.isfinite(qlm.real)) and np.abs(qlm).max() > 0


def test_run_qlms_masked_sky_cg_filters(tmp_path):
    """The driver over params/anisofilt_example.py (the structure of the reference's params/anisofilt_example.py: masked sky,
    cinv_t + cinv_p with the default multigrid chains and dense coarse preconditioners, library_ftl on top) at nside 512, lmax 1024:
    the filtering phase runs in block solves (library_cinv_sepTP.filter_sims through the library_ftl wrapper), the estimators are
    built from the cached filtered alms, and a one-by-one filtering of the same simulation gives the same alm."""
    env = dict(os.environ, PLENS=str(tmp_path), PLENS_NSIDE='512', PLENS_LMAX='1024', PLENS_NSIMS='10', PLENS_OPTIONS='cg_batch=3')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'examples', 'run_qlms.py'), os.path.join(ROOT, 'params', 'anisofilt_example.py'),
           '-imin', '0', '-imax', '2', '-k', 'p', '-ivt', '-ivp', '-dd']
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    log = out.stdout.decode()
    assert out.returncode == 0, log[-3000:]
    assert 'filtered sims [0, 1, 2] (t and p) in overlapped block solves' in log, log[-3000:]  # both fields of a simulation on one rank
    temp = os.path.join(str(tmp_path), 'temp', 'anisofilt_example')
    for f in ['ivfs/sim_0000_tlm.fits', 'ivfs/sim_0002_elm.fits', 'ivfs/sim_0001_blm.fits', 'cinv_t/dense.pk', 'cinv_p/dense.pk', 'cinv_t/ftl.dat',
              'qlms_dd/sim_p_0002.fits', 'qlms_dd/sim_x_0000.fits']:
        assert os.path.exists(os.path.join(temp, f)), f
    # the same simulation filtered by itself (fresh libraries over the same caches of masks and dense preconditioners)
    sys.path.insert(0, ROOT)
    os.environ.update({k: env[k] for k in ('PLENS', 'PLENS_NSIDE', 'PLENS_LMAX', 'PLENS_NSIMS')})
    try:
        from importlib.machinery import SourceFileLoader
        from plancklens_amd import dev, hp
        par = SourceFileLoader('aniso_par', os.path.join(ROOT, 'params', 'anisofilt_example.py')).load_module()
        blk = hp.read_alm(os.path.join(temp, 'ivfs', 'sim_0001_tlm.fits'))
        one = par.cinv_t.apply_ivf(par.sims.get_sim_tmap(1))
        assert np.abs(blk).max() > 0 and np.abs(np.asarray(one) - blk).max() < 1e-10 * np.abs(blk).max()
        eblk = hp.read_alm(os.path.join(temp, 'ivfs', 'sim_0002_elm.fits'))
        eone, _ = par.cinv_p.apply_ivf(par.sims.get_sim_pmap(2))
        assert np.abs(eblk).max() > 0 and np.abs(np.asarray(eone) - eblk).max() < 1e-10 * np.abs(eblk).max()
        qlm = hp.read_alm(os.path.join(temp, 'qlms_dd', 'sim_p_0001.fits'))
        assert hp.Alm.getlmax(qlm.size) == 2048 and np.all(np.isfinite(qlm.real)) and np.abs(qlm).max() > 0
    finally:
        for k in ('PLENS', 'PLENS_NSIDE', 'PLENS_LMAX', 'PLENS_NSIMS'):
            os.environ.pop(k, None)


def test_device_generator_matches_its_restatement(tmp_path):
    """pl_map_add_normal / pl_alm_unit_phases against oracle/philox_oracle.py (Philox4x32-10 pinned by the published known-answer vectors,
    tests/test_sims.py): same deviates to rounding (the integer part is exact; log / sincospi differ in the last bits), in place, out of
    place and with an odd length; the phase libraries built on them are pure functions of (seed, field, index)."""
    import ctypes
    import numpy as np
    import torch
    from oracle import philox_oracle as po
    from plancklens_amd import _lib, dev, hp
    from plancklens_amd.sims import phas
    L = _lib.lib()
    key = 0x9E3779B97F4A7C15
    for n in (2, 1001, 12 * 64 ** 2):
        base = torch.linspace(-1., 1., n, dtype=torch.float64, device='cuda')
        out = torch.empty_like(base)
        _lib.check(L.pl_map_add_normal(n, base.data_ptr(), out.data_ptr(), 2.5, ctypes.c_uint64(key), dev.stream_ptr()))
        ref = base.cpu().numpy() + 2.5 * po.normals(key, n)
        assert np.max(np.abs(out.cpu().numpy() - ref)) < 1e-13
        inpl = base.clone()
        _lib.check(L.pl_map_add_normal(n, inpl.data_ptr(), inpl.data_ptr(), 2.5, ctypes.c_uint64(key), dev.stream_ptr()))
        assert torch.equal(inpl, out)
        _lib.check(L.pl_map_add_normal(n, None, out.data_ptr(), 1.0, ctypes.c_uint64(key), dev.stream_ptr()))
        assert np.max(np.abs(out.cpu().numpy() - po.normals(key, n))) < 1e-13
    lmax = 37
    alm = torch.empty(hp.Alm.getsize(lmax), dtype=torch.complex128, device='cuda')
    _lib.check(L.pl_alm_unit_phases(lmax, alm.data_ptr(), ctypes.c_uint64(key), dev.stream_ptr()))
    assert np.max(np.abs(alm.cpu().numpy() - po.unit_phases(key, lmax))) < 1e-13
    lib = phas.pix_lib_phas_dev(str(tmp_path / 'pix'), 3, (12 * 16 ** 2,), seed=5)
    k = phas._dev_key(5, 2, 7)
    assert np.max(np.abs(lib.get_sim(7, idf=2).cpu().numpy() - po.normals(k, 12 * 16 ** 2))) < 1e-13
    m = torch.ones(12 * 16 ** 2, dtype=torch.float64, device='cuda')
    o = torch.empty_like(m)
    lib.add_scaled(m, 7, 2, 0.5, out=o)
    assert torch.equal(o, lib.add_scaled(m.clone(), 7, 2, 0.5)) and bool((m == 1.).all())
    assert phas._dev_key(5, 2, 7) != phas._dev_key(5, 1, 7) != phas._dev_key(5, 1, 8)
