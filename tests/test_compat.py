"""Import contract of the reference's parameter files (second table): with plancklens_amd.compat.install()
the reference's own params/idealized_example.py -- loaded from /root/reference, never copied -- imports `plancklens`,
`plancklens.filt`, `plancklens.n1`, `plancklens.sims.planck2018_sims`, `healpy` and instantiates the libraries of this
package.  Build-container only (the GPU box has no /root/reference).  Runs in a child process: install() edits sys.modules."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PARFILE = '/root/reference/params/idealized_example.py'

CHILD = r'''
import os, sys, traceback
sys.path.insert(0, %(root)r)
import numpy as np
import plancklens_amd.compat
names = plancklens_amd.compat.install()
assert 'plancklens' in names and 'healpy' in names and 'plancklens.n1.n1' in names
from importlib.machinery import SourceFileLoader
mode = sys.argv[1]
if mode == 'nodata':
    try:
        SourceFileLoader('par_ref', %(par)r).load_module()
    except NotImplementedError as e:
        tb = traceback.extract_tb(sys.exc_info()[2])
        where = [f for f in tb if f.filename == %(par)r][-1]
        print('STOPPED line %%d: %%s | %%s' %% (where.lineno, where.line, e))
        sys.exit(0)
    print('LOADED')
else:
    # healpy's window-function table is data: a stand-in table (all ones) in the documented file format
    from plancklens_amd import fitsio
    d = os.environ['PLENS_HEALPIX_DATA']
    fitsio.write_bintable(os.path.join(d, 'pixel_window_n2048.fits'), [('TEMPERATURE', np.ones(4 * 2048 + 1)), ('POLARIZATION', np.ones(4 * 2048 + 1))])
    par = SourceFileLoader('par_ref', %(par)r).load_module()
    import plancklens_amd as pa
    from plancklens_amd import qest, qecl, nhl, qresp
    from plancklens_amd.filt import filt_simple, filt_util
    from plancklens_amd.n1 import n1
    assert isinstance(par.ivfs, filt_simple.library_fullsky_sepTP) and isinstance(par.ivfs_d, filt_util.library_shuffle)
    for q in (par.qlms_dd, par.qlms_ds, par.qlms_ss):
        assert isinstance(q, qest.library) and q.get_lmax_qlm('p') == 4096
    assert isinstance(par.qcls_dd, qecl.library) and isinstance(par.nhl_dd, nhl.nhl_lib_simple)
    assert isinstance(par.n1_dd, n1.library_n1) and isinstance(par.qresp_dd, qresp.resp_lib_simple)
    assert par.cls_path == os.path.join(os.path.dirname(os.path.abspath(pa.__file__)), 'data', 'cls') and par.cl_len['tt'].size > 2048
    temp = os.path.join(os.environ['PLENS'], 'temp', 'idealized_example')
    for f in ('ivfs/filt_hash.pk', 'qlms_dd/qe_sim_hash.pk', 'qlms_dd/fskies.dat', 'qcls_dd/qcl_sim_hash.pk', 'n1_ffp10/n1_hash.pk'):
        assert os.path.exists(os.path.join(temp, f)), f
    try:  # the simulations themselves are NERSC data
        par.sims.get_sim_tmap(0)
    except RuntimeError as e:
        assert 'CFS' in str(e)
        print('LOADED; first data access stops at: %%s' %% e)
'''


@pytest.mark.skipif(not os.path.exists(PARFILE), reason='needs /root/reference (build container only)')
def test_reference_parameter_file_loads_unchanged(tmp_path):
    script = tmp_path / 'child.py'
    script.write_text(CHILD % {'root': ROOT, 'par': PARFILE})
    env = dict(os.environ, PLENS=str(tmp_path / 'plens'))
    env.pop('PLENS_HEALPIX_DATA', None); env.pop('HEALPY_DATAPATH', None); env.pop('HEALPIX', None); env.pop('CFS', None)
    # without healpy's data tables the file stops exactly at its hp.pixwin call (line 49), with an explanatory error
    out = subprocess.run([sys.executable, str(script), 'nodata'], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    txt = out.stdout.decode()
    assert out.returncode == 0 and 'STOPPED line 49' in txt and 'pixwin' in txt, txt[-2000:]
    # with a window-function table present it loads to the end and instantiates every library of the file
    data = tmp_path / 'healpix_data'
    data.mkdir()
    out = subprocess.run([sys.executable, str(script), 'data'], env=dict(env, PLENS_HEALPIX_DATA=str(data)), stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, timeout=900)
    txt = out.stdout.decode()
    assert out.returncode == 0 and 'LOADED; first data access stops at' in txt, txt[-3000:]
