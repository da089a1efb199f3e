"""Host-side logic: healpy-compatible helpers, FITS cache formats, utils (no GPU)."""
import os

import numpy as np
import pytest

from plancklens_amd import hp, utils
from helpers import random_alm, alm_size


def test_alm_indexing():
    lmax = 13
    assert hp.Alm.getsize(lmax) == alm_size(lmax) and hp.Alm.getlmax(alm_size(lmax)) == lmax
    assert hp.Alm.getlmax(alm_size(lmax) + 1) == -1
    l, m = hp.Alm.getlm(lmax)
    assert np.all(hp.Alm.getidx(lmax, l, m) == np.arange(alm_size(lmax)))
    assert l.min() == 0 and l.max() == lmax and np.all(m <= l)


def test_almxfl_alm2cl():
    rng = np.random.default_rng(0)
    lmax = 20
    a, b = random_alm(rng, lmax), random_alm(rng, lmax)
    fl = rng.standard_normal(12)  # shorter than lmax + 1 -> zero extended
    out = hp.almxfl(a, fl)
    l, m = hp.Alm.getlm(lmax)
    f = np.zeros(lmax + 1); f[:12] = fl
    assert np.allclose(out, a * f[l]) and not np.shares_memory(out, a)
    cl = hp.alm2cl(a, b)
    ref = np.zeros(lmax + 1)
    for i in range(a.size):
        ref[l[i]] += (1. if m[i] == 0 else 2.) * (a[i] * np.conj(b[i])).real
    assert np.allclose(cl, ref / (2 * np.arange(lmax + 1) + 1))
    assert np.allclose(hp.alm2cl(a, lmax_out=5), hp.alm2cl(a)[:6])


def test_alm_copy_and_cli():
    rng = np.random.default_rng(1)
    a = random_alm(rng, 15)
    b = utils.alm_copy(a, lmax=9)
    l, m = hp.Alm.getlm(9)
    assert np.all(b == a[hp.Alm.getidx(15, l, m)])
    assert np.all(utils.alm_copy(a) == a)
    with pytest.raises(AssertionError):
        utils.alm_copy(a, lmax=16)
    assert np.all(utils.cli(np.array([0., 2., -1., 4.])) == np.array([0., 0.5, 0., 0.25]))


def test_geometry_and_ud_grade():
    for nside in (1, 2, 8):
        cth, sth, nphi, phi0, ofs = hp.ring_info(nside)
        assert nphi.sum() == 12 * nside ** 2 and np.allclose(cth ** 2 + sth ** 2, 1.)
        assert np.allclose(cth, -cth[::-1]) and np.all(nphi == nphi[::-1])
        x, y, z = hp.pix2vec(nside)
        assert abs(z.sum()) < 1e-10 and abs(x.sum()) < 1e-10  # symmetric pixelisation
    p = hp.nest2ring(8, np.arange(768))
    assert sorted(p) == list(range(768))
    m = np.arange(768, dtype=float)
    d = hp.ud_grade(m, 4, power=-2)
    assert d.size == 192 and np.isclose(d.sum(), m.sum())  # power=-2: children are summed
    # children of a coarse pixel are spatially inside it: degrade of a smooth map is close to the coarse map
    x8, y8, z8 = hp.pix2vec(8)
    x4, y4, z4 = hp.pix2vec(4)
    assert np.abs(hp.ud_grade(z8, 4) - z4).max() < 0.02


def test_fits_roundtrip(tmp_path):
    rng = np.random.default_rng(2)
    a = random_alm(rng, 17)
    fn = str(tmp_path / 'a.fits')
    hp.write_alm(fn, a, overwrite=True)
    assert np.all(hp.read_alm(fn) == a)
    assert os.path.getsize(fn) % 2880 == 0
    with pytest.raises(OSError):
        hp.write_alm(fn, a, overwrite=False)
    m = rng.standard_normal((3, 12 * 16 ** 2))
    fm = str(tmp_path / 'm.fits')
    hp.write_map(fm, m, overwrite=True)
    assert np.all(np.array(hp.read_map(fm, field=None)) == m)
    assert np.all(hp.read_map(fm, field=1) == m[1])


def test_camb_clfile_and_hash():
    fn = os.path.join(os.path.dirname(utils.__file__), 'data', 'cls', 'FFP10_wdipole_lensedCls.dat')
    cl = utils.camb_clfile(fn, lmax=100)
    assert set(cl.keys()) >= {'tt', 'ee', 'bb', 'te'} and len(cl['tt']) == 101 and cl['tt'][0] == 0
    assert 1000 < cl['tt'][2] * 2 * 3 / (2 * np.pi) < 1100  # D_2^TT of FFP10 ~ 1035 muK^2
    utils.hash_check({'a': 1, 'b': {'c': np.ones(3)}}, {'a': 1, 'b': {'c': np.ones(3)}})
    with pytest.raises(AssertionError):
        utils.hash_check({'a': 1}, {'a': 2})
    assert utils.mchash([3, 1, 2]) == utils.mchash([1, 2, 3])


def test_bench_cpu_baseline_leg_runs_on_the_host():
    """bench.py's cpu_baseline leg (oracle, both stages in C) on a tiny configuration: the contract keys, a positive rate, every
    ring pair measured (no extrapolation when it fits the budget), and the usable-CPU count honours affinity / cgroup quota."""
    import bench
    n = bench.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    r = bench.cpu_baseline(16, 32, 20.)
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in r
    assert r['kind'] == 'port' and r['value'] > 0 and r['cores'] == n and r['extrapolated_from_ring_stride'] == 1
    assert bench.executed_flops(64, 64, 2) <= 24. * (65 * 66 // 2) * 128 and bench.executed_flops(64, 64, 0) > 0


def test_fits_reader_on_a_file_written_elsewhere():
    """The FITS reader against a binary table that this repository did not write: the STSDAS-written table shipped with
    numpy's test data (three rows of 1D, 1J, 5A columns), decoded independently by np.rec.fromfile as numpy's own test does."""
    import numpy
    from plancklens_amd import fitsio
    fname = os.path.join(os.path.dirname(numpy.__file__), '_core', 'tests', 'data', 'recarray_from_file.fits')
    if not os.path.exists(fname):
        pytest.skip('numpy test data not installed')
    cols, hdr = fitsio.read_bintable(fname)
    with open(fname, 'rb') as fd:
        fd.seek(2880 * 2)
        ref = np.rec.fromfile(fd, formats='f8,i4,S5', shape=3, byteorder='big')
    assert hdr['TFIELDS'] == 3 and hdr['NAXIS2'] == 3 and hdr['TFORM3'] == '5A'
    assert np.array_equal(cols['a'], ref['f0']) and np.allclose(cols['a'], [5.1, 5.2, 5.3])
    assert np.array_equal(cols['b'], ref['f1']) and list(cols['b']) == [61, 62, 63]
    assert [c.decode().strip() for c in cols['c']] == ['abcde', 'fghij', 'kl']


def test_alm_files_have_healpy_s_table_layout(tmp_path):
    """healpy.write_alm / read_alm (plancklens/qest.py:17,201; filt/filt_simple.py:97-99): the fixture is that table for a small
    alm, assembled byte by byte from the documented format by tests/golden/make_healpy_alm_fixture.py (independent of fitsio.py).
    read_alm must decode it; write_alm must produce the same rows -- index = l^2 + l + m + 1 as 32-bit, real and imag as 64-bit
    big-endian, m-major -- and the structural keywords a reader relies on."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import make_healpy_alm_fixture as mk
    from plancklens_amd import hp
    fix = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'healpy_alm_lmax3.fits')
    raw = open(fix, 'rb').read()
    assert raw == mk.build(), 'the committed fixture is what its script builds'
    rows = mk.alm_values()
    expect = np.array([re + 1j * im for _, re, im in rows])
    alm, mmax = hp.read_alm(fix, return_mmax=True)
    assert mmax == mk.LMAX and np.array_equal(alm, expect)
    out = str(tmp_path / 'ours.fits')
    hp.write_alm(out, expect)
    ours = open(out, 'rb').read()
    assert len(ours) == len(raw) and len(ours) % 2880 == 0
    assert ours[-2880:] == raw[-2880:], 'table rows differ from the healpy layout'

    def cards(b):
        hdr = b[2880:5760].decode('ascii')
        kv = {}
        for i in range(0, 2880, 80):
            c = hdr[i:i + 80]
            if c[8:10] == '= ':
                kv[c[:8].strip()] = c[10:].split(' /')[0].strip().strip("'").strip()
        return kv
    a, b = cards(ours), cards(raw)
    for k in ['XTENSION', 'BITPIX', 'NAXIS', 'NAXIS1', 'NAXIS2', 'PCOUNT', 'GCOUNT', 'TFIELDS', 'TTYPE1', 'TTYPE2', 'TTYPE3']:
        assert a[k] == b[k], (k, a[k], b[k])
    for k in ['TFORM1', 'TFORM2', 'TFORM3']:
        assert a[k].lstrip('1') == b[k], (k, a[k], b[k])  # '1J' and 'J' are the same format
    assert ours[:2880] == raw[:2880] or ours[:80].startswith(b'SIMPLE  =                    T')
    # a truncated write keeps healpy's semantics: rows with l <= lmax only
    hp.write_alm(out, expect, lmax=2, overwrite=True)
    assert np.array_equal(hp.read_alm(out), expect[[0, 1, 2, 4, 5, 7]])


def test_spectral_matrix_and_statistics_helpers():
    """utils.cls_dot / extcl (utils.py:367-416) and the utils.stats methods behind the spectra statistics (utils.py:181-260)
    against explicit numpy evaluations of their definitions"""
    from plancklens_amd import utils
    rng = np.random.default_rng(1)
    a = {'tt': rng.random(6), 'ee': rng.random(4), 'te': rng.random(6), 'bb': rng.random(5)}
    b = {'tt': rng.random(3), 'eb': rng.random(6), 'bb': rng.random(6)}
    A, B = utils._cldict2arr(a), utils._cldict2arr(b)
    assert A.shape == (3, 3, 6) and np.array_equal(A[0, 1], A[1, 0]) and np.array_equal(A[1, 1, :4], a['ee']) and not A[1, 1, 4:].any()
    ref = np.stack([[sum(A[i, k] * sum(B[k, m] * A[m, j] for m in range(3)) for k in range(3)) for j in range(3)] for i in range(3)])
    assert np.allclose(utils.cls_dot([a, b, a]), ref, rtol=1e-14, atol=0)
    d = utils.cls_dot([a, b], ret_dict=True)
    assert np.allclose(d['tt'], A[0, 0] * B[0, 0]) and np.allclose(d['eb'], (A[1, 1] * B[1, 2] + A[1, 0] * B[0, 2]))
    assert np.array_equal(utils.extcl(7, np.arange(3.)), [0, 1, 2, 0, 0, 0, 0, 0]) and np.array_equal(utils.extcl(1, np.arange(3.)), [0, 1])
    x = rng.standard_normal((40, 5)) @ rng.standard_normal((5, 5))
    st = utils.stats(5)
    for v in x:
        st.add(v)
    cov = np.cov(x.T)
    assert np.allclose(st.cov(), cov) and np.allclose(st.avg(), x.mean(0)) and np.allclose(st.corrcoeffs(), np.corrcoef(x.T))
    assert np.allclose(st.inverse(), (40 - 5 - 2.) / 39. * np.linalg.inv(cov))
    dx = np.ones(5) - x.mean(0)
    assert np.isclose(st.get_chisq(np.ones(5)), dx @ st.inverse() @ dx) and 0. <= st.get_chisq_pte(np.ones(5)) <= 1.
    rb = st.rebin_that_nooverlap(np.arange(5.), np.array([0., 2.]), np.array([1., 4.]), weights=np.array([1., 3., 1., 1., 2.]))
    xb = np.stack([(x[:, 0] + 3 * x[:, 1]) / 4., (x[:, 2] + x[:, 3] + 2 * x[:, 4]) / 4.], axis=1)
    assert np.allclose(rb.mean(), xb.mean(0)) and np.allclose(rb.cov(), np.cov(xb.T))
    alm = rng.standard_normal(10) + 1j * rng.standard_normal(10)
    alm[:4] = alm[:4].real
    assert np.allclose(utils.rlm2alm(utils.alm2rlm(alm)), alm) and utils.alm2rlm(alm).size == 16


def test_sql_tables_follow_the_reference_schema(tmp_path):
    """helpers.sql.npdb / fldb (plancklens/helpers/sql.py:28-106): same table and column names (files interchangeable with the reference's),
    add refuses duplicates without raising, get returns None for a missing id, remove deletes."""
    import sqlite3
    from plancklens_amd.helpers import sql
    db = sql.npdb(str(tmp_path / 'cl.db'))
    v = np.arange(5.) * 0.5
    assert db.get('a') is None
    db.add('a', v)
    db.add('a', 2 * v)  # refused ("npdb add failed!"), the first entry stays
    assert np.array_equal(db.get('a'), v) and db.get('a').ndim == 1
    db.remove('a')
    assert db.get('a') is None
    fl = sql.fldb(str(tmp_path / 'fl.db'), idtype='INTEGER')
    fl.add(3, 0.25)
    fl.add(3, 0.5)
    assert fl.get(3) == 0.25 and fl.get(4) is None
    fl.remove(3)
    assert fl.get(3) is None
    for fn, table, cols in (('cl.db', 'npdb', ['id', 'arr']), ('fl.db', 'fldb', ['id', 'fl'])):
        con = sqlite3.connect(str(tmp_path / fn))
        assert [r[1] for r in con.execute('PRAGMA table_info(%s)' % table)] == cols
        con.close()
