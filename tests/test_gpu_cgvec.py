"""GPU parity of the CG vector primitives of the C ABI (pl_alm_dot, pl_axpy_dev, pl_alm_splice, pl_almxfl_add,
pl_alm_copy) against the numpy statements of the same operations (hp.alm2cl weights, util_alm host path) and of the
single-matrix dense preconditioner against the alm -> rlm -> mat-vec -> alm route of the reference (dense.py:16-119)."""
import numpy as np
import pytest

from helpers import relrms

pytestmark = pytest.mark.gpu


def _rand_alm(rng, lmax):
    from plancklens_amd import hp
    n = hp.Alm.getsize(lmax)
    a = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    a[:lmax + 1] = a[:lmax + 1].real
    return a


@pytest.mark.parametrize('lmax', [0, 1, 5, 64, 300, 1500])
def test_alm_dot_axpy_almxfl_add(lmax):
    import torch
    from plancklens_amd import dev, hp
    rng = np.random.default_rng(lmax)
    a, b = _rand_alm(rng, lmax), _rand_alm(rng, lmax)
    da, db = dev.to_dev(a), dev.to_dev(b)
    for lmin in (0, 2):
        w = 2. * np.arange(lmax + 1) + 1.
        w[:lmin] = 0.
        ref = np.sum(hp.alm2cl(a, b) * w)
        scale = np.sum(np.abs(hp.alm2cl(a, a)) * w) + 1e-300
        assert abs(float(dev.alm_dot([(da, db)], lmin=lmin).sum()) - ref) < 1e-13 * max(scale, 1.)       # rounding only
        assert abs(float(dev.alm_dot([(da, db), (db, da)], lmin=lmin).sum()) - 2 * ref) < 2e-13 * max(scale, 1.)  # accumulation over pairs
    # same inputs, same launch: bit-identical (fixed reduction tree, no atomics)
    assert torch.equal(dev.alm_dot([(da, db)]), dev.alm_dot([(da, db)]))
    num = torch.zeros(dev.DOT_PARTS, dtype=torch.float64, device="cuda"); num[:3] = 1.0   # partial sums: 3
    den = torch.zeros(dev.DOT_PARTS, dtype=torch.float64, device="cuda"); den[-4:] = -1.0  # partial sums: -4
    y = da.clone()
    dev.axpy_dev(y, db, num, den, -1.0)
    assert np.allclose(dev.to_host(y), a + 0.75 * b, rtol=1e-14, atol=1e-14)
    y = da.clone()
    dev.axpy_dev(y, db, num, None, 1.0)
    assert np.allclose(dev.to_host(y), a + 3 * b, rtol=1e-14, atol=1e-14)
    fl = rng.standard_normal(max(lmax - 1, 1))  # shorter than lmax + 1: zero-extended as in hp.almxfl
    assert np.allclose(dev.to_host(dev.almxfl_add(da, db, fl)), a + hp.almxfl(b, fl), rtol=1e-14, atol=1e-14)
    out = da.clone()
    dev.almxfl_add(out, db, fl, out=out)  # in place on the first operand
    assert np.allclose(dev.to_host(out), a + hp.almxfl(b, fl), rtol=1e-14, atol=1e-14)


@pytest.mark.parametrize('lmax_lo,lmax_hi,lsplit', [(8, 8, 8), (8, 20, 5), (20, 8, 8), (64, 300, 64), (70, 300, 33)])
def test_alm_splice_and_copy(lmax_lo, lmax_hi, lsplit):
    from plancklens_amd import dev
    from plancklens_amd.qcinv import util_alm
    rng = np.random.default_rng(lmax_lo * 1000 + lmax_hi)
    lo, hi = _rand_alm(rng, lmax_lo), _rand_alm(rng, lmax_hi)
    ref = util_alm.alm_splice(lo, hi, lsplit)                                      # host route: per-entry index maps
    got = dev.to_host(util_alm.alm_splice(dev.to_dev(lo), dev.to_dev(hi), lsplit))  # device route: pl_alm_splice
    assert np.array_equal(ref, got)
    if lsplit < lmax_hi:
        assert np.array_equal(util_alm.alm_copy(hi, lsplit), dev.to_host(util_alm.alm_copy(dev.to_dev(hi), lsplit)))
    # pl_alm_splice_fl: the diagonal high-l preconditioner applied inside the splice = almxfl, then splice (bit-identical)
    fl = rng.uniform(0.5, 2., lmax_hi + 1)
    two = util_alm.alm_splice(dev.to_dev(lo), dev.almxfl(dev.to_dev(hi), fl), lsplit)
    assert np.array_equal(dev.to_host(two), dev.to_host(dev.alm_splice_fl(dev.to_dev(lo), dev.to_dev(hi), fl, lsplit)))


@pytest.mark.parametrize('nfields', [1, 2, 3])
def test_dense_single_matrix_equals_rlm_route(nfields):
    import torch
    from plancklens_amd import dev
    from plancklens_amd.qcinv import dense
    from plancklens_amd.qcinv.util_alm import eblm, teblm
    lmax = 9
    rng = np.random.default_rng(nfields)
    nr = (lmax + 1) ** 2 * nfields
    minv = rng.standard_normal((nr, nr))
    minv = dev.to_dev(minv + minv.T, torch.float64)
    parts = [dev.to_dev(_rand_alm(rng, lmax)) for _ in range(nfields)]
    rlm = torch.cat([dense.alm2rlm(p) for p in parts])
    out = torch.mv(minv, rlm)
    n = (lmax + 1) ** 2
    ref = [dev.to_host(dense.rlm2alm(out[k * n:(k + 1) * n])) for k in range(nfields)]
    cls = {1: dense.pre_op_dense_tt, 2: dense.pre_op_dense_pp, 3: dense.pre_op_dense_tp}[nfields]
    op = cls.__new__(cls)  # the operator around a given pseudo-inverse (no fwd_op needed)
    op.lmax, op.minv = lmax, minv
    vec = parts[0] if nfields == 1 else (eblm(parts) if nfields == 2 else teblm(parts))
    res = op.calc(vec)
    got = [res] if nfields == 1 else ([res.elm, res.blm] if nfields == 2 else [res.tlm, res.elm, res.blm])
    for r, g_ in zip(ref, got):
        assert np.allclose(r, dev.to_host(g_), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize('npix,nmodes', [(48, 1), (3072, 4), (196608, 4), (786432, 16)])
def test_template_project(npix, nmodes):
    import torch
    from plancklens_amd import dev
    rng = np.random.default_rng(npix + nmodes)
    t, ninv = rng.standard_normal(npix), rng.uniform(0.5, 2., npix) * (rng.uniform(size=npix) > 0.3)
    pm = rng.standard_normal((nmodes, npix))
    pinv = np.linalg.inv(pm @ (pm * ninv).T)
    rm = pinv @ (pm * ninv)
    u = ninv * t
    ref = u - rm.T @ (pm @ u)     # N^-1 t - N^-1 P (P^t N^-1 P)^-1 P^t N^-1 t
    dt = dev.to_dev(t)
    dev.template_project(dt, dev.to_dev(ninv), dev.to_dev(pm), dev.to_dev(rm))
    got = dev.to_host(dt)
    assert np.max(np.abs(got - ref)) < 1e-11 * np.max(np.abs(u))
    assert np.max(np.abs(pm @ got)) < 1e-8 * np.max(np.abs(pm @ u))  # the template modes are gone from the weighted map
    dt2 = dev.to_dev(t)
    dev.template_project(dt2, dev.to_dev(ninv), dev.to_dev(pm), dev.to_dev(rm))
    assert torch.equal(dt, dt2)  # bit-reproducible


@pytest.mark.parametrize('nrows,ncols', [(1, 1), (7, 5), (64, 257), (130, 4290), (4290, 4290), (33, 1000)])
def test_gemv_vs_numpy(nrows, ncols):
    """pl_gemv (the dense preconditioner mat-vec, dense.py:118-119): against numpy on the host, odd and even sizes (the
    16-byte and the scalar load variants), and bit-reproducible from call to call."""
    import torch
    from plancklens_amd import dev
    rng = np.random.default_rng(nrows * 7 + ncols)
    a = rng.standard_normal((nrows, ncols))
    x = rng.standard_normal(ncols)
    ad, xd = dev.to_dev(a), dev.to_dev(x)
    y = dev.gemv(ad, xd)
    ref = a @ x
    assert np.abs(dev.to_host(y) - ref).max() < 1e-13 * np.sqrt(ncols) * max(1., np.abs(ref).max())
    assert bool((dev.gemv(ad, xd) == y).all())
    if ncols > 3:  # a view with an odd leading dimension takes the scalar-load kernel
        sub = ad[:, :ncols - 1]
        y2 = torch.empty(nrows, dtype=torch.float64, device='cuda')
        from plancklens_amd import _lib
        _lib.check(_lib.lib().pl_gemv(nrows, ncols - 1, ncols, sub.data_ptr(), xd.data_ptr(), y2.data_ptr(), dev.stream_ptr()))
        ref2 = a[:, :ncols - 1] @ x[:ncols - 1]
        assert np.abs(dev.to_host(y2) - ref2).max() < 1e-13 * np.sqrt(ncols) * max(1., np.abs(ref2).max())


@pytest.mark.parametrize('lmaxs,lmin', [((32,), 0), ((256,), 0), ((2048,), 0), ((100, 100), 2), ((2048, 2048), 2), ((300, 200, 200), 0)])
def test_cg_dot_axpy_one_launch(lmaxs, lmin):
    """pl_cg_dot_axpy (two launches for all fields, or one with a grid barrier inside) against pl_alm_dot + pl_axpy_dev: bit-identical partial
    sums and vectors, in the conjugate-directions form (two products, two updates) and the re-orthogonalisation form (one, one)."""
    import torch
    from plancklens_amd import dev, hp
    rng = np.random.default_rng(sum(lmaxs) + lmin)

    def vec():
        return [dev.to_dev(rng.standard_normal(hp.Alm.getsize(l)) + 1j * rng.standard_normal(hp.Alm.getsize(l))) for l in lmaxs]
    d, q, r, x = vec(), vec(), vec(), vec()
    for update_r, one in ((True, False), (False, False), (True, True), (False, True)):
        x1, r1 = [t.clone() for t in x], [t.clone() for t in r]
        p1, p2 = dev.cg_dot_axpy(d, q, x1, d, 1.0, b2=r1, y2=r1 if update_r else None, x2=q if update_r else None, sign2=-1.0, lmin=lmin,
                                 one_launch=one)
        dtad = dev.alm_dot(list(zip(d, q)), lmin=lmin)
        delta = dev.alm_dot(list(zip(d, r)), lmin=lmin)
        x2, r2 = [t.clone() for t in x], [t.clone() for t in r]
        for k in range(len(lmaxs)):
            dev.axpy_dev(x2[k], d[k], delta, dtad, 1.0)
            if update_r:
                dev.axpy_dev(r2[k], q[k], delta, dtad, -1.0)
        assert torch.equal(p1, dtad) and torch.equal(p2, delta)
        for k in range(len(lmaxs)):
            assert torch.equal(x1[k], x2[k]) and torch.equal(r1[k], r2[k])
            assert not torch.equal(x1[k], x[k]) and (torch.equal(r1[k], r[k]) != update_r)
    for one in (False, True):
        s1, s2 = [t.clone() for t in x], [t.clone() for t in x]
        p1, none = dev.cg_dot_axpy(s1, q, s1, d, -1.0, den=dtad, lmin=lmin, one_launch=one)
        assert none is None
        num = dev.alm_dot(list(zip(s2, q)), lmin=lmin)
        for k in range(len(lmaxs)):
            dev.axpy_dev(s2[k], d[k], num, dtad, -1.0)
        assert torch.equal(p1, num)
        for k in range(len(lmaxs)):
            assert torch.equal(s1[k], s2[k])
    for _ in range(200):  # the barrier words are reusable back to back
        dev.cg_dot_axpy(s1, q, s1, d, -1.0, den=dtad, lmin=lmin, one_launch=True)
    assert not dev.cg_barrier_timed_out()


@pytest.mark.parametrize('nside', [1, 2, 16, 64, 512])
def test_monopole_dipole_projection_from_the_ring_geometry(nside):
    """pl_template_project_md_b (templates (1, x, y, z) evaluated per pixel from the plan's ring geometry) against pl_template_project
    on the stored template maps of template_removal.template_monopole / template_dipole (template_removal.py:116-150), single maps and
    a block of five (one full chunk of four and a remainder)."""
    import torch
    from plancklens_amd import dev, hp
    rng = np.random.default_rng(nside)
    npix = 12 * nside ** 2
    x, y, z = hp.pix2vec(nside, np.arange(npix))
    pm = np.stack([np.ones(npix), x, y, z])
    ninv = (rng.random(npix) + 0.5) * (np.abs(z) > 0.2 if nside > 2 else 1.)
    pinv = np.linalg.inv((pm * ninv) @ pm.T)
    rm = pinv @ (pm * ninv)
    t = rng.standard_normal((5, npix))
    ref = t * ninv
    ref = ref - (ref @ pm.T) @ rm
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    got = dev.template_project_md(d(t), d(ninv), nside, 2 * nside, d(pinv))
    assert relrms(dev.to_host(got), ref) < 1e-13
    one = dev.template_project_md(d(t[3]), d(ninv), nside, 2 * nside, d(pinv))
    assert bool((one == got[3]).all())
    if nside >= 16:
        mat = dev.template_project(d(t[3]), d(ninv), d(pm), d(rm))
        assert relrms(dev.to_host(one), dev.to_host(mat)) < 1e-13
