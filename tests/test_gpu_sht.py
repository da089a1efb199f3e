"""GPU parity tests of the four SHTs, through the C ABI (plancklens_amd.shts -> libplshts.so), against the CPU
oracle on identical seeded inputs.  Tolerance: relative rms 1e-11 on maps / alm (north_star asks < 1e-10 at the
SHT level, < 1e-8 on qlm); observed ~1e-14.  Full-size cases use size-independent properties (adjointness,
linearity) because the oracle would take minutes there."""
import numpy as np
import pytest

from helpers import random_alm, alm_dot, relrms, alm_size

pytestmark = pytest.mark.gpu

TOL = 1e-11


@pytest.fixture(scope='module')
def shts():
    import torch
    assert torch.cuda.is_available(), 'GPU tests need the MI355X'
    from plancklens_amd import shts as s
    return s


@pytest.mark.parametrize('nside,lmax', [(1, 2), (2, 5), (8, 16), (8, 23), (12, 30), (16, 47), (32, 64), (48, 100), (64, 191), (128, 256)])
def test_spin0_vs_oracle(shts, oracle, nside, lmax):
    """includes nside 1-2 (rings of 4 pixels), lmax up to 3 nside - 1 (strong aliasing in the polar caps) and nside that are
    not powers of two (12, 48: every ring, belt included, is a Bluestein transform in the generic FFT kernel)"""
    rng = np.random.default_rng(nside * 100 + lmax)
    a = random_alm(rng, lmax)
    assert relrms(shts.alm2map(a, nside, lmax=lmax), oracle.alm2map(a, nside, lmax=lmax)) < TOL
    m = rng.standard_normal(12 * nside ** 2)
    assert relrms(shts.map2alm(m, lmax=lmax, iter=0), oracle.map2alm(m, lmax=lmax)) < TOL


@pytest.mark.parametrize('spin', [1, 2, 3])
@pytest.mark.parametrize('nside,lmax', [(2, 5), (8, 16), (12, 30), (16, 47), (32, 64), (48, 100), (64, 150)])
def test_spin_vs_oracle(shts, oracle, spin, nside, lmax):
    rng = np.random.default_rng(spin * 10000 + nside * 100 + lmax)
    g, c = random_alm(rng, lmax, spin), random_alm(rng, lmax, spin)
    mg = shts.alm2map_spin([g, c], nside, spin, lmax)
    mo = oracle.alm2map_spin([g, c], nside, spin, lmax)
    assert relrms(np.stack(mg), np.stack(mo)) < TOL
    q, u = rng.standard_normal(12 * nside ** 2), rng.standard_normal(12 * nside ** 2)
    ag = shts.map2alm_spin([q, u], spin, lmax)
    ao = oracle.map2alm_spin([q, u], spin, lmax)
    assert relrms(np.stack(ag), np.stack(ao)) < TOL
    # entries below the spin are exactly zero
    ls = np.concatenate([np.arange(m, lmax + 1) for m in range(lmax + 1)])
    assert np.all(ag[0][ls < spin] == 0) and np.all(ag[1][ls < spin] == 0)


def test_longdouble_oracle_and_high_m_scaling(shts, oracle):
    """Rings near the pole with lmax = 1500 on nside 512 exercise the 2^(+-512) block scaling (sin^m underflows
    IEEE double for m > ~ 400 there); compared with the scaled-double oracle on the full sphere."""
    rng = np.random.default_rng(3)
    nside, lmax = 512, 1500
    a = random_alm(rng, lmax)
    assert relrms(shts.alm2map(a, nside, lmax=lmax), oracle.alm2map(a, nside, lmax=lmax)) < TOL
    g, c = random_alm(rng, lmax, 2), random_alm(rng, lmax, 2)
    assert relrms(np.stack(shts.alm2map_spin([g, c], nside, 2, lmax)), np.stack(oracle.alm2map_spin([g, c], nside, 2, lmax))) < TOL
    # unscaled long double (no polar pruning) on a small case
    nside, lmax = 16, 40
    a = random_alm(rng, lmax)
    assert relrms(shts.alm2map(a, nside, lmax=lmax), oracle.alm2map(a, nside, lmax=lmax, mode=0, use_pairs=False)) < TOL


def test_known_answers(shts):
    """monopole, xyz_to_alm dipole (template_removal.py:153-158), constant map."""
    from plancklens_amd import hp
    nside, lmax = 32, 64
    alm = np.zeros(alm_size(lmax), complex)
    alm[0] = np.sqrt(4 * np.pi)
    assert np.abs(shts.alm2map(alm, nside) - 1).max() < 1e-13
    x, y, z = 0.3, -0.7, 1.1
    alm[:] = 0
    alm[1] = z * np.sqrt(4 * np.pi / 3)
    alm[hp.Alm.getidx(lmax, 1, 1)] = (-x + 1j * y) * np.sqrt(2 * np.pi / 3)
    vx, vy, vz = hp.pix2vec(nside)
    assert np.abs(shts.alm2map(alm, nside) - (x * vx + y * vy + z * vz)).max() < 1e-13
    assert abs(shts.map2alm(np.ones(12 * nside ** 2), lmax=lmax, iter=0)[0] - np.sqrt(4 * np.pi)) < 1e-13


def test_fused_almxfl_and_inputs_untouched(shts, oracle):
    from plancklens_amd import hp
    rng = np.random.default_rng(5)
    nside, lmax = 16, 32
    a = random_alm(rng, lmax)
    a0 = a.copy()
    fl = rng.standard_normal(lmax + 1)
    assert relrms(shts.alm2map(a, nside, fl=fl), oracle.alm2map(hp.almxfl(a, fl), nside)) < TOL
    assert np.all(a == a0)
    m = rng.standard_normal((2, 12 * nside ** 2))
    m0 = m.copy()
    g, c = shts.map2alm_spin(m, 2, lmax, fl=fl[:20])  # short filter: zero-extended like hp.almxfl
    go, co = oracle.map2alm_spin(m, 2, lmax)
    assert relrms(g, hp.almxfl(go, fl[:20])) < TOL and relrms(c, hp.almxfl(co, fl[:20])) < TOL
    assert np.all(m == m0)


def test_pol_and_device_tensors(shts, oracle):
    """hp.alm2map(pol=True) / hp.map2alm(pol=True) call shapes (opfilt_tp.py:276,281) and the torch route."""
    import torch
    rng = np.random.default_rng(6)
    nside, lmax = 16, 32
    t, e, b = random_alm(rng, lmax), random_alm(rng, lmax, 2), random_alm(rng, lmax, 2)
    T, Q, U = shts.alm2map(np.array([t, e, b]), nside, lmax=lmax, pol=True)
    assert relrms(T, oracle.alm2map(t, nside)) < TOL
    assert relrms(np.stack([Q, U]), np.stack(oracle.alm2map_spin([e, b], nside, 2, lmax))) < TOL
    t2, e2, b2 = shts.map2alm([T, Q, U], lmax=lmax, pol=True, iter=0)
    assert relrms(t2, oracle.map2alm(T, lmax=lmax)) < TOL
    td = torch.from_numpy(t).cuda()
    out = shts.alm2map(td, nside)
    assert isinstance(out, torch.Tensor) and out.is_cuda and relrms(out.cpu().numpy(), T) < 1e-14
    with pytest.raises(AssertionError):
        shts.map2alm(T, lmax=lmax, iter=3)
    with pytest.raises(AssertionError):
        shts.alm2map(t[:-1], nside)


@pytest.mark.parametrize('spin', [0, 1, 2, 3])
def test_adjointness_and_linearity_full_size(shts, spin):
    """BASELINE size nside = lmax = 2048: <map, alm2map(a)> = npix/4pi <map2alm(map), a> (the identity the CG relies
    on, opfilt_tt.py:190) and linearity of the synthesis."""
    rng = np.random.default_rng(40 + spin)
    nside, lmax = 2048, 2048
    npix = 12 * nside ** 2
    if spin == 0:
        a, a2 = random_alm(rng, lmax), random_alm(rng, lmax)
        t = rng.standard_normal(npix)
        ma = shts.alm2map(a, nside)
        lhs = np.sum(t * ma)
        rhs = npix / (4 * np.pi) * alm_dot(shts.map2alm(t, lmax=lmax, iter=0), a, lmax)
        assert abs(lhs - rhs) < 1e-11 * np.sqrt(np.sum(t ** 2) * np.sum(ma ** 2))
        assert relrms(shts.alm2map(2 * a - 3 * a2, nside), 2 * ma - 3 * shts.alm2map(a2, nside)) < 1e-13
    else:
        g, c = random_alm(rng, lmax, spin), random_alm(rng, lmax, spin)
        q, u = rng.standard_normal(npix), rng.standard_normal(npix)
        mq, mu = shts.alm2map_spin([g, c], nside, spin, lmax)
        ga, ca = shts.map2alm_spin([q, u], spin, lmax)
        lhs = np.sum(q * mq + u * mu)
        rhs = npix / (4 * np.pi) * (alm_dot(ga, g, lmax) + alm_dot(ca, c, lmax))
        assert abs(lhs - rhs) < 1e-11 * np.sqrt((np.sum(q ** 2) + np.sum(u ** 2)) * (np.sum(mq ** 2) + np.sum(mu ** 2)))


def test_lmax_qlm_differs_from_lmax_ivf(shts, oracle):
    """map2alm_spin to a band-limit different from the one the map was synthesised with (qest.py:259)."""
    rng = np.random.default_rng(8)
    nside = 32
    q, u = rng.standard_normal(12 * nside ** 2), rng.standard_normal(12 * nside ** 2)
    for lmax in (20, 95):
        assert relrms(np.stack(shts.map2alm_spin([q, u], 1, lmax)), np.stack(oracle.map2alm_spin([q, u], 1, lmax))) < TOL


def _run_with_plan(shts, plan, fn):
    key = shts._plan_key(plan.nside, plan.lmax)
    old = shts._PLANS.get(key)
    shts._PLANS[key] = plan
    try:
        return fn()
    finally:
        if old is None:
            shts._PLANS.pop(key, None)
        else:
            shts._PLANS[key] = old


@pytest.mark.parametrize('nside,lmax', [(256, 300), (256, 512), (512, 512), (512, 1024), (768, 1000), (1024, 1400), (1024, 2048), (2048, 1024), (2048, 2048)])
def test_register_fft_kernels_match_generic_kernel(shts, nside, lmax):
    """The long rings go through the register-resident ring-FFT kernels (sizes 256 .. 4096, direct and band-limited
    Bluestein); the plan option fft_legacy (pl_plan_opts) sends every ring through the generic LDS kernel, which the
    small-nside tests above pin against the oracle.  Same inputs, both plans: maps and alm must agree to rounding.
    lmax = 2 nside (every coarse grid of the CG chains): the belt rings carry the order n / 2, which the direct classes treat on its own."""
    generic = shts.Plan(nside, lmax, opts={'fft_legacy': 1})
    # lmax = 2 nside: the belt in the direct classes at every size (default: sub-DFTs >= 2048 only); fft_generic_nside = 0: the register
    # classes on the small grids too (by default every ring of a grid up to nside 512 runs in the generic kernel)
    fast = shts.Plan(nside, lmax, opts={'fft_nyq_min': 256, 'fft_generic_nside': 0})
    rng = np.random.default_rng(nside + lmax)
    a = random_alm(rng, lmax)
    g, c = random_alm(rng, lmax, 2), random_alm(rng, lmax, 2)
    m = rng.standard_normal(12 * nside ** 2)
    qu = rng.standard_normal((2, 12 * nside ** 2))
    res = {}
    for name, plan in (('generic', generic), ('fast', fast)):
        res[name] = _run_with_plan(shts, plan, lambda: (
            shts.alm2map(a, nside, lmax=lmax), np.stack(shts.alm2map_spin([g, c], nside, 2, lmax)),
            shts.map2alm(m, lmax=lmax, iter=0), np.stack(shts.map2alm_spin(qu, 2, lmax))))
    for x, y in zip(res['generic'], res['fast']):
        assert relrms(y, x) < 1e-13


@pytest.mark.parametrize('nside,lmax', [(256, 383), (512, 512)])
def test_mid_size_vs_oracle(shts, oracle, nside, lmax):
    """mid sizes against the oracle (spin 0 and 2, both directions): with the default plan (grids up to nside 512: every ring in the generic
    ring-FFT kernel) and with the register-resident FFT classes switched on for these grids (plan option fft_generic_nside = 0)"""
    rng = np.random.default_rng(7 * nside + lmax)
    a = random_alm(rng, lmax)
    m = rng.standard_normal(12 * nside ** 2)
    g, c = random_alm(rng, lmax, 2), random_alm(rng, lmax, 2)
    qu = rng.standard_normal((2, 12 * nside ** 2))
    ref = (oracle.alm2map(a, nside, lmax=lmax), oracle.map2alm(m, lmax=lmax), np.stack(oracle.alm2map_spin([g, c], nside, 2, lmax)),
           np.stack(oracle.map2alm_spin(qu, 2, lmax)))
    for opts in ({}, {'fft_generic_nside': 0}):
        with shts.plan_options(**opts):
            out = (shts.alm2map(a, nside, lmax=lmax), shts.map2alm(m, lmax=lmax, iter=0), np.stack(shts.alm2map_spin([g, c], nside, 2, lmax)),
                   np.stack(shts.map2alm_spin(qu, 2, lmax)))
        for x, y in zip(out, ref):
            assert relrms(x, y) < TOL, opts


@pytest.mark.parametrize('spin', [1, 2, 3])
@pytest.mark.parametrize('nside,lmax', [(16, 40), (64, 150), (512, 700)])
def test_gradient_only_synthesis_equals_zero_curl(shts, spin, nside, lmax):
    """alm2map_spin([G, None]) (pl_alm2map_grad: 4 instead of 8 accumulation FMAs per step) against the general kernel fed
    with an explicit zero curl, host and device routes, with a fused l-filter."""
    import torch
    rng = np.random.default_rng(31 * spin + nside)
    g = random_alm(rng, lmax, spin)
    fl = 1. / (1. + np.arange(lmax + 1))
    ref = np.stack(shts.alm2map_spin([g, np.zeros_like(g)], nside, spin, lmax, fl=fl))
    out = np.stack(shts.alm2map_spin([g, None], nside, spin, lmax, fl=fl))
    assert relrms(out, ref) < 1e-13
    outd = shts.alm2map_spin([torch.from_numpy(g).cuda(), None], nside, spin, lmax, fl=fl)
    assert relrms(np.stack([o.cpu().numpy() for o in outd]), ref) < 1e-13


@pytest.mark.parametrize('spin', [1, 2, 3])
@pytest.mark.parametrize('nside,lmax', [(8, 16), (32, 64), (64, 150), (256, 300), (512, 700), (2048, 2048)])
def test_paired_synthesis_equals_two_calls(shts, spin, nside, lmax):
    """pl_alm2map_pair (general + gradient-only input on one recursion) against the two separate transforms -- including
    nside = lmax = 2048, the size and (for spin 1) the call the headline benchmark times (qest.lib_filt2map.get_gt_gp1maps)."""
    import torch
    from plancklens_amd import dev
    if nside == 2048 and spin != 1:
        pytest.skip('full size: the spin the estimator uses')
    rng = np.random.default_rng(spin * 7 + nside + lmax)
    g, c, g2 = (dev.to_dev(random_alm(rng, lmax, spin)) for _ in range(3))
    fl, fl2 = rng.uniform(0.5, 1.5, lmax + 1), rng.uniform(0.5, 1.5, lmax + 1)
    (q, u), (q2, u2) = shts.alm2map_spin_pair([g, c], g2, nside, spin, lmax, fl=fl, fl2=fl2)
    rq, ru = shts.alm2map_spin([g, c], nside, spin, lmax, fl=fl)
    rq2, ru2 = shts.alm2map_spin([g2, None], nside, spin, lmax, fl=fl2)
    for a, b in ((q, rq), (u, ru), (q2, rq2), (u2, ru2)):
        assert relrms(dev.to_host(a), dev.to_host(b)) < 1e-13


@pytest.mark.parametrize('nside,lmax', [(256, 767), (512, 1535)])
def test_adjointness_at_lmax_3nside_minus_1(shts, nside, lmax):
    """<m, Y a> = npix / 4 pi <Y^t m, a> (the identity opfilt_* relies on) where every ring is aliased (lmax = 3 nside - 1),
    all spins, and the paired synthesis against its two transforms at the same size."""
    import torch
    from plancklens_amd import dev, hp
    rng = np.random.default_rng(nside + lmax)
    n, npix = hp.Alm.getsize(lmax), 12 * nside ** 2
    w = torch.full((n,), 2., dtype=torch.float64, device='cuda')
    w[:lmax + 1] = 1.
    for spin in (0, 1, 2, 3):
        if spin == 0:
            a = dev.to_dev(random_alm(rng, lmax, 0))
            m = torch.randn(npix, dtype=torch.float64, device='cuda')
            lhs = float(torch.dot(m, shts.alm2map(a, nside, lmax=lmax)))
            rhs = float((w * (a.conj() * shts.map2alm(m, lmax=lmax, iter=0)).real).sum()) * npix / (4 * np.pi)
        else:
            g, c = dev.to_dev(random_alm(rng, lmax, spin)), dev.to_dev(random_alm(rng, lmax, spin))
            m = torch.randn((2, npix), dtype=torch.float64, device='cuda')
            q, u = shts.alm2map_spin([g, c], nside, spin, lmax)
            lhs = float(torch.dot(m[0], q) + torch.dot(m[1], u))
            bg, bc = shts.map2alm_spin([m[0], m[1]], spin, lmax=lmax)
            rhs = float((w * ((g.conj() * bg).real + (c.conj() * bc).real)).sum()) * npix / (4 * np.pi)
            (q1, u1), (q2, u2) = shts.alm2map_spin_pair([g, c], c, nside, spin, lmax)
            r2 = shts.alm2map_spin([c, None], nside, spin, lmax)
            for x, y in ((q1, q), (u1, u), (q2, r2[0]), (u2, r2[1])):
                assert float((x - y).abs().max() / y.abs().max()) < 1e-12
        assert abs(lhs / rhs - 1) < 1e-11, (spin, lhs, rhs)


@pytest.mark.parametrize('spin', [1, 2, 3])
@pytest.mark.parametrize('nside,lmax', [(8, 16), (32, 64), (64, 150), (256, 300), (512, 700), (2048, 2048)])
def test_batched_synthesis_is_bit_identical_to_two_calls(shts, spin, nside, lmax):
    """pl_alm2map_batch2 (two simulations on one recursion) against two pl_alm2map calls: every sum is formed in the same order,
    so the maps are equal bit for bit -- including nside = lmax = 2048, where the rings-per-lane 2 kernel with the three
    scaling phases runs."""
    import torch
    from plancklens_amd import dev
    rng = np.random.default_rng(spin * 11 + nside + lmax)
    g1, c1, g2, c2 = (dev.to_dev(random_alm(rng, lmax, spin)) for _ in range(4))
    fl = rng.uniform(0.5, 1.5, lmax + 1)
    (q1, u1), (q2, u2) = shts.alm2map_spin_batch2([g1, c1], [g2, c2], nside, spin, lmax, fl=fl)
    r1 = shts.alm2map_spin([g1, c1], nside, spin, lmax, fl=fl)
    r2 = shts.alm2map_spin([g2, c2], nside, spin, lmax, fl=fl)
    for a, b in ((q1, r1[0]), (u1, r1[1]), (q2, r2[0]), (u2, r2[1])):
        assert bool((a == b).all()), float((a - b).abs().max())


@pytest.mark.parametrize('nside,lmax', [(16, 40), (64, 128), (128, 300), (512, 512), (1024, 1500), (2048, 2048)])
def test_seed_tables_start_the_recursions_where_they_would_have_arrived(shts, nside, lmax):
    """A plan keeps, per kernel family, the recursion state of every (m, ring pair) at the step where the family's kernels stop recursing
    without accumulating (pl_plan_opts.seed_tables, DevSeedTab), made by the same arithmetic: every transform of a plan with the tables
    equals that of a plan without them (seed_tables = 0: every launch recurses from l = m) bit for bit -- all spins, both directions,
    the gradient-only, paired and two-simulation syntheses, a filter on the way."""
    import torch
    from plancklens_amd import dev
    rng = np.random.default_rng(11 * nside + lmax)
    plain = shts.Plan(nside, lmax, opts={'seed_tables': 0})
    seeded = shts.Plan(nside, lmax)
    assert seeded.bytes() > plain.bytes()
    npix = 12 * nside ** 2
    fl = 1. / (1. + np.arange(lmax + 1.)) ** .5
    a = dev.to_dev(random_alm(rng, lmax))
    gc = [[dev.to_dev(random_alm(rng, lmax, s)) for _ in range(2)] for s in (1, 2, 3)]
    gc2 = [dev.to_dev(random_alm(rng, lmax, 2)) for _ in range(2)]
    m = dev.to_dev(rng.standard_normal(npix))
    qu = dev.to_dev(rng.standard_normal((2, npix)))

    def everything():
        out = [shts.alm2map(a, nside, lmax=lmax), shts.alm2map(a, nside, lmax=lmax, fl=fl), shts.map2alm(m, lmax=lmax, iter=0)]
        for s in (1, 2, 3):
            g, c = gc[s - 1]
            out += shts.alm2map_spin([g, c], nside, s, lmax) + shts.alm2map_spin([g, None], nside, s, lmax, fl=fl)
            out += shts.map2alm_spin([qu[0], qu[1]], s, lmax) + shts.map2alm_spin([qu[1], qu[0]], s, lmax, fl=fl)
        g, c = gc[1]
        for pair in (shts.alm2map_spin_pair([g, c], gc2[0], nside, 2, lmax, fl=fl), shts.alm2map_spin_grad_pair(g, gc2[0], nside, 2, lmax),
                     shts.alm2map_spin_batch2([g, c], gc2, nside, 2, lmax)):
            out += pair[0] + pair[1]
        return [dev.to_host(x) if isinstance(x, torch.Tensor) else np.asarray(x) for x in out]
    r0 = _run_with_plan(shts, plain, everything)
    r1 = _run_with_plan(shts, seeded, everything)
    assert len(r0) == len(r1)
    for i, (x, y) in enumerate(zip(r0, r1)):
        assert x.shape == y.shape and np.array_equal(x, y), (i, float(np.max(np.abs(x - y))))
