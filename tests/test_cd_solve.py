"""cd_solve on the host with plain numpy vectors (the solver is generic over vector types; plancklens/qcinv/cd_solve.py:35-107):
a small symmetric positive-definite system, the three code paths of the solver (host scalars; device-style `dev` scalars; fused
`parts` / `axpy`, and the merged `step` / `ortho` form), the private right-hand side of the nested solves, the residual refresh."""
import numpy as np
import pytest

from plancklens_amd.qcinv import cd_solve


def _system(n=40, seed=0):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((n, n))
    amat = a @ a.T + n * np.eye(n)
    return amat, rng.standard_normal(n)


class _dot_host(object):
    def __call__(self, a, b):
        return float(np.dot(a, b))


class _dot_parts(_dot_host):
    """the interface of the device scalar products: `parts` returns an opaque object, `axpy` consumes two of them"""
    calls = 0

    def dev(self, a, b):
        return np.dot(a, b)

    def parts(self, a, b):
        return np.array([np.dot(a, b)])

    @staticmethod
    def axpy(y, x, num, den, sign):
        y += sign * num[0] / den[0] * x


class _dot_merged(_dot_parts):
    def step(self, x, d, r, q, update_r=True):
        _dot_merged.calls += 1
        dtad, delta = self.parts(d, q), self.parts(d, r)
        x += delta[0] / dtad[0] * d
        if update_r:
            r -= delta[0] / dtad[0] * q
        return dtad, delta

    def ortho(self, s, pq, pd, prev_dtad):
        s -= np.dot(s, pq) / prev_dtad[0] * pd


class _stop_after(object):
    def __init__(self, n):
        self.n = n

    def __call__(self, it, x, residual):
        return it >= self.n


@pytest.mark.parametrize('dot', [_dot_host, _dot_parts, _dot_merged])
@pytest.mark.parametrize('tr', [cd_solve.tr_cg, cd_solve.tr_cd])
def test_solves_and_all_paths_agree(dot, tr):
    amat, b = _system()
    fwd = lambda v: amat @ v
    pre = lambda v: v / np.diag(amat)
    x = np.zeros_like(b)
    its = cd_solve.cd_solve(x, b, fwd, [pre], dot(), _stop_after(60), tr, roundoff=25)
    assert its == 60
    assert np.allclose(x, np.linalg.solve(amat, b), rtol=1e-10, atol=1e-12)
    if dot is _dot_merged:
        assert _dot_merged.calls > 0


def test_private_right_hand_side_and_refresh_guard():
    amat, b = _system(seed=1)
    fwd = lambda v: amat @ v
    pre = lambda v: v / np.diag(amat)
    ref = np.zeros_like(b)
    cd_solve.cd_solve(ref, b.copy(), fwd, [pre], _dot_merged(), _stop_after(5), cd_solve.tr_cg, x_is_zero=True)
    x, rhs = np.zeros_like(b), b.copy()
    cd_solve.cd_solve(x, rhs, fwd, [pre], _dot_merged(), _stop_after(5), cd_solve.tr_cg, x_is_zero=True, b_scratch=True)
    assert np.array_equal(x, ref)                 # same arithmetic, one copy less
    assert not np.array_equal(rhs, b)             # the right-hand side was used as the residual
    with pytest.raises(AssertionError):           # a solve that reaches the residual refresh needs its right-hand side
        cd_solve.cd_solve(np.zeros_like(b), b.copy(), fwd, [pre], _dot_merged(), _stop_after(30), cd_solve.tr_cg, x_is_zero=True,
                          b_scratch=True)


def test_pace_rendezvous_of_two_solves():
    """multigrid.pace: two threads meet once per round; a party that leaves releases the other at once; a party alone is never held
    (beyond the time-out) -- pacing is an optimisation, not a condition for progress."""
    import threading
    import time
    from plancklens_amd.qcinv import multigrid
    pc = multigrid.pace(2, timeout=5.0)
    log = []

    def solve(name, rounds, work):
        for i in range(rounds):
            pc.wait()
            log.append((name, i, time.time()))
            time.sleep(work)
        pc.leave()
    a = threading.Thread(target=solve, args=('a', 6, 0.002))
    b = threading.Thread(target=solve, args=('b', 3, 0.02))
    t0 = time.time()
    a.start(); b.start(); a.join(); b.join()
    assert time.time() - t0 < 2.0, 'a party was held by one that had left'
    ta = {i: t for n, i, t in log if n == 'a'}
    tb = {i: t for n, i, t in log if n == 'b'}
    for i in range(1, 3):  # rounds both take part in start together: the fast party waited for the slow one
        assert abs(ta[i] - tb[i]) < 0.015, (i, ta[i] - tb[i])
        assert ta[i] >= tb[i - 1] + 0.015
    assert len(ta) == 6 and len(tb) == 3
    # a single party with a time-out is released by the time-out
    pc2 = multigrid.pace(2, timeout=0.05)
    t0 = time.time()
    pc2.wait()
    assert 0.04 < time.time() - t0 < 1.0


class _dot_pre(_dot_merged):
    """the merged interface plus the `pre` forms: scalar products handed over by whoever made the vector"""
    lmin = 0
    pre_steps = pre_orthos = inits = 0

    def step(self, x, d, r, q, update_r=True, active=None, pre=None, x_init=False):
        if pre is None:
            assert not x_init
            return _dot_merged.step(self, x, d, r, q, update_r=update_r)
        _dot_pre.pre_steps += 1
        dtad, delta = np.array([pre[0].sum()]), np.array([pre[1].sum()])
        if x_init:
            _dot_pre.inits += 1
            x[:] = delta[0] / dtad[0] * d  # written, never read
        else:
            x += delta[0] / dtad[0] * d
        if update_r:
            r -= delta[0] / dtad[0] * q
        return dtad, delta

    def ortho(self, s, pq, pd, prev_dtad, pre=None):
        if pre is None:
            return _dot_merged.ortho(self, s, pq, pd, prev_dtad)
        _dot_pre.pre_orthos += 1
        s -= pre.sum() / prev_dtad[0] * pd


class _fwd_with_dots(object):
    def __init__(self, amat):
        self.amat = amat

    def __call__(self, v):
        return self.amat @ v

    def with_dots(self, d, r):
        q = self.amat @ d
        # partial sums, as the kernels leave them: any split whose total is the scalar product
        return q, (np.array([np.dot(d[:7], q[:7]), np.dot(d[7:], q[7:])]), np.array([np.dot(d[:3], r[:3]), np.dot(d[3:], r[3:])]))


class _pre_with_dot(object):
    def __init__(self, diag, offer=True):
        self.diag, self.offer = diag, offer

    def __call__(self, v):
        return v / self.diag

    def with_dot(self, v, q, lmin):
        s = v / self.diag
        return s, (np.array([np.dot(s[:11], q[:11]), np.dot(s[11:], q[11:])]) if self.offer else None)


@pytest.mark.parametrize('offer', [True, False])
def test_scalar_products_handed_over_by_the_producing_operators(offer, monkeypatch):
    """fwd_op.with_dots / pre_op.with_dot / dot_op.step(pre=...) / dot_op.ortho(pre=...): the solver takes the scalar products from the
    operators that made the vectors (cd_solve.py:66-84,96-103 without scalar-product launches of their own) and lands on the same
    solution as with its own; a preconditioner that has none to offer falls back per call; x_uninit: the solution vector may hold anything
    on entry, the first step writes it (or it is zero-filled where no `pre` arrives); options.opts.cg_post_dots = False switches the protocol off."""
    amat, b = _system(seed=3)
    ref = np.zeros_like(b)
    cd_solve.cd_solve(ref, b.copy(), lambda v: amat @ v, [lambda v: v / np.diag(amat)], _dot_merged(), _stop_after(12), cd_solve.tr_cg, x_is_zero=True)
    _dot_pre.pre_steps = _dot_pre.pre_orthos = _dot_pre.inits = 0
    x = np.full_like(b, np.nan)  # uninitialised memory
    cd_solve.cd_solve(x, b.copy(), _fwd_with_dots(amat), [_pre_with_dot(np.diag(amat), offer)], _dot_pre(), _stop_after(12), cd_solve.tr_cg,
                      x_is_zero=True, b_scratch=True, x_uninit=True)
    assert np.allclose(x, ref, rtol=1e-12, atol=1e-14)
    assert _dot_pre.pre_steps == 12 and _dot_pre.inits == 1 and _dot_pre.pre_orthos == (11 if offer else 0)
    # a solve that ends before its first step still returns zeros
    x0 = np.full_like(b, np.nan)
    assert cd_solve.cd_solve(x0, b.copy(), _fwd_with_dots(amat), [_pre_with_dot(np.diag(amat))], _dot_pre(), _stop_after(0), cd_solve.tr_cg,
                             x_is_zero=True, x_uninit=True) == 0
    assert np.array_equal(x0, np.zeros_like(b))
    # switched off: the solver's own scalar products, the solution vector zero-filled before the first step
    from plancklens_amd import options
    monkeypatch.setattr(options.opts, 'cg_post_dots', False)
    _dot_pre.pre_steps = _dot_pre.pre_orthos = 0
    x1 = np.full_like(b, np.nan)
    cd_solve.cd_solve(x1, b.copy(), _fwd_with_dots(amat), [_pre_with_dot(np.diag(amat))], _dot_pre(), _stop_after(12), cd_solve.tr_cg,
                      x_is_zero=True, x_uninit=True)
    assert _dot_pre.pre_steps == 0 and _dot_pre.pre_orthos == 0 and np.allclose(x1, ref, rtol=1e-12, atol=1e-14)
