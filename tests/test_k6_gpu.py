"""GPU tests of K6, the SVD fallback for groups the fused kernels flag: rank-deficient / wide (n <= k) / ill-conditioned /
NaN groups.  Expected values: numpy's lstsq (LAPACK dgelsd -- what the reference's solve_ols_svd calls on linux,
src/least_squares.rs:183-191) and the CPU oracle; cases from reference tests/test_ols.py:272-360."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from polars_ols_amd import Engine

    e = Engine(0)
    yield e
    e.close()


def _cuda(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _np(t):
    return t.double().cpu().numpy() if hasattr(t, "cpu") else np.asarray(t, dtype=np.float64)


@pytest.mark.parametrize("k", [2, 10, 16, 31])
def test_fit_wide_min_norm(eng, k):                                  # tests/test_ols.py:272-312: n = 10 rows, k features
    from refdata import make_data

    d = make_data(n_samples=10, n_features=k, scale=1e-4)
    cols = [np.ascontiguousarray(d["x"][:, j]) for j in range(k)]
    out = eng.least_squares(d["y"], cols, [0, 10], want=("coef", "pred", "status"))
    exp = np.linalg.lstsq(d["x"], d["y"], rcond=None)[0]
    assert np.allclose(out["coef"][0], exp, rtol=1e-6, atol=1e-8)
    assert np.corrcoef(out["pred"], d["y"])[0, 1] == pytest.approx(1.0, rel=1e-5, abs=1e-5)
    assert out["status"][0] == (1 if k > 10 else 0)                 # n < k -> X'X is singular -> fallback taken (n == k: still PD)


@pytest.mark.parametrize("n_features,solve_method", [(10, "svd"), (30, "svd"), (10, "qr"), (10, None),
                                                     (99, "svd"), (1_000, "svd"), (90, "qr")])   # the reference's own four + small ones
def test_fit_multi_collinear(eng, n_features, solve_method):         # tests/test_ols.py:315-360
    from refdata import make_data

    d = make_data(n_samples=100, n_features=n_features, scale=1e-4)
    x = np.column_stack([d["x"], d["x"][:, -1] + 1e-12])
    cols = [np.ascontiguousarray(x[:, j]) for j in range(x.shape[1])]
    out = eng.least_squares(d["y"], cols, [0, 100], solve_method=solve_method, rcond=1e-16, want=("coef", "status"))
    coef = out["coef"][0]
    exp = np.linalg.lstsq(x, d["y"], rcond=1e-16)[0]
    assert out["status"][0] == 1 and np.isfinite(coef).all()
    assert np.allclose(x @ coef, x @ exp, rtol=1e-4, atol=1e-4)
    if solve_method == "svd":
        assert np.allclose(coef, exp, rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_degenerate_groups_do_not_disturb_healthy_ones(eng, dtype):
    from oracle import orc

    rng = np.random.default_rng(0)
    sizes = np.array([400, 5, 300, 250, 3, 500])
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N, k = int(offs[-1]), 6
    cols = [rng.standard_normal(N).astype(dtype) for _ in range(k)]
    y = (sum(c.astype(np.float64) for c in cols) + 0.1 * rng.standard_normal(N)).astype(dtype)
    cols[2][offs[2]:offs[3]] = 0.0                                   # group 2: an all-zero feature (rank deficient)
    cols[4][offs[3]:offs[4]] = cols[3][offs[3]:offs[4]]              # group 3: two identical features
    yd = _cuda(y); cd = [_cuda(c) for c in cols]
    out = eng.least_squares(yd, cd, offs, want=("coef", "pred", "status"))
    st = _np(out["status"]).astype(int)
    assert list(st) == [0, 1, 1, 1, 1, 0]                            # groups 1 and 4 have n < k
    tol = 1e-6 if dtype == np.float64 else 1e-4
    coef, pred = _np(out["coef"]), _np(out["pred"])
    # EVERY group against the oracle's dispatch (ls.rs:224-231): more rows than columns -> the pivoted QR, whose answer on a
    # rank-deficient group is the BASIC solution (dependent / zero columns get coefficient 0: the reference's notebook prints
    # {1.0, 2.0, -0.0} for it, cell 28); n <= k -> dgelsd's minimum-norm answer
    for g in range(6):
        sl = slice(offs[g], offs[g + 1])
        ref = orc.batched_least_squares(y[sl], [c[sl] for c in cols], [0, sizes[g]])
        assert np.allclose(coef[g], ref["coef"][0], rtol=tol, atol=tol), g
        assert np.allclose(pred[sl], ref["pred"], rtol=tol, atol=tol), g
    assert coef[2][2] == 0.0 and coef[3][4] == 0.0                   # the all-zero column; the SECOND of the two identical columns
    x3 = np.column_stack([c[offs[3]:offs[4]].astype(np.float64) for c in cols])
    mn = np.linalg.lstsq(x3, y[offs[3]:offs[4]].astype(np.float64), rcond=None)[0]
    assert not np.allclose(coef[3], mn, rtol=1e-3, atol=1e-3)        # ... which is NOT the minimum-norm solution ...
    svd = eng.least_squares(yd, cd, offs, solve_method="svd", want=("coef",))
    assert np.allclose(_np(svd["coef"])[3], mn, rtol=tol, atol=tol)  # ... that only solve_method="svd" returns (notebook cell 32)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("engine", [None, "k2", "stream"])
def test_notebook_collinear_frame_every_method(eng, dtype, engine):
    """The reference's own rank-deficient vectors (notebooks/polars_ols_demo.ipynb cells 26-34): x3 an exact copy of x2, y = x1 + x2 + x3.
    "qr" / default -> {1, 2, -0}; "svd" -> {1, 1, 1}; "chol" / "lu" -> Cholesky fails, LU divides by an exactly zero pivot -> nulls.
    Through every static engine (K1, K2, the streamed three-launch path), several groups at once."""
    from refdata import notebook_make_data

    d = notebook_make_data(n_samples=2_000, n_features=3, n_groups=5)
    n = 500
    cols, ys = [[], [], []], []
    for g in range(4):                                               # four groups of 500 rows of the frame
        sl = slice(g * n, (g + 1) * n)
        x1, x2 = d["x1"][sl].astype(dtype), d["x2"][sl].astype(dtype)
        for j, c in enumerate((x1, x2, x2)):
            cols[j].append(c)
        ys.append(((x1 + x2) + x2).astype(dtype))
    cols = [np.concatenate(c) for c in cols]
    y = np.concatenate(ys)
    offs = np.arange(5, dtype=np.int64) * n
    tol = 1e-9 if dtype == np.float64 else 1e-4
    eng.set_option("STATIC_ENGINE", engine)
    try:
        for m, exp in (("qr", [1.0, 2.0, 0.0]), (None, [1.0, 2.0, 0.0]), ("svd", [1.0, 1.0, 1.0])):
            out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, solve_method=m, want=("coef", "pred", "status"))
            assert np.allclose(_np(out["coef"]), np.tile(exp, (4, 1)), rtol=tol, atol=tol), (m, _np(out["coef"]))
            assert np.allclose(_np(out["pred"]), y, rtol=tol, atol=tol)
            assert (_np(out["status"]) == 1).all()
        for m in ("chol", "lu"):
            out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, solve_method=m, want=("coef", "pred"))
            assert np.isnan(_np(out["coef"])).all(), (m, _np(out["coef"]))
            assert np.isnan(_np(out["pred"])).all()
    finally:
        eng.set_option("STATIC_ENGINE", None)


def test_nan_group_gives_nan_like_reference(eng):
    """null_policy='ignore' turns a null into NaN (src/expressions.rs:53); the reference's QR then returns NaN
    coefficients for that group only."""
    rng = np.random.default_rng(1)
    offs = np.array([0, 200, 400], dtype=np.int64)
    cols = [rng.standard_normal(400) for _ in range(3)]
    y = sum(cols) + 0.1 * rng.standard_normal(400)
    cols[1][250] = np.nan
    out = eng.least_squares(y, cols, offs, want=("coef", "pred"))
    assert np.isfinite(out["coef"][0]).all() and np.isnan(out["coef"][1]).all()
    assert np.isfinite(out["pred"][:200]).all() and np.isnan(out["pred"][200:]).all()


def test_ill_conditioned_f32_group_is_rescued_in_f64(eng):
    """corr(x1, x2) = 0.99995: cond(X)^2 ~ 4e4 is beyond an f32 normal-equation solve; the pivot test sends the group
    to the f64 Jacobi fallback, which keeps the 1e-4 parity with the reference's (f64, QR) answer."""
    from oracle import orc

    rng = np.random.default_rng(2)
    n = 1000
    x1 = rng.standard_normal(n)
    x2 = x1 + 0.01 * rng.standard_normal(n)
    x3 = rng.standard_normal(n)
    y = x1 + x2 + x3 + 0.1 * rng.standard_normal(n)
    cols = [c.astype(np.float32) for c in (x1, x2, x3)]
    y32 = y.astype(np.float32)
    out = eng.least_squares(_cuda(y32), [_cuda(c) for c in cols], [0, n], want=("coef", "pred", "status"))
    ref = orc.batched_least_squares(y32, cols, [0, n])
    assert int(out["status"][0]) == 1
    assert np.allclose(_np(out["coef"]), ref["coef"], rtol=1e-4, atol=1e-4)
    assert np.allclose(_np(out["pred"]), ref["pred"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("k", [3, 12, 20, 40])
def test_f32_ridge_on_collinear_data_is_rescued_in_f64(eng, k):
    """Ridge on data with two identical columns, f32: cond(X'X + alpha I) * eps_f32 is beyond the 1e-4 tolerance, the pivot test flags
    the group and the fix-up pass runs the reference's own chain (Cholesky of X'X + alpha I, then LU) in f64 -- WITH the penalty on
    the diagonal.  Every engine width: K1, K2, K1w / K2w, K8."""
    from oracle import orc

    rng = np.random.default_rng(k)
    n = 900
    cols = [rng.standard_normal(n).astype(np.float32) for _ in range(k)]
    cols[1] = cols[0].copy()
    y = (sum(c.astype(np.float64) for c in cols) + 0.1 * rng.standard_normal(n)).astype(np.float32)
    offs = np.array([0, n], dtype=np.int64)
    out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, alpha=0.05, l1_ratio=0.0, want=("coef", "pred", "status"))
    ref = orc.batched_least_squares(y, cols, offs, alpha=0.05, l1_ratio=0.0)
    assert int(_np(out["status"])[0]) == 1
    assert np.allclose(_np(out["coef"]), ref["coef"], rtol=1e-4, atol=1e-4), float(np.abs(_np(out["coef"]) - ref["coef"]).max())
    assert np.allclose(_np(out["pred"]), ref["pred"], rtol=1e-4, atol=1e-4)


def test_ridge_on_collinear_data_needs_no_fallback(eng):
    from oracle import orc

    rng = np.random.default_rng(3)
    n = 300
    x1 = rng.standard_normal(n)
    cols = [x1, x1.copy(), rng.standard_normal(n)]
    y = x1 + cols[2] + 0.1 * rng.standard_normal(n)
    out = eng.least_squares(y, cols, [0, n], alpha=0.5, l1_ratio=0.0, want=("coef", "status"))
    ref = orc.batched_least_squares(y, cols, [0, n], alpha=0.5, l1_ratio=0.0)
    assert out["status"][0] == 0 and np.allclose(out["coef"], ref["coef"], rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("rcond", [0.1, 0.5, 1e-3])
def test_ridge_svd_honours_rcond_on_every_group(eng, dtype, rcond):
    """solve_ridge_svd (ls.rs:106-168): singular values below rcond * s_max are dropped (:143-148) on FULL-RANK groups too -- a
    truncated solve is not the normal-equation solution, so every group goes through the Jacobi-SVD kernel."""
    from oracle import orc

    rng = np.random.default_rng(7)
    sizes = np.array([300, 0, 41, 900, 7])
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N, k = int(offs[-1]), 5
    scale = np.array([1.0, 0.6, 0.3, 0.05, 2.0])                     # a spread of singular values: rcond cuts some of them
    cols = [(scale[j] * rng.standard_normal(N)).astype(dtype) for j in range(k)]
    y = (sum(c.astype(np.float64) for c in cols) + 0.1 * rng.standard_normal(N)).astype(dtype)
    kw = dict(alpha=0.5, l1_ratio=0.0, solve_method="svd", rcond=rcond)
    out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, want=("coef", "pred", "resid", "status"), **kw)
    assert eng.last_kernel == "k6_small_svd_all_groups"
    ref = orc.batched_least_squares(y, cols, offs, **kw)
    plain = orc.batched_least_squares(y, cols, offs, alpha=0.5, l1_ratio=0.0)
    if rcond >= 0.1:                                                 # the truncation really changes the answer
        assert np.abs(ref["coef"] - plain["coef"]).max() > 1e-2
    tol = 1e-6 if dtype == np.float64 else 1e-4
    st = _np(out["status"]).astype(int)
    assert list(st) == [1, 2, 1, 1, 1]                               # every non-empty group took the SVD pass
    for key in ("coef", "pred", "resid"):
        assert np.allclose(_np(out[key]), ref[key], rtol=tol, atol=tol), (key, float(np.abs(_np(out[key]) - ref[key]).max()))


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("k,rows,add_intercept,weights,method", [
    (8, (4, 4), False, False, None), (8, (1, 8), False, False, None), (8, (1, 9), True, True, None), (3, (1, 3), False, False, "svd"),
    (8, (2, 30), False, True, "svd"), (12, (1, 12), True, False, None), (16, (5, 16), False, False, None), (20, (1, 20), False, True, None),
    (31, (1, 31), False, False, None), (31, (20, 32), False, False, "svd"), (5, (1, 5), False, False, None),
    (31, (1, 32), True, True, None),              # 32 columns with the intercept: the wide path (K8) flags such groups by shape too (fuzz seed 101)
])
def test_short_groups_take_the_team_min_norm_solver(eng, dtype, tol, k, rows, add_intercept, weights, method):
    """K6s (k6s_small.hip): groups with no more rows than columns -- `solve_ols`'s own n <= k -> SVD branch (ls.rs:211-240,
    tests/test_ols.py:272-312) -- and short groups under "svd", a sub-wave team per group instead of K6's pool.  Every group against the
    oracle's dgelsd restatement; duplicated rows (rank below n) and an all-zero group included."""
    from oracle import orc

    rng = np.random.default_rng(k * 131 + rows[1])
    G = 3000
    sizes = rng.integers(rows[0], rows[1] + 1, size=G)
    sizes[5] = 0
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(offs[-1])
    cols = [rng.standard_normal(N).astype(dtype) for _ in range(k)]
    y = (sum(c.astype(np.float64) for c in cols) + 0.1 * rng.standard_normal(N)).astype(dtype)
    big = np.flatnonzero(sizes >= 2)
    g_dup, g_zero = int(big[0]), int(big[1])
    for c in cols:                                                    # a duplicated row: rank n - 1; an all-zero group: beta = 0
        c[offs[g_dup] + 1] = c[offs[g_dup]]
        c[offs[g_zero]:offs[g_zero + 1]] = 0
    y[offs[g_dup] + 1] = y[offs[g_dup]]
    w = (0.5 + rng.random(N)).astype(dtype) if weights else None
    kw = dict(add_intercept=add_intercept, solve_method=method)
    out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, weights=None if w is None else _cuda(w),
                            want=("coef", "pred", "resid", "status"), **kw)
    ref = orc.batched_least_squares(y, cols, offs, weights=w, **kw)
    kt = k + int(add_intercept)
    st = _np(out["status"]).astype(int)
    short = (sizes > 0) & (sizes < kt)
    bad = np.flatnonzero(short & (st != 1))
    assert bad.size == 0, (bad[:10], sizes[bad[:10]], st[bad[:10]])  # X'X singular -> fallback taken
    assert st[5] == 2                                                 # the empty group
    coef = _np(out["coef"])
    # the comparison scale: a minimum-norm solution of a 4 x 8 system is O(1); near-singular square groups (n == kt) are compared
    # through what they predict (the reference's own convention for ill-posed fits, tests/test_ols.py:355-360)
    loose = sizes >= kt
    assert np.allclose(coef[~loose], ref["coef"][~loose], rtol=tol, atol=tol), float(np.abs(coef[~loose] - ref["coef"][~loose]).max())
    pred, resid = _np(out["pred"]), _np(out["resid"])
    rowmask = np.repeat(~loose, sizes)
    assert np.allclose(pred[rowmask], ref["pred"][rowmask], rtol=tol, atol=tol)
    assert np.allclose(resid[rowmask], ref["resid"][rowmask], rtol=tol, atol=10 * tol if dtype == np.float32 else tol)
    wellposed = loose & (st == 0)
    if wellposed.any():
        rm = np.repeat(wellposed, sizes)
        assert np.allclose(pred[rm], ref["pred"][rm], rtol=1e-3, atol=1e-3)


def test_short_groups_full_size_min_norm(eng):
    """200 000 groups of 4 rows x 8 features (every one a minimum-norm problem) in one call, f32 and f64, every group
    against numpy's batched pinv (dgelsd's answer for full-row-rank groups); must take milliseconds, not the fix-up pool's minutes."""
    import time

    import torch

    G, n, k = 200_000, 4, 8
    for dt, tol in ((torch.float64, 1e-6), (torch.float32, 1e-4)):
        gen = torch.Generator(device="cuda").manual_seed(11)
        cols = [torch.randn(G * n, generator=gen, device="cuda", dtype=dt) for _ in range(k)]
        y = sum(cols) + 0.1 * torch.randn(G * n, generator=gen, device="cuda", dtype=dt)
        offs = np.arange(G + 1, dtype=np.int64) * n
        out = eng.least_squares(y, cols, offs, want=("coef", "pred", "status"))
        eng.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = eng.least_squares(y, cols, offs, want=("coef", "pred", "status"))
        eng.synchronize(); torch.cuda.synchronize()
        assert time.perf_counter() - t0 < 0.5
        X = torch.stack(cols, dim=1).double().reshape(G, n, k).cpu().numpy()
        Y = y.double().reshape(G, n).cpu().numpy()
        exp = np.einsum("gkn,gn->gk", np.linalg.pinv(X), Y)
        got = _np(out["coef"])
        assert (_np(out["status"]) == 1).all()
        assert np.allclose(got, exp, rtol=tol, atol=tol), float(np.abs(got - exp).max())
        assert np.allclose(_np(out["pred"]).reshape(G, n), Y, rtol=10 * tol, atol=10 * tol)     # n < k: the fit interpolates


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("method", [None, "svd"])
def test_short_groups_among_long_ones_use_their_own_team_size(eng, dtype, tol, method):
    """A frame of 1 000-row groups with hundreds of 3-, 6-, 12- and 20-row groups among them (assets that only just listed): K6s is launched once
    per team size the offsets hold (4 / 8 / 16 / 32 lanes), not once at the size of the largest short group -- 100 000 six-row groups among
    5 000 long ones took 1.83 ms in 32-lane teams, 0.2 ms now (profiles/r05_bench_shape_cliffs.txt).  Every short group against the oracle."""
    from oracle import orc

    rng = np.random.default_rng(11)
    k = 8
    sizes = np.array([1000] * 12 + [6] * 300 + [3] * 50 + [0] + [12] * 40 + [20] * 30 + [1000] * 3 + [7] * 9)
    rng.shuffle(sizes)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(offs[-1])
    cols = [rng.standard_normal(N).astype(dtype) for _ in range(k)]
    y = (sum(c.astype(np.float64) for c in cols) + 0.1 * rng.standard_normal(N)).astype(dtype)
    out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, want=("coef", "pred", "status"), solve_method=method)
    ref = orc.batched_least_squares(y, cols, offs, solve_method=method)
    st = _np(out["status"]).astype(int)
    assert (st[sizes == 0] == 2).all() and (st[(sizes > 0) & (sizes < k)] == 1).all() and (st[sizes == 1000] == 0).all()
    coef = _np(out["coef"])
    assert np.allclose(coef, ref["coef"], rtol=10 * tol, atol=10 * tol), float(np.abs(coef - ref["coef"]).max())
    short = sizes < k
    assert np.allclose(coef[short], ref["coef"][short], rtol=tol, atol=tol), float(np.abs(coef[short] - ref["coef"][short]).max())
    assert np.allclose(_np(out["pred"]), ref["pred"], rtol=10 * tol, atol=10 * tol)
