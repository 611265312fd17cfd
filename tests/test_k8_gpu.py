"""K8: static models for 32 .. 1024 columns (k8_wide.hip) through the C-ABI -- the reference's own wide cases:
tests/benchmark.py (10 000 x 100), test_elastic_net (100 .. 1 000 features), test_fit_wide (10 rows x up to 1 000)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import orc  # noqa: E402
from refdata import make_data  # noqa: E402


@pytest.fixture(scope="module")
def eng():
    from polars_ols_amd import Engine

    e = Engine(0)
    yield e
    e.close()


def _frame(seed, dtype, k, sizes, sparsity=0.0):
    rng = np.random.default_rng(seed)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(offs[-1])
    cols = [rng.normal(size=n).astype(dtype) for _ in range(k)]
    kk = max(1, int(k * (1 - sparsity)))
    y = (sum(c.astype(np.float64) for c in cols[:kk]) + 0.2 * rng.normal(size=n) + 0.3).astype(dtype)
    w = rng.uniform(0.2, 2.0, size=n).astype(dtype)
    return y, cols, offs, w


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("k,weights,icpt,alpha", [(32, False, False, 0.0), (40, True, True, 0.0), (100, False, True, 2.0), (130, True, False, 0.0)])
def test_wide_ols_ridge_vs_oracle(eng, dtype, tol, k, weights, icpt, alpha):
    y, cols, offs, w = _frame(k, dtype, k, [700, 0, 2_500, 333, 1_111])
    w = w if weights else None
    out = eng.least_squares(y, cols, offs, weights=w, add_intercept=icpt, alpha=alpha, want=("coef", "pred", "resid", "status"))
    assert eng.last_kernel.startswith("k8_wide_gram")
    ref = orc.batched_least_squares(y, cols, offs, weights=w, add_intercept=icpt, alpha=alpha)
    assert list(out["status"]) == [0, 2, 0, 0, 0]
    for key in ("coef", "pred", "resid"):
        assert np.allclose(out[key], ref[key], rtol=tol, atol=tol), (key, float(np.abs(out[key] - ref[key]).max()))


def test_reference_benchmark_shape(eng):                             # tests/benchmark.py:219 -- 10 000 rows x 100 features
    import torch

    d = make_data(n_samples=10_000, n_features=100)
    y = torch.from_numpy(d["y"]).cuda()
    cols = [torch.from_numpy(d[f"x{i + 1}"]).cuda() for i in range(100)]
    out = eng.least_squares(y, cols, np.array([0, 10_000], dtype=np.int64), want=("coef", "pred"))
    torch.cuda.synchronize()
    coef = np.linalg.lstsq(d["x"], d["y"], rcond=None)[0]
    assert np.allclose(out["coef"].cpu().numpy()[0], coef, rtol=1e-6, atol=1e-6)
    assert np.allclose(out["pred"].cpu().numpy(), d["x"] @ coef, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("k,sparsity,alpha,method", [(100, 0.5, 0.3, "cd"), (300, 0.5, 1.0, "cd"), (500, 0.9, 0.1, "cd"),
                                                     (1_000, 0.9, 0.3, "cd_active_set")])
def test_elastic_net_reference_cases(eng, k, sparsity, alpha, method):   # tests/test_ols.py:562-600
    d = make_data(n_features=k, sparsity=sparsity)
    cols = [d[f"x{i + 1}"] for i in range(k)]
    offs = np.array([0, len(d["y"])], dtype=np.int64)
    kw = dict(alpha=alpha, l1_ratio=0.5, max_iter=1_000, tol=1e-4, solve_method=method)
    out = eng.least_squares(d["y"], cols, offs, want=("coef", "pred", "status"), **kw)
    ref = orc.batched_least_squares(d["y"], cols, offs, **kw)
    assert int(out["status"][0]) == 0
    assert np.allclose(out["pred"], ref["pred"], rtol=1e-4, atol=1e-4)     # the reference's own tolerance vs sklearn
    assert np.allclose(out["coef"], ref["coef"], rtol=1e-4, atol=1e-4)


def test_elastic_net_wide_groups_tight(eng):
    y, cols, offs, w = _frame(3, np.float64, 48, [400, 900, 0, 650], sparsity=0.5)
    kw = dict(alpha=0.05, l1_ratio=0.7, positive=True, tol=1e-10, max_iter=20_000)
    out = eng.least_squares(y, cols, offs, weights=w, add_intercept=True, want=("coef", "pred"), **kw)
    ref = orc.batched_least_squares(y, cols, offs, weights=w, add_intercept=True, **kw)
    assert np.allclose(out["coef"], ref["coef"], rtol=1e-6, atol=1e-6) and np.allclose(out["pred"], ref["pred"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("k", [100, 1_000])
def test_fit_wide_reference_case(eng, k):                            # tests/test_ols.py:272-311 (10 rows, p >> n)
    d = make_data(n_samples=10, n_features=k, scale=1.0e-4)
    cols = [d[f"x{i + 1}"] for i in range(k)]
    offs = np.array([0, 10], dtype=np.int64)
    ols = eng.least_squares(d["y"], cols, offs, want=("coef", "status"))
    ridge = eng.least_squares(d["y"], cols, offs, want=("coef",), alpha=1e-5)
    lasso = eng.least_squares(d["y"], cols, offs, want=("coef",), alpha=1e-6, l1_ratio=1.0, tol=1e-8, max_iter=3_000)
    assert int(ols["status"][0]) == 1                               # flagged, solved by the minimum-norm path
    mn = np.linalg.lstsq(d["x"], d["y"], rcond=None)[0]
    assert np.allclose(ols["coef"][0], mn, rtol=1e-6, atol=1e-8)
    rr = np.linalg.solve(d["x"].T @ d["x"] + 1e-5 * np.eye(k), d["x"].T @ d["y"])
    assert np.allclose(ridge["coef"][0], rr, rtol=1e-5, atol=1e-7)
    for c in (ols["coef"][0], ridge["coef"][0], lasso["coef"][0]):   # the reference's assertion: predictions track y
        pred = eng.predict(cols, np.tile(c, (10, 1)))
        assert np.corrcoef(pred, d["y"])[0, 1] > 0.99


def test_wide_rank_deficient_tall_groups(eng):
    """Collinear columns in tall wide groups: flagged by the pivot test; default method and n > k -> the reference's pivoted QR, i.e.
    the basic solution (one twin carries both, the other is 0); solve_method="svd" -> the primal Jacobi pass, minimum norm (the twins
    share the weight); healthy groups untouched."""
    y, cols, offs, _ = _frame(21, np.float64, 50, [400, 900, 650])
    s, e = offs[1], offs[2]
    cols[7][s:e] = cols[3][s:e]                                      # group 1: two identical columns
    out = eng.least_squares(y, cols, offs, want=("coef", "pred", "status"))
    svd = eng.least_squares(y, cols, offs, solve_method="svd", want=("coef", "pred", "status"))
    ref = orc.batched_least_squares(y, cols, offs)
    assert list(out["status"]) == [0, 1, 0] and list(svd["status"]) == [0, 1, 0]
    assert np.allclose(out["coef"], ref["coef"], rtol=1e-6, atol=1e-6) and np.allclose(out["pred"], ref["pred"], rtol=1e-6, atol=1e-6)
    for g in range(3):
        a, b = offs[g], offs[g + 1]
        X = np.column_stack([c[a:b] for c in cols])
        exp = np.linalg.lstsq(X, y[a:b], rcond=None)[0]
        assert np.allclose(out["pred"][a:b], X @ exp, rtol=1e-6, atol=1e-6) and np.allclose(svd["pred"][a:b], X @ exp, rtol=1e-6, atol=1e-6)
        assert np.allclose(svd["coef"][g], exp, rtol=1e-5, atol=1e-6)
    assert (out["coef"][1][3] == 0.0) != (out["coef"][1][7] == 0.0)


def test_wide_device_matches_host_and_is_repeatable(eng):
    import torch

    y, cols, offs, w = _frame(9, np.float32, 70, [1_500, 800])
    host = eng.least_squares(y, cols, offs, weights=w, add_intercept=True, want=("coef", "pred"))
    for _ in range(2):
        dev = eng.least_squares(torch.from_numpy(y).cuda(), [torch.from_numpy(c).cuda() for c in cols], offs,
                                weights=torch.from_numpy(w).cuda(), add_intercept=True, want=("coef", "pred"))
        torch.cuda.synchronize()
        assert np.array_equal(host["coef"], dev["coef"].cpu().numpy()) and np.array_equal(host["pred"], dev["pred"].cpu().numpy())


def test_wide_limits(eng):
    """every entry takes up to 1 024 columns (POLS_MAX_FEATURES_STATIC / _DYNAMIC / _STATISTICS); one more is refused"""
    y2, cols2, offs2, _ = _frame(1, np.float64, 8, [40])
    cols2 = cols2 * 129                                               # 1 032 column pointers
    for call in (eng.recursive_least_squares, eng.least_squares, eng.least_squares_statistics):
        with pytest.raises(Exception, match="features|columns"):
            call(y2, cols2[:1025], offs2)
    with pytest.raises(Exception, match="features|columns"):
        eng.rolling_least_squares(y2, cols2[:1025], offs2, window_size=10)


# ------------------------------------------------------------------------------------------------ multi-target
def _mt_expected(ys, cols, offs, w, icpt, alpha):
    m, k = len(ys), len(cols) + int(icpt)
    G = len(offs) - 1
    coef = np.zeros((G, m, k))
    preds = [np.full(len(ys[0]), np.nan) for _ in range(m)]
    for g in range(G):
        s, e = offs[g], offs[g + 1]
        if e == s:
            continue
        X = np.column_stack([c[s:e] for c in cols]).astype(np.float64)
        if icpt:
            X = np.column_stack([X, np.ones(e - s)])
        sw = np.sqrt(w[s:e].astype(np.float64)) if w is not None else np.ones(e - s)
        Xs = X * sw[:, None]
        for t in range(m):
            yt = ys[t][s:e].astype(np.float64) * sw
            beta = np.linalg.solve(Xs.T @ Xs + alpha * np.eye(k), Xs.T @ yt) if alpha > 0 else np.linalg.lstsq(Xs, yt, rcond=None)[0]
            coef[g, t] = beta
            preds[t][s:e] = (Xs @ beta) / sw
    return coef, preds


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("k,m,weights,icpt,alpha", [(3, 2, False, False, 0.0), (8, 5, True, True, 0.1), (40, 3, False, True, 0.0)])
def test_multi_target_vs_numpy(eng, dtype, tol, k, m, weights, icpt, alpha):
    y0, cols, offs, w = _frame(50 + k, dtype, k, [500, 0, 1_300, 77])
    rng = np.random.default_rng(m)
    ys = [y0] + [(sum(rng.normal() * c.astype(np.float64) for c in cols) + 0.2 * rng.normal(size=len(y0))).astype(dtype) for _ in range(m - 1)]
    w = w if weights else None
    out = eng.multi_target_least_squares(ys, cols, offs, weights=w, add_intercept=icpt, alpha=alpha)
    coef, preds = _mt_expected(ys, cols, offs, w, icpt, alpha)
    assert list(out["status"]) == [0, 2, 0, 0]
    assert np.allclose(out["coef"], coef, rtol=tol, atol=tol), float(np.abs(out["coef"] - coef).max())
    for t in range(m):
        assert np.allclose(out["pred"][t], preds[t], rtol=tol, atol=tol, equal_nan=True), t


def _mt_oracle(ys, cols, offs, w, icpt, alpha, rcond=None):
    """The plugin body around solve_multi_target (ex.rs:521-591 with the Python pre-scaling, ls.py:190-196): per group, the
    sqrt(w)-scaled [X | 1] and the n x m target matrix go to the oracle's restatement of ls.rs:243-260."""
    from oracle import orc

    m, k = len(ys), len(cols) + int(icpt)
    G = len(offs) - 1
    coef = np.zeros((G, m, k))
    preds = [np.full(len(ys[0]), np.nan) for _ in range(m)]
    for g in range(G):
        s, e = offs[g], offs[g + 1]
        if e == s:
            continue
        X = np.column_stack([c[s:e] for c in cols]).astype(np.float64)
        if icpt:
            X = np.column_stack([X, np.ones(e - s)])
        sw = np.sqrt(w[s:e].astype(np.float64)) if w is not None else np.ones(e - s)
        Xs = X * sw[:, None]
        Y = np.column_stack([ys[t][s:e].astype(np.float64) * sw for t in range(m)])
        B = orc.solve_multi_target(Y, Xs, alpha=alpha, rcond=rcond)          # k x m
        coef[g] = B.T
        for t in range(m):
            preds[t][s:e] = (Xs @ B[:, t]) / sw
    return coef, preds


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("k,m,weights,icpt,alpha", [(3, 2, False, False, 0.0), (8, 5, True, True, 0.1), (40, 3, False, True, 0.0),
                                                    (12, 4, True, False, 3.0)])
def test_multi_target_vs_oracle(eng, dtype, tol, k, m, weights, icpt, alpha):
    """pols_multi_target_least_squares against the oracle's solve_multi_target (SVD form, ls.rs:243-260) on every group."""
    y0, cols, offs, w = _frame(70 + k, dtype, k, [500, 0, 1_300, 77])
    rng = np.random.default_rng(m)
    ys = [y0] + [(sum(rng.normal() * c.astype(np.float64) for c in cols) + 0.2 * rng.normal(size=len(y0))).astype(dtype) for _ in range(m - 1)]
    w = w if weights else None
    out = eng.multi_target_least_squares(ys, cols, offs, weights=w, add_intercept=icpt, alpha=alpha)
    coef, preds = _mt_oracle(ys, cols, offs, w, icpt, alpha)
    assert list(out["status"]) == [0, 2, 0, 0]
    assert np.allclose(out["coef"], coef, rtol=tol, atol=tol), float(np.abs(out["coef"] - coef).max())
    for t in range(m):
        assert np.allclose(out["pred"][t], preds[t], rtol=tol, atol=tol, equal_nan=True), t


def test_multi_target_reference_benchmark_shape(eng):               # tests/benchmark.py:220 -- 10 000 x 100 features, 20 targets
    import torch

    d = make_data(n_samples=10_000, n_features=100)
    rng = np.random.default_rng(1)
    B = rng.normal(size=(100, 20))
    Y = d["x"] @ B + 0.1 * rng.normal(size=(10_000, 20))
    ys = [torch.from_numpy(np.ascontiguousarray(Y[:, t])).cuda() for t in range(20)]
    cols = [torch.from_numpy(d[f"x{i + 1}"]).cuda() for i in range(100)]
    out = eng.multi_target_least_squares(ys, cols, np.array([0, 10_000], dtype=np.int64))
    torch.cuda.synchronize()
    exp = np.linalg.lstsq(d["x"], Y, rcond=None)[0]                  # [k, m]
    assert np.allclose(out["coef"].cpu().numpy()[0], exp.T, rtol=1e-6, atol=1e-6)
    for t in range(20):
        assert np.allclose(out["pred"][t].cpu().numpy(), d["x"] @ exp[:, t], rtol=1e-6, atol=1e-6)


def test_multi_target_rank_deficient_group_and_panics(eng):
    y0, cols, offs, _ = _frame(5, np.float64, 6, [300, 4, 200])       # group 1: 4 rows, 6 features -> minimum norm
    ys = [y0, (y0 * 0.5 - cols[0]).astype(np.float64)]
    out = eng.multi_target_least_squares(ys, cols, offs)
    assert list(out["status"]) == [0, 1, 0]
    coef, preds = _mt_expected(ys, cols, offs, None, False, 0.0)
    assert np.allclose(out["coef"], coef, rtol=1e-6, atol=1e-7)
    for t in range(2):
        assert np.allclose(out["pred"][t], preds[t], rtol=1e-6, atol=1e-7)
    with pytest.raises(Exception):
        eng.multi_target_least_squares(ys, cols, offs, solve_method="chol")


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("policy", ["drop", "zero", "drop_y_zero_x", "drop_zero", "ignore"])
@pytest.mark.parametrize("k,m,weights,icpt,alpha", [(3, 2, False, False, 0.0), (5, 3, True, True, 0.7), (40, 5, False, True, 0.0)])
def test_multi_target_null_policies_behind_the_cabi(eng, dtype, tol, policy, k, m, weights, icpt, alpha):
    """The plugin body under a null policy (src/expressions.rs:521-591) inside pols_multi_target_least_squares: the joint mask over
    every target (and, unless drop_y_zero_x, every feature), the fit on the rows it leaves, predictions for EVERY row from the
    zero-filled features, "drop" masked.  Expectation: those steps in numpy around solve_multi_target's oracle.  Host and device."""
    import torch

    rng = np.random.default_rng(31 * k + m)
    sizes = rng.integers(6 * (k + 1), 12 * (k + 1), size=9)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(offs[-1])
    cols = [rng.standard_normal(n) for _ in range(k)]
    ys = [sum((j + t + 1) * 0.2 * c for j, c in enumerate(cols)) + 0.3 * t + 0.1 * rng.standard_normal(n) for t in range(m)]
    w = rng.uniform(0.5, 2.0, size=n) if weights else None
    for y in ys[:2]:
        y[rng.random(n) < 0.04] = np.nan
    for c in cols[:2]:
        c[rng.random(n) < 0.03] = np.nan
    ys, cols = [y.astype(dtype) for y in ys], [c.astype(dtype) for c in cols]
    w = None if w is None else w.astype(dtype)
    # numpy expectation
    Y = np.column_stack(ys).astype(np.float64)
    X = np.column_stack(cols + ([np.ones(n, dtype=dtype)] if icpt else [])).astype(np.float64)
    sw = np.ones(n) if w is None else np.sqrt(w.astype(np.float64))
    ynull, xnull = np.isnan(Y).any(axis=1), np.isnan(X).any(axis=1)
    keep = ~ynull & ~xnull if policy in ("drop", "drop_zero") else (~ynull if policy == "drop_y_zero_x" else np.ones(n, dtype=bool))
    Xz = np.nan_to_num(X)
    exp = np.full((n, m), np.nan)
    for g in range(len(offs) - 1):
        s, e = offs[g], offs[g + 1]
        kk = keep[s:e]
        # "ignore" zero-fills too: both arrays come from construct_features_array(.., true) (ex.rs:546-547)
        Xf = (np.nan_to_num(X[s:e]) if policy in ("zero", "drop_y_zero_x", "ignore") else X[s:e])[kk] * sw[s:e][kk, None]
        Yf = (np.nan_to_num(Y[s:e]) if policy in ("zero", "ignore") else Y[s:e])[kk] * sw[s:e][kk, None]
        B = orc.solve_multi_target(Yf, Xf, alpha=alpha)                             # kt x m
        exp[s:e] = Xz[s:e] @ B
    if policy == "drop":
        exp[~keep] = np.nan
    kw = dict(weights=w, add_intercept=icpt, alpha=alpha, null_policy=policy, want=("pred", "coef"))
    host = eng.multi_target_least_squares(ys, cols, offs, **kw)
    t = lambda a: None if a is None else torch.from_numpy(a).cuda()  # noqa: E731
    dev = eng.multi_target_least_squares([t(y) for y in ys], [t(c) for c in cols], offs, **dict(kw, weights=t(w)))
    eng.synchronize()
    for res in (host, {"pred": [p.cpu().numpy() for p in dev["pred"]]}):
        got = np.column_stack([np.asarray(p, dtype=np.float64) for p in res["pred"]])
        assert np.array_equal(np.isnan(got), np.isnan(exp))
        assert np.allclose(got, exp, rtol=tol, atol=tol, equal_nan=True), float(np.nanmax(np.abs(got - exp)))


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-7), (np.float32, 1e-4)])
@pytest.mark.parametrize("k", [40, 150])
def test_wide_rank_deficient_groups_get_the_reference_solver(eng, dtype, tol, k):
    """Beyond 31 columns too, a flagged group is re-solved by what the REFERENCE runs for the (branch, method): default / "qr" with
    n > k -> pivoted QR, basic solution (a duplicated column: ONE twin carries the sum, the other is exactly 0; demo notebook cell 28);
    "svd" -> dgelsd's minimum norm (cell 32); "chol" / "lu" -> Cholesky fails, LU divides by an exactly zero pivot -> NaN (cell 30)."""
    y, cols, offs, _ = _frame(7 * k, dtype, k, [3 * k + 17, 2 * k + 5, 4 * k])
    s, e = offs[1], offs[2]
    cols[k - 3][s:e] = cols[5][s:e]                                  # group 1: two identical columns
    for m in (None, "qr"):
        out = eng.least_squares(y, cols, offs, solve_method=m, want=("coef", "pred", "status"))
        assert eng.last_kernel.startswith("k8_wide")
        ref = orc.batched_least_squares(y, cols, offs, solve_method=m)
        assert list(out["status"]) == [0, 1, 0]
        assert np.allclose(out["coef"], ref["coef"], rtol=tol, atol=tol), (m, float(np.abs(out["coef"] - ref["coef"]).max()))
        assert np.allclose(out["pred"], ref["pred"], rtol=tol, atol=tol)
        assert (out["coef"][1][5] == 0.0) != (out["coef"][1][k - 3] == 0.0)
    out = eng.least_squares(y, cols, offs, solve_method="svd", want=("coef",))
    X = np.column_stack([c[s:e].astype(np.float64) for c in cols])
    mn = np.linalg.lstsq(X, y[s:e].astype(np.float64), rcond=None)[0]
    assert np.allclose(out["coef"][1], mn, rtol=10 * tol, atol=10 * tol)
    assert abs(out["coef"][1][5] - out["coef"][1][k - 3]) < 10 * tol  # minimum norm: the twins share the coefficient
    for m in ("chol", "lu"):
        out = eng.least_squares(y, cols, offs, solve_method=m, want=("coef", "pred", "status"))
        ref = orc.batched_least_squares(y, cols, offs, solve_method=m)
        assert np.isnan(out["coef"][1]).all() and np.isnan(ref["coef"][1]).all(), m
        assert np.isnan(out["pred"][s:e]).all()
        for g in (0, 2):
            assert np.allclose(out["coef"][g], ref["coef"][g], rtol=tol, atol=tol)
