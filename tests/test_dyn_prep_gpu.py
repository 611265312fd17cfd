"""pols_recursive_least_squares / pols_rolling_least_squares on RAW columns: the sqrt(w) scaling, the ones column, the null
policy's validity mask, the zero-filling and the 1 / sqrt(w) un-scaling of the predictions that the reference does around the
plugin call (polars_ols/least_squares.py:184-196, 234-235; src/expressions.rs:201-228, 593-701) now happen on the device behind
the C-ABI (csrc/dyn_prep.hip).  Expectation: the same steps in numpy around the oracle's solvers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import orc  # noqa: E402


@pytest.fixture(scope="module")
def eng():
    from polars_ols_amd import Engine

    e = Engine(0)
    yield e
    e.close()


def _frame(seed, dtype, k, n_groups=7, lo=60, hi=400, nulls=True, null_weights=True):
    rng = np.random.default_rng(seed)
    sizes = rng.integers(lo, hi, size=n_groups)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(offs[-1])
    cols = [rng.standard_normal(n) for _ in range(k)]
    y = sum((j + 1) * 0.3 * c for j, c in enumerate(cols)) + 0.7 + 0.1 * rng.standard_normal(n)
    w = rng.uniform(0.5, 2.0, size=n)
    if nulls:
        y[rng.random(n) < 0.03] = np.nan
        for c in cols[: max(1, k // 2)]:
            c[rng.random(n) < 0.02] = np.nan
        if null_weights:
            w[rng.random(n) < 0.02] = np.nan
    return y.astype(dtype), [c.astype(dtype) for c in cols], offs, w.astype(dtype)


def _expected(kind, y, cols, offs, w, icpt, policy, **kw):
    """ls.py:184-196 + ex.rs:201-228 + the plugin body + ls.py:234-235, in float64."""
    y, cols = y.astype(np.float64), [c.astype(np.float64) for c in cols]
    if icpt:
        cols = cols + [np.ones_like(y)]
    sw = None
    if w is not None:
        sw = np.sqrt(np.where(np.isnan(w), 1e-24, w.astype(np.float64)))
        y, cols = y * sw, [c * sw for c in cols]
    if policy in ("drop", "drop_zero", "drop_window"):
        valid = ~np.isnan(y)
        for c in cols:
            valid &= ~np.isnan(c)
    elif policy == "drop_y_zero_x":
        valid = ~np.isnan(y)
    else:
        valid = None
    y0, cols0 = np.nan_to_num(y), [np.nan_to_num(c) for c in cols]
    if kind == "rls":
        ref = orc.batched_rls(y0, cols0, offs, is_valid=valid, **kw)
    else:
        ref = orc.batched_rolling(y0, cols0, offs, null_policy=policy, is_valid=valid, **kw)
    pred = ref["pred"]
    if valid is not None:
        pred = np.where(valid, pred, np.nan)
    if sw is not None:
        pred = pred * (1.0 / sw)
    return ref["coef"], pred


def _run(eng, kind, y, cols, offs, w, icpt, policy, device, **kw):
    import torch

    if device:
        t = lambda a: None if a is None else torch.from_numpy(a).cuda()  # noqa: E731
        yy, cc, ww = t(y), [t(c) for c in cols], t(w)
    else:
        yy, cc, ww = y, cols, w
    fn = eng.recursive_least_squares if kind == "rls" else eng.rolling_least_squares
    out = fn(yy, cc, offs, weights=ww, add_intercept=icpt, null_policy=policy, **kw)
    eng.synchronize()
    get = lambda a: a.cpu().numpy() if device else np.asarray(a)  # noqa: E731
    return get(out["coef"]).astype(np.float64), get(out["pred"]).astype(np.float64)


def _close(a, b, tol):
    return np.allclose(a, b, rtol=tol, atol=tol, equal_nan=True)


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("policy", ["drop", "zero", "drop_y_zero_x", "ignore"])
@pytest.mark.parametrize("weights,icpt", [(True, True), (True, False), (False, True), (False, False)])
def test_rls_raw_columns_vs_oracle(eng, dtype, tol, policy, weights, icpt):
    y, cols, offs, w = _frame(3, dtype, 3)
    w = w if weights else None
    kw = dict(half_life=30.0, initial_state_covariance=10.0)
    coef, pred = _expected("rls", y, cols, offs, w, icpt, policy, **kw)
    for device in (True, False):
        c, p = _run(eng, "rls", y, cols, offs, w, icpt, policy, device, **kw)
        assert c.shape == coef.shape
        assert _close(c, coef, tol), float(np.nanmax(np.abs(c - coef)))
        assert _close(p, pred, tol), float(np.nanmax(np.abs(p - pred)))
        assert np.array_equal(np.isnan(p), np.isnan(pred))


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("policy", ["drop", "drop_window", "zero", "drop_y_zero_x"])
@pytest.mark.parametrize("weights,icpt", [(True, True), (False, True), (True, False)])
def test_rolling_raw_columns_vs_oracle(eng, dtype, tol, policy, weights, icpt):
    y, cols, offs, w = _frame(5, dtype, 3, null_weights=False)
    w = w if weights else None
    kw = dict(window_size=40, min_periods=8)
    coef, pred = _expected("rolling", y, cols, offs, w, icpt, policy, **kw)
    for device in (True, False):
        c, p = _run(eng, "rolling", y, cols, offs, w, icpt, policy, device, **kw)
        assert c.shape == coef.shape
        assert np.array_equal(np.isnan(c), np.isnan(coef))
        assert _close(c, coef, tol), float(np.nanmax(np.abs(c - coef)))
        assert _close(p, pred, tol), float(np.nanmax(np.abs(p - pred)))


@pytest.mark.parametrize("kind,kw", [("rls", dict(half_life=None, initial_state_covariance=100.0)),
                                     ("rolling", dict(window_size=120, min_periods=45))])
@pytest.mark.parametrize("k", [20, 40])
def test_wide_dynamic_with_intercept_and_weights(eng, kind, kw, k):
    """k + intercept beyond the fixed kernel-argument column arrays (pointer-table kernels): the rewritten columns feed them too."""
    y, cols, offs, w = _frame(11, np.float64, k, n_groups=3, lo=300, hi=500)
    coef, pred = _expected(kind, y, cols, offs, w, True, "drop", **kw)
    c, p = _run(eng, kind, y, cols, offs, w, True, "drop", True, **kw)
    assert c.shape[1] == k + 1
    assert _close(c, coef, 1e-6), float(np.nanmax(np.abs(c - coef)))
    assert _close(p, pred, 1e-5), float(np.nanmax(np.abs(p - pred)))


def test_no_nulls_keeps_the_callers_columns_and_cached_tables(eng):
    """No weights, no intercept, no NaN: nothing is rewritten (the kernels read the caller's columns) and the mask-free chunk
    tables are reused across calls; results equal the run with an explicit all-ones validity column bit for bit."""
    import torch

    y, cols, offs, _ = _frame(7, np.float64, 4, nulls=False)
    t = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    yy, cc = t(y), [t(c) for c in cols]
    a = eng.rolling_least_squares(yy, cc, offs, window_size=30, min_periods=5, null_policy="drop")
    b = eng.rolling_least_squares(yy, cc, offs, window_size=30, min_periods=5, null_policy="drop")
    eng.synchronize()
    assert torch.equal(a["coef"].nan_to_num(7.0), b["coef"].nan_to_num(7.0)) and torch.equal(a["pred"].nan_to_num(7.0), b["pred"].nan_to_num(7.0))
    ref = orc.batched_rolling(y, cols, offs, 30, min_periods=5)
    assert _close(a["coef"].cpu().numpy(), ref["coef"], 1e-6)


def test_initial_state_mean_covers_the_intercept(eng):
    """initial_state_mean has kt = n_features + 1 entries when the entry appends the ones column itself."""
    y, cols, offs, w = _frame(9, np.float64, 2, nulls=False)
    mean = np.array([0.3, 0.6, 0.7])
    kw = dict(half_life=None, initial_state_covariance=1.0, initial_state_mean=mean)
    coef, pred = _expected("rls", y, cols, offs, None, True, "drop", **kw)
    c, p = _run(eng, "rls", y, cols, offs, None, True, "drop", True, **kw)
    assert _close(c, coef, 1e-8) and _close(p, pred, 1e-8)


def test_null_free_hint_skips_the_policy_work(eng):
    """pols_batch.null_free (what Arrow's null_count == 0 tells a caller for free): the static entry takes its policy-free
    kernel under a drop policy, the dynamic entries skip the validity scan; results are those of the scanned run bit for bit."""
    import torch

    y, cols, offs, _ = _frame(13, np.float32, 4, n_groups=50, lo=200, hi=300, nulls=False)
    t = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    yy, cc = t(y), [t(c) for c in cols]
    a = eng.least_squares(yy, cc, offs, null_policy="drop", want=("coef", "pred"))
    assert eng.last_kernel.endswith("_nulls"), eng.last_kernel
    b = eng.least_squares(yy, cc, offs, null_policy="drop", want=("coef", "pred"), null_free=True)
    assert not eng.last_kernel.endswith("_nulls"), eng.last_kernel
    eng.synchronize()
    assert torch.allclose(a["coef"], b["coef"], rtol=1e-6, atol=1e-6) and torch.allclose(a["pred"], b["pred"], rtol=1e-5, atol=1e-5)
    for fn, kw in ((eng.recursive_least_squares, dict(half_life=10.0)), (eng.rolling_least_squares, dict(window_size=25, min_periods=6))):
        a = fn(yy, cc, offs, null_policy="drop", add_intercept=True, **kw)
        b = fn(yy, cc, offs, null_policy="drop", add_intercept=True, null_free=True, **kw)
        eng.synchronize()
        assert torch.equal(a["coef"].nan_to_num(7.0), b["coef"].nan_to_num(7.0)) and torch.equal(a["pred"].nan_to_num(7.0), b["pred"].nan_to_num(7.0))


@pytest.mark.parametrize("kind,kw", [("rls", dict(half_life=15.0)), ("rolling", dict(window_size=25, min_periods=6))])
@pytest.mark.parametrize("device", [False, True])
def test_callers_validity_bytes_with_nans_left_in_the_columns(eng, kind, kw, device):
    """The caller passes its own validity bytes, NO weights and NO intercept, and its columns still hold NaNs on the masked rows (what a
    null looks like after a cast): the entry zero-fills them like everywhere else -- a NaN must never reach the prefix sums."""
    import torch

    y, cols, offs, _ = _frame(21, np.float64, 3, n_groups=4, lo=80, hi=200, null_weights=False)
    valid = ~np.isnan(y)
    for c in cols:
        valid &= ~np.isnan(c)
    assert (~valid).sum() > 5
    coef, pred = _expected(kind, y, cols, offs, None, False, "drop", **kw)
    t = (lambda a: torch.from_numpy(a).cuda()) if device else (lambda a: a)  # noqa: E731
    fn = eng.recursive_least_squares if kind == "rls" else eng.rolling_least_squares
    out = fn(t(y), [t(c) for c in cols], offs, valid=t(valid.astype(np.uint8)), null_policy="drop", **kw)
    eng.synchronize()
    get = lambda a: (a.cpu().numpy() if device else np.asarray(a)).astype(np.float64)  # noqa: E731
    assert _close(get(out["coef"]), coef, 1e-6), float(np.nanmax(np.abs(get(out["coef"]) - coef)))
    assert _close(get(out["pred"]), pred, 1e-6)
