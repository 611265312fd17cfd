"""The Arrow C Data Interface entries of the other seven plugin functions (src/expressions.rs:448-741) through pyarrow:
least_squares_statistics, multi_target_least_squares (STRUCT target in, struct of predictions out), recursive / rolling least squares
(predictions and the per-row coefficients struct) and predict (coefficients STRUCT in).  Columns as Polars holds them: validity
bitmaps, sliced arrays (offset != 0), several chunks, struct-level nulls.  Expectations: the CPU oracle on the same data with
NaN for null, and numpy for `predict`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

pa = pytest.importorskip("pyarrow")

from oracle import orc  # noqa: E402
from test_arrow_gpu import _arrow, _np  # noqa: E402
from test_dyn_prep_gpu import _expected as _dyn_expected  # noqa: E402


@pytest.fixture(scope="module")
def eng():
    from polars_ols_amd import Engine

    e = Engine(0)
    yield e
    e.close()


def _data(seed, n_groups, lo, hi, k, dtype=np.float64, nulls=0.0, weights=False):
    rng = np.random.default_rng(seed)
    sizes = rng.integers(lo, hi + 1, size=n_groups)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(offs[-1])
    cols = [rng.standard_normal(n) for _ in range(k)]
    y = sum((j + 1) * 0.5 * c for j, c in enumerate(cols)) + 0.3 + 0.1 * rng.standard_normal(n)
    if nulls:
        y[rng.random(n) < nulls] = np.nan
        for c in cols[:2]:
            c[rng.random(n) < nulls] = np.nan
    w = rng.uniform(0.5, 2.0, size=n) if weights else None
    return y.astype(dtype), [c.astype(dtype) for c in cols], offs, None if w is None else w.astype(dtype)


def _struct(cols, names, chunks=1, pad=0, null_rows=None):
    """StructArray of float fields (NaN -> null per field); optionally sliced out of a longer array (parent offset != 0), cut into
    chunks, with struct-level nulls on `null_rows`."""
    n = len(cols[0])
    fields = []
    for c in cols:
        m = np.isnan(c)
        v = np.where(m, 0, c)
        if pad:
            v = np.concatenate([np.full(pad, 9.0, dtype=v.dtype), v]); m = np.concatenate([np.zeros(pad, dtype=bool), m])
        fields.append(pa.array(v, mask=m))
    mask = None
    if null_rows is not None:
        mm = np.zeros(n + pad, dtype=bool); mm[np.asarray(null_rows) + pad] = True
        mask = pa.array(mm)
    arr = pa.StructArray.from_arrays(fields, names=list(names), mask=mask)
    if pad:
        arr = arr.slice(pad)
    if chunks == 1:
        return arr
    cuts = np.linspace(0, len(arr), chunks + 1).astype(int)
    return pa.chunked_array([arr.slice(cuts[i], cuts[i + 1] - cuts[i]) for i in range(chunks)])


def _struct_np(arr):
    """struct array -> dict field -> float64 column (null -> NaN)"""
    arr = arr.combine_chunks() if isinstance(arr, pa.ChunkedArray) else arr
    return {arr.type.field(i).name: _np(arr.field(i)) for i in range(arr.type.num_fields)}


# ----------------------------------------------------------------------------------------------------------------- statistics
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("k,weights,icpt,alpha,policy", [(2, False, True, 0.0, "ignore"), (5, True, True, 0.3, "drop"), (3, False, False, 0.0, "drop_y_zero_x")])
def test_statistics_struct(eng, dtype, tol, k, weights, icpt, alpha, policy):
    y, cols, offs, w = _data(3 + k, 7, 40, 90, k, dtype, nulls=0.0 if policy == "ignore" else 0.05, weights=weights)
    feats = {f"f{j}": _arrow(c, chunks=2 if j == 0 else 1, slice_pad=3 if j == 1 else 0) for j, c in enumerate(cols)}
    got = eng.least_squares_statistics_arrow(_arrow(y, chunks=3), feats, weights=None if w is None else _arrow(w), offsets=offs,
                                             add_intercept=icpt, alpha=alpha, null_policy=policy)
    kt = k + int(icpt)
    assert got.type.num_fields == 8 and len(got) == len(offs) - 1
    assert [got.type.field(i).name for i in range(8)] == ["r2", "mae", "mse", "feature_names", "coefficients", "standard_errors", "t_values", "p_values"]
    assert pa.types.is_large_list(got.type.field(3).type) and pa.types.is_large_string(got.type.field(3).type.value_type)
    rows = got.to_pylist()
    names = [f"f{j}" for j in range(k)] + (["const"] if icpt else [])
    for g, row in enumerate(rows):
        s, e = offs[g], offs[g + 1]
        yy = y[s:e].astype(np.float64)
        xx = np.column_stack([c[s:e].astype(np.float64) for c in cols])
        ww = None if w is None else w[s:e].astype(np.float64)
        keep = np.ones(e - s, dtype=bool)
        if policy == "drop":
            keep = ~np.isnan(yy) & ~np.isnan(xx).any(axis=1)
        elif policy == "drop_y_zero_x":
            keep = ~np.isnan(yy); xx = np.nan_to_num(xx)
        yy, xx = yy[keep], xx[keep]
        if icpt:
            xx = np.column_stack([xx, np.ones(len(yy))])
        if ww is not None:
            sw = np.sqrt(ww[keep]); yy, xx = yy * sw, xx * sw[:, None]
        ref = orc.statistics(yy, xx, alpha=alpha)
        assert row["feature_names"] == names
        for key in ("r2", "mae", "mse"):
            assert np.isclose(row[key], ref[key], rtol=tol, atol=tol), (g, key)
        assert np.allclose(row["coefficients"], ref["coefficients"], rtol=tol, atol=tol)
        for key in ("standard_errors", "t_values", "p_values"):
            assert np.allclose(row[key], ref[key], rtol=tol, atol=tol), (g, key)


def test_statistics_bad_dof_panics(eng):
    from polars_ols_amd import PolsPanic

    y, cols, offs, _ = _data(1, 1, 3, 3, 3)
    with pytest.raises(PolsPanic):
        eng.least_squares_statistics_arrow(_arrow(y), {f"f{j}": _arrow(c) for j, c in enumerate(cols)}, offsets=offs)


# --------------------------------------------------------------------------------------------------------------- multi-target
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("policy", ["ignore", "drop", "drop_y_zero_x"])
def test_multi_target_struct_in_struct_out(eng, dtype, tol, policy):
    y, cols, offs, w = _data(21, 6, 30, 70, 4, dtype, nulls=0.04, weights=True)
    n = len(y)
    ys = [y, (2.0 * y - cols[1]).astype(dtype), (cols[0] + cols[3]).astype(dtype)]
    null_rows = [5, 17, n - 2]                                        # struct-level nulls: null in EVERY target
    targets = _struct(ys, ["y1", "y2", "y3"], chunks=3, pad=4, null_rows=null_rows)
    ysn = [a.copy() for a in ys]
    for a in ysn:
        a[null_rows] = np.nan
    feats = {f"x{j + 1}": _arrow(c, chunks=2 if j % 2 else 1) for j, c in enumerate(cols)}
    got = eng.multi_target_least_squares_arrow(targets, feats, weights=_arrow(w), offsets=offs, add_intercept=True, alpha=0.2, null_policy=policy)
    assert got.type.num_fields == 3 and [got.type.field(i).name for i in range(3)] == ["y1", "y2", "y3"] and len(got) == n
    ref = eng.multi_target_least_squares(ysn, cols, offs, weights=w, add_intercept=True, alpha=0.2, null_policy=policy, want=("pred",))["pred"]
    out = _struct_np(got)
    for t, name in enumerate(("y1", "y2", "y3")):
        r = np.asarray(ref[t], dtype=np.float64)
        assert np.array_equal(np.isnan(out[name]), np.isnan(r))
        assert np.allclose(out[name], r, rtol=tol, atol=tol, equal_nan=True)
    if policy == "drop":
        assert np.isnan(out["y1"][null_rows]).all()
    # ... and against the oracle's solve_multi_target on one group (no nulls left after "drop")
    if policy == "drop" and dtype == np.float64:
        s, e = offs[1], offs[2]
        Y = np.column_stack([a[s:e] for a in ysn]); X = np.column_stack([c[s:e] for c in cols] + [np.ones(e - s)])
        keep = ~np.isnan(Y).any(axis=1) & ~np.isnan(X).any(axis=1)
        sw = np.sqrt(w[s:e])
        B = orc.solve_multi_target(Y[keep] * sw[keep, None], X[keep] * sw[keep, None], alpha=0.2)
        exp = X[keep] @ B
        assert np.allclose(np.column_stack([out[nm][s:e][keep] for nm in ("y1", "y2", "y3")]), exp, rtol=1e-6, atol=1e-6)


def test_multi_target_requires_a_struct(eng):
    from polars_ols_amd import PolsPanic

    y, cols, offs, _ = _data(2, 1, 50, 50, 2)
    with pytest.raises(PolsPanic):
        eng.multi_target_least_squares_arrow(_arrow(y), {"a": _arrow(cols[0])}, offsets=offs)


# ------------------------------------------------------------------------------------------------------------- rls / rolling
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("weights,icpt", [(False, False), (True, True)])
def test_recursive_least_squares_arrow(eng, dtype, tol, weights, icpt):
    y, cols, offs, w = _data(31, 5, 60, 120, 3, dtype, nulls=0.05, weights=weights)
    feats = {f"x{j + 1}": _arrow(c, chunks=3 if j == 1 else 1, slice_pad=2 if j == 0 else 0) for j, c in enumerate(cols)}
    kw = dict(weights=None if w is None else _arrow(w), offsets=offs, add_intercept=icpt, half_life=21.0)
    kt = 3 + int(icpt)
    mean = [-0.5] * kt
    ref_c, _ = _dyn_expected("rls", y, cols, offs, w, icpt, "drop", half_life=21.0, initial_state_mean=mean)
    got = eng.recursive_least_squares_arrow(_arrow(y, chunks=2), feats, mode="coefficients", initial_state_mean=mean, **kw)
    assert got.type.num_fields == kt and len(got) == len(y)
    assert [got.type.field(i).name for i in range(kt)] == ["x1", "x2", "x3"] + (["const"] if icpt else [])
    c = _struct_np(got)
    table = np.column_stack([c[nm] for nm in c])
    assert np.allclose(table, ref_c, rtol=tol, atol=tol, equal_nan=True), float(np.nanmax(np.abs(table - ref_c)))
    _, ref_p = _dyn_expected("rls", y, cols, offs, w, icpt, "drop", half_life=21.0)   # predictions: no prior mean (ex.rs:636)
    got = eng.recursive_least_squares_arrow(_arrow(y, chunks=2), feats, mode="predictions", target_name="tgt", initial_state_mean=mean, **kw)
    p = _np(got)
    assert got.type == (pa.float32() if dtype == np.float32 else pa.float64())
    assert np.array_equal(np.isnan(p), np.isnan(ref_p)) and got.null_count == int(np.isnan(ref_p).sum()) > 0
    assert np.allclose(p, ref_p, rtol=tol, atol=tol, equal_nan=True)


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("policy", ["drop", "drop_window"])
def test_rolling_least_squares_arrow(eng, dtype, tol, policy):
    y, cols, offs, _ = _data(41, 4, 150, 260, 3, dtype, nulls=0.04)
    feats = {f"x{j + 1}": _arrow(c, chunks=2 if j == 2 else 1) for j, c in enumerate(cols)}
    kw = dict(offsets=offs, window_size=40, min_periods=12, alpha=1e-3, null_policy=policy)
    ref_c, ref_p = _dyn_expected("rolling", y, cols, offs, None, False, policy, window_size=40, min_periods=12, alpha=1e-3)
    got = eng.rolling_least_squares_arrow(_arrow(y, slice_pad=7), feats, mode="coefficients", **kw)
    c = _struct_np(got)
    table = np.column_stack([c[nm] for nm in ("x1", "x2", "x3")])
    assert np.array_equal(np.isnan(table), np.isnan(ref_c))
    assert np.allclose(table, ref_c, rtol=tol, atol=tol, equal_nan=True), float(np.nanmax(np.abs(table - ref_c)))
    got = eng.rolling_least_squares_arrow(_arrow(y, slice_pad=7), feats, mode="predictions", **kw)
    p = _np(got)
    assert np.array_equal(np.isnan(p), np.isnan(ref_p))
    assert np.allclose(p, ref_p, rtol=tol, atol=tol, equal_nan=True)


# --------------------------------------------------------------------------------------------------------------------- predict
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-12), (np.float32, 1e-5)])
@pytest.mark.parametrize("policy", ["zero", "ignore", "drop"])
@pytest.mark.parametrize("icpt", [False, True])
def test_predict_from_a_coefficients_struct(eng, dtype, tol, policy, icpt):
    rng = np.random.default_rng(5)
    n, k = 777, 4
    cols = [rng.standard_normal(n).astype(dtype) for _ in range(k)]
    cols[1][rng.random(n) < 0.05] = np.nan
    cols[3][rng.random(n) < 0.05] = np.nan
    kt = k + int(icpt)
    coef = [rng.standard_normal(n).astype(dtype) for _ in range(kt)]
    coef[0][[3, 400]] = np.nan                                        # a null coefficient field
    names = [f"x{j + 1}" for j in range(k)] + (["const"] if icpt else [])
    cs = _struct(coef, names, chunks=2, pad=5, null_rows=[10, 11])
    feats = {f"x{j + 1}": _arrow(c, chunks=3 if j == 0 else 1, slice_pad=1 if j == 2 else 0) for j, c in enumerate(cols)}
    got = eng.predict_arrow(cs, feats, add_intercept=icpt, null_policy=policy, name="predictions_test")
    X = np.column_stack([c.astype(np.float64) for c in cols] + ([np.ones(n)] if icpt else []))
    C = np.column_stack([c.astype(np.float64) for c in coef])
    C[[10, 11]] = np.nan
    Xf = X if policy == "ignore" else np.nan_to_num(X)
    exp = (Xf * C).sum(axis=1)
    if policy == "drop":
        exp[np.isnan(X).any(axis=1) | np.isnan(C).any(axis=1)] = np.nan
    p = _np(got)
    assert len(got) == n and got.type == (pa.float32() if dtype == np.float32 else pa.float64())
    assert np.array_equal(np.isnan(p), np.isnan(exp))
    assert np.allclose(p, exp, rtol=tol, atol=tol, equal_nan=True)
    if policy == "drop":
        assert got.null_count == int(np.isnan(exp).sum()) > 0
    else:
        assert got.null_count == 0                                    # NaN values, not nulls (Series::from_vec, ex.rs:740)


def test_predict_field_count_must_match(eng):
    from polars_ols_amd import PolsPanic

    n = 20
    cs = _struct([np.ones(n), np.ones(n)], ["a", "b"])
    with pytest.raises(PolsPanic):
        eng.predict_arrow(cs, {"a": _arrow(np.ones(n))})
