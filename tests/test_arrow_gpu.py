"""pols_least_squares_arrow through pyarrow's Arrow C Data Interface export: the plugin bodies of src/expressions.rs:390-446 on
columns as Polars holds them -- validity BITMAPS (not NaN sentinels), sliced arrays (offset != 0), several chunks per column,
integer and f32 columns -- against the oracle on the rows the reference would have kept (same expectations as
tests/test_nulls_gpu.py), plus the coefficient struct (:114-143) and the nullable prediction column (:145-158)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

pa = pytest.importorskip("pyarrow")

from oracle import orc  # noqa: E402
from test_nulls_gpu import _expected, _frame  # noqa: E402


@pytest.fixture(scope="module")
def eng():
    from polars_ols_amd import Engine

    e = Engine(0)
    yield e
    e.close()


def _arrow(a, chunks=1, slice_pad=0, typ=None):
    """numpy column with NaN-as-null -> pyarrow column with a validity bitmap; optionally sliced out of a longer buffer
    (offset != 0) and split into chunks of unequal length."""
    mask = np.isnan(a)
    vals = np.where(mask, 0, a)                             # the value under a null is arbitrary: make sure nobody reads it as data
    if slice_pad:
        vals = np.concatenate([np.full(slice_pad, 7.0, dtype=vals.dtype), vals])
        mask = np.concatenate([np.zeros(slice_pad, dtype=bool), mask])
    arr = pa.array(vals, mask=mask, type=typ)
    if slice_pad:
        arr = arr.slice(slice_pad)                          # offset = slice_pad, shares the padded buffers
    if chunks == 1:
        return arr
    cuts = np.linspace(0, len(arr), chunks + 1).astype(int)
    cuts[1:-1] += np.arange(1, chunks) * 3 % 5              # unequal, not byte-aligned chunk boundaries
    return pa.chunked_array([arr.slice(cuts[i], cuts[i + 1] - cuts[i]) for i in range(chunks)])


def _np(arr):
    return np.array([np.nan if v is None else v for v in arr.to_pylist()], dtype=np.float64)


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("policy", ["zero", "drop", "drop_zero", "drop_y_zero_x"])
@pytest.mark.parametrize("k,weights,icpt,kw,chunks,pad", [
    (3, False, False, {}, 1, 0),
    (6, True, True, {"alpha": 0.5, "l1_ratio": 0.0}, 3, 5),
    (5, False, True, {"alpha": 0.01, "l1_ratio": 0.5, "tol": 1e-10, "max_iter": 20_000}, 4, 11),
])
def test_arrow_null_policies_vs_oracle(eng, dtype, tol, policy, k, weights, icpt, kw, chunks, pad):
    y, cols, offs, w = _frame(11 + k, dtype, k)
    w = w if weights else None
    coef, pred, resid = _expected(y, cols, offs, w, icpt, policy, **kw)
    feats = {f"x{j + 1}": _arrow(c, chunks=chunks if j % 2 == 0 else 1, slice_pad=pad if j % 3 == 0 else 0) for j, c in enumerate(cols)}
    args = dict(target_name="y", weights=None if w is None else _arrow(w, chunks=2), offsets=offs, add_intercept=icpt,
                null_policy=policy, **kw)
    got = eng.least_squares_arrow(_arrow(y, chunks=chunks, slice_pad=pad), feats, mode="coefficients", **args)
    assert got.type.num_fields == k + int(icpt) and [got.type.field(i).name for i in range(k)] == [f"x{j + 1}" for j in range(k)]
    if icpt:
        assert got.type.field(k).name == "const"                                    # appended LAST (least_squares.py:188)
    table = np.column_stack([_np(got.field(i)) for i in range(k + int(icpt))])
    assert np.allclose(table, coef, rtol=tol, atol=tol), float(np.nanmax(np.abs(table - coef)))
    got = eng.least_squares_arrow(_arrow(y, chunks=chunks, slice_pad=pad), feats, mode="predictions", **args)
    p = _np(got)
    assert np.allclose(p, pred, rtol=tol, atol=tol, equal_nan=True)
    if policy == "drop":                                                            # masked rows are NULLS (mask_predictions), not NaN values
        assert got.null_count == int(np.isnan(pred).sum()) > 0
    else:
        assert got.null_count == 0
    got = eng.least_squares_arrow(_arrow(y, chunks=chunks, slice_pad=pad), feats, mode="residuals", **args)
    assert np.allclose(_np(got), resid, rtol=tol, atol=tol, equal_nan=True)
    assert got.type == (pa.float32() if dtype == np.float32 else pa.float64())


def test_arrow_integer_and_mixed_columns_single_group_call(eng):
    """The per-group call a plugin receives (no offsets): int64 / int32 / uint8 / f32 / f64 inputs are cast like the reference
    casts every Series to Float64 (src/expressions.rs:33, 47, 80); a null in an integer column is a null, not a sentinel."""
    rng = np.random.default_rng(0)
    n = 5_000
    xi = rng.integers(-50, 50, size=n)
    xs = rng.integers(0, 200, size=n)
    xf = rng.standard_normal(n).astype(np.float32)
    xd = rng.standard_normal(n)
    y = 0.5 * xi - 0.1 * xs + 2.0 * xf + xd + 0.1 * rng.standard_normal(n)
    nulls = rng.random(n) < 0.05
    feats = {"i64": pa.array(xi, mask=nulls, type=pa.int64()), "u8": pa.array(xs.astype(np.uint8), type=pa.uint8()),
             "f32": pa.array(xf, type=pa.float32()), "f64": pa.chunked_array([pa.array(xd[:1234]), pa.array(xd[1234:])]),
             "i32": pa.array((xi * 2).astype(np.int32), type=pa.int32())}
    X = np.column_stack([xi, xs, xf.astype(np.float64), xd, xi * 2]).astype(np.float64)
    ok = ~nulls
    beta = orc.get_coefficients(y[ok], X[ok], alpha=1.0, l1_ratio=0.0)
    got = eng.least_squares_arrow(pa.array(y), feats, mode="coefficients", null_policy="drop", alpha=1.0, l1_ratio=0.0)
    assert len(got) == 1 and got.type.field(0).name == "i64"
    assert np.allclose([got.field(i)[0].as_py() for i in range(5)], beta, rtol=1e-6, atol=1e-8)
    pred = eng.least_squares_arrow(pa.array(y), feats, mode="predictions", null_policy="drop", alpha=1.0, l1_ratio=0.0)
    exp = np.where(ok, np.nan_to_num(X) @ beta, np.nan)
    assert pred.null_count == int(nulls.sum()) and np.allclose(_np(pred), exp, rtol=1e-6, atol=1e-6, equal_nan=True)


def test_arrow_ignore_policy_keeps_nan_values_and_unnamed_fields(eng):
    """null_policy="ignore" (the OLS default): a null becomes NaN and flows through (ex.rs:84-86) -- the group's predictions are NaN
    VALUES, not nulls; unnamed features are named by their index (ex.rs:126-131); all-NaN coefficients are nulls (:138)."""
    rng = np.random.default_rng(1)
    offs = np.array([0, 200, 400], dtype=np.int64)
    cols = [rng.standard_normal(400) for _ in range(3)]
    y = sum(cols) + 0.1 * rng.standard_normal(400)
    mask = np.zeros(400, dtype=bool)
    mask[250] = True                                                                  # one null feature value in group 1
    feats = [pa.array(cols[0]), pa.array(cols[1], mask=mask), pa.array(cols[2])]
    pred = eng.least_squares_arrow(pa.array(y), feats, offsets=offs)
    p = _np(pred)
    assert pred.null_count == 0 and np.isfinite(p[:200]).all() and np.isnan(np.array(pred.to_numpy(zero_copy_only=False))[200:]).all()
    coef = eng.least_squares_arrow(pa.array(y), feats, offsets=offs, mode="coefficients")
    assert [coef.type.field(i).name for i in range(3)] == ["0", "1", "2"]
    assert coef.field(0).null_count == 1 and coef.field(0)[1].as_py() is None and coef.field(0)[0].as_py() is not None
    ref = orc.batched_least_squares(y[:200], [c[:200] for c in cols], [0, 200])
    assert np.allclose([coef.field(i)[0].as_py() for i in range(3)], ref["coef"][0], rtol=1e-6, atol=1e-9)


def test_arrow_rejects_what_the_reference_cannot_cast(eng):
    from polars_ols_amd import PolsError

    y = pa.array([1.0, 2.0, 3.0])
    with pytest.raises(PolsError, match="arrow format"):
        eng.least_squares_arrow(y, {"s": pa.array(["a", "b", "c"])})
    with pytest.raises(PolsError, match="equal length"):
        eng.least_squares_arrow(y, {"x": pa.array([1.0, 2.0])})
