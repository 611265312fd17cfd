"""Null policies fused into the static kernels (src/expressions.rs:201-296 compute_is_valid_mask / handle_nulls and the
prediction rules of :398-427), through the C-ABI: a null is a NaN.  Expected values: the oracle's solver on the rows the
reference would have kept, predictions composed exactly as the reference composes them."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import orc  # noqa: E402

POLICIES = ("zero", "drop", "drop_zero", "drop_y_zero_x", "drop_window")


@pytest.fixture(scope="module")
def eng():
    from polars_ols_amd import Engine

    e = Engine(0)
    yield e
    e.close()


def _frame(seed, dtype, k, G=19, lo=30, hi=900, null_frac=0.08):
    rng = np.random.default_rng(seed)
    sizes = rng.integers(lo, hi, size=G)
    sizes[3] = 0
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(offs[-1])
    cols = [rng.normal(size=n).astype(dtype) for _ in range(k)]
    y = (sum(cols) + 0.2 * rng.normal(size=n) + 0.3).astype(dtype)
    for c in [y] + cols[: max(1, k // 2)]:
        c[rng.random(n) < null_frac] = np.nan
    s, e = offs[5], offs[6]
    y[s:e] = np.nan                                     # a group with no valid target at all
    w = rng.uniform(0.2, 2.0, size=n).astype(dtype)
    return y, cols, offs, w


def _expected(y, cols, offs, w, icpt, policy, **kw):
    n = len(y)
    k = len(cols) + int(icpt)
    coef = np.zeros((len(offs) - 1, k))
    pred = np.full(n, np.nan)
    for g in range(len(offs) - 1):
        s, e = offs[g], offs[g + 1]
        if e == s:
            continue
        X = np.column_stack([c[s:e] for c in cols]).astype(np.float64)
        if icpt:
            X = np.column_stack([X, np.ones(e - s)])
        yy = y[s:e].astype(np.float64)
        sw = np.sqrt(w[s:e].astype(np.float64)) if w is not None else np.ones(e - s)
        Xs, ys = X * sw[:, None], yy * sw
        if policy == "zero":
            ok = np.ones(e - s, dtype=bool)
            Xf, yf = np.nan_to_num(Xs), np.nan_to_num(ys)
        elif policy == "drop_y_zero_x":
            ok = ~np.isnan(ys)
            Xf, yf = np.nan_to_num(Xs), ys
        else:
            ok = ~np.isnan(ys) & ~np.isnan(Xs).any(axis=1)
            Xf, yf = Xs, ys
        beta = orc.get_coefficients(yf[ok], Xf[ok], **kw) if ok.any() else np.zeros(k)
        coef[g] = beta
        p = (np.nan_to_num(Xs) @ beta) / sw
        if policy == "drop":
            p[~ok] = np.nan
        pred[s:e] = p
    return coef, pred, y.astype(np.float64) - pred


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("policy", POLICIES)
@pytest.mark.parametrize("k,weights,icpt,kw", [
    (3, False, False, {}),
    (7, True, True, {"alpha": 0.5}),
    (9, True, True, {"alpha": 0.5}),
    (5, False, True, {"alpha": 0.01, "l1_ratio": 0.5, "tol": 1e-10, "max_iter": 20_000}),
    (20, True, False, {}),
    (40, True, True, {"alpha": 0.1}),                   # wide path (K8): row-mask pre-pass
    (36, False, False, {"alpha": 0.02, "l1_ratio": 0.5, "tol": 1e-10, "max_iter": 20_000}),
])
def test_null_policies_vs_oracle(eng, dtype, tol, policy, k, weights, icpt, kw):
    y, cols, offs, w = _frame(7 + k, dtype, k)
    w = w if weights else None
    out = eng.least_squares(y, cols, offs, weights=w, add_intercept=icpt, want=("coef", "pred", "resid", "status"),
                            null_policy=policy, **kw)
    coef, pred, resid = _expected(y, cols, offs, w, icpt, "drop_zero" if policy == "drop_window" else policy, **kw)
    # register-resident NULLS family (9+ columns: masked multi-pass Gram; 16-31 columns f32 up to 1 024 rows, f64 up to 512 -- these
    # groups reach 899) vs the streamed / wide kernels
    kt = k + int(icpt)
    k1 = "l1_ratio" not in kw and (kt <= 15 or (kt <= 31 and dtype == np.float32))
    assert eng.last_kernel.startswith("k1_gram_chol") == k1 and (not k1 or eng.last_kernel.endswith("_nulls")), eng.last_kernel
    assert eng.last_kernel.startswith("k8_wide") == (k + int(icpt) > 31)
    if policy != "zero":
        assert int(out["status"][3]) == 2 and int(out["status"][5]) == 2      # no rows / no row left in the fit
    assert np.allclose(out["coef"], coef, rtol=tol, atol=tol), float(np.nanmax(np.abs(out["coef"] - coef)))
    assert np.array_equal(np.isnan(out["pred"]), np.isnan(pred))
    assert np.allclose(out["pred"], pred, rtol=tol, atol=tol, equal_nan=True)
    assert np.array_equal(np.isnan(out["resid"]), np.isnan(resid))
    assert np.allclose(out["resid"], resid, rtol=tol, atol=tol, equal_nan=True)


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("policy", ["drop", "zero", "drop_y_zero_x"])
@pytest.mark.parametrize("k,weights,icpt,kw,lo,hi", [
    (11, False, False, {}, 80, 250),                   # one wave
    (11, True, True, {"alpha": 0.5}, 300, 500),        # two waves (f32) / the 256-thread team (f64)
    (13, False, True, {}, 120, 120),                   # equal groups of a vector multiple: the branch-free aligned form
    (14, True, False, {"alpha": 0.1}, 200, 900),       # 15 columns with the target: four passes, 256-thread team
    (15, False, False, {}, 600, 1000),
])
def test_null_policies_11_to_15_columns_stay_resident(eng, dtype, tol, policy, k, weights, icpt, kw, lo, hi):
    """src/expressions.rs:201-296 inside the register-resident kernels at 11-15 columns (masked multi-pass Gram)."""
    y, cols, offs, w = _frame(100 + k, dtype, k, G=23, lo=lo, hi=hi + 1, null_frac=0.03)
    w = w if weights else None
    out = eng.least_squares(y, cols, offs, weights=w, add_intercept=icpt, want=("coef", "pred", "status"), null_policy=policy, **kw)
    assert eng.last_kernel.startswith("k1_gram_chol") and eng.last_kernel.endswith("_nulls") and "_p" in eng.last_kernel, eng.last_kernel
    coef, pred, _ = _expected(y, cols, offs, w, icpt, policy, **kw)
    assert np.allclose(out["coef"], coef, rtol=tol, atol=tol), (eng.last_kernel, float(np.nanmax(np.abs(out["coef"] - coef))))
    assert np.array_equal(np.isnan(out["pred"]), np.isnan(pred))
    assert np.allclose(out["pred"], pred, rtol=tol, atol=tol, equal_nan=True)


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("policy", ["drop", "zero", "drop_y_zero_x"])
@pytest.mark.parametrize("k,weights,icpt,kw", [
    (16, False, False, {}),
    (19, True, True, {"alpha": 0.3}),
    (24, False, False, {}),
    (27, True, False, {"alpha": 0.1}),
    (30, False, True, {}),
])
def test_null_policies_16_to_31_columns(eng, dtype, tol, policy, k, weights, icpt, kw):
    """16-31 columns under a null policy: the register-resident masked kernels wherever the plain ones run (f32; f64 at 17-24)."""
    y, cols, offs, w = _frame(200 + k, dtype, k, G=17, lo=5 * k, hi=250, null_frac=0.02)
    w = w if weights else None
    out = eng.least_squares(y, cols, offs, weights=w, add_intercept=icpt, want=("coef", "pred", "status"), null_policy=policy, **kw)
    kt = k + int(icpt)
    resident = dtype == np.float32 or 17 <= kt <= 24
    assert (eng.last_kernel.startswith("k1_gram_chol") and eng.last_kernel.endswith("_nulls")) == resident, eng.last_kernel
    coef, pred, _ = _expected(y, cols, offs, w, icpt, policy, **kw)
    assert np.allclose(out["coef"], coef, rtol=tol, atol=tol), (eng.last_kernel, float(np.nanmax(np.abs(out["coef"] - coef))))
    assert np.array_equal(np.isnan(out["pred"]), np.isnan(pred))
    assert np.allclose(out["pred"], pred, rtol=tol, atol=tol, equal_nan=True)


def test_validity_bytes_drop_rows(eng):
    y, cols, offs, _ = _frame(3, np.float64, 4, null_frac=0.0)
    s, e = offs[5], offs[6]
    y[s:e] = 1.0
    rng = np.random.default_rng(1)
    valid = (rng.random(len(y)) > 0.2).astype(np.uint8)
    out = eng.least_squares(y, cols, offs, valid=valid, want=("coef", "pred"), null_policy="drop")
    y2 = y.copy()
    y2[valid == 0] = np.nan
    ref = eng.least_squares(y2, cols, offs, want=("coef", "pred"), null_policy="drop")
    assert np.array_equal(out["coef"], ref["coef"]) and np.array_equal(out["pred"], ref["pred"], equal_nan=True)
    with pytest.raises(Exception):
        eng.least_squares(y, cols, offs, valid=valid, want=("coef",), null_policy="ignore")


def test_device_matches_host_with_nulls(eng):
    import torch

    y, cols, offs, w = _frame(5, np.float32, 6)
    host = eng.least_squares(y, cols, offs, weights=w, add_intercept=True, want=("coef", "pred"), null_policy="drop")
    dev = eng.least_squares(torch.from_numpy(y).cuda(), [torch.from_numpy(c).cuda() for c in cols], offs,
                            weights=torch.from_numpy(w).cuda(), add_intercept=True, want=("coef", "pred"), null_policy="drop")
    torch.cuda.synchronize()
    assert np.array_equal(host["coef"], dev["coef"].cpu().numpy())
    assert np.array_equal(host["pred"], dev["pred"].cpu().numpy(), equal_nan=True)


def test_rank_deficient_group_under_drop_takes_svd(eng):
    """After dropping, a group keeps fewer rows than features: flagged, solved by K6 on the surviving rows only."""
    rng = np.random.default_rng(2)
    n, k = 40, 6
    cols = [rng.normal(size=n) for _ in range(k)]
    y = sum(cols) + 0.1 * rng.normal(size=n)
    y[4:] = np.nan                                      # 4 valid rows, 6 features
    offs = np.array([0, n], dtype=np.int64)
    out = eng.least_squares(y, cols, offs, want=("coef", "pred", "status"), null_policy="drop_zero")
    X = np.column_stack(cols)
    beta = np.linalg.lstsq(X[:4], y[:4], rcond=None)[0]
    assert int(out["status"][0]) == 1
    assert np.allclose(out["coef"][0], beta, rtol=1e-6, atol=1e-8)
    assert np.allclose(out["pred"], X @ beta, rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("k,n_lo,n_hi", [(3, 30, 900), (8, 900, 1100), (20, 300, 600)])
def test_null_weights_act_as_1e_24_behind_the_c_abi(eng, dtype, tol, k, n_lo, n_hi):
    """polars_ols/least_squares.py:193: sqrt_w = w.sqrt().fill_null(1e-12).  pols_least_squares itself does the fill (a device pass
    over the weights column; nothing in the Python front end): NaN weights straight into the C-ABI, device and host batches, must
    equal the oracle on the column with 1e-24 in their place -- and the caller's column must come back untouched."""
    import torch

    rng = np.random.default_rng(31 + k)
    sizes = rng.integers(n_lo, n_hi, size=9)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(offs[-1])
    cols = [rng.normal(size=n).astype(dtype) for _ in range(k)]
    y = (sum(cols) + 0.2 * rng.normal(size=n)).astype(dtype)
    w = rng.uniform(0.2, 2.0, size=n).astype(dtype)
    w[rng.random(n) < 0.1] = np.nan
    w_filled = np.where(np.isnan(w), dtype(1e-24), w).astype(dtype)
    ref = orc.batched_least_squares(y, cols, offs, weights=w_filled, add_intercept=True, alpha=0.1, l1_ratio=0.0)
    # (predictions on a null-weight row: (x sqrt_w) . beta / sqrt_w is evaluated with sqrt_w = 1e-12 in both -- compare where the weight is real)
    real = ~np.isnan(w)
    for device in (True, False):
        if device:
            wt = torch.from_numpy(w).cuda()
            out = eng.least_squares(torch.from_numpy(y).cuda(), [torch.from_numpy(c).cuda() for c in cols], offs, weights=wt,
                                    add_intercept=True, alpha=0.1, l1_ratio=0.0, want=("coef", "pred"))
            got_c, got_p = out["coef"].double().cpu().numpy(), out["pred"].double().cpu().numpy()
            back = wt.cpu().numpy()
            assert np.array_equal(np.isnan(back), np.isnan(w)) and np.array_equal(back[real], w[real])
        else:
            out = eng.least_squares(y, cols, offs, weights=w, add_intercept=True, alpha=0.1, l1_ratio=0.0, want=("coef", "pred"))
            got_c, got_p = np.asarray(out["coef"], dtype=np.float64), np.asarray(out["pred"], dtype=np.float64)
        assert np.allclose(got_c, ref["coef"], rtol=tol, atol=tol), (device, float(np.abs(got_c - ref["coef"]).max()))
        assert np.allclose(got_p[real], ref["pred"][real], rtol=tol, atol=tol), device
