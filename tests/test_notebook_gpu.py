"""The reference's demo notebook (notebooks/polars_ols_demo.ipynb) re-run cell by cell through this repo's mirror of the
`least_squares` namespace -- i.e. through the C-ABI and the HIP kernels -- against the outputs the REFERENCE printed
(tests/golden/notebook_kat.json, typed in from the notebook's output cells; 6 printed decimals).  Host (numpy) and device (torch)
frames, f64 like the reference; the static cells also as f32 at the 1e-4 bar."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P6 = 0.6e-6       # half a unit of the 6th printed decimal (+ slack)


def _frame(d, device, dtype=np.float64):
    from polars_ols_amd import Frame

    cols = {k: v for k, v in d.items() if k != "x"}
    out = {}
    for k, v in cols.items():
        v = v if k == "group" else v.astype(dtype)
        if device:
            import torch

            v = torch.from_numpy(np.ascontiguousarray(v)).cuda()
        out[k] = v
    return Frame(out)


def _np(a):
    return a.double().cpu().numpy() if hasattr(a, "cpu") else np.asarray(a, dtype=np.float64)


@pytest.mark.parametrize("device", [False, True])
@pytest.mark.parametrize("dtype,tol", [(np.float64, P6), (np.float32, 1e-4)])
def test_cells_7_9_11_36_49_static(notebook, device, dtype, tol):
    from polars_ols_amd import OLSKwargs, col, compute_least_squares

    kat, d = notebook["kat"], notebook["d"]
    df = _frame(d, device, dtype)
    # cell 5 / 7
    ols_expr = compute_least_squares(col("y"), col("x1"), col("x2"), col("x3"), mode="predictions",
                                     ols_kwargs=OLSKwargs(null_policy="drop", solve_method="svd"))
    wls_expr = col("y").least_squares.wls("x1", "x2", "x3", sample_weights=col("sample_weights"))
    out = df.select(ols_expr.over("group").alias("predictions_ols_group"), ols_expr.alias("predictions_ols"), wls_expr.alias("wls"))
    t = kat["cell7_tail10"]
    assert np.allclose(_np(out["predictions_ols_group"])[-10:], t["predictions_ols_group"], rtol=tol, atol=tol)
    assert np.allclose(_np(out["predictions_ols"])[-10:], t["predictions_ols"], rtol=tol, atol=tol)
    assert np.allclose(_np(out["wls"])[-10:] * (d["group"][-10:] == 2), t["predictions_wls_masked"], rtol=tol, atol=tol)
    # cell 9
    c = df.select(col("y").least_squares.ols("x1", "x2", "x3", add_intercept=True, mode="coefficients").alias("c"))["c"]
    assert c.names == ["x1", "x2", "x3", "const"]
    assert np.allclose(_np(c.values)[0, :2], kat["cell9_coefficients_first2"], rtol=tol, atol=tol)
    # cell 11
    c = df.select(col("y").least_squares.ols("x1", "x2", "x3", add_intercept=True, mode="coefficients").over("group").alias("c"))["c"]
    keys = [int(k) for k in _np(c.keys)]
    for key, exp in kat["cell11_coefficients_group"].items():
        assert np.allclose(_np(c.values)[keys.index(int(key))], exp, rtol=tol, atol=tol), key
    rows = _np(c.to_rows())                                                    # broadcast to the shape of the data (cell 11, head)
    for r, g in enumerate(d["group"][:5]):
        assert np.allclose(rows[r], kat["cell11_coefficients_group"][str(int(g))], rtol=tol, atol=tol)
    # cell 36
    out = df.select(
        col("y").least_squares.elastic_net(col("x1"), col("x2"), col("x3"), alpha=0.0001, l1_ratio=0.5, positive=True,
                                           mode="coefficients").alias("enet"),
        col("y").least_squares.ridge(col("x1"), col("x2"), col("x3"), alpha=100.0, sample_weights=col("sample_weights"),
                                     mode="coefficients").alias("ridge"))
    assert np.array_equal(_np(out["enet"].values)[0], kat["cell36_enet_non_negative"])
    assert np.allclose(_np(out["ridge"].values)[0], kat["cell36_ridge_alpha100_weighted"], rtol=tol, atol=tol)
    # cell 49
    c = df.select(col("y").least_squares.ols(col("x1"), col("x2"), mode="coefficients").over("group").alias("c"))["c"]
    keys = [int(k) for k in _np(c.keys)]
    for key, exp in kat["cell49_coefficients_group"].items():
        assert np.allclose(_np(c.values)[keys.index(int(key))], exp, rtol=tol, atol=tol), key


@pytest.mark.parametrize("device", [False, True])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_cells_26_to_34_collinear(notebook, device, dtype):
    """x3 := x2 (exact copy), y := x1 + x2 + x3: "qr" prints {1.0, 2.0, -0.0} with norm sqrt(5), "chol" {null, null, null}, "svd"
    {1.0, 1.0, 1.0}."""
    from polars_ols_amd import col

    kat, d = notebook["kat"], notebook["d"]
    x1, x2 = d["x1"].astype(dtype), d["x2"].astype(dtype)
    df = _frame({"x1": x1, "x2": x2, "x3": x2.copy(), "y": (x1 + x2) + x2, "group": d["group"]}, device, dtype)
    tol = 1e-9 if dtype == np.float64 else 1e-4

    def coef(method):
        return _np(df.select(col("y").least_squares.ols("x1", "x2", "x3", solve_method=method, mode="coefficients").alias("c"))["c"].values)[0]

    c = coef("qr")
    assert np.allclose(c, kat["cell28_collinear_qr"], rtol=tol, atol=tol) and c[2] == 0.0
    assert abs(np.linalg.norm(c) - kat["cell28_norm"]) < 10 * tol
    assert np.allclose(coef(None), kat["cell28_collinear_qr"], rtol=tol, atol=tol)        # the default method IS "qr" here (n > k)
    assert np.isnan(coef("chol")).all() and np.isnan(coef("lu")).all()                    # cell 30
    assert np.allclose(coef("svd"), kat["cell32_collinear_svd"], rtol=tol, atol=tol)      # cell 32 / 34


@pytest.mark.parametrize("device", [False, True])
def test_cell_47_dynamic_models(notebook, device):
    from polars_ols_amd import col

    kat, d = notebook["kat"], notebook["d"]
    df = _frame(d, device)
    out = df.select(
        col("y").least_squares.rolling_ols("x1", "x2", "x3", window_size=252, min_periods=5, alpha=0.0001,
                                           mode="coefficients").over("group").alias("rolling_ridge_coef"),
        col("y").least_squares.rls("x1", "x2", "x3", half_life=21.0, initial_state_mean=[-1.0, -1.0, -1.0],
                                   initial_state_covariance=10.0, mode="coefficients").over("group").alias("recursive_least_squares_coef"),
        col("y").least_squares.expanding_ols(col("x1"), col("x2"), col("x3"), mode="predictions").alias("expanding_ols_pred"))
    r = _np(out["rolling_ridge_coef"].to_rows())
    assert np.isnan(r[:5]).all()
    assert np.allclose(r[-5:], kat["cell47_rolling_ridge_tail5"], atol=P6)
    rl = _np(out["recursive_least_squares_coef"].to_rows())
    assert np.allclose(rl[:5], kat["cell47_rls_head5"], atol=P6) and np.allclose(rl[-5:], kat["cell47_rls_tail5"], atol=P6)
    ex = _np(out["expanding_ols_pred"])
    assert np.allclose(ex[:5], kat["cell47_expanding_pred_head5"], atol=P6)
    assert np.allclose(ex[-5:], kat["cell47_expanding_pred_tail5"], atol=P6)


@pytest.mark.parametrize("device", [False, True])
def test_cell_54_multi_target_residuals(notebook, device):
    from polars_ols_amd import col, struct

    kat, d = notebook["kat"], notebook["d"]
    df = _frame(d, device)
    x1, x2, x3 = df["x1"], df["x2"], df["x3"]
    df.update({"y1": x1 + x2 + x3, "y2": x1 - x2 + x3, "y3": -x1 + x2 - x3})
    res = df.select(struct("y1", "y2", "y3").least_squares.multi_target_ols("x1", "x2", "x3", sample_weights="sample_weights",
                                                                             mode="residuals").over("group").alias("residuals"))["residuals"]
    for k in ("y1", "y2", "y3"):
        assert float(np.abs(_np(res[k])).max()) < 100 * kat["cell54_multi_target_residual_bound"]
