"""Pins the CPU oracle (oracle/pols_oracle.c) against every known-answer vector the reference holds for
the hot path: the README's printed outputs, the literal Woodbury case of
src/lib.rs, the Rust unit-test scenarios of src/lib.rs:47-171 and the seeded `_make_data` cases of
tests/test_ols.py evaluated with the reference's own third-party oracles (committed under tests/golden/).
CPU only."""
import numpy as np
import pytest

from oracle import orc
from refdata import make_data, insert_nulls, sort_by_group


def _frame(kat):
    f = {k: np.asarray(v, dtype=np.float64) for k, v in kat["frame"].items()}
    f["group"] = f["group"].astype(np.int64)
    return f


# ----------------------------------------------------------------------------- README KATs

def test_readme_coefficients_full(golden):
    kat = golden["kat"]; f = _frame(kat)
    x = np.column_stack([f["x1"], f["x2"], np.ones(10)])  # intercept LAST (least_squares.py:188)
    coef = orc.get_coefficients(f["y"], x)
    assert np.allclose(np.round(coef, 6), kat["coefficients_full"], atol=1.1e-6)


def test_readme_coefficients_group(golden):
    kat = golden["kat"]; f = _frame(kat)
    order, offs, keys = sort_by_group(f["group"])
    out = orc.batched_least_squares(f["y"][order], [f["x1"][order], f["x2"][order]], offs, add_intercept=True)
    for g, key in enumerate(keys):
        assert np.allclose(np.round(out["coef"][g], 6), kat["coefficients_group"][str(key)], atol=1.1e-6)


def test_readme_lasso_predictions(golden):
    kat = golden["kat"]; f = _frame(kat)
    order, offs, _ = sort_by_group(f["group"])
    out = orc.batched_least_squares(f["y"][order], [f["x1"][order], f["x2"][order]], offs, add_intercept=True,
                                    alpha=0.0001, l1_ratio=1.0)
    assert np.array_equal(np.round(out["pred"][:5], 2), kat["predictions_lasso_head5_round2"])


def test_readme_wls_predictions(golden):
    kat = golden["kat"]; f = _frame(kat)
    out = orc.batched_least_squares(f["y"], [f["x1"], f["x2"]], [0, 10], weights=f["weights"])
    assert np.array_equal(np.round(out["pred"][:5], 2), kat["predictions_wls_head5_round2"])


def test_readme_rls_path(golden):
    kat = golden["kat"]; f = _frame(kat)
    order, offs, _ = sort_by_group(f["group"])
    out = orc.batched_rls(f["y"][order], [f["x1"][order], f["x2"][order]], offs)  # defaults: P0=10, no forgetting
    assert np.allclose(np.round(out["coef"][:5], 6), kat["rls_coefficients_head5"], atol=1.1e-6)


def test_readme_statistics(golden):
    kat = golden["kat"]; f = _frame(kat); s = kat["statistics"]
    x = np.column_stack([f["x1"], f["x2"], np.ones(10)])
    st = orc.statistics(f["y"], x)
    assert np.round(st["r2"], 5) == s["r2"] and np.round(st["mae"], 6) == s["mae"] and np.round(st["mse"], 5) == s["mse"]
    assert np.allclose(np.round(st["standard_errors"], 6), s["standard_errors"], atol=1.1e-6)
    assert np.allclose(np.round(st["t_values"], 6), s["t_values"], atol=1.1e-5)
    assert np.allclose(st["p_values"], s["p_values"], rtol=1e-4)


def test_lib_rs_woodbury_literal(golden):
    w = golden["kat"]["woodbury"]
    a, u, c, v = (np.array(w[k]) for k in "aucv")
    got = orc.woodbury_update(np.linalg.inv(a), u, c, v, c_is_diag=True)
    assert np.allclose(got, np.linalg.inv(a + u @ c @ v), rtol=1e-5)


# ----------------------------------------------------------------------------- src/lib.rs unit-test scenarios

def _lib_rs_data():
    rng = np.random.default_rng(123)
    x = rng.normal(size=(10_000, 2))
    return x.sum(1), x


def test_lib_rs_ols_ridge_enet():
    y, x = _lib_rs_data()
    assert np.allclose(orc.get_coefficients(y, x), [1, 1], rtol=1e-3)
    assert np.allclose(orc.get_coefficients(y, x, solve_method="svd"), [1, 1], rtol=1e-3)
    assert np.allclose(orc.get_coefficients(y, x, alpha=10.0), [0.999, 0.999], rtol=1e-3)
    assert np.allclose(orc.get_coefficients(y, x, alpha=10.0, solve_method="svd"), orc.get_coefficients(y, x, alpha=10.0), rtol=1e-9)
    w, _ = orc.solve_elastic_net(y, x, 0.001, l1_ratio=0.5)
    assert np.allclose(w, [0.999, 0.999], rtol=1e-3)


def test_lib_rs_rls_and_rolling():
    y, x = _lib_rs_data()
    c = orc.solve_rls(y, x, half_life=252.0, initial_state_covariance=0.01)
    assert np.allclose(c[-1], [1, 1], rtol=1e-4)
    c = orc.solve_rolling_ols(y, x, 1000, min_periods=100, use_woodbury=False, null_policy="drop_window")
    assert np.allclose(c[-1], [1, 1], rtol=1e-4)
    assert np.isnan(c[:99]).all() and not np.isnan(c[99:]).any()


def test_lib_rs_update_xtx_inv():
    rng = np.random.default_rng(5)
    x = rng.normal(size=(252, 5))
    xtx = x.T @ x
    x_new = np.array([0.5, 2.0, -0.3, 0.1, 0.2]); x_old = x[0]
    got = orc.update_xtx_inv(orc.inv(xtx, True), np.stack([x_old, x_new]), np.array([[-1.0, 0], [0, 1.0]]))
    exp = np.linalg.inv(xtx - np.outer(x_old, x_old) + np.outer(x_new, x_new))
    assert np.allclose(got, exp, rtol=1e-5, atol=1e-9)


# ----------------------------------------------------------------------------- tests/test_ols.py cases

@pytest.mark.parametrize("method", ["qr", "svd", "chol", "lu", None])
def test_ols_all_methods(golden, method):
    z = golden["npz"]
    coef = orc.get_coefficients(z["ols_y"], z["ols_x"], solve_method=method)
    assert np.allclose(coef, z["ols_coef"], rtol=1e-9, atol=1e-12)
    assert np.allclose(z["ols_x"] @ coef, z["ols_pred"], rtol=1e-9, atol=1e-10)


def test_ridge(golden):
    z = golden["npz"]
    for m in ("chol", "lu", None):
        assert np.allclose(orc.get_coefficients(z["ridge_y"], z["ridge_x"], alpha=0.01, solve_method=m),
                           z["ridge_coef_chol"], rtol=1e-10)
    assert np.allclose(orc.get_coefficients(z["ridge_y"], z["ridge_x"], alpha=0.01, solve_method="svd"),
                       z["ridge_coef_svd"], rtol=1e-9)
    assert np.allclose(orc.get_coefficients(z["ridge_y"], z["ridge_x"], alpha=10.0), z["ridge_coef_alpha10"], rtol=1e-10)


def test_wls_and_intercept(golden):
    z = golden["npz"]
    cols = [np.ascontiguousarray(z["ridge_x"][:, j]) for j in range(2)]
    out = orc.batched_least_squares(z["ridge_y"], cols, [0, 5000], weights=z["wls_w"], add_intercept=True)
    assert np.allclose(out["coef"][0], z["wls_coef"], rtol=1e-9)
    assert np.allclose(out["pred"], z["wls_pred"], rtol=1e-9, atol=1e-10)
    assert np.allclose(out["resid"], z["ridge_y"] - z["wls_pred"], rtol=1e-8, atol=1e-10)
    out = orc.batched_least_squares(z["ridge_y"], cols, [0, 5000], add_intercept=True)
    assert np.allclose(out["coef"][0], z["intercept_coef"], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("name,k,sparsity,alpha,method",
                         [("enet2", 2, 0.0, 0.1, "cd"), ("enet100", 100, 0.5, 0.3, "cd"),
                          ("enet100", 100, 0.5, 0.3, "cd_active_set")])
def test_elastic_net(golden, name, k, sparsity, alpha, method):
    z = golden["npz"]
    d = make_data(n_features=k, sparsity=sparsity)
    # reference tolerance for this comparison is 1e-4 at tol=1e-4 (tests/test_ols.py:599); we pin tighter
    coef = orc.get_coefficients(d["y"], d["x"], alpha=alpha, l1_ratio=0.5, tol=1e-9, max_iter=10_000, solve_method=method)
    if method == "cd":
        assert np.allclose(coef, z[f"{name}_coef"], atol=1e-7)
    assert np.allclose(d["x"] @ coef, z[f"{name}_pred"], atol=1e-4, rtol=1e-4)
    coef = orc.get_coefficients(d["y"], d["x"], alpha=alpha, l1_ratio=0.5, tol=1e-4, solve_method=method)
    assert np.allclose(d["x"] @ coef, z[f"{name}_pred"], atol=1e-4, rtol=1e-4)


def test_elastic_net_non_negative(golden):
    z = golden["npz"]
    d = make_data()
    xn = np.column_stack([d["x"][:, 0], -d["x"][:, 1]])
    coef = orc.get_coefficients(d["y"], xn, alpha=0.1, l1_ratio=0.5, tol=1e-9, positive=True)
    assert np.allclose(coef, z["nnls_coef"], atol=1e-7) and coef[1] == 0.0


def test_grouped_non_contiguous(golden):
    z = golden["npz"]
    d = make_data(n_groups=10)
    order, offs, keys = sort_by_group(d["group"])
    out = orc.batched_least_squares(d["y"][order], [d["x1"][order], d["x2"][order]], offs, n_threads=4)
    assert np.allclose(out["coef"], z["group_coef"], rtol=1e-9)
    pred = np.empty(5000); pred[order] = out["pred"]  # scatter back to row order
    assert np.allclose(pred, np.einsum("ij,ij->i", d["x"], z["group_coef"][d["group"]]), rtol=1e-9, atol=1e-10)


def test_rls_expanding_equals_ols(golden):
    z = golden["npz"]
    x, y = z["nulls_x"], z["nulls_y"]
    valid = ~np.isnan(x).any(axis=1) & ~np.isnan(y)
    c = orc.solve_rls(np.nan_to_num(y), np.nan_to_num(x), initial_state_covariance=1e6, is_valid=valid)
    assert np.allclose(c[-1], z["rls_expanding_last"], rtol=1e-4, atol=1e-4)
    # strong prior sticks (tests/test_ols.py:684-715)
    d = make_data()
    c = orc.solve_rls(d["y"], d["x"], initial_state_covariance=1e-6, initial_state_mean=[0.25, 0.25])
    assert np.allclose(c[0], 0.25, rtol=1e-3) and np.allclose(c[10], 0.25, rtol=1e-3)


@pytest.mark.parametrize("win,mp,woodbury", [(2, 2, False), (10, 2, False), (10, 2, True), (63, 5, False),
                                             (252, 5, False), (252, 5, True)])
def test_rolling_drop_window(golden, win, mp, woodbury):
    z = golden["npz"]
    x, y = z["roll_x"], z["roll_y"]
    valid = ~np.isnan(y)
    c = orc.solve_rolling_ols(np.nan_to_num(y), x, win, min_periods=mp, use_woodbury=woodbury,
                              is_valid=valid, null_policy="drop_window")
    exp = z[f"roll_{win}_{mp}"]
    # brute-force per-window lstsq (statsmodels RollingOLS semantics, tests/test_ols.py:751-764): the incremental add / subtract
    # state of ls.rs:707-734 (and the Woodbury form :737-787) agrees to 1e-8 on EVERY row, 2-row windows included
    assert np.allclose(c, exp, rtol=1e-8, atol=1e-8, equal_nan=True)


@pytest.mark.parametrize("mp,expected", [(999, 2), (1000, 1), (1001, 0)])
def test_rolling_insufficient_data(mp, expected):
    d = make_data(n_samples=1_000)
    c = orc.solve_rolling_ols(d["y"], d["x"], 2_000, min_periods=mp, use_woodbury=False, null_policy="drop_window")
    assert (~np.isnan(c[:, 0])).sum() == expected


@pytest.mark.parametrize("win", [21, 252])
def test_rolling_drop_equals_dropna(win):
    d = insert_nulls(make_data(n_samples=1_000), columns=("y",))
    valid = ~np.isnan(d["y"])
    c_full = orc.solve_rolling_ols(np.nan_to_num(d["y"]), d["x"], win, is_valid=valid, null_policy="drop")
    c_drop = orc.solve_rolling_ols(d["y"][valid], d["x"][valid], win, null_policy="drop")
    assert np.allclose(c_full[valid], c_drop, equal_nan=True)


def test_statistics(golden):
    z = golden["npz"]
    d = make_data()
    xi = np.column_stack([d["x"], np.ones(5000)])
    st = orc.statistics(d["y"], xi)
    assert np.allclose(st["coefficients"], z["stats_coef"]) and np.allclose(st["standard_errors"], z["stats_se"])
    assert np.allclose(st["t_values"], z["stats_t"]) and np.allclose(st["p_values"], z["stats_p"], rtol=1e-6, atol=1e-300)
    assert np.allclose([st["r2"], st["mse"]], z["stats_r2_mse"])


def test_svd_min_norm_wide_and_collinear():
    d = make_data(n_samples=10, n_features=100, scale=1e-4)
    c = orc.get_coefficients(d["y"], d["x"])  # n <= k -> SVD (least_squares.rs:225-229)
    assert np.allclose(c, np.linalg.lstsq(d["x"], d["y"], rcond=None)[0], atol=1e-8)
    d = make_data(n_samples=100, n_features=10, scale=1e-4)
    x = np.column_stack([d["x"], d["x"][:, -1] + 1e-12])
    c = orc.get_coefficients(d["y"], x, solve_method="svd")
    e = np.linalg.lstsq(x, d["y"], rcond=1e-16)[0]
    assert np.allclose(x @ c, x @ e, rtol=1e-4, atol=1e-4)


def test_empty_features_gives_zeros():
    assert np.array_equal(orc.get_coefficients(np.zeros(0), np.zeros((0, 3))), np.zeros(3))


@pytest.mark.parametrize("method", ["qr", "svd", "chol", "lu", None])
def test_configs0_oracle_matches_lstsq(method):
    """BASELINE configs[0] (ONE group, 10 000 rows x 4 f64 features, mode="coefficients"): the oracle's restatement of the
    dispatcher (src/expressions.rs:351-388) -> solve_ols / solve_ridge agrees with LAPACK's lstsq to 1e-12 for every
    solve_method."""
    d = make_data(n_samples=10_000, n_features=4)
    exp = np.linalg.lstsq(d["x"], d["y"], rcond=None)[0]
    got = orc.get_coefficients(d["y"], d["x"], solve_method=method)
    assert np.allclose(got, exp, rtol=1e-12, atol=1e-12), float(np.abs(got - exp).max())
    out = orc.batched_least_squares(d["y"], [d[f"x{i + 1}"] for i in range(4)], [0, 10_000], solve_method=method, want=("coef",))
    assert np.allclose(out["coef"][0], exp, rtol=1e-12, atol=1e-12)


# ----------------------------------------------------------------------------- demo-notebook KATs (reference-held vectors)
# notebooks/polars_ols_demo.ipynb: the printed outputs of cells 7 / 9 / 11 / 28 / 30 / 32 / 36 / 47 / 49 / 54 on the seeded frame
# of cell 1 (tests/golden/notebook_kat.json).  Printed to 6 decimals, so |oracle - printed| <= 0.5e-6 (+ rounding slack).

P6 = 0.6e-6


def _nan(a):
    return np.array([[np.nan if v is None else v for v in row] if isinstance(row, list) else (np.nan if row is None else row)
                     for row in a], dtype=np.float64)


def test_notebook_frame_is_the_printed_one(notebook):
    kat, d = notebook["kat"], notebook["d"]
    h, t = kat["cell53_head5"], kat["cell7_tail10"]
    for c in ("x1", "x2", "x3", "sample_weights"):
        assert np.allclose(d[c][:5], h[c], atol=P6) and np.allclose(d[c][-10:], t[c], atol=P6), c
    assert np.array_equal(d["group"][:5], h["group"])
    assert np.allclose(d["y"][-10:], t["y"], atol=P6)
    ys = np.column_stack([d["x1"] + d["x2"] + d["x3"], d["x1"] - d["x2"] + d["x3"], -d["x1"] + d["x2"] - d["x3"]])
    assert np.allclose(ys[:5], h["y_struct"], atol=P6)


def test_notebook_static_cells(notebook):
    kat, d = notebook["kat"], notebook["d"]
    x, y, w = d["x"], d["y"], d["sample_weights"]
    cols = [d["x1"], d["x2"], d["x3"]]
    order, offs, keys = sort_by_group(d["group"])
    inv = np.empty_like(order); inv[order] = np.arange(len(order))
    # cell 7
    t = kat["cell7_tail10"]
    g = orc.batched_least_squares(y[order], [c[order] for c in cols], offs, solve_method="svd", null_policy="drop")
    assert np.allclose(g["pred"][inv][-10:], t["predictions_ols_group"], atol=P6)
    f = orc.batched_least_squares(y, cols, [0, len(y)], solve_method="svd", null_policy="drop")
    assert np.allclose(f["pred"][-10:], t["predictions_ols"], atol=P6)
    wl = orc.batched_least_squares(y, cols, [0, len(y)], weights=w)
    assert np.allclose(wl["pred"][-10:] * (d["group"][-10:] == 2), t["predictions_wls_masked"], atol=P6)
    # cell 9 / 11 / 49
    c9 = orc.get_coefficients(y, np.column_stack([x, np.ones(len(y))]))
    assert np.allclose(c9[:2], kat["cell9_coefficients_first2"], atol=P6)
    g11 = orc.batched_least_squares(y[order], [c[order] for c in cols], offs, add_intercept=True)
    for gi, key in enumerate(keys):
        if str(key) in kat["cell11_coefficients_group"]:
            assert np.allclose(g11["coef"][gi], kat["cell11_coefficients_group"][str(key)], atol=P6)
    g49 = orc.batched_least_squares(y[order], [c[order] for c in cols[:2]], offs)
    for gi, key in enumerate(keys):
        assert np.allclose(g49["coef"][gi], kat["cell49_coefficients_group"][str(key)], atol=P6)
    # cell 36
    assert np.array_equal(orc.get_coefficients(y, x, alpha=0.0001, l1_ratio=0.5, positive=True), kat["cell36_enet_non_negative"])
    sw = np.sqrt(w)
    assert np.allclose(orc.get_coefficients(y * sw, x * sw[:, None], alpha=100.0), kat["cell36_ridge_alpha100_weighted"], atol=P6)


def test_notebook_collinear_cells(notebook):
    """Cells 26-34: x3 := x2 exactly, y := x1 + x2 + x3.  "qr" / default: the basic solution {1, 2, -0}; "chol" / "lu": Cholesky
    fails, LU meets an exactly zero pivot -> nulls; "svd": the minimum-norm {1, 1, 1}."""
    kat, d = notebook["kat"], notebook["d"]
    xc = np.column_stack([d["x1"], d["x2"], d["x2"]])
    yc = (xc[:, 0] + xc[:, 1]) + xc[:, 2]
    for m in ("qr", None):
        c = orc.get_coefficients(yc, xc, solve_method=m)
        assert np.allclose(c, kat["cell28_collinear_qr"], atol=1e-12)
        assert abs(np.linalg.norm(c) - kat["cell28_norm"]) < 1e-12
    for m in ("chol", "lu"):
        assert np.isnan(orc.get_coefficients(yc, xc, solve_method=m)).all()
    assert np.allclose(orc.get_coefficients(yc, xc, solve_method="svd"), kat["cell32_collinear_svd"], atol=1e-12)


def test_notebook_dynamic_cells(notebook):
    kat, d = notebook["kat"], notebook["d"]
    y = d["y"]
    cols = [d["x1"], d["x2"], d["x3"]]
    order, offs, _ = sort_by_group(d["group"])
    inv = np.empty_like(order); inv[order] = np.arange(len(order))
    r = orc.batched_rolling(y[order], [c[order] for c in cols], offs, window_size=252, min_periods=5, alpha=0.0001)["coef"][inv]
    assert np.isnan(r[:5]).all()
    assert np.allclose(r[-5:], kat["cell47_rolling_ridge_tail5"], atol=P6)
    rl = orc.batched_rls(y[order], [c[order] for c in cols], offs, half_life=21.0, initial_state_covariance=10.0,
                         initial_state_mean=[-1.0, -1.0, -1.0])["coef"][inv]
    assert np.allclose(rl[:5], kat["cell47_rls_head5"], atol=P6) and np.allclose(rl[-5:], kat["cell47_rls_tail5"], atol=P6)
    ex = orc.batched_rls(y, cols, [0, len(y)], half_life=None)["pred"]         # expanding_ols == rls(half_life=None), __init__.py:260-261
    assert np.allclose(ex[:5], kat["cell47_expanding_pred_head5"], atol=P6)
    assert np.allclose(ex[-5:], kat["cell47_expanding_pred_tail5"], atol=P6)


def test_notebook_multi_target_residuals(notebook):
    kat, d = notebook["kat"], notebook["d"]
    order, offs, _ = sort_by_group(d["group"])
    x = d["x"][order]
    sw = np.sqrt(d["sample_weights"][order])
    ys = np.column_stack([x[:, 0] + x[:, 1] + x[:, 2], x[:, 0] - x[:, 1] + x[:, 2], -x[:, 0] + x[:, 1] - x[:, 2]])
    for g in range(len(offs) - 1):
        s, e = offs[g], offs[g + 1]
        b = orc.solve_multi_target(ys[s:e] * sw[s:e, None], x[s:e] * sw[s:e, None])
        assert np.abs(ys[s:e] - x[s:e] @ b).max() < kat["cell54_multi_target_residual_bound"]
