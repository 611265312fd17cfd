#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.

Run in the BUILD container:  python tests/golden/make_golden.py

Two kinds of vectors:

1. ``readme_kat.json`` -- the literal known-answer outputs the REFERENCE prints in
   its own README (reference README.md:50-55 input frame; :72-76 lasso / WLS
   predictions; :104 full-sample coefficients; :112-113 per-group coefficients;
   :133-137 RLS coefficient path; :162-164 statistics table) and the literal 2x2
   Woodbury case of reference src/lib.rs:124-143.  These are data typed in from
   the reference's documentation, not computed here.

2. ``make_data_cases.npz`` -- the reference's own seeded fixture generator
   (``_make_data``, reference tests/test_ols.py:22-51, re-stated in
   tests/refdata.py) pushed through the SAME third-party numerics the
   reference's test-suite uses as its oracle (np.linalg.lstsq / solve,
   sklearn ElasticNet / Ridge; statsmodels is absent from the image so WLS uses
   lstsq on sqrt(w)-scaled data and RollingOLS is replaced by brute-force
   per-window lstsq).  Nothing from oracle/ or polars_ols_amd/ is used to make
   these numbers, so they pin both independently.

3. ``notebook_kat.json`` + ``notebook_frame.npz`` -- the outputs the REFERENCE prints in its demo notebook
   (reference notebooks/polars_ols_demo.ipynb cells 7, 9, 11, 28, 30, 32, 36, 47, 49, 53, 54) on the seeded frame of
   its cell 1 / 2 (``_make_data(n_samples=2_000, n_features=3, n_groups=5)``, re-stated as
   ``refdata.notebook_make_data``).  The printed values are typed in as data; the frame is regenerated from the seed and
   stored so that a numpy whose Generator stream ever changed would be noticed (the printed input rows of cells 7 / 53 pin it).

The reference itself cannot be imported here (needs polars + its Rust cdylib),
so no vector is produced by running reference code.
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))

from refdata import make_data, insert_nulls, notebook_make_data  # noqa: E402


def readme_kat() -> dict:
    return {
        "source": "reference README.md:50-55,72-76,104,112-113,133-137,162-164; src/lib.rs:124-143",
        "frame": {
            "y": [1.16, -2.16, -1.57, 0.21, 0.22, 1.6, -2.11, -2.92, -0.86, 0.47],
            "x1": [0.72, -2.43, -0.63, 0.05, -0.07, 0.65, -0.02, -1.64, -0.92, -0.27],
            "x2": [0.24, 0.18, -0.95, 0.23, 0.44, 1.01, -2.08, -1.36, 0.01, 0.75],
            "group": [1, 1, 1, 1, 1, 2, 2, 2, 2, 2],
            "weights": [0.34, 0.97, 0.39, 0.8, 0.57, 0.41, 0.19, 0.87, 0.06, 0.34],
        },
        # lasso("x1","x2", alpha=0.0001, add_intercept=True).over("group"), rounded to 2 dp, head(5)
        "predictions_lasso_head5_round2": [0.97, -2.23, -1.54, 0.29, 0.37],
        # from_formula("y ~ x1 + x2 -1", sample_weights=weights), rounded to 2 dp, head(5)
        "predictions_wls_head5_round2": [0.93, -2.18, -1.54, 0.27, 0.36],
        # from_formula("x1 + x2", mode="coefficients")  (x1, x2, const)
        "coefficients_full": [0.977375, 0.987413, 0.000757],
        "coefficients_group": {"1": [0.995157, 0.977495, 0.014344], "2": [0.939217, 0.997441, -0.017599]},
        # rls(x1, x2, mode="coefficients").over("group"), head(5)
        "rls_coefficients_head5": [[1.235503, 0.411834], [0.963515, 0.760769], [0.975484, 0.966029],
                                   [0.975657, 0.953735], [0.97898, 0.909793]],
        # ols(x1, x2, mode="statistics", add_intercept=True)
        "statistics": {
            "r2": 0.99631, "mae": 0.061732, "mse": 0.00794,
            "feature_names": ["x1", "x2", "const"],
            "coefficients": [0.977375, 0.987413, 0.000757],
            "standard_errors": [0.037286, 0.037321, 0.037474],
            "t_values": [26.212765, 26.457169, 0.02021],
            "p_values": [3.0095e-8, 2.8218e-8, 0.98444],
        },
        "woodbury": {"a": [[0.5, 0.2], [0.0, 0.5]], "u": [[1.0, 2.0], [3.0, 4.0]],
                     "c": [[1.0, 0.0], [0.0, 1.0]], "v": [[1.0, 0.0], [0.0, 1.0]]},
    }


def notebook_kat() -> dict:
    """Typed in from the output cells of reference notebooks/polars_ols_demo.ipynb (6 printed digits)."""
    nan = None
    return {
        "source": "reference notebooks/polars_ols_demo.ipynb, output cells 7, 9, 11, 28, 30, 32, 36, 47, 49, 53, 54",
        "frame": {"n_samples": 2000, "n_features": 3, "n_groups": 5, "file": "notebook_frame.npz"},
        # cell 53 head(5): the input rows as printed (x1, x2, x3, group, sample_weights); y struct = (x1+x2+x3, x1-x2+x3, -x1+x2-x3)
        "cell53_head5": {
            "x1": [0.12573, 0.1049, 1.304, -1.265421, -2.325031],
            "x2": [-0.132105, -0.535669, 0.947081, -0.623274, -0.218792],
            "x3": [0.640423, 0.361595, -0.703735, 0.041326, -1.245911],
            "group": [1, 0, 4, 3, 3],
            "sample_weights": [0.709215, 0.731333, 0.701351, 0.052596, 0.963254],
            "y_struct": [[0.634048, 0.898258, -0.898258], [-0.069174, 1.002165, -1.002165], [1.547346, -0.346816, 0.346816],
                         [-1.84737, -0.600821, 0.600821], [-3.789733, -3.35215, 3.35215]],
        },
        # cell 7 tail(10): inputs as printed + ols(svd, null_policy="drop").over("group"), the same over the whole frame,
        # and wls(sample_weights) * (group == 2)
        "cell7_tail10": {
            "x1": [-0.583369, 0.71304, 1.098849, -0.485594, 0.949438, 1.057735, -0.122949, -0.491295, -0.226812, 0.159845],
            "x2": [0.890726, 1.751887, 0.463944, -0.315542, 1.029228, 0.268385, 2.002523, 0.870951, 0.740164, -0.226334],
            "x3": [0.497755, -0.223204, -0.451817, 0.096269, 0.318868, 0.350553, 1.63392, 0.24026, 0.180547, -0.093559],
            "y": [-0.70099, -2.230821, -1.116165, 0.866697, -2.197234, -1.559323, -3.658936, -0.552929, -0.599317, 0.203998],
            "sample_weights": [0.871927, 0.776195, 0.01473, 0.507687, 0.062403, 0.559756, 0.585527, 0.367483, 0.119438, 0.082505],
            "predictions_ols_group": [-0.800004, -2.241771, -1.111206, 0.70668, -2.295554, -1.67483, -3.503794, -0.617277, -0.691848, 0.15888],
            "predictions_ols": [-0.802822, -2.239055, -1.110725, 0.704341, -2.294688, -1.674932, -3.506601, -0.6182, -0.692402, 0.159546],
            "predictions_wls_masked": [-0.800443, -0.0, -1.11138, 0.0, -0.0, -0.0, -0.0, -0.0, -0.0, 0.158951],
        },
        # cell 9: ols(x*, add_intercept=True, mode="coefficients"); the display shows the first two fields
        "cell9_coefficients_first2": [-0.999496, -0.998398],
        # cell 11: ols("x1","x2","x3", add_intercept=True, mode="coefficients").over("group"), (x1, x2, x3, const) by group key
        "cell11_coefficients_group": {"1": [-0.993655, -0.997968, -1.006029, 0.003129], "0": [-1.002313, -1.0001, -0.993786, -0.006384],
                                      "4": [-1.000859, -0.998017, -0.998356, 0.004797], "3": [-0.999962, -0.999101, -0.995044, 0.004376]},
        # cells 26-32: x3 := x2 (exact copy), y := x1 + x2 + x3 (sum_horizontal, left to right)
        "cell28_collinear_qr": [1.0, 2.0, -0.0],
        "cell28_norm": 2.23606797749979,
        "cell30_collinear_chol": [nan, nan, nan],
        "cell32_collinear_svd": [1.0, 1.0, 1.0],
        # cell 36: elastic_net(alpha=0.0001, l1_ratio=0.5, positive=True) and ridge(alpha=100, sample_weights)
        "cell36_enet_non_negative": [0.0, 0.0, 0.0],
        "cell36_ridge_alpha100_weighted": [-0.912021, -0.898583, -0.908957],
        # cell 47: rolling_ols(window_size=252, min_periods=5, alpha=0.0001).over("group") | rls(half_life=21, initial_state_mean=
        # [-1,-1,-1], initial_state_covariance=10).over("group") | expanding_ols(...) predictions over the whole frame
        "cell47_rolling_ridge_head5": [[nan, nan, nan]] * 5,
        "cell47_rolling_ridge_tail5": [[-0.995486, -1.003153, -0.998155], [-0.995348, -1.004411, -0.99898], [-1.001142, -1.002267, -1.002736],
                                       [-1.001257, -1.001316, -1.003019], [-1.002536, -0.993688, -0.998938]],
        "cell47_rls_head5": [[-1.031467, -0.966938, -1.160281], [-1.013228, -0.932454, -1.045596], [-1.003344, -1.002429, -0.998195],
                             [-1.053291, -1.026248, -0.99826], [-1.059251, -1.017933, -1.014118]],
        "cell47_rls_tail5": [[-0.987406, -1.004047, -1.004306], [-0.985392, -1.011036, -1.00789], [-1.001857, -0.980484, -1.007321],
                             [-1.002182, -0.978417, -1.006454], [-0.968934, -1.00696, -1.010813]],
        "cell47_expanding_pred_head5": [-0.627675, -0.127212, -1.48872, 1.852231, 3.941405],
        "cell47_expanding_pred_tail5": [-1.674772, -3.506569, -0.618197, -0.692359, 0.159536],
        # cell 49: ols("x1","x2", mode="coefficients").over("group") by group key
        "cell49_coefficients_group": {"2": [-0.948578, -0.917406], "1": [-1.094156, -0.946028], "0": [-1.028893, -1.004733],
                                      "3": [-1.074243, -1.050159], "4": [-1.037308, -0.981468]},
        # cell 54: multi_target_ols(x1, x2, x3, sample_weights, mode="residuals").over("group"): every printed residual is ~1e-16
        "cell54_multi_target_residual_bound": 1e-14,
    }


def rolling_bruteforce(y, x, window, min_periods):
    """statsmodels RollingOLS(window, min_nobs, expanding=True, missing='drop') + forward fill,
    re-stated with per-window lstsq (reference tests/test_ols.py:751-764)."""
    n, k = x.shape
    valid = ~np.isnan(y) & ~np.isnan(x).any(axis=1)
    out = np.full((n, k), np.nan)
    last = None
    for i in range(n):
        lo = max(0, i - window + 1)
        idx = np.arange(lo, i + 1)
        idx = idx[valid[idx]]
        if len(idx) >= min_periods:
            last = np.linalg.lstsq(x[idx], y[idx], rcond=None)[0]
        if last is not None:
            out[i] = last
    return out


def main() -> None:
    from sklearn.linear_model import ElasticNet, Ridge

    out = {}

    # --- test_ols (tests/test_ols.py:54-73): n=1000, k=2, lstsq predictions
    d = make_data(n_samples=1_000, n_features=2)
    x, y = d["x"], d["y"]
    coef = np.linalg.lstsq(x, y, rcond=None)[0]
    out["ols_x"], out["ols_y"], out["ols_coef"], out["ols_pred"] = x, y, coef, x @ coef

    # --- test_ridge (:475-503): n=5000, alpha=0.01
    d = make_data()
    x, y = d["x"], d["y"]
    out["ridge_x"], out["ridge_y"] = x, y
    out["ridge_coef_chol"] = np.linalg.solve(x.T @ x + np.eye(2) * 0.01, x.T @ y)
    out["ridge_coef_svd"] = Ridge(fit_intercept=False, solver="svd", alpha=0.01).fit(x, y).coef_
    # ridge alpha=10 (src/lib.rs:57-65 scenario shape)
    out["ridge_coef_alpha10"] = np.linalg.solve(x.T @ x + np.eye(2) * 10.0, x.T @ y)

    # --- test_least_squares_from_formula / test_wls (:456-472,:506-541): WLS + intercept
    rng = np.random.default_rng(1)
    w = rng.uniform(0, 1, size=5_000)
    w /= w.mean()
    xi = np.column_stack([x, np.ones(len(y))])
    sw = np.sqrt(w)
    coef = np.linalg.lstsq(xi * sw[:, None], y * sw, rcond=None)[0]
    out["wls_w"], out["wls_coef"], out["wls_pred"] = w, coef, xi @ coef

    # --- test_ols_intercept (:446-453)
    coef = np.linalg.lstsq(xi, y, rcond=None)[0]
    out["intercept_coef"], out["intercept_pred"] = coef, xi @ coef

    # --- test_elastic_net (:561-599) & non-negative (:602-630); sklearn solved to tight tol
    for name, (k, sparsity, alpha) in {"enet2": (2, 0.0, 0.1), "enet100": (100, 0.5, 0.3)}.items():
        d = make_data(n_features=k, sparsity=sparsity)
        mdl = ElasticNet(fit_intercept=False, alpha=alpha, l1_ratio=0.5, max_iter=100_000, tol=1e-13)
        mdl.fit(d["x"], d["y"])
        out[f"{name}_coef"] = mdl.coef_
        out[f"{name}_pred"] = mdl.predict(d["x"])
    d = make_data()
    xn = np.column_stack([d["x"][:, 0], -d["x"][:, 1]])
    mdl = ElasticNet(fit_intercept=False, alpha=0.1, l1_ratio=0.5, max_iter=100_000, tol=1e-13, positive=True)
    mdl.fit(xn, d["y"])
    out["nnls_coef"] = mdl.coef_

    # --- grouped (:377-401): n=5000, k=2, 10 non-contiguous groups -> per-group lstsq
    d = make_data(n_groups=10)
    coefs = np.zeros((10, 2))
    for g in range(10):
        m = d["group"] == g
        coefs[g] = np.linalg.lstsq(d["x"][m], d["y"][m], rcond=None)[0]
    out["group_coef"] = coefs

    # --- null policies (:130-249): drop / drop_zero / drop_y_zero_x predictions
    d = insert_nulls(make_data(), columns=("x1", "x2", "y"))
    x, y = d["x"], d["y"]
    out["nulls_x"], out["nulls_y"] = x, y
    for pol in ("drop", "drop_zero", "drop_y_zero_x"):
        if pol == "drop_y_zero_x":
            ok = ~np.isnan(y)
            xf, yf = np.nan_to_num(x[ok]), y[ok]
        else:
            ok = ~np.isnan(x).any(axis=1) & ~np.isnan(y)
            xf, yf = x[ok], y[ok]
        c = np.linalg.lstsq(xf, yf, rcond=None)[0]
        pred = np.nan_to_num(x) @ c
        if pol == "drop":
            pred[~ok] = np.nan
        out[f"nulls_{pol}_coef"], out[f"nulls_{pol}_pred"] = c, pred

    # --- expanding RLS == OLS at the last row (:633-681), nulls in all columns, P0 = 1e6
    ok = ~np.isnan(x).any(axis=1) & ~np.isnan(y)
    out["rls_expanding_last"] = np.linalg.lstsq(x[ok], y[ok], rcond=None)[0]

    # --- rolling, drop_window (:718-772): n=1000, nulls in y only
    d = insert_nulls(make_data(n_samples=1_000), columns=("y",))
    out["roll_x"], out["roll_y"] = d["x"], d["y"]
    for (win, mp) in [(2, 2), (10, 2), (63, 5), (252, 5)]:
        out[f"roll_{win}_{mp}"] = rolling_bruteforce(d["y"], d["x"], win, mp)

    # --- statistics (:998-1029): OLS + intercept on make_data(); closed-form numpy/scipy
    from scipy import stats as sps
    d = make_data()
    xi = np.column_stack([d["x"], np.ones(5_000)])
    y = d["y"]
    c = np.linalg.lstsq(xi, y, rcond=None)[0]
    res = y - xi @ c
    dfree = len(y) - 3
    cov = np.linalg.inv(xi.T @ xi) * (res @ res / dfree)
    se = np.sqrt(np.diag(cov))
    out["stats_coef"], out["stats_se"], out["stats_t"] = c, se, c / se
    out["stats_p"] = 2 * sps.t.sf(np.abs(c / se), dfree)
    out["stats_r2_mse"] = np.array([1 - res @ res / ((y - y.mean()) @ (y - y.mean())), (res ** 2).mean()])

    np.savez_compressed(HERE / "make_data_cases.npz", **out)
    (HERE / "readme_kat.json").write_text(json.dumps(readme_kat(), indent=1))
    (HERE / "notebook_kat.json").write_text(json.dumps(notebook_kat(), indent=1))
    nb = notebook_make_data(n_samples=2_000, n_features=3, n_groups=5)
    np.savez_compressed(HERE / "notebook_frame.npz", x=nb["x"], y=nb["y"], group=nb["group"], sample_weights=nb["sample_weights"])
    print("wrote", HERE / "make_data_cases.npz", "and readme_kat.json;", len(out), "arrays")


if __name__ == "__main__":
    main()
