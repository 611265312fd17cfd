"""K7 (mode="statistics", src/statistics.rs + src/expressions.rs:468-509) through the C-ABI against the oracle and the
reference's README / test_ols.py known answers; multi-target regression (src/expressions.rs:521-591) through the mirror
of the reference's namespace."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import orc  # noqa: E402
from refdata import insert_nulls, make_data, synthetic_groups  # noqa: E402

@pytest.fixture(scope="module")
def engine():
    from polars_ols_amd import Engine

    e = Engine(0)
    yield e
    e.close()


KEYS = ("r2", "mae", "mse")
MATS = (("std_err", "standard_errors"), ("t_values", "t_values"), ("p_values", "p_values"))


def _oracle_stats(d, weights=None, add_intercept=False, **kw):
    """Per-group loop over the oracle on the sqrt(w)-scaled rows (least_squares.py:190-196)."""
    offs = d["offsets"]
    out = {k: [] for k in KEYS + ("coef", "standard_errors", "t_values", "p_values")}
    for g in range(len(offs) - 1):
        s, e = offs[g], offs[g + 1]
        x = np.column_stack([c[s:e] for c in d["cols"]]).astype(np.float64)
        y = d["y"][s:e].astype(np.float64)
        if add_intercept:
            x = np.column_stack([x, np.ones(e - s)])
        if weights is not None:
            sw = np.sqrt(weights[s:e].astype(np.float64))
            x, y = x * sw[:, None], y * sw
        alpha = kw.get("alpha", 0.0)
        coef = orc.get_coefficients(y, x, **kw)
        st = orc.statistics(y, x, alpha=alpha)
        # orc.statistics solves with the default dispatcher at (alpha); residual metrics must use the requested model
        pred = x @ coef
        err = y - pred
        out["r2"].append(1.0 - (err ** 2).sum() / ((y - y.mean()) ** 2).sum())
        out["mae"].append(np.abs(err).mean())
        out["mse"].append((err ** 2).mean())
        out["coef"].append(coef)
        for k in ("standard_errors", "t_values", "p_values"):
            out[k].append(st[k])
    return {k: np.array(v) for k, v in out.items()}


def _check(res, exp, rtol, atol):
    for k in KEYS:
        assert np.allclose(np.asarray(res[k]), exp[k], rtol=rtol, atol=atol), k
    assert np.allclose(np.asarray(res["coef"], dtype=np.float64), exp["coef"], rtol=rtol, atol=atol)
    for mine, ref in MATS:
        assert np.allclose(np.asarray(res[mine]), exp[ref], rtol=rtol, atol=atol, equal_nan=True), mine


def _ragged(seed, dtype, G=37, k=4, lo=12, hi=700):
    rng = np.random.default_rng(seed)
    sizes = rng.integers(lo, hi, size=G)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(offs[-1])
    cols = [rng.normal(size=n).astype(dtype) for _ in range(k)]
    y = (sum(cols) + 0.3 * rng.normal(size=n) + 0.5).astype(dtype)
    w = rng.uniform(0.2, 2.0, size=n).astype(dtype)
    return {"y": y, "cols": cols, "offsets": offs, "w": w}


@pytest.mark.parametrize("dtype,rtol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("weights", [False, True])
@pytest.mark.parametrize("add_intercept", [False, True])
@pytest.mark.parametrize("alpha", [0.0, 2.5])
def test_statistics_vs_oracle(engine, dtype, rtol, weights, add_intercept, alpha):
    d = _ragged(5, dtype)
    w = d["w"] if weights else None
    res = engine.least_squares_statistics(d["y"], d["cols"], d["offsets"], weights=w, add_intercept=add_intercept, alpha=alpha)
    exp = _oracle_stats(d, weights=w, add_intercept=add_intercept, alpha=alpha)
    _check(res, exp, rtol, rtol)
    assert (np.asarray(res["status"]) == 0).all()


def test_statistics_device_matches_host(engine):
    import torch

    d = _ragged(9, np.float64)
    host = engine.least_squares_statistics(d["y"], d["cols"], d["offsets"], add_intercept=True)
    dev = engine.least_squares_statistics(torch.from_numpy(d["y"]).cuda(), [torch.from_numpy(c).cuda() for c in d["cols"]],
                                          d["offsets"], add_intercept=True)
    torch.cuda.synchronize()
    for k in KEYS + ("std_err", "t_values", "p_values", "coef"):
        assert np.array_equal(np.asarray(host[k]), dev[k].cpu().numpy(), equal_nan=True), k


def test_statistics_wide_features_and_large_groups(engine):
    d = synthetic_groups(6, 9_000, 20, seed=3, dtype=np.float64)      # streamed path, 21 columns with the intercept
    res = engine.least_squares_statistics(d["y"], d["cols"], d["offsets"], add_intercept=True)
    exp = _oracle_stats(d, add_intercept=True)
    _check(res, exp, 1e-6, 1e-6)


@pytest.mark.parametrize("dtype,rtol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("k,weights,add_intercept,alpha", [(32, False, False, 0.0), (60, True, True, 1.5), (126, False, True, 0.0)])
def test_statistics_wide_32_to_127_columns(engine, dtype, rtol, k, weights, add_intercept, alpha):
    """The K8 kernels solve, the wide statistics kernel works from their Gram matrix (sweep-operator inverse in LDS)."""
    d = _ragged(40 + k, dtype, G=5, k=k, lo=3 * k, hi=6 * k)
    w = d["w"] if weights else None
    res = engine.least_squares_statistics(d["y"], d["cols"], d["offsets"], weights=w, add_intercept=add_intercept, alpha=alpha)
    exp = _oracle_stats(d, weights=w, add_intercept=add_intercept, alpha=alpha)
    _check(res, exp, rtol, rtol)
    assert (np.asarray(res["status"]) == 0).all()


@pytest.mark.parametrize("k,weights,add_intercept,alpha,G", [(127, False, True, 0.0, 3), (200, True, False, 0.5, 3), (640, False, True, 0.0, 2)])
def test_statistics_beyond_127_columns(engine, k, weights, add_intercept, alpha, G):
    """128 .. 1 024 columns: the same sweep-operator inverse, the matrix in an HBM / L2 work area owned by one workgroup."""
    d = _ragged(400 + k, np.float64, G=G, k=k, lo=2 * k + 50, hi=3 * k)
    w = d["w"] if weights else None
    res = engine.least_squares_statistics(d["y"], d["cols"], d["offsets"], weights=w, add_intercept=add_intercept, alpha=alpha)
    exp = _oracle_stats(d, weights=w, add_intercept=add_intercept, alpha=alpha)
    _check(res, exp, 1e-6, 1e-6)
    assert (np.asarray(res["status"]) == 0).all()


def test_statistics_elastic_net_uses_alpha_as_lambda(engine):
    d = _ragged(11, np.float64, G=9)
    res = engine.least_squares_statistics(d["y"], d["cols"], d["offsets"], alpha=0.01, l1_ratio=0.5)
    exp = _oracle_stats(d, alpha=0.01, l1_ratio=0.5)
    _check(res, exp, 1e-5, 1e-6)


def test_statistics_failed_inverse_gives_nan(engine):                # src/statistics.rs:101-111
    d = _ragged(13, np.float64, G=5, k=3)
    d["cols"].append(np.zeros_like(d["cols"][0]))                     # an all-zero feature: X'X is singular
    res = engine.least_squares_statistics(d["y"], d["cols"], d["offsets"])
    for k in ("std_err", "t_values", "p_values"):
        assert np.isnan(res[k]).all()
    exp = _oracle_stats({**d, "cols": d["cols"][:3]})
    assert np.allclose(res["r2"], exp["r2"], rtol=1e-6) and np.allclose(res["mse"], exp["mse"], rtol=1e-6)
    assert np.allclose(res["coef"][:, :3], exp["coef"], rtol=1e-6, atol=1e-9) and np.allclose(res["coef"][:, 3], 0.0)


def test_statistics_bad_degrees_of_freedom_is_marked(engine):        # src/statistics.rs:131-134 asserts df > 0
    d = _ragged(17, np.float64, G=4, k=3, lo=30, hi=60)
    res = engine.least_squares_statistics(d["y"], d["cols"], d["offsets"], alpha=1e-3)
    assert (np.asarray(res["status"]) == 0).all()
    tiny = {"y": d["y"][:3], "cols": [c[:3] for c in d["cols"]], "offsets": np.array([0, 3], dtype=np.int64)}
    res = engine.least_squares_statistics(tiny["y"], tiny["cols"], tiny["offsets"], alpha=1e-9)   # df = 3 - trace(inv) < 0
    assert int(res["status"][0]) == 4 and np.isnan(res["std_err"]).all()


def test_readme_statistics_kat(golden):                              # README.md:145-152
    from polars_ols_amd import Frame, col

    kat = golden["kat"]
    f = Frame({k: np.array(v, dtype=np.float64) for k, v in kat["frame"].items() if k in ("y", "x1", "x2")})
    s = kat["statistics"]
    st = f.select(col("y").least_squares.ols(col("x1"), col("x2"), mode="statistics", add_intercept=True))["statistics"]
    assert st["feature_names"] == s["feature_names"]
    assert np.round(st["r2"][0], 5) == s["r2"] and np.round(st["mae"][0], 6) == s["mae"] and np.round(st["mse"][0], 5) == s["mse"]
    assert np.allclose(np.round(st["coefficients"][0], 6), s["coefficients"], atol=1.1e-6)
    assert np.allclose(np.round(st["standard_errors"][0], 6), s["standard_errors"], atol=1.1e-6)
    assert np.allclose(np.round(st["t_values"][0], 6), s["t_values"], atol=1.1e-5)
    assert np.allclose(st["p_values"][0], s["p_values"], rtol=1e-4)


def test_least_squares_statistics_reference_case(golden):            # tests/test_ols.py:998-1030 (statsmodels values)
    from polars_ols_amd import Frame, col

    z = golden["npz"]
    d = make_data()
    df = Frame({k: v for k, v in d.items() if k != "x"})
    st = df.select(col("y").least_squares.ols(col("x1"), col("x2"), mode="statistics", add_intercept=True))["statistics"]
    assert np.allclose(st["coefficients"][0], z["stats_coef"]) and np.allclose(st["standard_errors"][0], z["stats_se"])
    assert np.allclose(st["t_values"][0], z["stats_t"]) and np.allclose(st["p_values"][0], z["stats_p"], rtol=1e-6, atol=1e-300)
    assert np.allclose([st["r2"][0], st["mse"][0]], z["stats_r2_mse"])


def test_statistics_over_groups_and_null_policy():
    from polars_ols_amd import Frame, col

    d = make_data(n_groups=7)
    dn = insert_nulls(d, ["y", "x1"], 0.05, seed=1)
    df = Frame({k: v for k, v in dn.items() if k != "x"})
    st = df.select(col("y").least_squares.ols(col("x1"), col("x2"), mode="statistics", null_policy="drop").over("group"))["statistics"]
    assert st["r2"].shape == (7,) and st["coefficients"].shape == (7, 2)
    for gi, key in enumerate(st.keys_):
        m = (dn["group"] == key) & ~np.isnan(dn["y"]) & ~np.isnan(dn["x1"]) & ~np.isnan(dn["x2"])
        x = np.column_stack([dn["x1"][m], dn["x2"][m]])
        ref = orc.statistics(dn["y"][m], x)
        assert np.allclose(st["standard_errors"][gi], ref["standard_errors"], rtol=1e-6)
        assert np.allclose(st["t_values"][gi], ref["t_values"], rtol=1e-6)
        assert np.isclose(st["r2"][gi], ref["r2"], rtol=1e-6)


def test_statistics_rejected_for_dynamic_models():                   # least_squares.py:357, 394
    from polars_ols_amd import col

    with pytest.raises(AssertionError):
        col("y").least_squares.rls(col("x1"), mode="statistics")
    with pytest.raises(AssertionError):
        col("y").least_squares.rolling_ols(col("x1"), window_size=5, mode="statistics")


# ------------------------------------------------------------------------------------------------ multi-target
@pytest.mark.parametrize("alpha", [0.0, 0.1])
@pytest.mark.parametrize("mode", ["predictions", "residuals"])
def test_multi_target_regression(alpha, mode):                       # tests/test_ols.py:80-119
    from polars_ols_amd import Frame, OLSKwargs, col, compute_multi_target_least_squares, struct

    d = make_data(n_samples=2_000, n_features=3)
    rng = np.random.default_rng(4)
    y2 = d["x"] @ np.array([0.5, -1.0, 2.0]) + 0.1 * rng.normal(size=2_000)
    df = Frame({**{k: v for k, v in d.items() if k != "x"}, "y2": y2})
    out = df.select(compute_multi_target_least_squares(struct("y", "y2"), col("x1"), col("x2"), col("x3"), mode=mode,
                                                       ols_kwargs=OLSKwargs(alpha=alpha)).alias("predictions"))["predictions"]
    for name, y in (("y", d["y"]), ("y2", y2)):
        coef = np.linalg.solve(d["x"].T @ d["x"] + alpha * np.eye(3), d["x"].T @ y)
        exp = d["x"] @ coef
        assert np.allclose(out[name], exp if mode == "predictions" else y - exp, rtol=1e-6, atol=1e-6)


def test_multi_target_joint_validity_mask():                         # src/expressions.rs:539 (mask over ALL targets)
    from polars_ols_amd import Frame, col, struct

    d = make_data(n_samples=1_000, n_features=2)
    rng = np.random.default_rng(8)
    y2 = d["x"] @ np.array([2.0, -1.0]) + 0.1 * rng.normal(size=1_000)
    y1 = d["y"].copy()
    y1[rng.random(1_000) < 0.1] = np.nan
    y2[rng.random(1_000) < 0.1] = np.nan
    df = Frame({"y": y1, "y2": y2, "x1": d["x1"], "x2": d["x2"]})
    out = df.select(struct("y", "y2").least_squares.multi_target_ols(col("x1"), col("x2"), null_policy="drop"))["predictions"]
    valid = ~np.isnan(y1) & ~np.isnan(y2)
    for name, y in (("y", y1), ("y2", y2)):
        coef = np.linalg.lstsq(d["x"][valid], y[valid], rcond=None)[0]
        assert np.allclose(out[name][valid], d["x"][valid] @ coef, rtol=1e-6, atol=1e-6)
        assert np.isnan(out[name][~valid]).all()


def test_multi_target_rejects_unsupported():                         # least_squares.py:303-318
    from polars_ols_amd import OLSKwargs, col, compute_multi_target_least_squares, struct

    with pytest.raises(AssertionError):
        compute_multi_target_least_squares(struct("y", "y2"), col("x1"), ols_kwargs=OLSKwargs(alpha=0.1, l1_ratio=0.5))
    with pytest.raises(AssertionError):
        compute_multi_target_least_squares(struct("y", "y2"), col("x1"), ols_kwargs=OLSKwargs(solve_method="chol"))
    with pytest.raises(NotImplementedError):
        compute_multi_target_least_squares(struct("y", "y2"), col("x1"), mode="coefficients")


@pytest.mark.parametrize("dtype,rtol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("policy", ["drop", "zero", "drop_y_zero_x", "drop_zero"])
@pytest.mark.parametrize("weights,add_intercept,k", [(False, False, 4), (True, True, 4), (False, True, 36)])
def test_statistics_null_policies_behind_the_cabi(engine, dtype, rtol, policy, weights, add_intercept, k):
    """handle_nulls ahead of the statistics code (src/expressions.rs:469-471, 255-296): the entry filters / zero-fills the rows
    on the device (csrc/dyn_prep.hip: compact_*); expectation = the oracle on the numpy-filtered frame.  Host and device batches."""
    import torch

    d = _ragged(70 + k, dtype, G=11, k=k, lo=4 * k, hi=9 * k)
    rng = np.random.default_rng(k)
    n = len(d["y"])
    d["y"][rng.random(n) < 0.04] = np.nan
    for c in d["cols"][:3]:
        c[rng.random(n) < 0.03] = np.nan
    w = d["w"] if weights else None
    # the reference's filter, in numpy
    null_y = np.isnan(d["y"])
    null_x = np.zeros(n, dtype=bool)
    for c in d["cols"]:
        null_x |= np.isnan(c)
    keep = ~null_y & ~null_x if policy in ("drop", "drop_zero") else (~null_y if policy == "drop_y_zero_x" else np.ones(n, dtype=bool))
    gid = np.repeat(np.arange(len(d["offsets"]) - 1), np.diff(d["offsets"]))
    f = {"y": d["y"][keep], "cols": [c[keep] for c in d["cols"]],
         "offsets": np.concatenate([[0], np.cumsum(np.bincount(gid[keep], minlength=len(d["offsets"]) - 1))]).astype(np.int64)}
    if policy in ("zero", "drop_y_zero_x"):
        f["y"], f["cols"] = np.nan_to_num(f["y"]), [np.nan_to_num(c) for c in f["cols"]]
    exp = _oracle_stats(f, weights=None if w is None else w[keep], add_intercept=add_intercept)
    res = engine.least_squares_statistics(d["y"], d["cols"], d["offsets"], weights=w, add_intercept=add_intercept, null_policy=policy)
    _check(res, exp, rtol, rtol)
    t = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    dev = engine.least_squares_statistics(t(d["y"]), [t(c) for c in d["cols"]], d["offsets"], weights=None if w is None else t(w),
                                          add_intercept=add_intercept, null_policy=policy)
    torch.cuda.synchronize()
    _check({k_: v.cpu().numpy() for k_, v in dev.items()}, exp, rtol, rtol)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("sizes,k,weights,icpt,alpha", [([600_000], 6, False, True, 0.0), ([250_000, 40, 0, 180_001], 4, True, False, 0.5)])
def test_statistics_of_long_groups_run_per_segment(engine, dtype, sizes, k, weights, icpt, alpha):
    """ONE model summary over a whole frame: the row passes of the statistics run one workgroup per segment (shifted one-pass sums,
    added per group in segment order).  Against the unsplit kernel (NO_SPLIT) on the same frame and against the oracle."""
    import torch

    eng = engine

    def _cuda(x):
        return torch.from_numpy(np.ascontiguousarray(x)).cuda()

    rng = np.random.default_rng(len(sizes) + k)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(offs[-1])
    cols = [rng.standard_normal(N).astype(dtype) for _ in range(k)]
    y = (sum(c.astype(np.float64) for c in cols) + 3.0 + 0.5 * rng.standard_normal(N)).astype(dtype)
    w = rng.uniform(0.3, 2.0, N).astype(dtype) if weights else None
    kw = dict(weights=None if w is None else _cuda(w), add_intercept=icpt, alpha=alpha, l1_ratio=0.0 if alpha else None)
    kw = {a: b for a, b in kw.items() if b is not None}
    out = eng.least_squares_statistics(_cuda(y), [_cuda(c) for c in cols], offs, **kw)
    eng.set_option("NO_SPLIT", "1")
    try:
        one = eng.least_squares_statistics(_cuda(y), [_cuda(c) for c in cols], offs, **kw)
    finally:
        eng.set_option("NO_SPLIT", None)
    tol = 1e-6 if dtype == np.float64 else 2e-4
    live = np.asarray(sizes) > k + 2
    for key in ("r2", "mae", "mse", "std_err", "t_values", "p_values"):
        f = lambda v: v.double().cpu().numpy() if hasattr(v, "cpu") else np.asarray(v, dtype=np.float64)  # noqa: E731
        a, b2 = f(out[key])[live], f(one[key])[live]
        assert np.allclose(a, b2, rtol=tol, atol=tol, equal_nan=True), (key, a, b2)


@pytest.mark.parametrize("split", [True, False])
def test_f32_multi_million_row_group_with_nonzero_means(engine, split):
    """ONE f32 group of 4 000 000 rows whose columns have means of 1 .. 3 (nearly every product is positive: an f32 running sum of a million of
    them loses five digits) through least_squares and least_squares_statistics, cut into segments and unsplit (NO_SPLIT: the whole group is
    one item of the VALU Gram kernel, k5v_gram.hip) -- the lanes' f32 partials are flushed into f64 every 128 rows, so the coefficients
    hold north_star's f32 bound of 1e-4 against the f64 oracle on the same f32 data."""
    import torch

    eng = engine
    rng = np.random.default_rng(77)
    n, k = 4_000_000, 6
    cols = [(rng.standard_normal(n) + 1.0 + 0.4 * j).astype(np.float32) for j in range(k)]
    beta = np.linspace(0.5, 1.5, k)
    y = (sum(b * c.astype(np.float64) for b, c in zip(beta, cols)) + 2.0 + rng.standard_normal(n)).astype(np.float32)
    offs = np.array([0, n], dtype=np.int64)
    X = np.column_stack([c.astype(np.float64) for c in cols] + [np.ones(n)])
    ref = np.linalg.lstsq(X, y.astype(np.float64), rcond=None)[0]
    dy, dc = torch.from_numpy(y).cuda(), [torch.from_numpy(c).cuda() for c in cols]
    eng.set_option("NO_SPLIT", None if split else "1")
    try:
        out = eng.least_squares(dy, dc, offs, add_intercept=True, want=("coef", "pred"))
        name = eng.last_kernel
        st = eng.least_squares_statistics(dy, dc, offs, add_intercept=True)
    finally:
        eng.set_option("NO_SPLIT", None)
    coef = out["coef"].double().cpu().numpy()[0]
    assert np.allclose(coef, ref, rtol=1e-4, atol=1e-4), (name, coef, ref)
    pred = out["pred"].double().cpu().numpy()
    assert np.allclose(pred, X @ ref, rtol=1e-4, atol=1e-3)
    sc = st["coef"].double().cpu().numpy()[0] if hasattr(st["coef"], "cpu") else np.asarray(st["coef"], dtype=np.float64)[0]
    assert np.allclose(sc, ref, rtol=1e-4, atol=1e-4), (sc, ref)
    resid = y.astype(np.float64) - X @ ref
    r2 = 1.0 - (resid ** 2).sum() / ((y.astype(np.float64) - y.astype(np.float64).mean()) ** 2).sum()
    got_r2 = float(st["r2"].cpu().numpy()[0]) if hasattr(st["r2"], "cpu") else float(np.asarray(st["r2"])[0])
    assert abs(got_r2 - r2) < 1e-4
