"""GPU parity of K5 (elastic net: streamed MFMA Gram + Gram-form coordinate descent + prediction pass) and of
pols_predict, through the C-ABI, against the CPU oracle (residual-form CD of src/least_squares.rs:386-492), the
README lasso known answer and the sklearn golden fixtures."""
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


def _frame(rng, offs, k, dtype, sparsity=0.5, weights=False):
    N = int(offs[-1])
    cols = [rng.standard_normal(N).astype(dtype) for _ in range(k)]
    kk = max(1, int(k * (1 - sparsity)))
    y = (sum(c.astype(np.float64) for c in cols[:kk]) + 0.1 * rng.standard_normal(N)).astype(dtype)
    w = rng.uniform(0.2, 2.0, N).astype(dtype) if weights else None
    return y, cols, w


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("k,alpha,l1,positive,method,weights,icpt", [
    (2, 0.1, 0.5, False, "cd", False, False),
    (8, 0.05, 1.0, False, None, False, True),          # lasso + intercept
    (8, 0.3, 0.5, True, "cd", True, False),            # non-negative + weights
    (15, 0.01, 0.5, False, "cd_active_set", False, True),
    (16, 0.001, 0.5, False, "cd", False, False),       # cfg5's feature count: two MFMA tiles for [X | y]
    (16, 0.2, 0.9, False, "cd_active_set", True, False),
    (17, 0.05, 0.5, False, "cd", False, False),        # > 16 columns: 32 lanes per group
    (24, 0.1, 1.0, True, "cd", True, True),
    (31, 0.02, 0.5, False, "cd_active_set", False, False),
])
def test_elastic_net_tight_tolerance(eng, dtype, tol, k, alpha, l1, positive, method, weights, icpt):
    """Converged to tol = 1e-10 both paths must land on the same (unique) fixed point."""
    from oracle import orc

    rng = np.random.default_rng(k + int(100 * alpha))
    sizes = rng.integers(40, 900, size=29)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    y, cols, w = _frame(rng, offs, k - int(icpt), dtype, weights=weights)
    kw = dict(alpha=alpha, l1_ratio=l1, positive=positive, solve_method=method, tol=1e-10, max_iter=20_000)
    eng.set_option("STATIC_ENGINE", "stream")             # the three-launch path (K2 takes these shapes by default: tests/test_k2_gpu.py)
    try:
        out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, weights=None if w is None else _cuda(w),
                                add_intercept=icpt, want=("coef", "pred", "resid", "status"), **kw)
    finally:
        eng.set_option("STATIC_ENGINE", None)
    ref = orc.batched_least_squares(y, cols, offs, weights=w, add_intercept=icpt, **kw)
    assert eng.last_kernel.startswith("k5_gram_stream")
    assert int(_np(out["status"]).sum()) == 0
    for key in ("coef", "pred", "resid"):
        assert np.allclose(_np(out[key]), ref[key], rtol=tol, atol=tol), (key, float(np.abs(_np(out[key]) - ref[key]).max()))


def test_elastic_net_default_tol_matches_oracle(eng):
    """Default tol = 1e-5 / max_iter = 1000: same sweep count, same answer (the stop rule is ||dw||_2 < tol)."""
    from oracle import orc
    from refdata import synthetic_groups

    d = synthetic_groups(300, 2000, 16, seed=5, dtype=np.float64)
    out = eng.least_squares(_cuda(d["y"]), [_cuda(c) for c in d["cols"]], d["offsets"], alpha=0.001, l1_ratio=0.5,
                            want=("coef", "pred"))
    ref = orc.batched_least_squares(d["y"], d["cols"], d["offsets"], alpha=0.001, l1_ratio=0.5)
    assert np.allclose(_np(out["coef"]), ref["coef"], rtol=1e-6, atol=1e-6)
    assert np.allclose(_np(out["pred"]), ref["pred"], rtol=1e-6, atol=1e-6)


def test_readme_lasso_and_golden_sklearn(eng, golden):
    from refdata import make_data, sort_by_group

    kat, z = golden["kat"], golden["npz"]
    f = {k: np.asarray(v, dtype=np.float64) for k, v in kat["frame"].items()}
    order, offs, _ = sort_by_group(f["group"].astype(np.int64))
    out = eng.least_squares(f["y"][order], [f["x1"][order], f["x2"][order]], offs, add_intercept=True, alpha=0.0001,
                            l1_ratio=1.0, want=("pred",))                       # README.md:58,72-76
    assert np.array_equal(np.round(out["pred"][:5], 2), kat["predictions_lasso_head5_round2"])
    d = make_data(n_features=2)                                                 # tests/test_ols.py:561-599, k = 2
    out = eng.least_squares(d["y"], [d["x1"], d["x2"]], [0, 5000], alpha=0.1, l1_ratio=0.5, tol=1e-9, want=("coef", "pred"))
    assert np.allclose(out["coef"][0], z["enet2_coef"], atol=1e-7) and np.allclose(out["pred"], z["enet2_pred"], atol=1e-6)
    out = eng.least_squares(d["y"], [d["x1"], -d["x2"]], [0, 5000], alpha=0.1, l1_ratio=0.5, tol=1e-9, positive=True,
                            want=("coef",))                                     # tests/test_ols.py:602-630
    assert np.allclose(out["coef"][0], z["nnls_coef"], atol=1e-7) and out["coef"][0][1] == 0.0


def test_max_iter_reached_is_reported(eng):
    rng = np.random.default_rng(0)
    offs = np.array([0, 500], dtype=np.int64)
    y, cols, _ = _frame(rng, offs, 6, np.float64)
    cols[1] = cols[0] + 1e-3 * cols[1]                                          # strongly correlated pair: slow CD
    out = eng.least_squares(y, cols, offs, alpha=1e-6, l1_ratio=0.5, tol=1e-14, max_iter=3, want=("coef", "status"))
    assert out["status"][0] == 3                                                # POLS_GROUP_NOT_CONVERGED


def test_reference_panics(eng):
    from polars_ols_amd import PolsPanic

    y = np.ones(8); x = [np.arange(8.0)]
    with pytest.raises(PolsPanic, match="coordinate descent"):
        eng.least_squares(y, x, [0, 8], alpha=0.1, l1_ratio=0.5, solve_method="qr")     # least_squares.rs:404
    with pytest.raises(PolsPanic, match="l1_ratio"):
        eng.least_squares(y, x, [0, 8], alpha=0.1, l1_ratio=1.5)                        # least_squares.rs:410


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_predict_rowwise_coefficients(eng, dtype):
    """pols_predict == (features * coefficients).sum(axis=1) (src/expressions.rs:728), with and without intercept."""
    rng = np.random.default_rng(1)
    n, k = 10_007, 5
    cols = [rng.standard_normal(n).astype(dtype) for _ in range(k)]
    coef = rng.standard_normal((n, k + 1)).astype(dtype)
    exp = sum(c.astype(np.float64) * coef[:, j] for j, c in enumerate(cols)) + coef[:, k]
    tol = 1e-4 if dtype == np.float32 else 1e-9
    got = eng.predict([_cuda(c) for c in cols], _cuda(coef), add_intercept=True)
    assert np.allclose(_np(got), exp, rtol=tol, atol=tol)
    got = eng.predict(cols, coef[:, :k].copy())                                 # host buffers
    assert np.allclose(got, exp - coef[:, k], rtol=tol, atol=tol)


def test_cfg5_shape_medium(eng):
    """BASELINE configs[4] shape (2 000 rows x 16 feats, alpha = 0.001, l1_ratio = 0.5, f64) on 4 000 groups generated
    on the device; oracle parity on a sample of groups, normal-equation-style optimality on all."""
    import torch
    from oracle import orc

    G, n, k = 4_000, 2_000, 16
    g = torch.Generator(device="cuda").manual_seed(5)
    cols = [torch.randn(G * n, generator=g, device="cuda", dtype=torch.float64) for _ in range(k)]
    y = sum(cols) + 0.1 * torch.randn(G * n, generator=g, device="cuda", dtype=torch.float64)
    offs = np.arange(G + 1, dtype=np.int64) * n
    out = eng.least_squares(y, cols, offs, alpha=0.001, l1_ratio=0.5, want=("coef", "pred", "status"))
    assert int(out["status"].sum()) == 0
    pick = np.array([0, 7, 1999, 3999])
    yh = y.cpu().numpy().reshape(G, n)[pick].reshape(-1)
    ch = [c.cpu().numpy().reshape(G, n)[pick].reshape(-1) for c in cols]
    ref = orc.batched_least_squares(yh, ch, np.arange(len(pick) + 1) * n, alpha=0.001, l1_ratio=0.5)
    assert np.allclose(_np(out["coef"])[pick], ref["coef"], rtol=1e-6, atol=1e-6)
    assert np.allclose(_np(out["pred"]).reshape(G, n)[pick].reshape(-1), ref["pred"], rtol=1e-6, atol=1e-6)
    # KKT of the elastic net at an interior (all non-zero) solution: x_j.(y - Xw) = n*alpha*(l1*sign(w_j) + (1-l1)*w_j)
    w = out["coef"]
    r = (y - out["pred"]).view(G, n)
    for j in range(k):
        lhs = (cols[j].view(G, n) * r).sum(1)
        rhs = n * 0.001 * (0.5 * torch.sign(w[:, j]) + 0.5 * w[:, j])
        assert float((lhs - rhs).abs().max()) < 5e-2    # tol = 1e-5 on w times ||x_j||^2 ~ 2000


def test_cfg5_full_size(eng):
    """BASELINE configs[4] at FULL size on one GPU: 100 000 groups x 2 000 rows x 16 feats f64 (27 GB resident), elastic net
    alpha = 0.001, l1_ratio = 0.5.  Size-independent properties on every group (KKT stationarity of the elastic net, every group
    converged) + oracle parity on a sample of groups."""
    import torch
    from oracle import orc

    G, n, k = 100_000, 2_000, 16
    g = torch.Generator(device="cuda").manual_seed(11)
    cols = [torch.randn(G * n, generator=g, device="cuda", dtype=torch.float64) for _ in range(k)]
    y = torch.zeros(G * n, device="cuda", dtype=torch.float64)
    for c in cols:
        y += c
    y += 0.1 * torch.randn(G * n, generator=g, device="cuda", dtype=torch.float64)
    offs = np.arange(G + 1, dtype=np.int64) * n
    out = eng.least_squares(y, cols, offs, alpha=0.001, l1_ratio=0.5, want=("coef", "pred", "status"))
    torch.cuda.synchronize()
    assert int(out["status"].sum()) == 0
    w = out["coef"]
    r = (y - out["pred"]).view(G, n)
    worst = 0.0
    for j in range(k):
        lhs = (cols[j].view(G, n) * r).sum(1)
        rhs = n * 0.001 * (0.5 * torch.sign(w[:, j]) + 0.5 * w[:, j])
        worst = max(worst, float((lhs - rhs).abs().max()))
    assert worst < 5e-2, worst                                  # tol = 1e-5 on w times ||x_j||^2 ~ 2000
    pick = np.array([0, 12_345, 50_000, 99_999])
    yh = np.concatenate([y[p * n:(p + 1) * n].cpu().numpy() for p in pick])
    ch = [np.concatenate([c[p * n:(p + 1) * n].cpu().numpy() for p in pick]) for c in cols]
    ref = orc.batched_least_squares(yh, ch, np.arange(len(pick) + 1) * n, alpha=0.001, l1_ratio=0.5)
    assert np.allclose(_np(out["coef"])[pick], ref["coef"], rtol=1e-6, atol=1e-6)
    got_p = np.concatenate([out["pred"][p * n:(p + 1) * n].cpu().numpy() for p in pick])
    assert np.allclose(got_p, ref["pred"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("k,alpha,weights,icpt", [(16, 0.0, False, False), (20, 1.0, True, True), (31, 0.0, False, False),
                                                  (24, 0.3, False, True)])
def test_static_ols_ridge_wide_features_streamed(eng, dtype, tol, k, alpha, weights, icpt):
    """16..31 features (incl. intercept): streamed two-tile MFMA Gram + wave-cooperative Cholesky + prediction pass."""
    from oracle import orc

    rng = np.random.default_rng(k)
    sizes = rng.integers(200, 1500, size=13)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    y, cols, w = _frame(rng, offs, k - int(icpt), dtype, sparsity=0.0, weights=weights)
    kw = dict(alpha=alpha, l1_ratio=0.0) if alpha else {}
    eng.set_option("STATIC_ENGINE", "stream")             # 16 columns fit K2 by default; this test is about the streamed kernels
    try:
        out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, weights=None if w is None else _cuda(w),
                                add_intercept=icpt, want=("coef", "pred", "resid", "status"), **kw)
    finally:
        eng.set_option("STATIC_ENGINE", None)
    ref = orc.batched_least_squares(y, cols, offs, weights=w, add_intercept=icpt, **kw)
    assert eng.last_kernel.startswith("k5_gram_stream") and int(_np(out["status"]).sum()) == 0
    for key in ("coef", "pred", "resid"):
        assert np.allclose(_np(out[key]), ref[key], rtol=tol, atol=tol), (key, float(np.abs(_np(out[key]) - ref[key]).max()))


def test_static_huge_groups_streamed(eng):
    """Groups far beyond K1's registers and K1m's LDS tile (60 000 rows x 8 features f64) take the streamed path."""
    from oracle import orc

    rng = np.random.default_rng(3)
    offs = np.array([0, 60_000, 60_003, 135_000], dtype=np.int64)
    y, cols, w = _frame(rng, offs, 8, np.float64, sparsity=0.0, weights=True)
    out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, weights=_cuda(w), alpha=2.0, l1_ratio=0.0,
                            want=("coef", "pred"))
    ref = orc.batched_least_squares(y, cols, offs, weights=w, alpha=2.0, l1_ratio=0.0)
    assert eng.last_kernel.startswith("k5_gram_stream")
    assert np.allclose(_np(out["coef"]), ref["coef"], rtol=1e-6, atol=1e-6)
    assert np.allclose(_np(out["pred"]), ref["pred"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("sizes,k,kind", [
    ([500_003], 8, "ols"), ([200_000, 0, 5, 100_001, 17_000, 3], 4, "ridge_w_icpt"), ([300_000, 150_000], 12, "enet"),
    ([120_001, 90_000, 64], 6, "drop_nulls"), ([1_000_000], 20, "ols"),
])
def test_static_few_long_groups_are_split_into_segments(eng, dtype, tol, sizes, k, kind):
    """ONE regression over a whole frame (the reference's first README example) or a few long groups: the streamed path cuts them
    into segments (one workgroup each in the Gram and the prediction pass; the segments' Gram matrices are summed per group in
    segment order).  Against the oracle and against the unsplit launch (NO_SPLIT) on the same frame."""
    from oracle import orc

    rng = np.random.default_rng(len(sizes) * 1000 + k)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    y, cols, w = _frame(rng, offs, k, dtype, sparsity=0.0, weights=(kind == "ridge_w_icpt"))
    kw, okw = {}, {}
    if kind == "ridge_w_icpt":
        kw = dict(weights=_cuda(w), alpha=0.5, l1_ratio=0.0, add_intercept=True)
        okw = dict(weights=w, alpha=0.5, l1_ratio=0.0, add_intercept=True)
    elif kind == "enet":
        kw = okw = dict(alpha=0.01, l1_ratio=0.5, tol=1e-10, max_iter=20_000)
    elif kind == "drop_nulls":
        y = y.copy()
        y[rng.random(len(y)) < 0.02] = np.nan
        cols[1] = cols[1].copy()
        cols[1][rng.random(len(y)) < 0.01] = np.nan
        kw = dict(null_policy="drop")
    out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, want=("coef", "pred"), **kw)
    assert eng.last_kernel.startswith("k5_gram_stream") and eng.last_kernel.endswith("_split"), eng.last_kernel
    eng.set_option("NO_SPLIT", "1")
    try:
        one = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, want=("coef", "pred"), **kw)
        assert not eng.last_kernel.endswith("_split")
    finally:
        eng.set_option("NO_SPLIT", None)
    assert np.allclose(_np(out["coef"]), _np(one["coef"]), rtol=tol, atol=tol, equal_nan=True)
    assert np.allclose(_np(out["pred"]), _np(one["pred"]), rtol=tol, atol=tol, equal_nan=True)
    if kind != "drop_nulls":
        ref = orc.batched_least_squares(y, cols, offs, **okw)
        assert np.allclose(_np(out["coef"]), ref["coef"], rtol=tol, atol=tol), float(np.abs(_np(out["coef"]) - ref["coef"]).max())
        assert np.allclose(_np(out["pred"]), ref["pred"], rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("k,icpt,kind", [(1, False, "ols"), (1, True, "ols"), (2, False, "ridge"), (3, True, "ols"), (4, False, "lu"), (5, True, "ridge"),
                                         (6, False, "ols"), (7, True, "ols"), (8, False, "ols"), (8, True, "ridge"), (9, True, "ols"), (10, False, "lu"),
                                         (9, False, "enet")])
def test_static_long_groups_valu_gram(eng, dtype, tol, k, icpt, kind):
    """Round 5: plain frames (no weights, no null policy) of up to ten columns take the VALU Gram pass (K5v, k5v_gram.hip) and the lean
    prediction kernel whenever a group is too long for the register-resident kernels -- ragged, unaligned groups next to short ones,
    every solver the streamed path serves, against the oracle."""
    from oracle import orc

    rng = np.random.default_rng(100 * k + int(icpt))
    sizes = [5_001, 12_345, 7, 9_000, 0, 4_097 + k, 3, 20_011]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    y, cols, _ = _frame(rng, offs, k, dtype, sparsity=0.0)
    kw = {"ols": {}, "ridge": dict(alpha=0.7, l1_ratio=0.0), "lu": dict(solve_method="lu"),
          "enet": dict(alpha=0.01, l1_ratio=0.5, tol=1e-10, max_iter=20_000)}[kind]
    out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, add_intercept=icpt, want=("coef", "pred", "resid", "status"), **kw)
    assert eng.last_kernel.startswith("k5_gram_stream") and "_valu_" in eng.last_kernel, eng.last_kernel
    ref = orc.batched_least_squares(y, cols, offs, add_intercept=icpt, **kw)
    st = _np(out["status"]).astype(int)
    # (the 3-row group has fewer rows than columns from k = 4 on and the empty one is flagged empty; everything else factors)
    long_rows = np.repeat(np.asarray(sizes) >= 1000, sizes)
    long_groups = np.asarray(sizes) >= 1000
    assert (st[long_groups] == 0).all(), st
    kt = k + int(icpt)
    got_c, ref_c = _np(out["coef"]).reshape(-1, kt), np.asarray(ref["coef"]).reshape(-1, kt)
    assert np.allclose(got_c[long_groups], ref_c[long_groups], rtol=tol, atol=tol), float(np.abs(got_c[long_groups] - ref_c[long_groups]).max())
    for key in ("pred", "resid"):
        assert np.allclose(_np(out[key])[long_rows], np.asarray(ref[key])[long_rows], rtol=tol, atol=tol), key
    if kind in ("ols", "ridge"):      # the short groups too: same reference branch per group (n <= k groups get the minimum-norm fix-up)
        assert np.allclose(got_c, ref_c, rtol=10 * tol, atol=10 * tol), float(np.abs(got_c - ref_c).max())


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("k,icpt,kind", [(1, False, "ols"), (3, True, "ridge"), (8, False, "ols"), (8, True, "ols"), (10, False, "ridge"), (6, True, "lu")])
def test_static_long_groups_valu_gram_with_weights(eng, dtype, tol, k, icpt, kind):
    """WLS on groups too long for the registers: the VALU Gram pass scales every column by sqrt(w) as it loads it, the lean prediction kernel
    keeps the reference's arithmetic, (sqrt(w) x) . c * (1 / sqrt(w)) (least_squares.py:190-196, 234-235).  Against the oracle."""
    from oracle import orc

    rng = np.random.default_rng(200 * k + int(icpt))
    sizes = [6_001, 14_345, 9_000, 4_097 + k, 23_011, 5_555]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    y, cols, w = _frame(rng, offs, k, dtype, sparsity=0.0, weights=True)
    kw = {"ols": {}, "ridge": dict(alpha=0.7, l1_ratio=0.0), "lu": dict(solve_method="lu")}[kind]
    out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, weights=_cuda(w), add_intercept=icpt, want=("coef", "pred", "resid", "status"), **kw)
    assert eng.last_kernel.startswith("k5_gram_stream") and "_valu_w_" in eng.last_kernel, eng.last_kernel
    ref = orc.batched_least_squares(y, cols, offs, weights=w, add_intercept=icpt, **kw)
    assert int(_np(out["status"]).sum()) == 0
    for key in ("coef", "pred", "resid"):
        assert np.allclose(_np(out[key]), ref[key], rtol=tol, atol=tol), (key, float(np.abs(_np(out[key]) - ref[key]).max()))


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("policy", ["zero", "drop", "drop_zero", "drop_y_zero_x"])
@pytest.mark.parametrize("k,weights,icpt,kw", [(8, False, False, {}), (5, True, True, {"alpha": 0.5}), (10, False, False, {}), (2, True, False, {})])
def test_static_long_groups_valu_gram_null_policies(eng, dtype, tol, policy, k, weights, icpt, kw):
    """Null policies on groups too long for the registers (round 5): K5v applies them to the rows as it loads them -- dropped rows become zero rows,
    nulls that stay become 0, the rows left in the fit are counted -- and the lean prediction kernel masks the dropped rows under "drop".
    "drop" on ONE 10M-row group took 0.38 ms (f32) before, 2.5 x the plain call (profiles/r05_bench_long_nulls.txt).  Expected values composed
    like the reference composes them (tests/test_nulls_gpu.py::_expected)."""
    from test_nulls_gpu import _expected

    rng = np.random.default_rng(300 * k + len(policy))
    sizes = [7_003, 15_345, 9_000, 4_097 + k, 5, 21_011]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    y, cols, w = _frame(rng, offs, k, dtype, sparsity=0.0, weights=weights)
    y = y.copy()
    y[rng.random(len(y)) < 0.02] = np.nan
    for j in range(0, k, 3):
        cols[j] = cols[j].copy()
        cols[j][rng.random(len(y)) < 0.01] = np.nan
    y[offs[4]:offs[5]] = np.nan                                      # the 5-row group: nothing left to fit under the drop family
    out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, weights=None if w is None else _cuda(w), add_intercept=icpt,
                            want=("coef", "pred", "resid"), null_policy=policy, **kw)
    assert eng.last_kernel.startswith("k5_gram_stream") and "_valu" in eng.last_kernel and "_nulls_" in eng.last_kernel, eng.last_kernel
    coef, pred, resid = _expected(y, cols, offs, w, icpt, policy, **kw)
    long_g = np.asarray(sizes) >= 1000
    rows = np.repeat(long_g, sizes)
    got_c = _np(out["coef"]).reshape(-1, k + int(icpt))
    assert np.allclose(got_c[long_g], coef[long_g], rtol=tol, atol=tol), float(np.abs(got_c[long_g] - coef[long_g]).max())
    gp = _np(out["pred"])
    assert np.array_equal(np.isnan(gp[rows]), np.isnan(pred[rows]))
    assert np.allclose(gp[rows], pred[rows], rtol=tol, atol=tol, equal_nan=True)
    assert np.allclose(_np(out["resid"])[rows], resid[rows], rtol=tol, atol=tol, equal_nan=True)


@pytest.mark.parametrize("k,icpt,weights,policy", [(11, False, False, None), (12, True, False, None), (13, False, True, None), (12, False, False, "drop")])
def test_static_long_groups_valu_gram_f32_up_to_13_columns(eng, k, icpt, weights, policy):
    """f32 frames keep the VALU Gram pass up to 13 columns (105 accumulators + 14 vectors in flight: 254 registers); f64 and 14+ columns take the MFMA tiles."""
    from oracle import orc
    from test_nulls_gpu import _expected

    kt = k + int(icpt)
    if kt > 13:
        pytest.skip("14 columns: MFMA")
    rng = np.random.default_rng(400 + k)
    sizes = [9_001, 14_345, 11_000, 12_097]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    y, cols, w = _frame(rng, offs, k, np.float32, sparsity=0.0, weights=weights)
    kw = {}
    if policy:
        y = y.copy(); y[rng.random(len(y)) < 0.02] = np.nan
        kw["null_policy"] = policy
    out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, weights=None if w is None else _cuda(w), add_intercept=icpt, want=("coef", "pred"), **kw)
    assert "_valu" in eng.last_kernel, eng.last_kernel
    if policy:
        coef, pred, _ = _expected(y, cols, offs, w, icpt, policy)
    else:
        ref = orc.batched_least_squares(y, cols, offs, weights=w, add_intercept=icpt)
        coef, pred = np.asarray(ref["coef"]).reshape(-1, kt), np.asarray(ref["pred"])
    assert np.allclose(_np(out["coef"]).reshape(-1, kt), coef, rtol=1e-4, atol=1e-4)
    assert np.allclose(_np(out["pred"]), pred, rtol=1e-4, atol=1e-4, equal_nan=True)
    eng.set_option("STATIC_ENGINE", "stream")
    try:                                                              # f64, same width: the MFMA Gram pass
        out64 = eng.least_squares(_cuda(y.astype(np.float64)), [_cuda(c.astype(np.float64)) for c in cols], offs,
                                  weights=None if w is None else _cuda(w.astype(np.float64)), add_intercept=icpt, want=("coef",), **kw)
        assert "_valu" not in eng.last_kernel, eng.last_kernel
    finally:
        eng.set_option("STATIC_ENGINE", None)
    assert np.allclose(_np(out64["coef"]).reshape(-1, kt), coef, rtol=1e-4, atol=1e-4)
