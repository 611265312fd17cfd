"""GPU parity of K2 (rows resident in registers, X'X on the matrix cores, solver in the same workgroup: X read once) through the
C-ABI against the CPU oracle: OLS / ridge beyond K1's eight columns, solve_method="lu" (src/least_squares.rs:264-273), the
Cholesky -> LU fallback of solve_ridge (:358-363), elastic net / lasso / non-negative (:386-492) -- every workgroup shape, ragged
and unaligned groups, weights, intercept, empty groups, groups flagged for the SVD pass.

Tolerances (BASELINE.json north_star): 1e-6 for f64, 1e-4 for f32, |a - b| <= tol + tol |b| (tests/test_ols.py:73)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = {np.float32: 1e-4, np.float64: 1e-6}


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


def _offsets(rng, n_groups, lo, hi):
    sizes = rng.integers(lo, hi + 1, size=n_groups)
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


def _frame(rng, offs, k, dtype, weights=False, sparsity=0.0):
    N = int(offs[-1])
    cols = [rng.standard_normal(N).astype(dtype) for _ in range(k)]
    kk = max(1, int(k * (1 - sparsity)))
    beta = rng.uniform(0.5, 1.5, size=kk)
    y = (sum(b * c.astype(np.float64) for b, c in zip(beta, cols[:kk])) + 0.1 * rng.standard_normal(N)).astype(dtype)
    w = rng.uniform(0.2, 2.0, N).astype(dtype) if weights else None
    return y, cols, w


def _check(out, ref, dtype, keys=("coef", "pred", "resid")):
    tol = TOL[dtype]
    for k in keys:
        got = _np(out[k])
        assert got.shape == ref[k].shape, k
        assert np.allclose(got, ref[k], rtol=tol, atol=tol), (k, float(np.abs(got - ref[k]).max()))


# (lo, hi) group sizes chosen to land on every workgroup shape: one wave (1 or 2 chunks per lane), four waves, eight waves
SHAPES = [(20, 60, "_w1_rc1"), (70, 250, "_w1_rc2"), (300, 510, "_w4_rc1"), (600, 1000, "_w4_rc2"), (1100, 2040, "_w8_rc2")]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("lo,hi,shape", SHAPES)
@pytest.mark.parametrize("kt,weights,icpt,alpha", [(9, False, True, 0.0), (12, True, False, 0.7), (16, False, True, 0.0), (16, True, False, 2.0)])
def test_ols_ridge_every_shape(eng, dtype, lo, hi, shape, kt, weights, icpt, alpha):
    """9..16 columns (the intercept counted): Cholesky on the register-resident rows' Gram matrix, fused predictions."""
    from oracle import orc

    rng = np.random.default_rng(hi + kt)
    if dtype == np.float32:                                   # an f32 lane holds twice the rows of an f64 lane
        lo, hi = 2 * lo, 2 * hi
    offs = _offsets(rng, 23, lo, hi)
    y, cols, w = _frame(rng, offs, kt - int(icpt), dtype, weights=weights)
    kw = dict(alpha=alpha, l1_ratio=0.0) if alpha else {}
    eng.set_option("STATIC_ENGINE", "k2")                   # 9..15 columns that fit K1m's LDS tile go there by default
    try:
        out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, weights=None if w is None else _cuda(w), add_intercept=icpt,
                                want=("coef", "pred", "resid", "status"), **kw)
    finally:
        eng.set_option("STATIC_ENGINE", None)
    ref = orc.batched_least_squares(y, cols, offs, weights=w, add_intercept=icpt, **kw)
    assert eng.last_kernel.startswith("k2_gram_mfma_resident") and shape in eng.last_kernel and eng.last_kernel.endswith("_chol"), eng.last_kernel
    assert int(_np(out["status"]).sum()) == 0
    _check(out, ref, dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("k,alpha,icpt", [(2, 0.0, False), (5, 0.3, True), (8, 0.0, True), (13, 1.5, False), (16, 0.0, False)])
def test_solve_method_lu(eng, dtype, k, alpha, icpt):
    """solve_method="lu" is a partial-pivot LU of X'X + alpha I (solve_ols_lu, ls.rs:264-273; reached through solve_ridge,
    ex.rs:374-375), not an alias of the Cholesky kernel."""
    from oracle import orc

    rng = np.random.default_rng(k)
    offs = _offsets(rng, 31, 30, 900)
    y, cols, w = _frame(rng, offs, k - int(icpt), dtype, weights=True)
    kw = dict(alpha=alpha, l1_ratio=0.0, solve_method="lu")
    out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, weights=_cuda(w), add_intercept=icpt,
                            want=("coef", "pred", "resid", "status"), **kw)
    ref = orc.batched_least_squares(y, cols, offs, weights=w, add_intercept=icpt, **kw)
    assert eng.last_kernel.startswith("k2_gram_mfma_resident") and eng.last_kernel.endswith("_lu"), eng.last_kernel
    assert int(_np(out["status"]).sum()) == 0
    _check(out, ref, dtype)


def test_lu_on_groups_beyond_the_registers(eng):
    """Groups too long for K2 take the streamed Gram pass; "lu" is still an LU there (gram_solve's LU branch)."""
    from oracle import orc

    rng = np.random.default_rng(4)
    offs = np.array([0, 9_000, 9_007, 21_000], dtype=np.int64)
    y, cols, w = _frame(rng, offs, 5, np.float64, weights=True)
    kw = dict(alpha=0.4, l1_ratio=0.0, solve_method="lu")
    out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, weights=_cuda(w), want=("coef", "pred", "status"), **kw)
    ref = orc.batched_least_squares(y, cols, offs, weights=w, **kw)
    assert eng.last_kernel.startswith("k5_gram_stream")
    assert int(_np(out["status"]).sum()) == 0
    _check(out, ref, np.float64, keys=("coef", "pred"))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("engine", [None, "stream"])
@pytest.mark.parametrize("k,alpha,l1,positive,method,weights,icpt", [
    (2, 0.1, 0.5, False, "cd", False, False),
    (8, 0.05, 1.0, False, None, False, True),          # lasso + intercept
    (8, 0.3, 0.5, True, "cd", True, False),            # non-negative + weights
    (15, 0.01, 0.5, False, "cd_active_set", False, True),
    (16, 0.001, 0.5, False, "cd", False, False),       # BASELINE configs[4]'s feature count: X'y on the VALU
    (16, 0.2, 0.9, False, "cd_active_set", True, False),
])
def test_elastic_net_both_engines(eng, dtype, engine, k, alpha, l1, positive, method, weights, icpt):
    """The fused kernel and the three-launch path land on the oracle's fixed point (tol = 1e-10: the unique minimiser)."""
    from oracle import orc

    rng = np.random.default_rng(k + int(100 * alpha))
    offs = _offsets(rng, 29, 40, 1900)
    y, cols, w = _frame(rng, offs, k - int(icpt), dtype, weights=weights, sparsity=0.5)
    kw = dict(alpha=alpha, l1_ratio=l1, positive=positive, solve_method=method, tol=1e-10, max_iter=20_000)
    eng.set_option("STATIC_ENGINE", engine)
    try:
        out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, weights=None if w is None else _cuda(w),
                                add_intercept=icpt, want=("coef", "pred", "resid", "status"), **kw)
    finally:
        eng.set_option("STATIC_ENGINE", None)
    ref = orc.batched_least_squares(y, cols, offs, weights=w, add_intercept=icpt, **kw)
    assert eng.last_kernel.startswith("k5_gram_stream" if engine else "k2_gram_mfma_resident"), eng.last_kernel
    assert int(_np(out["status"]).sum()) == 0
    _check(out, ref, dtype)


def test_elastic_net_default_tolerance_same_sweeps(eng):
    """Default tol = 1e-5 / max_iter = 1000 on BASELINE configs[4]'s shape: the stop rule fires on the same sweep as the oracle's."""
    from oracle import orc
    from refdata import synthetic_groups

    d = synthetic_groups(300, 2000, 16, seed=5, dtype=np.float64)
    out = eng.least_squares(_cuda(d["y"]), [_cuda(c) for c in d["cols"]], d["offsets"], alpha=0.001, l1_ratio=0.5,
                            want=("coef", "pred", "resid"))
    ref = orc.batched_least_squares(d["y"], d["cols"], d["offsets"], alpha=0.001, l1_ratio=0.5)
    assert "_k16yv_w8_rc2" in eng.last_kernel and eng.last_kernel.endswith("_cd"), eng.last_kernel
    _check(out, ref, np.float64)


def test_max_iter_reached_is_reported(eng):
    rng = np.random.default_rng(0)
    offs = np.array([0, 500], dtype=np.int64)
    y, cols, _ = _frame(rng, offs, 6, np.float64)
    cols[1] = cols[0] + 1e-3 * cols[1]                                          # strongly correlated pair: slow CD
    out = eng.least_squares(y, cols, offs, alpha=1e-6, l1_ratio=0.5, tol=1e-14, max_iter=3, want=("coef", "status"))
    assert eng.last_kernel.startswith("k2_") and out["status"][0] == 3          # POLS_GROUP_NOT_CONVERGED


def test_empty_tiny_and_flagged_groups(eng):
    """Empty groups give zeros (ex.rs:357-359); rank-deficient groups on the OLS branch are flagged and re-solved by the fix-up pass
    with the reference's own solver for them (ls.rs:224-231: pivoted QR -> basic solution when n > k, dgelsd's minimum norm when
    n <= k), their neighbours are untouched."""
    from oracle import orc

    rng = np.random.default_rng(12)
    offs = np.array([0, 0, 40, 40, 43, 700, 700, 1500], dtype=np.int64)
    k = 11
    y, cols, _ = _frame(rng, offs, k, np.float64)
    cols[7][43:700] = cols[2][43:700]                                           # group 4: two identical columns
    eng.set_option("STATIC_ENGINE", "k2")
    try:
        out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, want=("coef", "pred", "status"))
    finally:
        eng.set_option("STATIC_ENGINE", None)
    assert eng.last_kernel.startswith("k2_")
    st = _np(out["status"]).astype(int)
    assert list(st) == [2, 0, 2, 1, 1, 2, 0], st                                # 3 rows x 11 columns and the twin columns: fallback
    coef = _np(out["coef"])
    assert np.array_equal(coef[[0, 2, 5]], np.zeros((3, k)))
    ref = orc.batched_least_squares(y, cols, offs)
    for g in (1, 3, 4, 6):
        assert np.allclose(coef[g], ref["coef"][g], rtol=1e-6, atol=1e-9), g
    assert np.allclose(_np(out["pred"]), ref["pred"], rtol=1e-6, atol=1e-8)
    assert (coef[4][2] == 0.0) != (coef[4][7] == 0.0)                           # one of the twins carries both, the other is exactly 0
    X = np.stack([c[40:43] for c in cols], axis=1)
    assert np.allclose(coef[3], np.linalg.lstsq(X, y[40:43], rcond=None)[0], rtol=1e-6, atol=1e-8)   # n < k: dgelsd's minimum norm


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_forced_on_k1_shapes_matches_k1(eng, dtype):
    """POLS_STATIC_ENGINE=k2 runs K2 on the shapes K1 owns by default (<= 8 columns): both agree with the oracle."""
    from oracle import orc
    from refdata import synthetic_groups

    d = synthetic_groups(64, 1000, 8, seed=3, dtype=dtype, with_weights=True)
    args = (_cuda(d["y"]), [_cuda(c) for c in d["cols"]], d["offsets"])
    kw = dict(weights=_cuda(d["w"]), alpha=1.0, l1_ratio=0.0, want=("coef", "pred", "resid"))
    a = eng.least_squares(*args, **kw)
    k1_name = eng.last_kernel
    eng.set_option("STATIC_ENGINE", "k2")
    try:
        b = eng.least_squares(*args, **kw)
    finally:
        eng.set_option("STATIC_ENGINE", None)
    assert k1_name.startswith("k1_") and eng.last_kernel.startswith("k2_"), (k1_name, eng.last_kernel)
    ref = orc.batched_least_squares(d["y"], d["cols"], d["offsets"], weights=d["w"], alpha=1.0, l1_ratio=0.0)
    _check(a, ref, dtype)
    _check(b, ref, dtype)


def test_host_path_and_determinism(eng):
    rng = np.random.default_rng(8)
    offs = _offsets(rng, 37, 100, 1500)
    y, cols, w = _frame(rng, offs, 12, np.float64, weights=True)
    kw = dict(weights=w, alpha=0.01, l1_ratio=0.5, want=("coef", "pred"))
    a = eng.least_squares(y, cols, offs, **kw)
    b = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, weights=_cuda(w), alpha=0.01, l1_ratio=0.5, want=("coef", "pred"))
    c = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, weights=_cuda(w), alpha=0.01, l1_ratio=0.5, want=("coef", "pred"))
    assert np.array_equal(a["coef"], _np(b["coef"])) and np.array_equal(a["pred"], _np(b["pred"]))
    assert np.array_equal(_np(b["pred"]), _np(c["pred"]))                       # fixed reduction order: run-to-run identical


@pytest.mark.parametrize("dtype,kt,weights,icpt,kw", [
    (np.float64, 12, True, True, dict(alpha=0.3, l1_ratio=0.0)),
    (np.float64, 16, False, False, dict(alpha=0.01, l1_ratio=0.5, tol=1e-10, max_iter=20_000)),
    (np.float32, 12, True, False, {}),          # (9-10 columns at these row counts stay with K1 since round 5)
])
def test_persistent_workgroups_and_prefetch(eng, dtype, kt, weights, icpt, kw):
    """More groups than the chip holds eight-wave workgroups: every workgroup walks several groups and the first chunks of its
    next group arrive through the LDS prefetch buffer (ragged, unaligned groups: prefetched and directly loaded chunks mix)."""
    from oracle import orc

    rng = np.random.default_rng(kt)
    lo, hi = (1100, 2040) if dtype == np.float64 else (2200, 4080)
    offs = _offsets(rng, 700, lo, hi)
    y, cols, w = _frame(rng, offs, kt - int(icpt), dtype, weights=weights, sparsity=0.3)
    out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, weights=None if w is None else _cuda(w), add_intercept=icpt,
                            want=("coef", "pred", "resid", "status"), **kw)
    assert "_w8_rc2" in eng.last_kernel, eng.last_kernel
    ref = orc.batched_least_squares(y, cols, offs, weights=w, add_intercept=icpt, **kw)
    assert int(_np(out["status"]).sum()) == 0
    _check(out, ref, dtype)
    eng.set_option("K2_NOPREFETCH", "1")
    try:
        out2 = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, weights=None if w is None else _cuda(w), add_intercept=icpt,
                                 want=("coef", "pred"), **kw)
    finally:
        eng.set_option("K2_NOPREFETCH", None)
    assert np.array_equal(_np(out["coef"]), _np(out2["coef"])) and np.array_equal(_np(out["pred"]), _np(out2["pred"]))


# ----------------------------------------------------------------------------------------------------------------- K2w (17..31 columns)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("kt,weights,icpt,alpha", [(17, False, False, 0.0), (24, True, True, 0.0), (31, True, False, 0.7), (31, False, True, 0.0),
                                                    (20, False, True, 2.0)])
@pytest.mark.parametrize("shape", ["w4", "w8"])
def test_k2w_two_tile_resident_kernel(eng, dtype, kt, weights, icpt, alpha, shape):
    """17..31 columns with every row resident in registers and Z'Z as three 16 x 16 tiles on the matrix cores (k2w_kernel.inl): ragged,
    unaligned groups up to the capacity of the four- / eight-wave workgroup, an empty group, a rank-deficient one (fix-up pass),
    weights, intercept, ridge -- against the oracle.  POLS_STATIC_ENGINE=k2w takes the shapes the resident K1 kernels would."""
    from oracle import orc

    vec = 4 if dtype == np.float32 else 2
    lo, hi = (kt + 9, 256 * vec - vec) if shape == "w4" else (256 * vec + 1, 512 * vec - vec)
    rng = np.random.default_rng(kt * 10 + (1 if shape == "w8" else 0))
    sizes = rng.integers(lo, hi + 1, size=260 if shape == "w4" else 300)
    sizes[3] = 0
    sizes[-1] = hi                                                   # the capacity itself
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    y, cols, w = _frame(rng, offs, kt - int(icpt), dtype, weights=weights)
    s7, e7 = offs[7], offs[8]
    cols[2][s7:e7] = cols[5][s7:e7]                                  # rank-deficient group: flagged, re-solved by the fix-up pass
    eng.set_option("STATIC_ENGINE", "k2w")
    try:
        out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, weights=None if w is None else _cuda(w), add_intercept=icpt,
                                alpha=alpha, l1_ratio=0.0 if alpha else None, want=("coef", "pred", "resid", "status"))
        name = eng.last_kernel
    finally:
        eng.set_option("STATIC_ENGINE", None)
    assert name.startswith(f"k2w_gram_mfma_resident2_{'f32' if dtype == np.float32 else 'f64'}_k{kt}_w{4 if shape == 'w4' else 8}"), name
    ref = orc.batched_least_squares(y, cols, offs, weights=w, add_intercept=icpt, alpha=alpha, l1_ratio=0.0 if alpha else None)
    st = _np(out["status"]).astype(int)
    # (the twin-column group: rank deficient without a penalty -> flagged and re-solved by the fix-up pass; with one it is positive
    # definite -- an f32 batch may still flag it when cond * eps_f32 is beyond the tolerance, and gets the f64 answer from the fix-up)
    assert st[3] == 2 and (st[7] == 1 or alpha > 0) and (np.delete(st, [3, 7]) == 0).all(), st[:10]
    _check(out, ref, dtype)


@pytest.mark.parametrize("k,family", [(31, "k2w_gram_mfma_resident2_f64_k31_w8"), (22, "k2w_gram_mfma_resident2_f64_k22_w8"), (20, "k2w_gram_mfma_resident2_f64_k20_w8"),
                                      (18, "k2w_gram_mfma_resident2_f64_k18_w8")])
def test_k2w_is_the_default_beyond_the_resident_k1_shapes(eng, k, family):
    """f64, 20..31 columns x 1 000 rows (up to 248 KB per group): used to take the three-launch streamed path (X read twice); 20..24 columns
    since the solvers padded to the next of 20 / 24 / 28 / 32 columns, 17..19 since their padded system is built by independent loads."""
    from oracle import orc

    rng = np.random.default_rng(5)
    offs = np.arange(0, 41 * 1000, 1000, dtype=np.int64)
    y, cols, _ = _frame(rng, offs, k, np.float64)
    out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, want=("coef", "pred", "resid", "status"))
    assert eng.last_kernel.startswith(family), eng.last_kernel
    _check(out, orc.batched_least_squares(y, cols, offs), np.float64)


@pytest.mark.parametrize("dtype,k,rows,family", [(np.float64, 12, 1300, "k2_gram_mfma_resident_f64_k16_w8_rc2"), (np.float64, 9, 1100, "k1_gram_chol_f64_k9_team256_rc4"),
                                                 (np.float32, 14, 2600, "k2_gram_mfma_resident_f32_k16_w8_rc2"), (np.float32, 9, 2600, "k1_gram_chol_f32_k9_team256_rc4"),
                                                 # round 5, up to eight columns: four chunks per lane of the eight-wave K2 (8 192 f32 / 4 096 f64 rows)
                                                 (np.float64, 8, 4090, "k2_gram_mfma_resident_f64_k8_w8_rc4"), (np.float64, 7, 2300, "k2_gram_mfma_resident_f64_k8_w8_rc4"),
                                                 (np.float32, 8, 8190, "k2_gram_mfma_resident_f32_k8_w8_rc4"), (np.float32, 7, 6000, "k2_gram_mfma_resident_f32_k8_w8_rc4"),
                                                 (np.float32, 3, 4300, "k5_gram_stream_f32_valu_k3"), (np.float64, 5, 2300, "k5_gram_stream_f64_valu_k5")])
def test_over_resident_groups_route_to_k2_where_it_wins(eng, dtype, k, rows, family):
    """Rows beyond K1's registers, tile within LDS: K2 by default (it beats the LDS-tile engine there since round 3) except f32 with 9-10
    columns; ragged groups whose upper waves own no row of the second chunk (they skip its tile stages).  Round 5: up to 10 columns K1's own
    registers reach 4 096 f32 / 2 048 f64 rows (four chunks per lane), so K2 / K1m start beyond that."""
    from oracle import orc

    rng = np.random.default_rng(rows + k)
    offs = _offsets(rng, 40, rows - 200, rows)
    y, cols, w = _frame(rng, offs, k, dtype, weights=(k % 2 == 0))
    out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, weights=None if w is None else _cuda(w), want=("coef", "pred", "resid", "status"))
    assert eng.last_kernel.startswith(family), eng.last_kernel
    ref = orc.batched_least_squares(y, cols, offs, weights=w)
    assert int(_np(out["status"]).sum()) == 0
    _check(out, ref, dtype)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("kt", list(range(9, 32)))
def test_every_padded_solver_width(eng, dtype, kt):
    """The K2 / K2w Cholesky runs on a system padded to the next of 10 / 12 / 14 / 16 and 20 / 24 / 28 / 32 columns: every column count from
    9 to 31 (the intercept counted, weights on the odd ones, ridge on every third), ragged groups, against the oracle."""
    from oracle import orc

    rng = np.random.default_rng(100 + kt)
    icpt = kt % 2 == 0
    offs = _offsets(rng, 9, 3 * kt, 900)
    y, cols, w = _frame(rng, offs, kt - int(icpt), dtype, weights=(kt % 2 == 1))
    kw = dict(alpha=0.4, l1_ratio=0.0) if kt % 3 == 0 else {}
    eng.set_option("STATIC_ENGINE", "k2" if kt <= 16 else "k2w")
    try:
        out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, weights=None if w is None else _cuda(w), add_intercept=icpt,
                                want=("coef", "pred", "resid", "status"), **kw)
    finally:
        eng.set_option("STATIC_ENGINE", None)
    assert eng.last_kernel.startswith("k2_gram_mfma_resident_" if kt <= 16 else "k2w_gram_mfma_resident2_"), eng.last_kernel
    ref = orc.batched_least_squares(y, cols, offs, weights=w, add_intercept=icpt, **kw)
    assert int(_np(out["status"]).sum()) == 0
    _check(out, ref, dtype)
