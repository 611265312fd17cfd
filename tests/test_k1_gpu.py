"""GPU parity tests of K1 (gram_chol_predict) through the C-ABI, against the CPU oracle, the committed golden
fixtures and -- at BASELINE.json's full sizes -- size-independent properties.

Tolerances (BASELINE.json north_star): 1e-6 relative for f64, 1e-4 for f32; predictions cross zero, so the
comparison is |a-b| <= atol + rtol*|b| with atol = rtol, the reference's own convention (tests/test_ols.py:73).
"""
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


@pytest.fixture(params=["valu", "mfma"])
def engine_kind(request, eng):
    """A/B both engines of the static path: K1 (register-resident VALU Gram) and K1m (LDS tile + MFMA Gram); shapes neither
    takes by default go to K2 / the streamed kernels whatever the knob says."""
    eng.set_option("K1_ENGINE", request.param)
    yield request.param
    eng.set_option("K1_ENGINE", None)


def _cuda(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _ragged_offsets(rng, n_groups, lo, hi):
    sizes = rng.integers(lo, hi + 1, size=n_groups)
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


def _frame(rng, offsets, k, dtype, weights=False):
    N = int(offsets[-1])
    cols = [rng.standard_normal(N).astype(dtype) for _ in range(k)]
    beta = rng.uniform(0.5, 1.5, size=k)
    y = (sum(b * c.astype(np.float64) for b, c in zip(beta, cols)) + 0.1 * rng.standard_normal(N)).astype(dtype)
    w = rng.uniform(0.1, 2.0, N).astype(dtype) if weights else None
    return y, cols, w


def _check(out, ref, dtype, keys=("coef", "pred", "resid")):
    tol = TOL[dtype]
    for k in keys:
        got = out[k].double().cpu().numpy() if hasattr(out[k], "cpu") else np.asarray(out[k], dtype=np.float64)
        assert got.shape == ref[k].shape, k
        assert np.allclose(got, ref[k], rtol=tol, atol=tol), (k, float(np.abs(got - ref[k]).max()))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("k", [1, 2, 4, 8, 10])
def test_ols_equal_groups(eng, engine_kind, dtype, k):
    from oracle import orc

    rng = np.random.default_rng(k)
    offs = np.arange(33, dtype=np.int64) * 1000
    y, cols, _ = _frame(rng, offs, k, dtype)
    out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, want=("coef", "pred", "resid", "status"))
    ref = orc.batched_least_squares(y, cols, offs)   # reference default: pivoted QR (least_squares.rs:195-240)
    _check(out, ref, dtype)
    assert int(out["status"].abs().sum()) == 0
    assert eng.last_kernel.startswith("k1m_gram_mfma" if engine_kind == "mfma" else "k1_gram_chol"), eng.last_kernel


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("lo,hi,variant", [(12, 60, "k1t_"), (12, 120, "team64"), (130, 250, "team64"), (200, 1000, "team256"), (900, 1150, "team256"),
                                           (900, 5000, "team256")])
def test_ragged_groups_unaligned_offsets(eng, engine_kind, dtype, lo, hi, variant):
    """Group starts are not multiples of the 16-byte vector width; sizes straddle every kernel variant, incl.
    groups larger than the register-resident capacity (overflow rows are streamed twice)."""
    from oracle import orc

    rng = np.random.default_rng(hi)
    offs = _ragged_offsets(rng, 97, lo, hi)
    y, cols, w = _frame(rng, offs, 5, dtype, weights=True)
    out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, weights=_cuda(w), add_intercept=True,
                            want=("coef", "pred", "resid", "status"))
    ref = orc.batched_least_squares(y, cols, offs, weights=w, add_intercept=True)
    _check(out, ref, dtype)
    if 120 < hi <= 1000:                                 # up to 1 024 rows, aligned or not: f32 one wave per group up to 512 rows, the
        # 256-thread team with one chunk per lane beyond; f64 (6+ columns) two waves per group
        variant = ("team64" if hi <= 512 else "team256_rc1") if dtype == np.float32 else ("team64" if hi <= 256 else "team128")
    if hi == 120:                                        # up to 128 rows: four groups per wave (f64, 6+ columns: four chunks per lane)
        variant = "sub16_rc2" if dtype == np.float32 else "sub16_rc4"
    if hi == 1150 and dtype == np.float32:               # beyond one chunk per lane: two chunks per lane of the 256-thread team
        variant = "team256_rc2"
    big_f64 = hi > 4000 and dtype == np.float64        # neither registers nor the LDS tile hold 5000 f64 rows: streamed path
    ok = eng.last_kernel.startswith("k5_gram_stream") if big_f64 else ((variant in eng.last_kernel) if engine_kind == "valu" else eng.last_kernel.startswith("k1m_"))
    assert ok, eng.last_kernel


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_ridge_weights_intercept_cfg3_shape(eng, engine_kind, dtype):
    """BASELINE configs[2] shape at a size the oracle finishes in seconds: ridge alpha=1 + sample_weights."""
    from oracle import orc
    from refdata import synthetic_groups

    d = synthetic_groups(200, 1000, 8, seed=11, dtype=dtype, with_weights=True)
    out = eng.least_squares(_cuda(d["y"]), [_cuda(c) for c in d["cols"]], d["offsets"], weights=_cuda(d["w"]),
                            alpha=1.0, l1_ratio=0.0, want=("coef", "pred", "resid"))
    ref = orc.batched_least_squares(d["y"], d["cols"], d["offsets"], weights=d["w"], alpha=1.0, l1_ratio=0.0)
    _check(out, ref, dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("k", [12, 15])
@pytest.mark.parametrize("engine", [None, "k2"])
def test_wide_features_mfma_engines(eng, dtype, k, engine):
    """11..15 columns while the rows stay resident: K1 with the three- / four-pass Gram (K1m, the LDS-tile engine, takes them once
    they no longer do, and under POLS_K1_ENGINE=mfma); K2 (rows resident in registers, MFMA Gram) on request."""
    from oracle import orc

    rng = np.random.default_rng(k)
    offs = _ragged_offsets(rng, 21, 300, 900)
    y, cols, w = _frame(rng, offs, k - 1, dtype, weights=True)
    eng.set_option("STATIC_ENGINE", engine)
    try:
        out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, weights=_cuda(w), add_intercept=True,
                                alpha=0.5, want=("coef", "pred", "resid"))
    finally:
        eng.set_option("STATIC_ENGINE", None)
    ref = orc.batched_least_squares(y, cols, offs, weights=w, add_intercept=True, alpha=0.5)
    _check(out, ref, dtype)
    assert eng.last_kernel.startswith("k2_gram_mfma_resident" if engine else f"k1_gram_chol_{'f32' if dtype == np.float32 else 'f64'}_k{k}_w_team"), eng.last_kernel


@pytest.mark.parametrize("method", ["qr", "svd", "chol", "lu", None])
def test_all_solve_methods_agree_with_oracle(eng, golden, method):
    """tests/test_ols.py:54-73: every solve_method gives lstsq's predictions on the seeded frame."""
    z = golden["npz"]
    x, y = z["ols_x"], z["ols_y"]
    out = eng.least_squares(_cuda(y), [_cuda(x[:, 0]), _cuda(x[:, 1])], [0, 1000], solve_method=method,
                            want=("coef", "pred"))
    assert np.allclose(out["coef"].cpu().numpy()[0], z["ols_coef"], rtol=1e-9)
    assert np.allclose(out["pred"].cpu().numpy(), z["ols_pred"], rtol=1e-6, atol=1e-6)


def test_golden_readme_and_make_data(eng, golden):
    from refdata import make_data, sort_by_group

    kat, z = golden["kat"], golden["npz"]
    f = {k: np.asarray(v, dtype=np.float64) for k, v in kat["frame"].items()}
    # README.md:104 -- from_formula("x1 + x2", mode="coefficients"): intercept last
    out = eng.least_squares(f["y"], [f["x1"], f["x2"]], [0, 10], add_intercept=True, want=("coef",))   # host buffers
    assert np.allclose(np.round(out["coef"][0], 6), kat["coefficients_full"], atol=1.1e-6)
    # README.md:112-113 -- .over("group")
    order, offs, keys = sort_by_group(f["group"].astype(np.int64))
    out = eng.least_squares(f["y"][order], [f["x1"][order], f["x2"][order]], offs, add_intercept=True, want=("coef",))
    for g, key in enumerate(keys):
        assert np.allclose(np.round(out["coef"][g], 6), kat["coefficients_group"][str(key)], atol=1.1e-6)
    # README.md:72-76 -- WLS predictions rounded to 2 dp
    out = eng.least_squares(f["y"], [f["x1"], f["x2"]], [0, 10], weights=f["weights"], want=("pred",))
    assert np.array_equal(np.round(out["pred"][:5], 2), kat["predictions_wls_head5_round2"])
    # tests/test_ols.py:475-503 ridge, :456-472 WLS + intercept, :377-401 grouped
    x, y = z["ridge_x"], z["ridge_y"]
    cols = [np.ascontiguousarray(x[:, 0]), np.ascontiguousarray(x[:, 1])]
    out = eng.least_squares(y, cols, [0, 5000], alpha=0.01, solve_method="chol", want=("coef",))
    assert np.allclose(out["coef"][0], z["ridge_coef_chol"], rtol=1e-9)
    out = eng.least_squares(y, cols, [0, 5000], weights=z["wls_w"], add_intercept=True, want=("coef", "pred"))
    assert np.allclose(out["coef"][0], z["wls_coef"], rtol=1e-8) and np.allclose(out["pred"], z["wls_pred"], rtol=1e-6, atol=1e-6)
    d = make_data(n_groups=10)
    order, offs, _ = sort_by_group(d["group"])
    out = eng.least_squares(d["y"][order], [d["x1"][order], d["x2"][order]], offs, want=("coef",))
    assert np.allclose(out["coef"], z["group_coef"], rtol=1e-8)


def test_host_and_device_paths_are_bit_identical(eng, engine_kind):
    rng = np.random.default_rng(5)
    offs = _ragged_offsets(rng, 40, 300, 1200)
    y, cols, w = _frame(rng, offs, 6, np.float32, weights=True)
    a = eng.least_squares(y, cols, offs, weights=w, want=("coef", "pred"))
    b = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, weights=_cuda(w), want=("coef", "pred"))
    assert np.array_equal(a["coef"], b["coef"].cpu().numpy()) and np.array_equal(a["pred"], b["pred"].cpu().numpy())
    c = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, weights=_cuda(w), want=("coef", "pred"))
    assert np.array_equal(b["pred"].cpu().numpy(), c["pred"].cpu().numpy())   # deterministic reduction order


def test_empty_and_tiny_groups(eng, engine_kind):
    from oracle import orc

    rng = np.random.default_rng(9)
    offs = np.array([0, 0, 5, 5, 9, 300, 300], dtype=np.int64)   # empty groups at the front, middle and end
    y, cols, _ = _frame(rng, offs, 3, np.float64)
    out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, want=("coef", "pred", "status"))
    ref = orc.batched_least_squares(y, cols, offs)
    st = out["status"].cpu().numpy()
    assert list(st[[0, 2, 5]]) == [2, 2, 2]                      # POLS_GROUP_EMPTY -> zeros (expressions.rs:357-359)
    assert np.array_equal(out["coef"].cpu().numpy()[[0, 2, 5]], np.zeros((3, 3)))
    ok = [1, 3, 4]
    assert np.allclose(out["coef"].cpu().numpy()[ok], ref["coef"][ok], rtol=1e-6, atol=1e-9)
    assert np.allclose(out["pred"].cpu().numpy(), ref["pred"], rtol=1e-6, atol=1e-8)


def test_weight_zero_gives_nan_prediction_like_reference(eng, engine_kind):
    """sqrt(0) = 0 -> (0*x).beta * (1/0) = NaN in the reference's Python un-scaling (least_squares.py:234-235)."""
    rng = np.random.default_rng(2)
    offs = np.array([0, 64], dtype=np.int64)
    y, cols, w = _frame(rng, offs, 2, np.float64, weights=True)
    w[7] = 0.0
    out = eng.least_squares(y, cols, offs, weights=w, want=("pred",))
    assert np.isnan(out["pred"][7]) and np.isfinite(np.delete(out["pred"], 7)).all()


def test_reference_panics_surface_as_errors(eng):
    from polars_ols_amd import PolsError, PolsPanic

    y = np.ones(8); x = [np.arange(8.0)]
    with pytest.raises(PolsPanic, match="supported solver methods for Ridge"):
        eng.least_squares(y, x, [0, 8], alpha=0.1, solve_method="qr")          # least_squares.rs:366
    with pytest.raises(PolsPanic, match="strictly positive"):
        eng.least_squares(y, x, [0, 8], alpha=-1.0)                           # least_squares.rs:409
    with pytest.raises(PolsError):
        eng.least_squares(y, x, [0, 5], want=("coef",))                       # offsets do not end at n_rows
    with pytest.raises(ValueError):
        eng.least_squares(y, [], [0, 8])                                      # expressions.rs:72


@pytest.mark.parametrize("dtype,groups,rows,k", [(np.float32, 10_000, 1_000, 8)])
def test_full_size_cfg2_properties(eng, engine_kind, dtype, groups, rows, k):
    """BASELINE configs[1] at full size (10k x 1k x 8, f32, predictions): checked through properties that do not
    need the oracle at 10^7 rows: normal equations X^T(y - yhat) = 0 per group, pred + resid == y, linearity in y,
    and oracle parity on a sample of groups."""
    import torch
    from oracle import orc

    g = torch.Generator(device="cuda").manual_seed(0)
    N = groups * rows
    cols = [torch.randn(N, generator=g, device="cuda", dtype=torch.float32) for _ in range(k)]
    y = sum(cols) + 0.1 * torch.randn(N, generator=g, device="cuda", dtype=torch.float32)
    offs = np.arange(groups + 1, dtype=np.int64) * rows
    out = eng.least_squares(y, cols, offs, want=("coef", "pred", "resid", "status"))
    assert int(out["status"].abs().sum()) == 0
    assert torch.equal(out["pred"] + out["resid"], y) or torch.allclose(out["pred"] + out["resid"], y, atol=1e-6)
    r = out["resid"].double().view(groups, rows)
    for c in cols:   # X^T r == 0 (scaled by ||x|| ||r|| ~ sqrt(n) * 0.1 * sqrt(n))
        dot = (c.double().view(groups, rows) * r).sum(1)
        assert float(dot.abs().max()) < 1e-4 * rows * 0.1 * 10
    out2 = eng.least_squares(2.0 * y, cols, offs, want=("pred",))
    assert torch.allclose(out2["pred"], 2.0 * out["pred"], rtol=1e-5, atol=1e-5)
    pick = np.array([0, 1, 4999, 9998, 9999])
    yh = y.cpu().numpy().reshape(groups, rows)[pick].reshape(-1)
    ch = [c.cpu().numpy().reshape(groups, rows)[pick].reshape(-1) for c in cols]
    ref = orc.batched_least_squares(yh, ch, np.arange(len(pick) + 1) * rows)
    assert np.allclose(out["coef"].cpu().numpy()[pick], ref["coef"], rtol=1e-4, atol=1e-4)
    assert np.allclose(out["pred"].cpu().numpy().reshape(groups, rows)[pick].reshape(-1), ref["pred"], rtol=1e-4, atol=1e-4)


def test_full_size_cfg3_properties(eng):
    """BASELINE configs[2] at full size (10k x 1k x 8, f64, ridge alpha = 1 + sample_weights, predictions): the weighted ridge
    normal equations X'W(y - yhat) = alpha * beta hold for every group, pred + resid == y, and sampled groups match the oracle."""
    import torch
    from oracle import orc

    groups, rows, k = 10_000, 1_000, 8
    g = torch.Generator(device="cuda").manual_seed(3)
    N = groups * rows
    cols = [torch.randn(N, generator=g, device="cuda", dtype=torch.float64) for _ in range(k)]
    y = sum(cols) + 0.1 * torch.randn(N, generator=g, device="cuda", dtype=torch.float64)
    w = torch.rand(N, generator=g, device="cuda", dtype=torch.float64) + 0.05
    offs = np.arange(groups + 1, dtype=np.int64) * rows
    out = eng.least_squares(y, cols, offs, weights=w, alpha=1.0, l1_ratio=0.0, want=("coef", "pred", "resid", "status"))
    assert eng.last_kernel.startswith("k1_gram_chol_f64_k8_w")
    assert int(out["status"].abs().sum()) == 0
    assert torch.allclose(out["pred"] + out["resid"], y, atol=1e-12)
    wr = (w * out["resid"]).view(groups, rows)
    for j, c in enumerate(cols):
        lhs = (c.view(groups, rows) * wr).sum(1)                  # x_j' W (y - X beta) = alpha * beta_j
        assert float((lhs - 1.0 * out["coef"][:, j]).abs().max()) < 1e-9 * rows
    pick = np.array([0, 17, 5_000, 9_999])
    sl = lambda t: np.concatenate([t[p * rows:(p + 1) * rows].cpu().numpy() for p in pick])  # noqa: E731
    ref = orc.batched_least_squares(sl(y), [sl(c) for c in cols], np.arange(len(pick) + 1) * rows, weights=sl(w), alpha=1.0, l1_ratio=0.0)
    assert np.allclose(out["coef"].cpu().numpy()[pick], ref["coef"], rtol=1e-6, atol=1e-6)
    assert np.allclose(sl(out["pred"]), ref["pred"], rtol=1e-6, atol=1e-6)


def test_contexts_are_independent_across_host_threads():
    """The boundary's threading contract (Polars calls plugins from its rayon pool): one context per host thread,
    concurrent calls, private streams and scratch.  Four threads run different models on different shapes at once; every
    result must be bit-identical to the same call made alone."""
    import threading

    import torch

    from polars_ols_amd import Engine

    def make(seed, G, n, k, dt):
        g = torch.Generator(device="cuda").manual_seed(seed)
        cols = [torch.randn(G * n, device="cuda", dtype=dt, generator=g) for _ in range(k)]
        y = sum(cols) + 0.1 * torch.randn(G * n, device="cuda", dtype=dt, generator=g)
        return y, cols, np.arange(G + 1, dtype=np.int64) * n

    jobs = [
        (make(1, 3000, 500, 8, torch.float32), dict(want=("coef", "pred"))),
        (make(2, 2000, 700, 6, torch.float64), dict(want=("coef", "pred"), alpha=1.0)),
        (make(3, 500, 1500, 16, torch.float64), dict(want=("coef", "pred"), alpha=0.001, l1_ratio=0.5)),
        (make(4, 40, 2000, 40, torch.float64), dict(want=("coef", "pred"))),                 # wide path (device column tables)
    ]
    torch.cuda.synchronize()

    def run(job, reps, sink):
        (y, cols, offs), kw = job
        e = Engine(0)
        e.use_private_stream()                  # threads must not serialise on torch's default stream
        try:
            for _ in range(reps):
                out = e.least_squares(y, cols, offs, **kw)
                e.synchronize()
            sink.append({k: v.clone() for k, v in out.items() if k in ("coef", "pred")})
        except Exception as exc:  # surfaced in the main thread
            sink.append(exc)
        finally:
            e.close()

    alone = []
    for job in jobs:
        run(job, 1, alone)
    together = [[] for _ in jobs]
    threads = [threading.Thread(target=run, args=(job, 20, sink)) for job, sink in zip(jobs, together)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for a, sink in zip(alone, together):
        assert not isinstance(a, Exception), a
        assert len(sink) == 1 and not isinstance(sink[0], Exception), sink
        for key in ("coef", "pred"):
            assert torch.equal(a[key], sink[0][key]) or torch.allclose(a[key], sink[0][key], rtol=0, atol=0, equal_nan=True), key


@pytest.mark.parametrize("dtype,lo,hi,tag", [(np.float32, 12, 40, "sub8_rc2"), (np.float32, 14, 28, "sub8_rc1"), (np.float32, 40, 120, "sub16_rc2"),
                                             (np.float64, 12, 30, "sub8_rc2"), (np.float64, 20, 60, "sub16_rc2")])
def test_many_tiny_groups_four_per_wave(eng, dtype, lo, hi, tag):
    """K1t at scale: 300 000 ragged groups, four or eight per wave.  Size-independent checks on every group (X'(y - yhat) = 0 through a
    segmented sum, pred + resid == y, exact scaling in y) and oracle parity on a sample that includes both ends of the frame."""
    import torch
    from oracle import orc

    G, k = 300_000, 5
    rng = np.random.default_rng(hi)
    sizes = rng.integers(lo, hi + 1, size=G)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(offs[-1])
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    g = torch.Generator(device="cuda").manual_seed(hi)
    cols = [torch.randn(N, generator=g, device="cuda", dtype=tdt) for _ in range(k)]
    y = sum(cols) * 0.5 + 0.1 * torch.randn(N, generator=g, device="cuda", dtype=tdt) + 0.3
    out = eng.least_squares(y, cols, offs, add_intercept=True, want=("coef", "pred", "resid", "status"))
    family = "k1t_"
    assert tag in eng.last_kernel and eng.last_kernel.startswith(family), eng.last_kernel
    assert int(out["status"].abs().sum()) == 0
    tol = 1e-4 if dtype == np.float32 else 1e-9
    assert torch.allclose(out["pred"] + out["resid"], y, atol=tol)
    gid = torch.repeat_interleave(torch.arange(G, device="cuda"), torch.as_tensor(sizes, device="cuda"))
    r = out["resid"].double()
    for c in cols + [torch.ones_like(y)]:                                          # normal equations, every group
        dot = torch.zeros(G, device="cuda", dtype=torch.float64).index_add_(0, gid, c.double() * r)
        assert float(dot.abs().max()) < (2e-3 if dtype == np.float32 else 1e-9) * hi
    out2 = eng.least_squares(-3.0 * y, cols, offs, add_intercept=True, want=("coef",))
    assert torch.allclose(out2["coef"], -3.0 * out["coef"], rtol=10 * tol, atol=10 * tol)
    pick = np.array([0, 1, 2, 3, 4, 15, 16, 17, G // 2, G - 17, G - 3, G - 2, G - 1])
    yh, ch, oh = [], [[] for _ in cols], [0]
    for p in pick:
        s, e = offs[p], offs[p + 1]
        yh.append(y[s:e].cpu().numpy())
        for j, c in enumerate(cols):
            ch[j].append(c[s:e].cpu().numpy())
        oh.append(oh[-1] + int(e - s))
    ref = orc.batched_least_squares(np.concatenate(yh), [np.concatenate(c) for c in ch], np.array(oh, dtype=np.int64), add_intercept=True)
    rt = TOL[dtype]
    assert np.allclose(out["coef"].cpu().numpy()[pick], ref["coef"], rtol=rt, atol=rt)


@pytest.mark.parametrize("method", ["qr", "svd", "chol", "lu", None])
@pytest.mark.parametrize("mem", ["host", "device"])
def test_configs0_single_group_coefficients(eng, method, mem):
    """BASELINE configs[0]: ONE group, 10 000 rows x 4 f64 features, mode="coefficients" -- the call the reference's plugin
    receives per group (src/expressions.rs:430-446 -> _get_least_squares_coefficients :351-388), every solve_method, against the
    oracle's restatement of the same dispatch at 1e-6 (and lstsq, which the oracle matches to 1e-12 on this shape)."""
    from oracle import orc
    from refdata import make_data

    d = make_data(n_samples=10_000, n_features=4)
    cols = [d[f"x{i + 1}"] for i in range(4)]
    y = d["y"]
    ref = orc.get_coefficients(y, d["x"], solve_method=method)
    if mem == "device":
        out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], [0, 10_000], solve_method=method, want=("coef", "status"))
    else:
        out = eng.least_squares(y, cols, [0, 10_000], solve_method=method, want=("coef", "status"))
    coef = out["coef"].cpu().numpy() if hasattr(out["coef"], "cpu") else out["coef"]
    assert coef.shape == (1, 4) and int(np.asarray(out["status"].cpu() if hasattr(out["status"], "cpu") else out["status"])[0]) == 0
    assert np.allclose(coef[0], ref, rtol=1e-6, atol=1e-6)
    assert np.allclose(coef[0], np.linalg.lstsq(d["x"], y, rcond=None)[0], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("lo,hi,sub", [(100, 250, 32), (100, 250, 64), (130, 252, 32), (30, 60, 16), (30, 60, 32), (30, 60, 64), (40, 120, 16),
                                        (200, 500, 64)])
@pytest.mark.parametrize("k,weights,icpt", [(8, False, False), (5, True, True), (1, False, False), (9, True, True)])
def test_persistent_wave_kernel_ragged_frames(eng, lo, hi, sub, k, weights, icpt):
    """K1p (persistent waves, 64 / sub groups per wave, next groups' rows DMA'd into LDS while these are solved): ragged,
    unaligned groups -- incl. empty ones, a rank-deficient one, and a last chunk that crosses the end of the columns -- against
    the oracle; more groups than persistent waves so that every wave walks several rounds."""
    from oracle import orc

    rng = np.random.default_rng(lo * 7 + k)
    G = 40_000 if hi <= 260 else 12_000
    sizes = rng.integers(lo, hi + 1, size=G)
    sizes[[5, 777, G - 2]] = 0
    if sizes.sum() % 4 == 0:
        sizes[G - 1] += 1                                            # the last chunk crosses the end of the columns
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    y, cols, w = _frame(rng, offs, k, np.float32, weights=weights)
    if k > 1:
        s, e = offs[11], offs[12]
        cols[1][s:e] = cols[0][s:e]                                  # rank-deficient group: flagged, re-solved by the SVD pass
    eng.set_option("K1_PERSIST", "1")
    eng.set_option("K1_PERSIST_SUB", str(sub))
    try:
        out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, weights=None if w is None else _cuda(w), add_intercept=icpt,
                                want=("coef", "pred", "resid", "status"))
        name = eng.last_kernel
    finally:
        eng.set_option("K1_PERSIST", None)
        eng.set_option("K1_PERSIST_SUB", None)
    assert name.startswith(f"k1p_gram_chol_persistent_f32_k{k + int(icpt)}") and f"_sub{sub}_" in name, name
    ref = orc.batched_least_squares(y, cols, offs, weights=w, add_intercept=icpt)
    st = out["status"].cpu().numpy()
    assert list(st[[5, 777, G - 2]]) == [2, 2, 2]
    if k > 1:
        assert st[11] == 1 and (np.delete(st, [5, 11, 777, G - 2]) == 0).all()
    # every group, the rank-deficient one included: the fix-up pass returns the reference's pivoted-QR basic solution there
    got_c, got_p, got_r = (out[q].double().cpu().numpy() for q in ("coef", "pred", "resid"))
    assert np.allclose(got_c, ref["coef"], rtol=1e-4, atol=1e-4), float(np.abs(got_c - ref["coef"]).max())
    assert np.allclose(got_p, ref["pred"], rtol=1e-4, atol=1e-4), float(np.abs(got_p - ref["pred"]).max())
    assert np.allclose(got_r, ref["resid"], rtol=1e-4, atol=1e-4)


def test_persistent_wave_kernel_matches_one_shot_kernel(eng):
    """The persistent kernel (on request) against the default one-shot kernel of the same frame."""
    import torch

    rng = np.random.default_rng(3)
    offs = np.arange(0, 30_001 * 200, 200, dtype=np.int64)
    y, cols, _ = _frame(rng, offs, 8, np.float32)
    yy, cc = _cuda(y), [_cuda(c) for c in cols]
    eng.set_option("K1_PERSIST", "0")
    a = eng.least_squares(yy, cc, offs, want=("coef", "pred"))
    ka = eng.last_kernel
    eng.set_option("K1_PERSIST", "1")
    eng.set_option("K1_PERSIST_SUB", "64")
    b = eng.least_squares(yy, cc, offs, want=("coef", "pred"))
    kb = eng.last_kernel
    eng.set_option("K1_PERSIST", None)
    eng.set_option("K1_PERSIST_SUB", None)
    eng.synchronize()
    assert ka.startswith("k1_gram_chol_f32_k8_team64_rc1") and kb.startswith("k1p_"), (ka, kb)
    assert torch.allclose(a["coef"], b["coef"], rtol=1e-5, atol=1e-6) and torch.allclose(a["pred"], b["pred"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("k,icpt", [(8, True), (9, True), (10, False), (10, True), (12, False), (13, False), (14, True)])
@pytest.mark.parametrize("lo,hi", [(1000, 1000), (600, 1010), (150, 250), (300, 500)])
def test_nine_and_ten_columns_take_the_multi_pass_valu_kernels(eng, dtype, k, icpt, lo, hi):
    """8 features + intercept (the smoke() shape) and 10 columns while the rows stay resident: K1 with the Gram accumulated in
    passes (a third / half of the accumulators live at a time), not the LDS-tile engine -- aligned and ragged frames, weights."""
    from oracle import orc

    rng = np.random.default_rng(k * 100 + hi)
    offs = _ragged_offsets(rng, 300, lo, hi)
    y, cols, w = _frame(rng, offs, k, dtype, weights=True)
    out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, weights=_cuda(w), add_intercept=icpt,
                            want=("coef", "pred", "resid", "status"))
    name = eng.last_kernel
    assert name.startswith(f"k1_gram_chol_{'f32' if dtype == np.float32 else 'f64'}_k{k + int(icpt)}_w_team"), name
    if hi > 256:
        assert any(tag in name for tag in ("_p2", "_p3", "_p4")), name
    ref = orc.batched_least_squares(y, cols, offs, weights=w, add_intercept=icpt)
    _check(out, ref, dtype)
    assert int(out["status"].abs().sum()) == 0


def test_calls_under_alternating_torch_streams_stay_ordered(eng):
    """The context's scratch (offsets, Gram matrices, status words) is shared by its calls: launches issued under different
    torch streams must not overlap on it -- pols_set_stream makes the new stream wait for the old one."""
    import torch
    from oracle import orc

    rng = np.random.default_rng(5)
    frames = []
    for i, (G, n, k) in enumerate(((3000, 1000, 8), (40_000, 64, 4), (2000, 1500, 12))):
        offs = np.arange(0, (G + 1) * n, n, dtype=np.int64)
        y, cols, _ = _frame(rng, offs, k, np.float32)
        frames.append((offs, y, cols, _cuda(y), [_cuda(c) for c in cols]))
    streams = [torch.cuda.Stream() for _ in range(3)]
    torch.cuda.synchronize()
    outs = []
    for rep in range(4):
        for i, (offs, y, cols, yy, cc) in enumerate(frames):
            with torch.cuda.stream(streams[(i + rep) % 3]):
                outs.append((i, eng.least_squares(yy, cc, offs, want=("coef",))["coef"]))
    torch.cuda.synchronize()
    refs = [orc.batched_least_squares(y, cols, offs)["coef"] for offs, y, cols, _, _ in frames]
    for i, c in outs:
        assert np.allclose(c.double().cpu().numpy(), refs[i], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("k,icpt", [(16, False), (17, True), (20, True), (24, False), (27, True), (31, False), (30, True)])
@pytest.mark.parametrize("lo,hi", [(500, 500), (200, 505), (80, 250)])
def test_sixteen_to_thirtyone_columns_resident_multi_pass(eng, dtype, k, icpt, lo, hi):
    """16..31 columns while every row stays resident (one chunk per lane: up to 1 024 f32 / 512 f64 rows): K1 with the Gram in up to
    15 passes and the row-resident right-looking Cholesky -- X is read once, where the streamed path reads it twice.  Aligned and
    ragged frames, weights, intercept, a rank-deficient group for the SVD fix-up."""
    from oracle import orc

    rng = np.random.default_rng(k * 100 + hi)
    offs = _ragged_offsets(rng, 120, lo, hi)
    y, cols, w = _frame(rng, offs, k, dtype, weights=True)
    s, e = offs[7], offs[8]
    cols[2][s:e] = cols[5][s:e]                                      # rank-deficient group
    out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, weights=_cuda(w), add_intercept=icpt,
                            want=("coef", "pred", "resid", "status"))
    name = eng.last_kernel
    kt = k + int(icpt)
    mx = int(np.diff(offs).max())
    # f64, round 5 (scripts/ab_wide_f64.py): K2w is ahead of the VALU passes at 23-24 columns at every length, at 22 from ~200 rows, at 20-21 from ~224
    # round 6: groups of 257 .. 512 rows (K1's 256-thread team at three waves per SIMD) stay with K1 through 23 columns
    k2w_wins = dtype == np.float64 and (kt >= 24 or (mx <= 256 and (kt >= 23 or (kt == 22 and mx >= 192) or (kt >= 20 and mx >= 224))))
    if k2w_wins and kt <= 24:
        assert name.startswith(f"k2w_gram_mfma_resident2_f64_k{kt}_w"), name
    elif dtype == np.float32 or 17 <= kt <= 24:                      # f64: K2 keeps 16 columns, K2w 25+
        assert name.startswith(f"k1_gram_chol_{'f32' if dtype == np.float32 else 'f64'}_k{kt}_w_team"), name
    ref = orc.batched_least_squares(y, cols, offs, weights=w, add_intercept=icpt)
    st = out["status"].cpu().numpy()
    assert st[7] == 1 and (np.delete(st, 7) == 0).all(), st[:12]
    tol = TOL[dtype]
    # every group, the rank-deficient one included (the fix-up pass returns the reference's pivoted-QR basic solution there)
    got_c, got_p, got_r = (out[q].double().cpu().numpy() for q in ("coef", "pred", "resid"))
    assert np.allclose(got_c, ref["coef"], rtol=tol, atol=tol), float(np.abs(got_c - ref["coef"]).max())
    assert np.allclose(got_p, ref["pred"], rtol=tol, atol=tol), float(np.abs(got_p - ref["pred"]).max())
    assert np.allclose(got_r, ref["resid"], rtol=tol, atol=tol)


@pytest.mark.parametrize("k,icpt,weights", [(16, False, False), (17, True, True), (18, False, True), (23, False, False), (24, True, True), (25, False, False),
                                            (26, False, True), (27, True, False), (28, False, False), (30, True, True), (31, False, True)])
@pytest.mark.parametrize("lo,hi", [(1000, 1000), (520, 1021)])
def test_f32_wide_columns_on_the_256_thread_team(eng, k, icpt, weights, lo, hi):
    """Round 6: f32 groups of 513 .. 1 024 rows at 16 .. 31 columns (the 256-thread team, one 16-byte chunk per lane).  The Gram passes are as
    short as three (23+) / four (16-18) waves per SIMD allow -- up to 27 passes at 31 columns -- and from 26 columns the solving wave parks its
    resident rows (the sqrt(w) vector included) in LDS while it solves and reads them back for the predictions.  Aligned and ragged frames,
    weights, intercept, a rank-deficient group for the fix-up pass; every group against the oracle."""
    from oracle import orc

    rng = np.random.default_rng(k * 37 + hi)
    offs = _ragged_offsets(rng, 40, lo, hi)
    y, cols, w = _frame(rng, offs, k, np.float32, weights=weights)
    s, e = offs[5], offs[6]
    cols[1][s:e] = cols[3][s:e]                                      # rank-deficient group
    out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, weights=None if w is None else _cuda(w), add_intercept=icpt,
                            want=("coef", "pred", "resid", "status"))
    kt = k + int(icpt)
    assert eng.last_kernel.startswith(f"k1_gram_chol_f32_k{kt}{'_w' if weights else ''}_team256_rc1"), eng.last_kernel
    ref = orc.batched_least_squares(y, cols, offs, weights=w, add_intercept=icpt)
    st = out["status"].cpu().numpy()
    assert st[5] == 1 and (np.delete(st, 5) == 0).all(), st[:10]
    tol = TOL[np.float32]
    got_c, got_p, got_r = (out[q].double().cpu().numpy() for q in ("coef", "pred", "resid"))
    assert np.allclose(got_c, ref["coef"], rtol=tol, atol=tol), float(np.abs(got_c - ref["coef"]).max())
    assert np.allclose(got_p, ref["pred"], rtol=tol, atol=tol), float(np.abs(got_p - ref["pred"]).max())
    assert np.allclose(got_r, ref["resid"], rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("k,weights,icpt", [(8, False, False), (5, True, True), (7, False, True), (6, True, False), (4, False, False), (1, False, False)])
@pytest.mark.parametrize("slots", [8, 16])
def test_eight_lane_teams(eng, dtype, k, weights, icpt, slots):
    """K1t with eight groups per wave (eight-lane teams, the team Gram reduce-scattered over DPP pairs): the default for frames whose
    every group fits 16 chunk slots (two chunks per lane beyond eight); POLS_K1T_SUB8=0 is the 16-lane form.  Ragged,
    unaligned groups incl. empty ones and a rank-deficient one; against the oracle and against the 16-lane form."""
    from oracle import orc

    vec = 4 if dtype == np.float32 else 2
    rng = np.random.default_rng(17 + k)
    G = 9_000
    sizes = rng.integers(min(k + 3, slots * vec - vec), slots * vec - vec + 1, size=G)
    sizes[[7, 4_000, G - 1]] = 0
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    y, cols, w = _frame(rng, offs, k, dtype, weights=weights)
    if k > 1:
        s, e = offs[11], offs[12]
        cols[1][s:e] = cols[0][s:e]                                  # rank-deficient group: flagged, re-solved by the fix-up pass
    args = dict(weights=None if w is None else _cuda(w), add_intercept=icpt, want=("coef", "pred", "resid", "status"))
    kt = k + int(icpt)
    try:
        out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, **args)
        name = eng.last_kernel
        want = "_sub8_rc1" if slots == 8 else ("_sub8_rc2" if kt >= 6 or kt <= 3 else "_sub16_rc1")   # (4-5 columns keep 16-lane teams beyond 8 slots)
        assert name.startswith("k1t_gram_chol_") and name.endswith(want), name
        eng.set_option("K1T_SUB8", "0")
        one = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, **args)
        assert "_sub16_rc" in eng.last_kernel, eng.last_kernel
    finally:
        eng.set_option("K1T_SUB8", None)
    ref = orc.batched_least_squares(y, cols, offs, weights=w, add_intercept=icpt)
    tol = TOL[dtype]
    st = out["status"].cpu().numpy()
    assert list(st[[7, 4_000, G - 1]]) == [2, 2, 2] and np.array_equal(st, one["status"].cpu().numpy())
    for q in ("coef", "pred", "resid"):
        got = out[q].double().cpu().numpy()
        assert np.allclose(got, ref[q], rtol=tol, atol=tol), (q, float(np.abs(got - ref[q]).max()))
        assert np.allclose(got, one[q].double().cpu().numpy(), rtol=tol, atol=tol), q


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("k,weights,icpt,alpha,ragged", [(8, False, False, None, False), (8, True, False, 1.0, True), (7, False, True, None, True),
                                                        (4, True, True, None, False), (1, False, False, None, True), (6, False, False, 0.5, False)])
def test_decade_sized_groups_stay_register_resident(eng, dtype, k, weights, icpt, alpha, ragged):
    """Round 5: groups of 2 049..4 096 f32 / 1 025..2 048 f64 rows (ten years of trading days per asset) at up to 8 columns take FOUR chunks per
    lane of K1's 256-thread team -- X read once -- instead of K2 / K1m / the streamed path.  Aligned (FAST) and ragged (EDGE) frames, weights,
    intercept, ridge; every group against the oracle; an empty and a short group in the frame."""
    from oracle import orc

    vec = 4 if dtype == np.float32 else 2
    cap = 256 * 4 * vec
    rng = np.random.default_rng(k * 17 + int(ragged))
    if ragged:
        sizes = np.concatenate([[cap - 3, 0, 37, cap // 2 + 5], rng.integers(cap // 2 + 1, cap - 2, size=9)])
    else:
        sizes = np.array([cap, cap - vec * 8, cap // 2 + vec * 4, cap, cap - vec * 100] * 2)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    y, cols, w = _frame(rng, offs, k, dtype, weights=weights)
    kw = dict(add_intercept=icpt)
    if alpha is not None:
        kw.update(alpha=alpha, l1_ratio=0.0)
    out = eng.least_squares(_cuda(y), [_cuda(c) for c in cols], offs, weights=None if w is None else _cuda(w),
                            want=("coef", "pred", "resid", "status"), **kw)
    assert "team256_rc4" in eng.last_kernel, eng.last_kernel
    ref = orc.batched_least_squares(y, cols, offs, weights=w, **kw)
    _check(out, ref, dtype)
    st = out["status"].cpu().numpy()
    assert (st[sizes > 0] == 0).all() and (st[sizes == 0] == 2).all()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("dist,k,weights,icpt,policy", [
    ("lognormal", 8, False, False, None), ("lognormal", 5, True, True, None), ("bimodal", 8, False, True, None),
    ("bimodal", 3, False, False, "drop"), ("lognormal", 12, False, False, None), ("bimodal", 20, False, False, None),
    ("with_empty_and_tiny", 6, False, False, None),
])
def test_size_classes_on_widely_spread_group_sizes(eng, dtype, dist, k, weights, icpt, policy):
    """Round 5: a frame whose group sizes spread widely (most groups a few hundred rows, a tail of thousands; or many short groups next to a
    few long ones) is served by TWO launches of the K1 family, each sized for its own size class -- the workgroups of the other class exit
    after reading their offsets (`pick_size_classes`, api.hip).  Every group against the oracle, and against the one-launch form (NO_CLASSES)."""
    from oracle import orc

    if dtype == np.float64 and k > 16:
        pytest.skip("f64 beyond 16 columns x 1 000 rows is K2w's, one launch")
    rng = np.random.default_rng(len(dist) * 100 + k)
    G = 12_000                                                        # (an extra launch has to pay for itself: small frames keep one)
    if dist == "lognormal":
        top = 1000 if k > 10 else (1900 if dtype == np.float64 else 3900)        # (what the register-resident K1 forms hold at this width)
        sizes = np.clip(rng.lognormal(np.log(200 if k <= 10 else 60), 0.8, size=G).astype(np.int64), 5, top)
        sizes[17] = top
    elif dist == "bimodal":
        sizes = np.where(rng.random(G) < 0.9, rng.integers(30, 60, size=G), rng.integers(900, 1001, size=G))
    else:
        sizes = np.where(rng.random(G) < 0.9, rng.integers(0, 40, size=G), rng.integers(700, 1001, size=G))
        sizes[3] = 0
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    y, cols, w = _frame(rng, offs, k, dtype, weights=weights)
    kw = dict(add_intercept=icpt)
    if policy:
        y = y.copy(); y[rng.random(len(y)) < 0.02] = np.nan
        kw["null_policy"] = policy
    args = (_cuda(y), [_cuda(c) for c in cols], offs)
    wd = None if w is None else _cuda(w)
    out = eng.least_squares(*args, weights=wd, want=("coef", "pred", "resid", "status"), **kw)
    name = eng.last_kernel
    assert " | " in name and name.count("k1") >= 2, name             # two launches: the long groups' kernel | the short groups' kernel
    eng.set_option("NO_CLASSES", "1")
    try:
        one = eng.least_squares(*args, weights=wd, want=("coef", "pred", "resid", "status"), **kw)
        assert " | " not in eng.last_kernel
    finally:
        eng.set_option("NO_CLASSES", None)
    tol = TOL[dtype]
    kt = k + int(icpt)
    full = sizes > 3 * kt                                             # (shorter groups: near-singular or minimum-norm fits, compared through the oracle below)
    fr = np.repeat(full, sizes)
    for key in ("coef", "pred", "resid"):
        a_, b_ = out[key].double().cpu().numpy(), one[key].double().cpu().numpy()
        if key == "coef":
            a_, b_ = a_.reshape(-1, kt)[full], b_.reshape(-1, kt)[full]
        else:
            a_, b_ = a_[fr], b_[fr]
        assert np.allclose(a_, b_, rtol=10 * tol, atol=10 * tol, equal_nan=True), key
    assert (out["status"].cpu().numpy() == one["status"].cpu().numpy()).all()
    if not policy:
        ref = orc.batched_least_squares(y, cols, offs, weights=w, **kw)
        got = out["coef"].double().cpu().numpy().reshape(-1, kt)
        assert np.allclose(got[full], np.asarray(ref["coef"]).reshape(-1, kt)[full], rtol=tol, atol=tol), float(np.abs(got[full] - np.asarray(ref["coef"]).reshape(-1, kt)[full]).max())
        assert np.allclose(out["pred"].double().cpu().numpy()[fr], np.asarray(ref["pred"])[fr], rtol=tol, atol=tol)
        short = (sizes > 0) & (sizes < kt)
        if short.any():                                                # the minimum-norm branch (K6s) still sees the short groups of both classes
            assert np.allclose(got[short], np.asarray(ref["coef"]).reshape(-1, kt)[short], rtol=10 * tol, atol=10 * tol)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("k,weights,icpt,policy", [(8, False, False, None), (5, True, True, None), (3, False, False, "drop"), (10, False, True, None)])
def test_size_classes_with_a_streamed_top_class(eng, dtype, k, weights, icpt, policy):
    """The largest groups do not fit the K1 family's registers (beyond 4 096 f32 / 2 048 f64 rows) but most groups are short: the long ones form a
    class of their own on the streamed path (segment tables over THEIR list, gram_solve mapping list positions back to group ids), the others are
    classed among the K1 kernels.  One frame-wide streamed call was 2.2 TB/s on such a frame.  Against the oracle and the one-launch form."""
    from oracle import orc

    rng = np.random.default_rng(1000 + k)
    G = 9000
    cap = 9000 if dtype == np.float32 else 5000
    sizes = np.clip(rng.lognormal(np.log(250), 0.9, size=G).astype(np.int64), 0, cap)
    sizes[rng.integers(0, G, size=25)] = rng.integers(cap // 2 + 200, cap, size=25)
    sizes[11] = 0
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    y, cols, w = _frame(rng, offs, k, dtype, weights=weights)
    kw = dict(add_intercept=icpt)
    if policy:
        y = y.copy(); y[rng.random(len(y)) < 0.02] = np.nan
        kw["null_policy"] = policy
    args = (_cuda(y), [_cuda(c) for c in cols], offs)
    wd = None if w is None else _cuda(w)
    out = eng.least_squares(*args, weights=wd, want=("coef", "pred", "resid", "status"), **kw)
    name = eng.last_kernel
    assert name.startswith("k5_gram_stream") and " | k1" in name, name     # the streamed top class | the K1 launches
    eng.set_option("NO_CLASSES", "1")
    try:
        one = eng.least_squares(*args, weights=wd, want=("coef", "pred", "resid", "status"), **kw)
        assert " | " not in eng.last_kernel
    finally:
        eng.set_option("NO_CLASSES", None)
    tol = TOL[dtype]
    kt = k + int(icpt)
    full = sizes > 3 * kt
    fr = np.repeat(full, sizes)
    assert (out["status"].cpu().numpy() == one["status"].cpu().numpy()).all()
    for key in ("coef", "pred", "resid"):
        a_, b_ = out[key].double().cpu().numpy(), one[key].double().cpu().numpy()
        a_, b_ = (a_.reshape(-1, kt)[full], b_.reshape(-1, kt)[full]) if key == "coef" else (a_[fr], b_[fr])
        assert np.allclose(a_, b_, rtol=10 * tol, atol=10 * tol, equal_nan=True), key
    if not policy:
        ref = orc.batched_least_squares(y, cols, offs, weights=w, **kw)
        got = out["coef"].double().cpu().numpy().reshape(-1, kt)
        assert np.allclose(got[full], np.asarray(ref["coef"]).reshape(-1, kt)[full], rtol=tol, atol=tol)
        assert np.allclose(out["pred"].double().cpu().numpy()[fr], np.asarray(ref["pred"])[fr], rtol=tol, atol=tol)
