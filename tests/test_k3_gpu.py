"""GPU parity of K3 (recursive least squares) through the C-ABI against the CPU oracle
(oracle restates src/least_squares.rs:494-598) and the README's RLS known answers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from polars_ols_amd import Engine

    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(params=["seq", "scan"])
def rls_engine(request, eng):
    """K3 (wave-per-sequence P-form recursion) and K3s (chunk-parallel information-form scan) must both match."""
    eng.set_option("RLS_ENGINE", request.param)
    yield request.param
    eng.set_option("RLS_ENGINE", None)


def _cuda(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _np(t):
    return t.double().cpu().numpy() if hasattr(t, "cpu") else np.asarray(t, dtype=np.float64)


def _masked(pred, valid):
    """make_predictions with the validity mask (src/expressions.rs:640-645): rows the mask leaves out are nulls (NaN here)"""
    return np.where(np.asarray(valid).astype(bool), pred, np.nan)


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("k", [1, 2, 6, 8])
@pytest.mark.parametrize("half_life,p0,mean", [(None, 10.0, None), (21.0, 10.0, None), (252.0, 0.01, 0.25), (None, 1e6, None)])
def test_rls_many_groups(eng, rls_engine, dtype, tol, k, half_life, p0, mean):
    from oracle import orc

    rng = np.random.default_rng(k)
    sizes = rng.integers(1, 400, size=37)
    sizes[3] = 0                                     # an empty group
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(offs[-1])
    cols = [rng.standard_normal(N).astype(dtype) for _ in range(k)]
    y = (sum(cols).astype(np.float64) + 0.1 * rng.standard_normal(N)).astype(dtype)
    valid = (rng.random(N) > 0.1).astype(np.uint8)
    mean0 = None if mean is None else [mean] * k
    out = eng.recursive_least_squares(_cuda(y), [_cuda(c) for c in cols], offs, valid=_cuda(valid), half_life=half_life,
                                      initial_state_covariance=p0, initial_state_mean=mean0)
    ref = orc.batched_rls(y, cols, offs, half_life=half_life, initial_state_covariance=p0, initial_state_mean=mean0,
                          is_valid=valid)
    assert eng.last_kernel.startswith("k3s_" if rls_engine == "scan" else "k3_rls")
    # north_star's bound on every row, the diffuse prior of tests/test_ols.py:633-681 (p0 = 1e6) included: the scan engine solves the
    # information matrix on every row (cond(A) eps ~ 1e-9), it never propagates an inverted ill-conditioned matrix
    assert np.allclose(_np(out["coef"]), ref["coef"], rtol=tol, atol=tol), float(np.abs(_np(out["coef"]) - ref["coef"]).max())
    assert np.allclose(_np(out["pred"]), _masked(ref["pred"], valid), rtol=tol, atol=tol, equal_nan=True)


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("k", [9, 10, 16, 32])
@pytest.mark.parametrize("half_life,p0,mean", [(None, 10.0, None), (252.0, 1.0, 0.25)])
def test_rls_wide_features(eng, dtype, tol, k, half_life, p0, mean):
    """9..32 features: 9 and 10 on the row-parallel kernel, 11+ on the wave-per-chunk P-form kernel (k4p_wide.hip), several chunks per sequence
    (decayed totals + scan + one inversion per chunk start), invalid rows."""
    from oracle import orc

    rng = np.random.default_rng(100 + k)
    sizes = rng.integers(1, 300, size=11)
    sizes[2] = 0
    sizes[5] = 2_500
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(offs[-1])
    cols = [rng.standard_normal(N).astype(dtype) for _ in range(k)]
    y = (sum(cols).astype(np.float64) + 0.1 * rng.standard_normal(N)).astype(dtype)
    valid = (rng.random(N) > 0.1).astype(np.uint8)
    mean0 = None if mean is None else [mean] * k
    out = eng.recursive_least_squares(_cuda(y), [_cuda(c) for c in cols], offs, valid=_cuda(valid), half_life=half_life,
                                      initial_state_covariance=p0, initial_state_mean=mean0)
    ref = orc.batched_rls(y, cols, offs, half_life=half_life, initial_state_covariance=p0, initial_state_mean=mean0,
                          is_valid=valid)
    assert eng.last_kernel.startswith("k3s_rls_rows" if k <= 10 else "k3p_")     # (up to 10 features fit the row-parallel kernel; beyond: k4p_wide.hip)
    assert np.allclose(_np(out["coef"]), ref["coef"], rtol=tol, atol=tol), float(np.abs(_np(out["coef"]) - ref["coef"]).max())
    assert np.allclose(_np(out["pred"]), _masked(ref["pred"], valid), rtol=tol, atol=tol, equal_nan=True)
    if k > 10:                                    # the same cut sequences with several chunks per wave (totals, scan and walk)
        eng.set_option("K4P_LPS", "16" if k <= 16 else "32")
        try:
            out = eng.recursive_least_squares(_cuda(y), [_cuda(c) for c in cols], offs, valid=_cuda(valid), half_life=half_life,
                                              initial_state_covariance=p0, initial_state_mean=mean0)
            assert eng.last_kernel.endswith("_x4" if k <= 16 else "_x2"), eng.last_kernel
        finally:
            eng.set_option("K4P_LPS", None)
        assert np.allclose(_np(out["coef"]), ref["coef"], rtol=tol, atol=tol), float(np.abs(_np(out["coef"]) - ref["coef"]).max())
        assert np.allclose(_np(out["pred"]), _masked(ref["pred"], valid), rtol=tol, atol=tol, equal_nan=True)


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("k,half_life,p0,mean", [(33, None, 10.0, None), (100, 252.0, 1.0, 0.25), (128, None, 10.0, None),
                                                 (129, 300.0, 10.0, None), (200, None, 2.0, 0.1)])
def test_rls_inverse_propagation_33_features_and_up(eng, dtype, tol, k, half_life, p0, mean):
    """33+ features (k4x_inverse.hip): the reference's P-form update, chunk-parallel (README benchmark shape: 100 features); beyond
    128 the K x K state of a chunk lives in HBM / L2 instead of LDS."""
    from oracle import orc

    rng = np.random.default_rng(300 + k)
    sizes = np.array([900, 0, 2_300, 40])
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(offs[-1])
    cols = [rng.standard_normal(N).astype(dtype) for _ in range(k)]
    y = (sum(cols).astype(np.float64) + 0.1 * rng.standard_normal(N)).astype(dtype)
    valid = (rng.random(N) > 0.1).astype(np.uint8)
    mean0 = None if mean is None else [mean] * k
    out = eng.recursive_least_squares(_cuda(y), [_cuda(c) for c in cols], offs, valid=_cuda(valid), half_life=half_life,
                                      initial_state_covariance=p0, initial_state_mean=mean0)
    ref = orc.batched_rls(y, cols, offs, half_life=half_life, initial_state_covariance=p0, initial_state_mean=mean0, is_valid=valid)
    assert eng.last_kernel.startswith("k3y_" if k > 128 else "k3x_")
    assert np.allclose(_np(out["coef"]), ref["coef"], rtol=tol, atol=tol), float(np.abs(_np(out["coef"]) - ref["coef"]).max())
    assert np.allclose(_np(out["pred"]), _masked(ref["pred"], valid), rtol=tol, atol=tol, equal_nan=True)


def test_rls_readme_known_answer(eng, golden):
    """README.md:133-137: rls(x1, x2, mode="coefficients").over("group") with the default prior."""
    from refdata import sort_by_group

    kat = golden["kat"]
    f = {k: np.asarray(v, dtype=np.float64) for k, v in kat["frame"].items()}
    order, offs, _ = sort_by_group(f["group"].astype(np.int64))
    out = eng.recursive_least_squares(f["y"][order], [f["x1"][order], f["x2"][order]], offs, want=("coef",))   # host buffers
    assert np.allclose(np.round(out["coef"][:5], 6), kat["rls_coefficients_head5"], atol=1.1e-6)


def test_rls_expanding_equals_ols_on_golden(eng, golden):
    """tests/test_ols.py:633-681: expanding RLS with a diffuse prior and nulls ends at the full-sample OLS."""
    z = golden["npz"]
    x, y = z["nulls_x"], z["nulls_y"]
    valid = (~np.isnan(x).any(axis=1) & ~np.isnan(y)).astype(np.uint8)
    out = eng.recursive_least_squares(np.nan_to_num(y), [np.nan_to_num(x[:, 0]), np.nan_to_num(x[:, 1])], [0, len(y)],
                                      valid=valid, initial_state_covariance=1e6, want=("coef",))
    assert np.allclose(out["coef"][-1], z["rls_expanding_last"], rtol=1e-4, atol=1e-4)


def test_rls_cfg4_full_size_single_sequence(eng, rls_engine):
    """BASELINE configs[3]: one sequence of 1 000 000 rows, 6 features, half_life = 21, f64 -- the oracle is a
    sequential C loop, fast enough to check every row at full size."""
    from oracle import orc

    rng = np.random.default_rng(4)
    n, k = 1_000_000, 6
    cols = [rng.standard_normal(n) for _ in range(k)]
    y = sum(cols) + 0.1 * rng.standard_normal(n)
    out = eng.recursive_least_squares(_cuda(y), [_cuda(c) for c in cols], [0, n], half_life=21.0)
    ref = orc.batched_rls(y, cols, [0, n], half_life=21.0)
    assert np.allclose(_np(out["coef"]), ref["coef"], rtol=1e-6, atol=1e-6)
    assert np.allclose(_np(out["pred"]), ref["pred"], rtol=1e-6, atol=1e-6)


def test_rls_cfg4_full_size_default_route_is_one_launch(eng):
    """BASELINE configs[3] on the DEFAULT route: half_life = 21 gives ff^768 = 2^-36.6, so a tile's carry-in is a function of the rows right in
    front of it.  Up to 6 features that is tile t - 1's own aggregate: the LOOK-BACK-ONE form (k3c_scan.hip MODE 3: every tile publishes its
    aggregate -- it falls out of the tile's scan; POLS_RLS_EARLY=1: computed and published before the scan -- and picks up its predecessor's);
    POLS_RLS_ENGINE=halo re-accumulates the 768 rows instead (MODE 2), and
    POLS_RLS_SPINS=0 makes every wave of the look-back form take its fallback (the slow halo) -- the path a dispatch order that starts tile t
    before tile t - 1 would take.  Every row of the 1 000 000 of all three against the sequential oracle at north_star's 1e-6, and against the
    exact scan (POLS_RLS_ENGINE=scan) far inside it."""
    from oracle import orc

    rng = np.random.default_rng(4)
    n, k = 1_000_000, 6
    cols = [rng.standard_normal(n) for _ in range(k)]
    y = sum(cols) + 0.1 * rng.standard_normal(n)
    dy, dc = _cuda(y), [_cuda(c) for c in cols]
    ref = orc.batched_rls(y, cols, [0, n], half_life=21.0)
    eng.set_option("RLS_ENGINE", "scan")
    try:
        ex = eng.recursive_least_squares(dy, dc, [0, n], half_life=21.0, null_free=True)
        assert eng.last_kernel == "k3s_rls_rows_f64"
    finally:
        eng.set_option("RLS_ENGINE", None)
    for opts, name in (({}, "k3s_rls_rows_lookback_f64"), ({"RLS_ENGINE": "halo"}, "k3s_rls_rows_halo_f64"), ({"RLS_SPINS": "0"}, "k3s_rls_rows_lookback_f64"),
                       ({"RLS_EARLY": "1"}, "k3s_rls_rows_lookback_f64"), ({"RLS_EARLY": "1", "RLS_SPINS": "0"}, "k3s_rls_rows_lookback_f64")):
        for key, v in opts.items():
            eng.set_option(key, v)
        try:
            for rep in range(3):                              # (repeated launches: the record granules of one launch must never satisfy the next)
                out = eng.recursive_least_squares(dy, dc, [0, n], half_life=21.0, null_free=True)
                assert eng.last_kernel == name
                assert np.allclose(_np(out["coef"]), ref["coef"], rtol=1e-6, atol=1e-6), (opts, rep)
                assert np.allclose(_np(out["pred"]), ref["pred"], rtol=1e-6, atol=1e-6)
        finally:
            for key in opts:
                eng.set_option(key, None)
        # what these forms drop is <= 2^-36.6 = 1e-11 of the state 768 (1 024) rows back
        assert float(np.abs(_np(out["coef"]) - _np(ex["coef"])).max()) < 1e-9, opts


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("k", [1, 3, 6, 7, 9, 10])
@pytest.mark.parametrize("half_life,halo", [(5.0, True), (21.0, True), (21.5, True), (43.0, True), (56.5, True), (57.0, False), (252.0, False), (None, False)])
def test_rls_halo_route_boundary(eng, dtype, tol, k, half_life, halo):
    """The halo form's route: taken iff ff^H <= 2^-36 for some H = 256 .. 2 048 rows (half_life <= 56.9: 5 -> 256 rows, 21 -> 768, 21.5 -> 1 024,
    43 -> 1 792, 56.5 -> 2 048), on null-free frames with a sequence longer than a tile, up to 9 features; half_life = None / longer half-lives keep the scan.  Frame: sequences of 1 row to
    20 000 rows, several starting exactly on tile boundaries, one straddling four tiles; a diffuse prior with a mean (the prior's own decay
    up to a tile is exact: first row of the sequence from the per-tile table) -- every row against the oracle."""
    from oracle import orc

    rng = np.random.default_rng(31 * k + int(half_life or 0))
    sizes = np.array([20_000, 1, 3, 1020, 1024, 2048, 5, 4099, 700, 2, 3500, 1024 * 3 - 7, 7, 9000], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(offs[-1])
    cols = [rng.standard_normal(N).astype(dtype) for _ in range(k)]
    y = (sum(cols).astype(np.float64) + 0.1 * rng.standard_normal(N)).astype(dtype)
    mean0 = [0.25] * k
    out = eng.recursive_least_squares(_cuda(y), [_cuda(c) for c in cols], offs, half_life=half_life, initial_state_covariance=1e3,
                                      initial_state_mean=mean0, null_free=True)
    name = eng.last_kernel
    assert name.startswith("k3s_rls_rows")
    one_launch = halo and k <= 9
    lookback = one_launch and k <= 6 and half_life * 36.0 <= 1024.0      # the rows that matter lie inside the ONE tile in front
    assert ("lookback" in name) == lookback and ("halo" in name) == (one_launch and not lookback), (name, half_life, k)
    if lookback:                                   # ... and the same frame with every wave on the look-back form's fallback
        eng.set_option("RLS_SPINS", "0")
        try:
            slow = eng.recursive_least_squares(_cuda(y), [_cuda(c) for c in cols], offs, half_life=half_life, initial_state_covariance=1e3,
                                               initial_state_mean=mean0, null_free=True)
        finally:
            eng.set_option("RLS_SPINS", None)
        assert np.allclose(_np(slow["coef"]), _np(out["coef"]), rtol=tol, atol=tol)
    ref = orc.batched_rls(y, cols, offs, half_life=half_life, initial_state_covariance=1e3, initial_state_mean=mean0)
    assert np.allclose(_np(out["coef"]), ref["coef"], rtol=tol, atol=tol), float(np.abs(_np(out["coef"]) - ref["coef"]).max())
    assert np.allclose(_np(out["pred"]), ref["pred"], rtol=tol, atol=tol)


def test_rls_halo_prior_decay_is_exact_on_small_features(eng):
    """Returns-sized features (sigma = 1e-4) under the default prior (p0 = 10): the prior 1 / p0 outweighs the data sums (~30 sigma^2 = 3e-7) by
    five orders of magnitude for hundreds of rows and fades as ff^t -- the halo form carries that term in closed form from the sequence's
    first row, so the coefficients follow the oracle through the whole fade-out (tiles 1 .. 3 of the long sequence are the test)."""
    from oracle import orc

    rng = np.random.default_rng(8)
    n, k = 6_000, 4
    cols = [1e-4 * rng.standard_normal(n) for _ in range(k)]
    y = sum(cols) + 1e-5 * rng.standard_normal(n)
    offs = np.array([0, 37, 37 + n - 37], dtype=np.int64)
    out = eng.recursive_least_squares(_cuda(y), [_cuda(c) for c in cols], offs, half_life=10.0, null_free=True)
    assert eng.last_kernel == "k3s_rls_rows_lookback_f64"
    ref = orc.batched_rls(y, cols, offs, half_life=10.0)
    assert np.allclose(_np(out["coef"]), ref["coef"], rtol=1e-6, atol=1e-9), float(np.abs(_np(out["coef"]) - ref["coef"]).max())


def test_rls_halo_not_taken_with_validity_bytes(eng):
    """A masked row does not decay the state (least_squares.rs:589-591 updates on valid rows only), so the distance to the tile is not the row distance: masked frames keep the scan."""
    from oracle import orc

    rng = np.random.default_rng(9)
    n, k = 5_000, 3
    cols = [rng.standard_normal(n) for _ in range(k)]
    y = sum(cols) + 0.1 * rng.standard_normal(n)
    valid = (rng.random(n) > 0.2).astype(np.uint8)
    out = eng.recursive_least_squares(_cuda(y), [_cuda(c) for c in cols], [0, n], valid=_cuda(valid), half_life=5.0)
    assert eng.last_kernel == "k3s_rls_rows_f64"
    ref = orc.batched_rls(y, cols, [0, n], half_life=5.0, is_valid=valid)
    assert np.allclose(_np(out["coef"]), ref["coef"], rtol=1e-6, atol=1e-6)


def test_rls_many_sequences_full_size(eng):
    """The dynamic models the way the reference is used (README.md:119-137, `.rls(...).over("group")`), bench.py --config rlsg:
    10 000 sequences x 1 000 rows x 6 features, half_life = 21, f64 -- sampled sequences against the oracle, every row of them."""
    from oracle import orc
    import torch

    import synth

    G, n, k = 10_000, 1_000, 6
    # the frame is generated ON the device by the counter-based generator (synth.py); the host regenerates the sampled sequences' rows bit for
    # bit from (seed, row, column) -- no input column is copied back
    y, cols, _ = synth.frame_columns(77, k, 0, G * n, dtype=torch.float64, device="cuda")
    offs = np.arange(G + 1, dtype=np.int64) * n
    out = eng.recursive_least_squares(y, cols, offs, half_life=21.0, null_free=True)
    assert eng.last_kernel.startswith("k3s_rls_rows")
    coef, pred = out["coef"], out["pred"]
    assert bool(torch.isfinite(coef).all()) and bool(torch.isfinite(pred).all())
    rng = np.random.default_rng(5)
    pick = np.unique(np.concatenate([[0, 1, 2, G - 1], rng.integers(0, G, size=60)]))   # sequences 0..2 straddle the first tiles
    hy0, hc0, _ = synth.frame_columns(77, k, 0, n)
    assert np.array_equal(hy0, _np(y[:n])) and all(np.array_equal(h, _np(c[:n])) for h, c in zip(hc0, cols))   # host == device, bit for bit
    for g in pick:
        s, e = int(offs[g]), int(offs[g + 1])
        hy, hc, _ = synth.frame_columns(77, k, s, e)
        ref = orc.batched_rls(hy, hc, [0, n], half_life=21.0)
        assert np.allclose(_np(coef[s:e]), ref["coef"], rtol=1e-6, atol=1e-6), g
        assert np.allclose(_np(pred[s:e]), ref["pred"], rtol=1e-6, atol=1e-6), g


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("k", [3, 6, 7, 9, 10])
def test_rls_lookback_long_and_short_sequences_mixed(eng, dtype, tol, k):
    """K3c's look-back across tiles and groups of tiles: a frame of one 40 000-row sequence (several look-back groups), thousands
    of rows of tiny sequences (every tile closed), and sequences that start exactly on tile / run boundaries; with validity bytes."""
    from oracle import orc

    rng = np.random.default_rng(900 + k)
    sizes = np.concatenate([[40_000], rng.integers(1, 9, size=700), [1024, 2048, 4, 4, 4096, 3], rng.integers(200, 3000, size=12), [1]])
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(offs[-1])
    cols = [rng.standard_normal(N).astype(dtype) for _ in range(k)]
    y = (sum(cols).astype(np.float64) + 0.1 * rng.standard_normal(N)).astype(dtype)
    valid = (rng.random(N) > 0.07).astype(np.uint8)
    for v in (None, valid):
        out = eng.recursive_least_squares(_cuda(y), [_cuda(c) for c in cols], offs, valid=None if v is None else _cuda(v), half_life=63.0,
                                          initial_state_covariance=5.0, initial_state_mean=[0.1] * k, null_free=v is None)
        assert eng.last_kernel.startswith("k3s_rls_rows")
        ref = orc.batched_rls(y, cols, offs, half_life=63.0, initial_state_covariance=5.0, initial_state_mean=[0.1] * k, is_valid=v)
        assert np.allclose(_np(out["coef"]), ref["coef"], rtol=tol, atol=tol), float(np.abs(_np(out["coef"]) - ref["coef"]).max())
        exp_p = ref["pred"] if v is None else _masked(ref["pred"], v)
        assert np.allclose(_np(out["pred"]), exp_p, rtol=tol, atol=tol, equal_nan=True)


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("k", [2, 6, 8, 9, 10])
def test_rls_packed_tiles_ragged_sequences(eng, dtype, tol, k):
    """K3c's single-pass form: no sequence longer than a tile, so tiles are cut at sequence starts (whole sequences, starts at any
    row -- a tile's lanes begin up to three rows before its first sequence) and nothing is carried between them.  Ragged lengths up
    to the largest a tile holds, with and without validity bytes; the same frame through the two-pass form (RLS_ENGINE=scan) must
    give the same numbers."""
    from oracle import orc

    rng = np.random.default_rng(4100 + k)
    top = (1024 if k <= 6 else 512) - 3
    sizes = np.concatenate([[top, 1, top - 1, 2, 3, 5, top, top], rng.integers(1, top + 1, size=120), rng.integers(300, top + 1, size=200), [1, 1, 7]])
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(offs[-1])
    cols = [rng.standard_normal(N).astype(dtype) for _ in range(k)]
    y = (sum(cols).astype(np.float64) + 0.1 * rng.standard_normal(N)).astype(dtype)
    valid = (rng.random(N) > 0.05).astype(np.uint8)
    for v in (None, valid):
        kw = dict(valid=None if v is None else _cuda(v), half_life=40.0, initial_state_covariance=3.0, initial_state_mean=[0.05] * k,
                  null_free=v is None)
        out = eng.recursive_least_squares(_cuda(y), [_cuda(c) for c in cols], offs, **kw)
        assert eng.last_kernel.startswith("k3s_rls_rows")
        ref = orc.batched_rls(y, cols, offs, half_life=40.0, initial_state_covariance=3.0, initial_state_mean=[0.05] * k, is_valid=v)
        assert np.allclose(_np(out["coef"]), ref["coef"], rtol=tol, atol=tol), float(np.abs(_np(out["coef"]) - ref["coef"]).max())
        exp_p = ref["pred"] if v is None else _masked(ref["pred"], v)
        assert np.allclose(_np(out["pred"]), exp_p, rtol=tol, atol=tol, equal_nan=True)
        eng.set_option("RLS_ENGINE", "scan")
        try:
            two = eng.recursive_least_squares(_cuda(y), [_cuda(c) for c in cols], offs, **kw)
        finally:
            eng.set_option("RLS_ENGINE", None)
        assert np.allclose(_np(two["coef"]), _np(out["coef"]), rtol=tol, atol=tol)
        assert np.allclose(_np(two["pred"]), _np(out["pred"]), rtol=tol, atol=tol, equal_nan=True)


def test_rls_more_than_64_blocks_of_tiles(eng):
    """One 4.4M-row sequence = 67 blocks of 64 tiles: the one-wave top scan between the passes chains two 64-block windows (a 1M-row
    sequence is 16 blocks: one window); a 1M-row neighbour in the same frame.  Every row against the oracle."""
    from oracle import orc

    rng = np.random.default_rng(6464)
    k = 3
    sizes = np.array([4_400_000, 700, 1_000_003], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(offs[-1])
    cols = [rng.standard_normal(N) for _ in range(k)]
    y = sum(cols) + 0.1 * rng.standard_normal(N)
    out = eng.recursive_least_squares(_cuda(y), [_cuda(c) for c in cols], offs, half_life=2000.0, null_free=True)
    assert eng.last_kernel.startswith("k3s_rls_rows")
    ref = orc.batched_rls(y, cols, offs, half_life=2000.0)
    assert np.allclose(_np(out["coef"]), ref["coef"], rtol=1e-6, atol=1e-6), float(np.abs(_np(out["coef"]) - ref["coef"]).max())
    assert np.allclose(_np(out["pred"]), ref["pred"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("k,half_life,p0,mean,null_frac", [(11, 21.0, 10.0, None, 0.0), (12, None, 1e6, None, 0.0), (16, 63.0, 10.0, 0.5, 0.05),
                                                           (17, 21.0, 10.0, None, 0.2), (24, None, 10.0, None, 0.0), (32, 100.0, 1.0, 0.1, 0.1)])
def test_rls_wave_per_chunk_single_chunk_sequences(eng, dtype, tol, k, half_life, p0, mean, null_frac):
    """K3p (k4p_wide.hip) on the `.rls().over(group)` shape: no sequence longer than 1 024 rows, so every sequence is ONE chunk that runs
    the reference's recursion from the prior (no totals, no scan) -- both padded widths (16, 32), a diffuse prior (p0 = 1e6, the reference's
    own test setting tests/test_ols.py:633-681), validity masks, empty and one-row sequences, every row against the oracle; the same frame
    through the old wave-per-chunk kernels (POLS_RLS_ENGINE=chunk) must agree."""
    from oracle import orc

    rng = np.random.default_rng(500 + k)
    sizes = np.concatenate([[1024, 0, 1, 2, k - 1, k, k + 1, 33, 64, 1000], rng.integers(1, 700, size=30)])
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(offs[-1])
    cols = [rng.standard_normal(N).astype(dtype) for _ in range(k)]
    y = (sum(cols).astype(np.float64) + 0.1 * rng.standard_normal(N)).astype(dtype)
    valid = (rng.random(N) >= null_frac).astype(np.uint8) if null_frac > 0 else None
    mean0 = None if mean is None else [mean] * k
    kw = dict(half_life=half_life, initial_state_covariance=p0, initial_state_mean=mean0)
    out = eng.recursive_least_squares(_cuda(y), [_cuda(c) for c in cols], offs, valid=None if valid is None else _cuda(valid), **kw)
    assert eng.last_kernel.startswith("k3p_")
    ref = orc.batched_rls(y, cols, offs, is_valid=valid, **kw)
    got_c, got_p = _np(out["coef"]), _np(out["pred"])
    assert np.allclose(got_c, ref["coef"], rtol=tol, atol=tol), float(np.abs(got_c - ref["coef"]).max())
    assert np.allclose(got_p, _masked(ref["pred"], valid) if valid is not None else ref["pred"], rtol=tol, atol=tol, equal_nan=True)
    eng.set_option("RLS_ENGINE", "chunk")
    try:
        old = eng.recursive_least_squares(_cuda(y), [_cuda(c) for c in cols], offs, valid=None if valid is None else _cuda(valid), **kw)
        assert eng.last_kernel.startswith("k3sw_")
    finally:
        eng.set_option("RLS_ENGINE", None)
    assert np.allclose(_np(old["coef"]), got_c, rtol=tol, atol=tol)
    # several sequences per wave (what a frame of >= 16 384 / 8 192 chunks takes by itself): four on 16-lane rows up to 16 features, two on
    # 32 lanes beyond -- ragged lengths, so the sub-waves of a wave finish at different rows
    eng.set_option("K4P_LPS", "16" if k <= 16 else "32")
    try:
        packed = eng.recursive_least_squares(_cuda(y), [_cuda(c) for c in cols], offs, valid=None if valid is None else _cuda(valid), **kw)
        assert eng.last_kernel.startswith("k3p_") and eng.last_kernel.endswith("_x4" if k <= 16 else "_x2"), eng.last_kernel
    finally:
        eng.set_option("K4P_LPS", None)
    assert np.allclose(_np(packed["coef"]), ref["coef"], rtol=tol, atol=tol), float(np.abs(_np(packed["coef"]) - ref["coef"]).max())
    assert np.allclose(_np(packed["pred"]), _masked(ref["pred"], valid) if valid is not None else ref["pred"], rtol=tol, atol=tol, equal_nan=True)
