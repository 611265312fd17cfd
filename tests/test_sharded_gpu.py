"""pols_least_squares_sharded: ONE process, several contexts -- the form a Polars plugin process takes.  The GPU box has one device,
so: (a) three contexts on device 0 with host outputs exercise the partition, the per-device host threads, the shard batches and the
slice-wise copies home with a real three-way split; (b) a world of one with device outputs runs every collective of the
device-side assembly (pols_comm_create_all, all-gather of the coefficient table, gathers of predictions / residuals / status)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import orc  # noqa: E402


def _frame(seed, dtype, k, sizes, weights=True):
    rng = np.random.default_rng(seed)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(offs[-1])
    cols = [rng.standard_normal(n).astype(dtype) for _ in range(k)]
    y = (sum(c.astype(np.float64) for c in cols) + 0.1 * rng.standard_normal(n) + 0.2).astype(dtype)
    w = rng.uniform(0.3, 2.0, size=n).astype(dtype) if weights else None
    return y, cols, offs, w


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
@pytest.mark.parametrize("k,kw", [(8, {}), (5, {"alpha": 0.7, "l1_ratio": 0.0}), (16, {"alpha": 0.001, "l1_ratio": 0.5, "tol": 1e-10, "max_iter": 10_000}),
                                  (40, {})])
def test_three_contexts_host_outputs(dtype, tol, k, kw):
    from polars_ols_amd import Engine, least_squares_sharded
    from polars_ols_amd.engine import partition_groups_native

    rng = np.random.default_rng(k)
    sizes = rng.integers(60, 900, size=57)
    sizes[[4, 30]] = 0                                               # empty groups inside shards
    y, cols, offs, w = _frame(k, dtype, k, sizes)
    engines = [Engine(0) for _ in range(3)]
    try:
        out = least_squares_sharded(engines, y, cols, offs, weights=w, add_intercept=True, want=("coef", "pred", "resid", "status"), **kw)
        one = engines[0].least_squares(y, cols, offs, weights=w, add_intercept=True, want=("coef", "pred", "resid", "status"), **kw)
    finally:
        for e in engines:
            e.close()
    b = partition_groups_native(offs, 3)
    assert 0 < b[1] < b[2] < len(sizes)                              # a real three-way split
    assert np.array_equal(out["status"], one["status"])
    for key in ("coef", "pred", "resid"):                            # (a shard's own largest group / alignment may pick another kernel shape
        assert np.allclose(out[key], one[key], rtol=tol, atol=tol), key    #  than the whole frame's: same answer, not the same bits)
    ref = orc.batched_least_squares(y, cols, offs, weights=w, add_intercept=True, **kw)
    for key in ("coef", "pred", "resid"):
        assert np.allclose(out[key], ref[key], rtol=tol, atol=tol), key
    assert list(np.flatnonzero(out["status"] == 2)) == [4, 30]


def test_more_contexts_than_groups():
    from polars_ols_amd import Engine, least_squares_sharded

    y, cols, offs, _ = _frame(3, np.float64, 3, [50, 70], weights=False)
    engines = [Engine(0) for _ in range(4)]
    try:
        out = least_squares_sharded(engines, y, cols, offs, want=("coef", "pred"))
    finally:
        for e in engines:
            e.close()
    ref = orc.batched_least_squares(y, cols, offs)
    assert np.allclose(out["coef"], ref["coef"], rtol=1e-9, atol=1e-9) and np.allclose(out["pred"], ref["pred"], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-6), (np.float32, 1e-4)])
def test_world_of_one_device_side_assembly(dtype, tol):
    """out="device": every collective of the assembly runs (RCCL world of one: ncclCommInitAll, all-gather, grouped send / recv)."""
    from polars_ols_amd import Engine, comm_create_all, least_squares_sharded

    y, cols, offs, w = _frame(11, dtype, 6, [300, 0, 120, 999, 41])
    eng = Engine(0)
    comms = comm_create_all([eng])
    try:
        out = least_squares_sharded([eng], y, cols, offs, comms=comms, out="device", weights=w, alpha=0.3, l1_ratio=0.0,
                                    want=("coef", "pred", "resid", "status"))
        got = {k: v.cpu().numpy() for k, v in out.items()}
    finally:
        for c in comms:
            c.close()
        eng.close()
    assert out["pred"].is_cuda and out["coef"].shape == (5, 6)
    ref = orc.batched_least_squares(y, cols, offs, weights=w, alpha=0.3, l1_ratio=0.0)
    for key in ("coef", "pred", "resid"):
        assert np.allclose(got[key], ref[key], rtol=tol, atol=tol), key
    assert list(got["status"]) == [0, 2, 0, 0, 0]


def test_sharded_argument_errors():
    from polars_ols_amd import Engine, PolsError, least_squares_sharded

    y, cols, offs, _ = _frame(1, np.float64, 2, [30, 30], weights=False)
    eng = Engine(0)
    try:
        with pytest.raises(PolsError):                               # device-side assembly without communicators
            least_squares_sharded([eng], y, cols, offs, out="device")
    finally:
        eng.close()


def test_device_side_assembly_returns_a_rank_failure_instead_of_waiting():
    """A rank that fails before its collectives (here: the reference's panic for solve_method="qr" with alpha > 0, src/least_squares.rs:366)
    reports through the host-side rendezvous and the call returns its error -- no rank enters RCCL (with N > 1 the peers would wait for
    a send / receive that never comes); the contexts and communicators stay usable."""
    from polars_ols_amd import Engine, PolsError, comm_create_all, least_squares_sharded

    y, cols, offs, _ = _frame(5, np.float64, 3, [200, 150, 90], weights=False)
    eng = Engine(0)
    comms = comm_create_all([eng])
    try:
        with pytest.raises(PolsError):
            least_squares_sharded([eng], y, cols, offs, comms=comms, out="device", alpha=0.5, l1_ratio=0.0, solve_method="qr", want=("coef", "pred"))
        out = least_squares_sharded([eng], y, cols, offs, comms=comms, out="device", want=("coef", "pred"))
        got = out["coef"].cpu().numpy()
    finally:
        for c in comms:
            c.close()
        eng.close()
    ref = orc.batched_least_squares(y, cols, offs)
    assert np.allclose(got, ref["coef"], rtol=1e-9, atol=1e-9)
