"""The `predict` plugin's null policies (src/expressions.rs:725, :732-738) through pols_predict_policy: the zero fill of "zero" and the
masking of "drop" are the kernel's, not front-end passes over the columns."""
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


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("k", [3, 40])
@pytest.mark.parametrize("device", [False, True])
def test_predict_null_policies(eng, dtype, k, device):
    """predict(null_policy=...): "zero" counts a null feature as 0 (ex.rs:725), "drop" and "ignore" leave the rows with a null anywhere
    null (:732-738; un-filled product); narrow (predict kernel) and wide (column-pointer table) forms, host and device buffers."""
    rng = np.random.default_rng(k)
    n = 5_003
    cols = [rng.standard_normal(n).astype(dtype) for _ in range(k)]
    coef = rng.standard_normal((n, k + 1)).astype(dtype)
    holes = rng.random((n, k)) < 0.02
    for j, c in enumerate(cols):
        c[holes[:, j]] = np.nan
    x = np.stack(cols, axis=1).astype(np.float64)
    zero = (np.nan_to_num(x, nan=0.0) * coef[:, :k]).sum(axis=1) + coef[:, k]
    anynull = holes.any(axis=1)
    tol = 1e-4 if dtype == np.float32 else 1e-9
    xs = [_cuda(c) for c in cols] if device else cols
    cf = _cuda(coef) if device else coef
    got = _np(eng.predict(xs, cf, add_intercept=True, null_policy="zero"))
    assert np.allclose(got, zero, rtol=tol, atol=tol * np.sqrt(k))
    for pol in ("drop", "ignore"):
        got = _np(eng.predict(xs, cf, add_intercept=True, null_policy=pol))
        assert np.array_equal(np.isnan(got), anynull)
        assert np.allclose(got[~anynull], zero[~anynull], rtol=tol, atol=tol * np.sqrt(k))
    with pytest.raises(Exception):
        eng.predict(xs, cf, add_intercept=True, null_policy="drop_window")
