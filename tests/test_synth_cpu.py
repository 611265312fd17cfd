"""synth.py: the counter-based synthetic-frame generator of SURVEY.md section 8d.  Philox-4x32-10 against Random123's known-answer
vectors (kat_vectors: zero / all-ones / pi counters and keys), the numpy and torch code paths bit-identical, slices independent of
where they start, and the frame recipe (beta = 1, noise 0.1)."""
import numpy as np
import pytest
import torch

import synth

KAT = [((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
       ((0xFFFFFFFF,) * 4, (0xFFFFFFFF, 0xFFFFFFFF), (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
       ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0), (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1))]


@pytest.mark.parametrize("ctr,key,want", KAT)
def test_philox4x32_10_known_answers(ctr, key, want):
    for mk in (lambda v: np.array([v], dtype=np.int64), lambda v: torch.tensor([v], dtype=torch.int64)):
        out = synth.philox4x32(*[mk(c) for c in ctr], key[0], key[1])
        assert tuple(int(o[0]) for o in out) == want


def test_host_and_torch_paths_are_bit_identical_and_slices_compose():
    lo, hi = 2 ** 33 - 1000, 2 ** 33 + 5000                      # rows beyond 2^32: the second counter word is live
    a = synth.normal(11, 5, lo, hi)
    t = synth.normal(11, 5, lo, hi, dtype=torch.float64, device="cpu").numpy()
    assert np.array_equal(a, t)
    assert np.array_equal(a[1234:2345], synth.normal(11, 5, lo + 1234, lo + 2345))    # a slice is a function of (seed, column, row) alone
    assert not np.array_equal(a, synth.normal(12, 5, lo, hi)) and not np.array_equal(a, synth.normal(11, 6, lo, hi))
    u = synth.uniform(11, synth.COL_WEIGHT, 0, 100_000)
    assert 0.0 < u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 5e-3
    f32 = synth.normal(11, 5, 0, 1000, dtype=np.float32)
    assert f32.dtype == np.float32 and np.array_equal(f32, synth.normal(11, 5, 0, 1000).astype(np.float32))


def test_frame_recipe_moments_and_regression():
    y, cols, w = synth.frame_columns(3, 4, 0, 400_000, weights=True)
    for c in cols:
        assert abs(c.mean()) < 6e-3 and abs(c.std() - 1.0) < 5e-3 and np.abs(c).max() < 5.0
    X = np.stack(cols, axis=1)
    assert np.abs(np.corrcoef(X.T) - np.eye(4)).max() < 6e-3
    beta = np.linalg.lstsq(X, y, rcond=None)[0]
    assert np.allclose(beta, 1.0, atol=2e-3)
    assert abs((y - X @ beta).std() - 0.1) < 1e-3
    yt, colst, wt = synth.frame_columns(3, 4, 1000, 3000, dtype=torch.float32, device="cpu", weights=True)
    assert np.array_equal(yt.numpy(), y[1000:3000].astype(np.float32)) and np.array_equal(wt.numpy(), w[1000:3000].astype(np.float32))
