"""synth.py on the device: the frame bench.py generates in HBM is, bit for bit, the frame a host regenerates from (seed, row, column)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tdt,ndt", [("float64", np.float64), ("float32", np.float32)])
def test_device_frame_equals_host_frame(tdt, ndt):
    import torch

    import synth

    n, k = 1_000_003, 5
    y, cols, w = synth.frame_columns(21, k, 0, n, dtype=getattr(torch, tdt), device="cuda", weights=True)
    for lo, hi in ((0, 4096), (n - 5000, n), (500_000, 500_777)):
        hy, hc, hw = synth.frame_columns(21, k, lo, hi, dtype=ndt, weights=True)
        assert np.array_equal(hy, y[lo:hi].cpu().numpy()) and np.array_equal(hw, w[lo:hi].cpu().numpy())
        for a, b in zip(hc, cols):
            assert np.array_equal(a, b[lo:hi].cpu().numpy())


def test_bench_frame_is_the_synth_frame():
    import torch

    import bench
    import synth

    y, cols, _ = bench.make_columns(50_000, 3, torch.float32, 1234)
    hy, hc, _ = synth.frame_columns(1234, 3, 0, 50_000, dtype=np.float32)
    assert np.array_equal(hy, y.cpu().numpy()) and all(np.array_equal(a, b.cpu().numpy()) for a, b in zip(hc, cols))
