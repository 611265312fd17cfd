"""Where the row-parallel dynamic kernels hand over to the lane-per-chunk ones (10 000 sequences x 1 000 rows, f64, null-free): feature
counts 6 / 7 / 8 / 12, windows 252 / 508 / 600 -- wall clock per call and the kernel that took it."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine  # noqa: E402

eng = Engine(0)
G, n = 10_000, 1_000
offs = np.arange(G + 1, dtype=np.int64) * n


def timed(fn, reps=5):
    fn(); eng.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    eng.synchronize(); torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


for k in [int(v) for v in os.environ.get('KS', '6,7,8,12').split(',')]:
    gen = torch.Generator(device="cuda").manual_seed(3)
    cols = [torch.randn(G * n, generator=gen, device="cuda", dtype=torch.float64) for _ in range(k)]
    y = sum(cols) + 0.1 * torch.randn(G * n, generator=gen, device="cuda", dtype=torch.float64)
    for w in (252, 508, 600):
        ms = timed(lambda: eng.rolling_least_squares(y, cols, offs, window_size=w, min_periods=k, null_policy="drop", null_free=True))
        print(f"rolling k={k:2d} window={w:3d} {ms:8.3f} ms per call  {eng.last_kernel}", flush=True)
    ms = timed(lambda: eng.recursive_least_squares(y, cols, offs, half_life=21.0, null_free=True))
    print(f"rls     k={k:2d}            {ms:8.3f} ms per call  {eng.last_kernel}", flush=True)
    del cols, y
