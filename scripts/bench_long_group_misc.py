"""ONE long group (5M rows x 8 features f64) through the other static entries: statistics, weights + ridge, elastic net, 20 features,
null policy -- wall clock per call (each call bounded; the script stops a shape after a slow first call)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine  # noqa: E402

eng = Engine(0)
N = int(os.environ.get("N", 5_000_000))
gen = torch.Generator(device="cuda").manual_seed(0)
cols20 = [torch.randn(N, device="cuda", generator=gen, dtype=torch.float64) for _ in range(20)]
y = sum(cols20[:8]) + 0.1 * torch.randn(N, device="cuda", generator=gen, dtype=torch.float64)
w = torch.rand(N, device="cuda", generator=gen, dtype=torch.float64) + 0.5
offs = np.array([0, N], dtype=np.int64)


def timed(name, fn):
    t0 = time.perf_counter(); fn(); eng.synchronize(); torch.cuda.synchronize()
    first = 1e3 * (time.perf_counter() - t0)
    if first > 500:
        print(f"{name:34s} {first:10.3f} ms (first call; not repeated)  {eng.last_kernel}", flush=True)
        return
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    eng.synchronize(); torch.cuda.synchronize()
    print(f"{name:34s} {1e3 * (time.perf_counter() - t0) / 3:10.3f} ms  {eng.last_kernel}", flush=True)


c8 = cols20[:8]
timed("ols 8 feats pred", lambda: eng.least_squares(y, c8, offs, want=("pred",)))
timed("ols 8 feats coef only", lambda: eng.least_squares(y, c8, offs, want=("coef",)))
timed("ridge + weights + intercept", lambda: eng.least_squares(y, c8, offs, weights=w, alpha=1.0, l1_ratio=0.0, add_intercept=True, want=("pred",)))
timed("elastic net", lambda: eng.least_squares(y, c8, offs, alpha=0.01, l1_ratio=0.5, want=("pred",)))
timed("ols 20 feats pred", lambda: eng.least_squares(y, cols20, offs, want=("pred",)))
timed("statistics 8 feats", lambda: eng.least_squares_statistics(y, c8, offs))
timed("multi-target (3) 8 feats", lambda: eng.multi_target_least_squares([y, y * 2, y + 1], c8, offs))
