"""Host-side cost of one plan.run() of the dynamic entries next to the kernel time (is the stream ever waiting for the host?)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine  # noqa: E402

eng = Engine(0)
G, n, k = 10_000, 1_000, 6
gen = torch.Generator(device="cuda").manual_seed(1)
cols = [torch.randn(G * n, generator=gen, device="cuda", dtype=torch.float64) for _ in range(k)]
y = sum(cols) + 0.1 * torch.randn(G * n, generator=gen, device="cuda", dtype=torch.float64)
offs = np.arange(G + 1, dtype=np.int64) * n
out = {"pred": torch.empty(G * n, device="cuda", dtype=torch.float64), "coef": torch.empty(G * n, k, device="cuda", dtype=torch.float64)}
for name, plan in (("rls", eng.plan_recursive_least_squares(y, cols, offs, half_life=21.0, out=out, null_free=True)),
                   ("rolling", eng.plan_rolling_least_squares(y, cols, offs, window_size=252, min_periods=6, null_policy="drop", out=out, null_free=True))):
    for _ in range(5):
        plan.run()
    eng.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        plan.run()
    t1 = time.perf_counter()
    eng.synchronize()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: host {1e6 * (t1 - t0) / 50:.1f} us per call issued, {1e6 * (t2 - t0) / 50:.1f} us per call completed ({eng.last_kernel})")
