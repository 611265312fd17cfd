"""One frame of tiny groups through the static path, N calls: the workload rocprofv3 is pointed at when the K1t kernels are profiled
(MODE=default|sub16|sub8x1, LO / HI = group length range, DT=f32|f64)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine  # noqa: E402

lo, hi = int(os.environ.get("LO", 12)), int(os.environ.get("HI", 40))
dt = torch.float64 if os.environ.get("DT") == "f64" else torch.float32
eng = Engine(0)
rng = np.random.default_rng(0)
G = 500_000
sizes = rng.integers(lo, hi + 1, size=G)
offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
n = int(offs[-1])
g = torch.Generator(device="cuda").manual_seed(0)
cols = [torch.randn(n, device="cuda", generator=g, dtype=dt) for _ in range(8)]
y = sum(cols) + 0.1 * torch.randn(n, device="cuda", generator=g, dtype=dt)
plan = eng.plan_least_squares(y, cols, offs, want=("pred", "coef"))
eng.set_option("K1T_SUB8", {"sub16": "0", "sub8x1": "1"}.get(os.environ.get("MODE", "default")))
for _ in range(int(os.environ.get("N", 6))):
    plan.run()
eng.synchronize()
print(eng.last_kernel, n)
