"""Same size distribution, group boundaries aligned to 4 rows or not: how much of the ragged-frame gap is the unaligned code path."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine  # noqa: E402

eng = Engine(0)
rng = np.random.default_rng(0)
for lo, hi, G in ((100, 300, 50_000), (130, 252, 500_000), (900, 1020, 10_000)):
    base = rng.integers(lo, hi + 1, size=G)
    for label, sizes in (("unaligned", base), ("aligned x4", (base // 4) * 4)):
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        N = int(offs[-1])
        g = torch.Generator(device="cuda").manual_seed(0)
        cols = [torch.randn(N, device="cuda", generator=g) for _ in range(8)]
        y = sum(cols) + 0.1 * torch.randn(N, device="cuda", generator=g)
        plan = eng.plan_least_squares(y, cols, offs, want=("pred",))
        for _ in range(5):
            plan.run()
        eng.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            plan.run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 30
        print(f"{lo}..{hi} x {G} {label:11s} {us:8.1f} us {N * 40 / us / 1e6:5.2f} TB/s {eng.last_kernel}", flush=True)
        del cols, y, plan
