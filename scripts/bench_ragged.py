"""K1 on ragged frames (group sizes are not multiples of the vector width and differ): what real `.over(key)` frames look like."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine  # noqa: E402


def main():
    eng = Engine(0)
    rng = np.random.default_rng(0)
    res = {}
    for name, lo, hi, dt in (("equal_1000", 1000, 1000, torch.float32), ("ragged_900_1020", 900, 1020, torch.float32),
                             ("ragged_950_1100", 950, 1100, torch.float32), ("ragged_100_300", 100, 300, torch.float32),
                             ("f64_equal_1000", 1000, 1000, torch.float64), ("f64_ragged_900_1020", 900, 1020, torch.float64),
                             ("f64_ragged_400_500", 400, 500, torch.float64), ("tiny_12_40", 12, 40, torch.float32),
                             ("f64_tiny_12_40", 12, 40, torch.float64), ("small_40_120", 40, 120, torch.float32),
                             ("year_130_252", 130, 252, torch.float32), ("f64_small_40_120", 40, 120, torch.float64)):
        G = 10_000 if hi > 400 else (50_000 if hi > 260 else 500_000)
        sizes = rng.integers(lo, hi + 1, size=G)
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        n = int(offs[-1])
        g = torch.Generator(device="cuda").manual_seed(0)
        cols = [torch.randn(n, device="cuda", generator=g, dtype=dt) for _ in range(8)]
        y = sum(cols) + 0.1 * torch.randn(n, device="cuda", generator=g, dtype=dt)
        plan = eng.plan_least_squares(y, cols, offs, want=("pred",))
        for _ in range(10):
            plan.run()
        eng.timing(1)
        for _ in range(40):
            plan.run()
        ms = eng.timing_collect()
        eng.timing(False)
        us = float(np.mean(ms) * 1e3)
        res[name] = {"kernel": eng.last_kernel, "us": round(us, 1), "TBps": round(n * (40 if dt == torch.float32 else 80) / us / 1e6, 2),
                     "Mgroups_per_s": round(G / us, 1)}
        del cols, y, plan
    print(json.dumps(res))


if __name__ == "__main__":
    main()
