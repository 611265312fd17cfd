"""Sweep of the static path: every column count 1..15, f32 / f64, plain / weights / null policy, 10 000 x 1 000 rows -- to spot a
kernel variant that fell off (register allocation tipping into AGPRs halves occupancy without any other symptom)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine  # noqa: E402

eng = Engine(0)
n = int(os.environ.get("ROWS", "1000")); G = 10_000_000 // n
offs = np.arange(0, (G + 1) * n, n, dtype=np.int64)
N = G * n
for dt, dname, b in ((torch.float32, "f32", 4), (torch.float64, "f64", 8)):
    g = torch.Generator(device="cuda").manual_seed(0)
    allc = [torch.randn(N, device="cuda", generator=g, dtype=dt) for _ in range(15)]
    w = torch.rand(N, device="cuda", generator=g, dtype=dt) + 0.5
    y = sum(allc[:4]) + 0.1 * torch.randn(N, device="cuda", generator=g, dtype=dt)
    for k in range(1, 16):
        row = []
        for name, kw in (("plain", {}), ("w", {"weights": w}), ("drop", {"null_policy": "drop"}), ("w+drop", {"weights": w, "null_policy": "drop"})):
            plan = eng.plan_least_squares(y, allc[:k], offs, want=("pred",), **kw)
            for _ in range(3):
                plan.run()
            # back-to-back launches, wall clock per call (an isolated launch bracketed by events can look faster than the same
            # kernel does in a stream of launches: 70 against 79 us for the 256-thread team at 8 features)
            eng.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(30):
                plan.run()
            ev1.record()
            torch.cuda.synchronize()
            us = ev0.elapsed_time(ev1) * 1e3 / 30
            bytes_ = N * (k + 2 + (1 if "weights" in kw else 0)) * b
            row.append(f"{name} {us:7.1f}us {bytes_ / us / 1e6:4.2f}TB/s {eng.last_kernel.split('_k')[-1][:28]}")
        print(f"{dname} k={k:2d} | " + " | ".join(row), flush=True)
