"""Over-resident groups with 9..15 columns (rows beyond what the K1 kernels keep in registers): the default route (K1m, LDS-tile MFMA
engine) against K2 (rows resident in the registers of up to eight waves), wall clock per call."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine  # noqa: E402

eng = Engine(0)
for dt, n, G in ((torch.float64, 1100, 10000), (torch.float64, 1500, 8000), (torch.float64, 2000, 6000), (torch.float32, 2200, 7000), (torch.float32, 3000, 5000), (torch.float32, 4000, 4000)):
    for k in (9, 12, 15):
        offs = np.arange(G + 1, dtype=np.int64) * n
        N = G * n
        g = torch.Generator(device="cuda").manual_seed(0)
        cols = [torch.randn(N, device="cuda", generator=g, dtype=dt) for _ in range(k)]
        y = sum(cols) + 0.1 * torch.randn(N, device="cuda", generator=g, dtype=dt)
        row = []
        for engine in (None, "k2"):
            eng.set_option("STATIC_ENGINE", engine)
            plan = eng.plan_least_squares(y, cols, offs, want=("pred", "coef"))
            for _ in range(3):
                plan.run()
            eng.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                plan.run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100.0
            row.append(f"{engine or 'default'}: {us:7.1f} us {N * (k + 2) * (4 if dt == torch.float32 else 8) / us / 1e6:4.2f} TB/s {eng.last_kernel}")
        eng.set_option("STATIC_ENGINE", None)
        print(f"{'f32' if dt == torch.float32 else 'f64'} k={k} {G}x{n} | " + " | ".join(row), flush=True)
        del cols, y, plan
