"""16..31 columns OLS / ridge: which engine takes them and at what rate (10 000 groups x 1 000 rows, 50 000 x 200)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine  # noqa: E402

eng = Engine(0)
if os.environ.get("ENGINE"):                                  # ENGINE=k2w | stream: force an engine (A/B)
    eng.set_option("STATIC_ENGINE", os.environ["ENGINE"])
KS = tuple(int(v) for v in os.environ.get("KS", "15,16,20,24,31").split(","))
for dt, dname in ((torch.float32, "f32"), (torch.float64, "f64")):
    if os.environ.get("ONLY_F64") and dt != torch.float64:
        continue
    for k in KS:
        for G, n in ((10_000, 1000), (50_000, 200)):
            if dt == torch.float64 and k > 20 and n == 1000:
                G = 5_000
            offs = np.arange(0, (G + 1) * n, n, dtype=np.int64)
            N = G * n
            g = torch.Generator(device="cuda").manual_seed(0)
            cols = [torch.randn(N, device="cuda", generator=g, dtype=dt) for _ in range(k)]
            y = sum(cols) + 0.1 * torch.randn(N, device="cuda", generator=g, dtype=dt)
            plan = eng.plan_least_squares(y, cols, offs, want=("pred",))
            for _ in range(3):
                plan.run()
            eng.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(10):
                plan.run()
            ev1.record()
            torch.cuda.synchronize()
            us = ev0.elapsed_time(ev1) * 100.0
            print(f"{dname} k={k} {G}x{n}: {us:8.1f} us/call {N * (k + 2) * (4 if dt == torch.float32 else 8) / us / 1e6:5.2f} TB/s {eng.last_kernel}", flush=True)
            del cols, y, plan
