"""9..15 columns (8 features + intercept = the smoke() shape, and up): K1 VALU multi-pass vs K1m vs K2, f32 and f64."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine  # noqa: E402

eng = Engine(0)
res = {}
for dt, dname in ((torch.float32, "f32"), (torch.float64, "f64")):
    for k, icpt in ((8, True), (10, False), (11, False), (12, False), (14, False)):
        for G, n in ((10_000, 1000), (50_000, 200)):
            offs = np.arange(0, (G + 1) * n, n, dtype=np.int64)
            N = G * n
            g = torch.Generator(device="cuda").manual_seed(0)
            cols = [torch.randn(N, device="cuda", generator=g, dtype=dt) for _ in range(k)]
            y = sum(cols) + 0.1 * torch.randn(N, device="cuda", generator=g, dtype=dt)
            plan = eng.plan_least_squares(y, cols, offs, add_intercept=icpt, want=("pred",))
            row = {}
            for engine in (None, "valu", "mfma"):
                eng.set_option("K1_ENGINE", engine)
                try:
                    for _ in range(5):
                        plan.run()
                    eng.timing(1)
                    for _ in range(20):
                        plan.run()
                    ms = eng.timing_collect()
                    eng.timing(False)
                    us = float(np.mean(ms) * 1e3)
                    row[str(engine)] = (round(us, 1), round(N * (k + 2) * (4 if dt == torch.float32 else 8) / us / 1e6, 2), eng.last_kernel)
                except Exception as ex:  # noqa: BLE001
                    row[str(engine)] = str(ex)[:60]
            eng.set_option("K1_ENGINE", None)
            res[f"{dname}_k{k}{'+1' if icpt else ''}_{G}x{n}"] = row
            print(f"{dname} k={k}{'+1' if icpt else ''} {G}x{n}: " + " | ".join(f"{e}: {v}" for e, v in row.items()), flush=True)
            del cols, y, plan
