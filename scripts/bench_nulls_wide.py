"""Null policies at 11-31 columns: the register-resident masked kernels next to the plain kernel and the streamed three-launch path."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine  # noqa: E402


def main():
    G, n = 20_000, 500
    eng = Engine(0)
    g = torch.Generator(device="cuda").manual_seed(0)
    res = {}
    for dt in (torch.float32, torch.float64):
        for k in (12, 15, 20):
            cols = [torch.randn(G * n, device="cuda", generator=g, dtype=dt) for _ in range(k)]
            y = sum(cols) + 0.1 * torch.randn(G * n, device="cuda", generator=g, dtype=dt)
            yn = y.clone()
            yn[torch.rand(G * n, device="cuda", generator=g) < 0.05] = float("nan")
            offs = np.arange(G + 1, dtype=np.int64) * n
            for name, yy, kw in (("plain", y, {}), ("drop", yn, {"null_policy": "drop"}), ("drop_streamed", yn, {"null_policy": "drop"})):
                eng.set_option("POLS_STATIC_ENGINE", "stream" if name == "drop_streamed" else None)
                plan = eng.plan_least_squares(yy, cols, offs, want=("pred",), **kw)
                for _ in range(5):
                    plan.run()
                torch.cuda.synchronize()
                t0 = time.perf_counter()                 # wall clock over back-to-back calls: the streamed path is three launches
                for _ in range(30):
                    plan.run()
                torch.cuda.synchronize()
                us = (time.perf_counter() - t0) * 1e6 / 30
                byts = G * n * (k + 2) * (4 if dt == torch.float32 else 8)
                res[f"{'f32' if dt == torch.float32 else 'f64'}_k{k}_{name}"] = {"kernel": eng.last_kernel, "us": round(us, 1), "TBps": round(byts / us / 1e6, 2)}
            del cols, y, yn
    print(json.dumps(res))


if __name__ == "__main__":
    main()
