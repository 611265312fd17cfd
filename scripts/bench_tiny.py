"""Frames of tiny groups (per-asset-per-month sized regressions) through K1t: four groups per wave (16-lane teams, POLS_K1T_SUB8=0), eight-lane
teams only where every group fits eight chunk slots (=1), and the default rule (eight-lane teams up to 16 chunk slots, two chunks per lane); wall clock
per call over back-to-back calls, each mode measured twice in alternation (the first timed loop of a process runs on cold clocks)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine  # noqa: E402

eng = Engine(0)
rng = np.random.default_rng(0)
res = {}
for name, lo, hi, dt, k in (("tiny_12_40", 12, 40, torch.float32, 8), ("tiny_14_28", 14, 28, torch.float32, 8), ("tiny_20_64", 20, 64, torch.float32, 8),
                            ("f64_tiny_12_40", 12, 40, torch.float64, 8), ("f64_tiny_10_16", 10, 16, torch.float64, 8),
                            ("k3_tiny_12_40", 12, 40, torch.float32, 3), ("k4_tiny_12_40", 12, 40, torch.float32, 4), ("k5_tiny_12_40", 12, 40, torch.float32, 5), ("k6_tiny_12_40", 12, 40, torch.float32, 6),
                            ("f64_k3_tiny_10_28", 10, 28, torch.float64, 3)):
    G = 500_000
    sizes = rng.integers(lo, hi + 1, size=G)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(offs[-1])
    g = torch.Generator(device="cuda").manual_seed(0)
    cols = [torch.randn(n, device="cuda", generator=g, dtype=dt) for _ in range(k)]
    y = sum(cols) + 0.1 * torch.randn(n, device="cuda", generator=g, dtype=dt)
    plan = eng.plan_least_squares(y, cols, offs, want=("pred", "coef"))
    row = {}
    outs = {}
    for mode, opt in (("sub16", "0"), ("sub8x1", "1"), ("default", None), ("sub16", "0"), ("sub8x1", "1"), ("default", None)):
        eng.set_option("K1T_SUB8", opt)
        for _ in range(5):
            plan.run()
        eng.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            plan.run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        if mode in row and row[mode]["us"] <= us:
            continue
        row[mode] = {"kernel": eng.last_kernel, "us": round(us, 1), "TBps": round(n * (k + 2) * (4 if dt == torch.float32 else 8) / us / 1e6, 2)}
        outs[mode] = (plan.results["pred"].clone(), plan.results["coef"].clone())
    eng.set_option("K1T_SUB8", None)
    row["max_abs_diff_pred"] = float((outs["default"][0] - outs["sub16"][0]).abs().max())
    res[name] = row
    del cols, y, plan
print(json.dumps(res))
