"""Interleaved A/B of f64 kernel shapes on BASELINE configs[2] (ridge + weights) and its unweighted twin, one process."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine  # noqa: E402

eng = Engine(0)
G, n, k = 10_000, 1000, int(os.environ.get("K", "8"))
offs = np.arange(0, (G + 1) * n, n, dtype=np.int64)
g = torch.Generator(device="cuda").manual_seed(0)
cols = [torch.randn(G * n, device="cuda", generator=g, dtype=torch.float64) for _ in range(k)]
y = sum(cols) + 0.1 * torch.randn(G * n, device="cuda", generator=g, dtype=torch.float64)
w = torch.rand(G * n, device="cuda", generator=g, dtype=torch.float64) + 0.5
variants = {"team128_rc4_p2": {}, "team256_rc2_p2": {"K1_F64_TEAM": "256", "K1_PASSES": "2"}, "team256_rc2_p3": {"K1_F64_TEAM": "256", "K1_PASSES": "3"},
            "team128_rc4_p3": {"K1_PASSES": "3"}}
for label, kw in (("ridge+weights", dict(weights=w, alpha=1.0, l1_ratio=0.0)), ("plain", {})):
    plan = eng.plan_least_squares(y, cols, offs, want=("pred",), **kw)
    res = {v: [] for v in variants}
    names = {}
    for rnd in range(8):
        for v, opts in variants.items():
            for key in ("K1_F64_TEAM", "K1_PASSES"):
                eng.set_option(key, opts.get(key))
            for _ in range(5):
                plan.run()
            eng.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                plan.run()
            e1.record()
            torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) * 1e3 / 100)
            names[v] = eng.last_kernel
    for v in variants:
        a = np.array(res[v][2:])
        print(f"{label:14s} {v:16s} median {np.median(a):7.2f} us  min {a.min():7.2f}  {names[v]}")
