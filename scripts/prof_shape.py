"""One synthetic frame through the static path, N calls: the workload rocprofv3 is pointed at when a shape's kernel is profiled.
env: G groups, LO / HI rows per group (equal when HI == LO), K columns, DT=f32|f64, W=1 weights, POLICY=drop (5 % null targets),
ICPT=1, N calls, plus any POLS_* option."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine  # noqa: E402

E = os.environ
G, lo, hi, k = int(E.get("G", 10_000)), int(E.get("LO", 1000)), int(E.get("HI", E.get("LO", 1000))), int(E.get("K", 8))
dt = torch.float64 if E.get("DT") == "f64" else torch.float32
eng = Engine(0)
rng = np.random.default_rng(0)
sizes = rng.integers(lo, hi + 1, size=G) if hi > lo else np.full(G, lo)
offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
n = int(offs[-1])
g = torch.Generator(device="cuda").manual_seed(0)
cols = [torch.randn(n, device="cuda", generator=g, dtype=dt) for _ in range(k)]
y = sum(cols) + 0.1 * torch.randn(n, device="cuda", generator=g, dtype=dt)
kw = {}
if E.get("W"):
    kw["weights"] = torch.rand(n, device="cuda", generator=g, dtype=dt) + 0.5
if E.get("POLICY"):
    y[torch.rand(n, device="cuda", generator=g) < 0.05] = float("nan")
    kw["null_policy"] = E["POLICY"]
plan = eng.plan_least_squares(y, cols, offs, want=("pred", "coef"), add_intercept=bool(E.get("ICPT")), **kw)
for _ in range(int(E.get("N", 6))):
    plan.run()
eng.synchronize()
print(eng.last_kernel, n)
