"""Per-call wall clock of the first calls of a fresh process (one 10M-row group, 8 features f32): is there a slow call after the warm-up?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine
eng = Engine(0)
N, k = 10_000_000, 8
gen = torch.Generator(device="cuda").manual_seed(0)
cols = [torch.randn(N, device="cuda", generator=gen, dtype=torch.float32) for _ in range(k)]
y = sum(cols) + 0.1 * torch.randn(N, device="cuda", generator=gen, dtype=torch.float32)
offs = np.array([0, N], dtype=np.int64)
torch.cuda.synchronize()
plan = eng.plan_least_squares(y, cols, offs, want=("pred",))
ts = []
for i in range(12):
    t0 = time.perf_counter(); plan.run(); eng.synchronize(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print("per call ms:", " ".join(f"{t:.3f}" for t in ts))
t0 = time.perf_counter()
for _ in range(5): plan.run()
eng.synchronize(); torch.cuda.synchronize()
print("5 back to back, per call ms:", 1e3 * (time.perf_counter() - t0) / 5)
