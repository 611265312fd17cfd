import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine
eng = Engine(0)
N, k = 10_000_000, 8
dt = torch.float32
gen = torch.Generator(device="cuda").manual_seed(0)
cols = [torch.randn(N, device="cuda", generator=gen, dtype=dt) for _ in range(k)]
y = sum(cols) + 0.1 * torch.randn(N, device="cuda", generator=gen, dtype=dt)
offs = np.array([0, N], dtype=np.int64)
plan = eng.plan_least_squares(y[:N], [c[:N] for c in cols], offs, want=("pred",))
mode = sys.argv[1] if len(sys.argv) > 1 else "a"
if mode == "a":          # what bench_shape_cliffs does: warm-up runs back to back, one sync, then timed runs
    for _ in range(30): plan.run()
    eng.synchronize(); torch.cuda.synchronize()
for rep in range(3):
    ts = []
    for i in range(5):
        t0 = time.perf_counter(); plan.run(); eng.synchronize(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    t0 = time.perf_counter()
    for _ in range(5): plan.run()
    eng.synchronize(); torch.cuda.synchronize()
    print(mode, "rep", rep, "single calls:", " ".join(f"{t:.3f}" for t in ts), "| 5 back to back per call:", f"{1e3 * (time.perf_counter() - t0) / 5:.3f}", flush=True)
