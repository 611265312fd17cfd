"""ONE regression over a 10M-row frame (and 100 x 100k-row groups) under the null policies / with weights: what do they cost on the streamed path?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine
eng = Engine(0)
N, k = 10_000_000, 8
for dt, nm, b in ((torch.float32, "f32", 4), (torch.float64, "f64", 8)):
    gen = torch.Generator(device="cuda").manual_seed(0)
    cols = [torch.randn(N, device="cuda", generator=gen, dtype=dt) for _ in range(k)]
    y = sum(cols) + 0.1 * torch.randn(N, device="cuda", generator=gen, dtype=dt)
    yn = y.clone(); yn[torch.rand(N, device="cuda", generator=gen) < 0.02] = float("nan")
    w = torch.rand(N, device="cuda", generator=gen, dtype=dt) + 0.5
    first = True
    for shape, sizes in (("1 x 10M", [N]), ("100 x 100k", [100_000] * 100)):
        offs = np.concatenate([[0], np.cumsum(np.asarray(sizes, dtype=np.int64))])
        for what, kw, yy in (("plain", {}, y), ("weights", dict(weights=w), y), ("zero", dict(null_policy="zero"), yn), ("drop", dict(null_policy="drop"), yn),
                             ("drop + weights", dict(null_policy="drop", weights=w), yn), ("drop, no nulls", dict(null_policy="drop"), y)):
            plan = eng.plan_least_squares(yy, cols, offs, want=("pred",), **kw)
            for _ in range(30 if first else 3): plan.run()
            first = False
            eng.synchronize(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): plan.run()
            eng.synchronize(); torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / 5
            print(f"{nm} {shape:11s} {what:15s} {ms:7.3f} ms {N * (k + 2) * b / ms / 1e9:5.2f} TB/s  {eng.last_kernel}", flush=True)
