"""Static OLS (predictions) over group LENGTH x column count x dtype at ~10M rows per frame: TB/s of algorithmic bytes and the kernel that took the
frame -- where does the dispatcher hand a shape to something slow?  (aligned, equal groups; weights off)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine
eng = Engine(0)
ROWS = [int(v) for v in os.environ.get("ROWS", "64,200,500,1000,1500,2500,4000,6000,10000").split(",")]
COLS = [int(v) for v in os.environ.get("COLS", "2,4,6,8,9,10,12,15,16,20,24,31").split(",")]
for dt, nm, b in ((torch.float32, "f32", 4), (torch.float64, "f64", 8)):
    N = 8_000_000 if dt == torch.float32 else 4_000_000
    gen = torch.Generator(device="cuda").manual_seed(1)
    allc = [torch.randn(N, generator=gen, device="cuda", dtype=dt) for _ in range(max(COLS))]
    for k in COLS:
        cols = allc[:k]
        y = sum(cols[:4]) + 0.1 * torch.randn(N, generator=gen, device="cuda", dtype=dt)
        line = []
        for n in ROWS:
            G = N // n
            offs = np.arange(G + 1, dtype=np.int64) * n
            M = G * n
            try:
                plan = eng.plan_least_squares(y[:M], [c[:M] for c in cols], offs, want=("pred",))
                for _ in range(2): plan.run()
                eng.synchronize(); torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(5): plan.run()
                eng.synchronize(); torch.cuda.synchronize()
                ms = 1e3 * (time.perf_counter() - t0) / 5
                tb = M * (k + 2) * b / ms / 1e9
                fam = eng.last_kernel.split("_")[0] + ("w" if "resident2" in eng.last_kernel else "")
                line.append(f"{tb:4.1f}{fam:>4s}")
            except Exception as exc:
                line.append(" ERR    ")
        print(f"{nm} k={k:2d} | " + " | ".join(f"{n:>5d}: {c}" for n, c in zip(ROWS, line)), flush=True)
