"""f64, 17..24 columns, groups of 200..512 rows: K1 wide (VALU passes) against K2w (two MFMA tiles) -- where is the crossover?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine
eng = Engine(0)
dt = torch.float64
N = 4_000_000
gen = torch.Generator(device="cuda").manual_seed(1)
allc = [torch.randn(N, generator=gen, device="cuda", dtype=dt) for _ in range(24)]
first = True
for k in (17, 20, 22, 24):
    cols = allc[:k]
    y = sum(cols[:4]) + 0.1 * torch.randn(N, generator=gen, device="cuda", dtype=dt)
    for n in (128, 200, 256, 300, 384, 512):
        G = N // n
        offs = np.arange(G + 1, dtype=np.int64) * n
        res = []
        for engine in (None, "k2w"):
            eng.set_option("STATIC_ENGINE", engine)
            plan = eng.plan_least_squares(y[:G * n], [c[:G * n] for c in cols], offs, want=("pred",))
            for _ in range(30 if first else 3): plan.run()
            first = False
            eng.synchronize(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): plan.run()
            eng.synchronize(); torch.cuda.synchronize()
            res.append((1e3 * (time.perf_counter() - t0) / 5, eng.last_kernel))
        eng.set_option("STATIC_ENGINE", None)
        print(f"f64 k={k} rows={n}: default {res[0][0]:.3f} ms ({res[0][1][:40]})   k2w {res[1][0]:.3f} ms ({res[1][1][:44]})", flush=True)
