"""f64, 8 features + intercept (+ weights) and 9 + intercept, groups of 500 / 250 rows: the two-wave team's two-chunk kernels, where round 6 picks three (four)
Gram passes to stay under 168 registers.  POLS_K1_PASSES=2 gives the two-pass builds (170-196 VGPRs, two waves per SIMD) for the A/B."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine
eng = Engine(0)
N = 4_000_000
gen = torch.Generator(device="cuda").manual_seed(1)
allc = [torch.randn(N, generator=gen, device="cuda", dtype=torch.float64) for _ in range(9)]
w = torch.rand(N, generator=gen, device="cuda", dtype=torch.float64) + 0.5
for k, wt in ((8, True), (8, False), (9, True), (9, False)):
    cols = allc[:k]
    y = sum(cols[:4]) + 0.1 * torch.randn(N, generator=gen, device="cuda", dtype=torch.float64)
    for n in (500, 250):
        G = N // n
        offs = np.arange(G + 1, dtype=np.int64) * n
        plan = eng.plan_least_squares(y, cols, offs, weights=w if wt else None, add_intercept=True, want=("pred",))
        for _ in range(3): plan.run()
        eng.synchronize(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): plan.run()
        eng.synchronize(); torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 10
        print(f"f64 {k} features + intercept{' + weights' if wt else ''} rows={n}: {ms:.3f} ms {N * (k + 2 + wt) * 8 / ms / 1e9:.2f} TB/s {eng.last_kernel}", flush=True)
