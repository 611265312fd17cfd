"""k = 12 features, 10 000 sequences x 1 000 rows: one RLS and one rolling call under rocprofv3 (which kernels take the 40 ms)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine
eng = Engine(0)
G, n, k = 10_000, 1_000, int(os.environ.get("K", 12))
gen = torch.Generator(device="cuda").manual_seed(3)
cols = [torch.randn(G * n, generator=gen, device="cuda", dtype=torch.float64) for _ in range(k)]
y = sum(cols) + 0.1 * torch.randn(G * n, generator=gen, device="cuda", dtype=torch.float64)
offs = np.arange(G + 1, dtype=np.int64) * n
for _ in range(3):
    eng.recursive_least_squares(y, cols, offs, half_life=21.0, null_free=True)
    eng.rolling_least_squares(y, cols, offs, window_size=252, min_periods=k, null_policy="drop", null_free=True)
eng.synchronize(); torch.cuda.synchronize()
