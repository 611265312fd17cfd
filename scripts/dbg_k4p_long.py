"""K4p on CUT sequences (1 000 x 10 000 rows x 12 features): where the time goes."""
import numpy as np, torch, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine
eng = Engine(0)
G, n, k = int(os.environ.get("G", 1000)), int(os.environ.get("N", 10_000)), 12
gen = torch.Generator(device="cuda").manual_seed(3)
cols = [torch.randn(G * n, generator=gen, device="cuda", dtype=torch.float64) for _ in range(k)]
y = sum(cols) + 0.1 * torch.randn(G * n, generator=gen, device="cuda", dtype=torch.float64)
offs = np.arange(G + 1, dtype=np.int64) * n
def timed(f):
    f(); eng.synchronize(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): f()
    eng.synchronize(); torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / 3
for w in (252, 1_000_000):
    ms = timed(lambda: eng.rolling_least_squares(y, cols, offs, window_size=w, min_periods=k, null_policy="drop", null_free=True))
    print(f"{G} x {n} rows x 12, rolling window {w}: {ms:8.3f} ms  {eng.last_kernel}")
ms = timed(lambda: eng.recursive_least_squares(y, cols, offs, half_life=21.0, null_free=True))
print(f"{G} x {n} rows x 12, rls: {ms:8.3f} ms  {eng.last_kernel}")
