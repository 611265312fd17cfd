import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from polars_ols_amd import Engine
eng = Engine(0)
n, k = 10_000, 100
gen = torch.Generator(device="cuda").manual_seed(1)
cols = [torch.randn(n, generator=gen, device="cuda", dtype=torch.float64) for _ in range(k)]
y = sum(cols) + 0.1 * torch.randn(n, generator=gen, device="cuda", dtype=torch.float64)
offs = np.array([0, n], dtype=np.int64)
for name, plan in (("rls k=100", eng.plan_recursive_least_squares(y, cols, offs)),
                   ("rolling k=100 w=1000", eng.plan_rolling_least_squares(y, cols, offs, window_size=1000, min_periods=100, null_policy="drop"))):
    for _ in range(3): plan.run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): plan.run()
    torch.cuda.synchronize()
    print(name, eng.last_kernel, "ms/call=%.3f" % ((time.perf_counter() - t0) * 100))
