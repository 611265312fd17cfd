import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from polars_ols_amd import Engine
eng = Engine(0)
for G in (64, 10_000):
    n, k = 1000, 8
    cols = [torch.randn(G * n, device="cuda") for _ in range(k)]
    y = sum(cols)
    offs = np.arange(G + 1, dtype=np.int64) * n
    plan = eng.plan_least_squares(y, cols, offs, want=("pred", "coef"))
    for timing in (False, True):
        eng.timing(timing)
        for _ in range(50): plan.run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        N = 2000
        for _ in range(N): plan.run()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize(); t_all = time.perf_counter() - t0
        if timing: eng.timing_collect()
        eng.timing(False)
        print(f"G={G} timing={timing}: host-side issue {1e6 * t_host / N:.1f} us/call, end-to-end {1e6 * t_all / N:.1f} us/call")
