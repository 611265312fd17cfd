import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from polars_ols_amd import Engine
eng = Engine(0)
G, n, k = 10_000, 1_000, 8
gen = torch.Generator(device="cuda").manual_seed(1)
cols = [torch.randn(G * n, generator=gen, device="cuda") for _ in range(k)]
y = sum(cols) + 0.1 * torch.randn(G * n, generator=gen, device="cuda")
y[torch.rand(G * n, generator=gen, device="cuda") < 0.05] = float("nan")
offs = np.arange(G + 1, dtype=np.int64) * n
for pol in ("ignore", "drop", "zero", "drop_zero"):
    plan = eng.plan_least_squares(y, cols, offs, want=("pred", "coef"), null_policy=pol)
    for _ in range(5): plan.run()
    torch.cuda.synchronize(); eng.timing(True)
    for _ in range(30): plan.run()
    torch.cuda.synchronize()
    ms = eng.timing_collect(); eng.timing(False)
    print(pol, eng.last_kernel, "kernel_ms=%.4f GB/s=%.0f" % (ms.mean(), 400e6 / (ms.mean() * 1e-3) / 1e9))
