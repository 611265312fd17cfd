import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from polars_ols_amd import Engine
eng = Engine(0)
n, k = 1_000_000, 6
gen = torch.Generator(device="cuda").manual_seed(1)
cols = [torch.randn(n, generator=gen, device="cuda", dtype=torch.float64) for _ in range(k)]
y = sum(cols) + 0.1 * torch.randn(n, generator=gen, device="cuda", dtype=torch.float64)
offs = np.array([0, n], dtype=np.int64)
for want in (("coef", "pred"), ("pred",), ("coef",)):
    plan = eng.plan_recursive_least_squares(y, cols, offs, half_life=21.0, want=want)
    for _ in range(5): plan.run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): plan.run()
    torch.cuda.synchronize()
    print(want, "ms/call=%.4f" % ((time.perf_counter() - t0) * 20))
