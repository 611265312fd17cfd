"""K2 on BASELINE configs[4]'s shape: phase timeline (POLS_TIMELINE stamps) as a function of max_iter -- what one sweep of the
in-workgroup coordinate descent costs."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from polars_ols_amd import Engine

eng = Engine(0)
G, n, k = 20_000, 2_000, 16
g = torch.Generator(device="cuda").manual_seed(1)
cols = [torch.randn(G * n, generator=g, device="cuda", dtype=torch.float64) for _ in range(k)]
y = sum(cols) + 0.1 * torch.randn(G * n, generator=g, device="cuda", dtype=torch.float64)
offs = np.arange(G + 1, dtype=np.int64) * n
for mi in (1, 2, 3, 6, 1000):
    plan = eng.plan_least_squares(y, cols, offs, alpha=0.001, l1_ratio=0.5, max_iter=mi, want=("coef", "pred", "status"))
    for _ in range(2):
        plan.run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        out = plan.run()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    print("max_iter", mi, "ms %.3f" % ms, "not converged:", int((out["status"] == 3).sum()), flush=True)
    eng.set_option("TIMELINE", "1")
    plan.run(); torch.cuda.synchronize()
    eng.set_option("TIMELINE", None)
