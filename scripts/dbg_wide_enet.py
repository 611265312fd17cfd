import sys, time; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from polars_ols_amd import Engine
from refdata import make_data
eng = Engine(0)
for n, k, sp, alpha, method in ((1000, 1000, 0.9, 0.3, "cd"), (1000, 1000, 0.9, 0.3, "cd_active_set"), (10_000, 100, 0.5, 0.3, "cd"), (5000, 500, 0.9, 0.1, "cd")):
    d = make_data(n_samples=n, n_features=k, sparsity=sp)
    y = torch.from_numpy(d["y"]).cuda(); cols = [torch.from_numpy(d[f"x{i+1}"]).cuda() for i in range(k)]
    offs = np.array([0, n], dtype=np.int64)
    plan = eng.plan_least_squares(y, cols, offs, alpha=alpha, l1_ratio=0.5, max_iter=1000, tol=1e-4, solve_method=method, want=("pred", "coef", "status"))
    plan.run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): plan.run()
    torch.cuda.synchronize()
    print(f"enet n={n} k={k} {method}: {(time.perf_counter() - t0) * 200:.2f} ms/call, status {int(plan.results['status'][0])}")
