import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from polars_ols_amd import Engine
eng = Engine(0)
rng = np.random.default_rng(0)
for dtype in (np.float32, np.float64):
    for n in (8, 64, 300):
        x = [rng.normal(size=n).astype(dtype) for _ in range(2)]
        y = (x[0] + x[1]).astype(dtype)
        y[1] = np.nan; x[0][3] = np.nan; y[n - 1] = np.nan
        offs = np.array([0, n], dtype=np.int64)
        for pol in ("drop", "drop_zero"):
            out = eng.least_squares(y, x, offs, want=("coef", "pred", "status"), null_policy=pol)
            print(dtype.__name__, n, pol, eng.last_kernel, "coef", out["coef"][0], "pred[:5]", out["pred"][:5], "last", out["pred"][-1], "st", out["status"])
