"""K3p debug: error against the oracle by row for one sequence (k = 11 / 17, half_life 21 / 10 / 63)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from polars_ols_amd.engine import Engine
from oracle import orc
eng = Engine(0)
for k in (11, 17):
    for hl in (63.0, 21.0, 10.0):
        n = 1000
        rng = np.random.default_rng(511)
        cols = [rng.standard_normal(n) for _ in range(k)]
        y = sum(cols) + 0.1 * rng.standard_normal(n)
        offs = np.array([0, n], dtype=np.int64)
        ref = orc.batched_rls(y, cols, offs, half_life=hl, initial_state_covariance=10.0)
        out = eng.recursive_least_squares(torch.from_numpy(y).cuda(), [torch.from_numpy(c).cuda() for c in cols], offs, half_life=hl, initial_state_covariance=10.0)
        err = np.abs(out["coef"].cpu().numpy() - ref["coef"]).max(axis=1)
        print(k, hl, eng.last_kernel, " ".join(f"{err[i]:.1e}" for i in (0, 10, 31, 32, 50, 100, 200, 300, 400, 500, 600, 700, 800, 900, 999)), flush=True)
