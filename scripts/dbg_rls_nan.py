import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from polars_ols_amd.engine import Engine
eng = Engine(0)
G, n, k = 10_000, 1_000, 6
gen = torch.Generator(device="cuda").manual_seed(3)
cols = [torch.randn(G * n, generator=gen, device="cuda", dtype=torch.float64) for _ in range(k)]
y = sum(cols) + 0.1 * torch.randn(G * n, generator=gen, device="cuda", dtype=torch.float64)
offs = np.arange(G + 1, dtype=np.int64) * n
valid = (torch.rand(G * n, generator=gen, device="cuda") > 0.03)
y_nan = torch.where(valid, y, torch.full_like(y, float("nan")))
for _ in range(4):
    eng.recursive_least_squares(y_nan, cols, offs, half_life=21.0)
eng.synchronize(); torch.cuda.synchronize()
