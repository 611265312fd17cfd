import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from polars_ols_amd import Engine
from oracle import orc
eng = Engine(0)
rng = np.random.default_rng(0)
G = 2_000_000
sizes = rng.integers(0, 12, size=G)
offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
n = int(offs[-1]); k = 2
cols = [rng.normal(size=n) for _ in range(k)]
y = cols[0] - 2 * cols[1] + 0.1 * rng.normal(size=n)
for kw in ({}, {"alpha": 0.3}, {"alpha": 0.05, "l1_ratio": 0.5}):
    out = eng.least_squares(torch.from_numpy(y).cuda(), [torch.from_numpy(c).cuda() for c in cols], offs, want=("coef", "pred", "status"), **kw)
    torch.cuda.synchronize()
    pick = rng.choice(G, size=2000, replace=False)
    bad = 0
    for g in pick:
        s, e = offs[g], offs[g + 1]
        if e - s < 4: continue
        X = np.column_stack([c[s:e] for c in cols])
        ref = orc.get_coefficients(y[s:e], X, **kw)
        if not np.allclose(out["coef"][g].cpu().numpy(), ref, rtol=1e-6, atol=1e-6): bad += 1
    st = out["status"].cpu().numpy()
    print(kw, eng.last_kernel, "rows", n, "status counts", np.bincount(st, minlength=4).tolist(), "bad", bad)
# dynamic with many short sequences
m = int(offs[200_000])
out = eng.recursive_least_squares(torch.from_numpy(y[:m]).cuda(), [torch.from_numpy(c[:m]).cuda() for c in cols], offs[:200_001], half_life=10.0)
torch.cuda.synchronize()
ref = orc.batched_rls(y[:m], [c[:m] for c in cols], offs[:200_001], half_life=10.0)
print("rls many groups", eng.last_kernel, "max diff", float(np.abs(out["coef"].cpu().numpy() - ref["coef"]).max()))
out = eng.rolling_least_squares(torch.from_numpy(y[:m]).cuda(), [torch.from_numpy(c[:m]).cuda() for c in cols], offs[:200_001], window_size=5, min_periods=3)
torch.cuda.synchronize()
print("rolling many groups", eng.last_kernel, "finite rows", int(torch.isfinite(out["pred"]).sum()))
