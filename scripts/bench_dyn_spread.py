"""RLS and rolling OLS over groups whose lengths spread widely (log-normal with a tail, short + long mixes): do the dynamic kernels keep their rate?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine
eng = Engine(0)
rng = np.random.default_rng(0)
NT = 8_000_000
def sizes_for(name):
    if name == "equal 1000": s = np.full(NT // 1000, 1000)
    elif name == "U(10,1000)": s = rng.integers(10, 1001, size=40000)
    elif name == "lognormal(300)": s = np.clip(rng.lognormal(np.log(300), 0.8, size=60000).astype(np.int64), 5, 4000)
    elif name == "90% 50 + 10% 1000": s = np.where(rng.random(200000) < 0.9, 50, 1000)
    elif name == "lognormal(300) + 3 x 500k": s = np.concatenate([np.clip(rng.lognormal(np.log(300), 0.8, size=60000).astype(np.int64), 5, 4000)[:15000], [500_000] * 3])
    c = np.cumsum(s); return s[: int(np.searchsorted(c, NT))]
first = True
for k in (6, 12):
    gen = torch.Generator(device="cuda").manual_seed(1)
    cols = [torch.randn(NT, generator=gen, device="cuda", dtype=torch.float64) for _ in range(k)]
    y = sum(cols) + 0.1 * torch.randn(NT, generator=gen, device="cuda", dtype=torch.float64)
    for name in ("equal 1000", "U(10,1000)", "lognormal(300)", "90% 50 + 10% 1000", "lognormal(300) + 3 x 500k"):
        sizes = sizes_for(name)
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        M = int(offs[-1])
        out = {"pred": torch.empty(M, device="cuda", dtype=torch.float64), "coef": torch.empty(M, k, device="cuda", dtype=torch.float64)}
        for what in ("rls", "rolling"):
            if what == "rls":
                plan = eng.plan_recursive_least_squares(y[:M], [c[:M] for c in cols], offs, half_life=21.0, out=out, null_free=True)
            else:
                plan = eng.plan_rolling_least_squares(y[:M], [c[:M] for c in cols], offs, window_size=252, min_periods=k, null_policy="drop", out=out, null_free=True)
            for _ in range(20 if first else 3): plan.run()
            first = False
            eng.synchronize(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): plan.run()
            eng.synchronize(); torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / 5
            print(f"k={k:2d} {name:26s} groups={len(sizes):6d} {what:8s} {ms:8.3f} ms {M / ms / 1e6:8.1f} G rows/s  {eng.last_kernel}", flush=True)
