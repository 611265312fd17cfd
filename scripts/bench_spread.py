"""Static OLS (predictions, 8 features) on frames whose group sizes SPREAD widely -- what `.over(key)` on real panels looks like (assets with 20 rows next to
assets with 1 000): the dispatcher sizes its kernel for the largest group; what does that cost the small ones?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine
eng = Engine(0)
rng = np.random.default_rng(0)
N_TARGET = 8_000_000
def sizes_for(name):
    if name == "U(10,1000)": draw = lambda m: rng.integers(10, 1001, size=m)
    elif name == "U(100,1000)": draw = lambda m: rng.integers(100, 1001, size=m)
    elif name == "U(500,1000)": draw = lambda m: rng.integers(500, 1001, size=m)
    elif name == "U(10,250)": draw = lambda m: rng.integers(10, 251, size=m)
    elif name == "U(10,4000)": draw = lambda m: rng.integers(10, 4001, size=m)
    elif name == "lognormal(300)": draw = lambda m: np.clip(rng.lognormal(np.log(300), 0.8, size=m).astype(np.int64), 5, 4000)
    elif name == "90% 50 + 10% 1000": draw = lambda m: np.where(rng.random(m) < 0.9, 50, 1000)
    elif name == "99% 1000 + 1% 20": draw = lambda m: np.where(rng.random(m) < 0.99, 1000, 20)
    elif name == "50% 30 + 50% 1000": draw = lambda m: np.where(rng.random(m) < 0.5, 30, 1000)
    s = draw(200_000)
    c = np.cumsum(s)
    return s[: int(np.searchsorted(c, N_TARGET))]
NAMES = os.environ.get("ONLY", "").split(";") if os.environ.get("ONLY") else ["U(500,1000)", "U(100,1000)", "U(10,1000)", "U(10,250)", "U(10,4000)", "lognormal(300)", "90% 50 + 10% 1000", "99% 1000 + 1% 20", "50% 30 + 50% 1000"]
k = 8
for dt, nm, b in ((torch.float32, "f32", 4), (torch.float64, "f64", 8)):
    gen = torch.Generator(device="cuda").manual_seed(1)
    cols = [torch.randn(N_TARGET, generator=gen, device="cuda", dtype=dt) for _ in range(k)]
    y = sum(cols) + 0.1 * torch.randn(N_TARGET, generator=gen, device="cuda", dtype=dt)
    first = True
    for name in NAMES:
        sizes = sizes_for(name)
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        M = int(offs[-1])
        plan = eng.plan_least_squares(y[:M], [c[:M] for c in cols], offs, want=("pred",))
        for _ in range(30 if first else 3): plan.run()
        first = False
        eng.synchronize(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): plan.run()
        eng.synchronize(); torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 5
        print(f"{nm} {name:20s} groups={len(sizes):7d} {ms:7.3f} ms {M * (k + 2) * b / ms / 1e9:5.2f} TB/s  {eng.last_kernel}", flush=True)
