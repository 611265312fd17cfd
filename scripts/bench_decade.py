"""Groups of 1 025..4 096 rows (ten years of trading days per asset) at up to 8 columns: the four-chunk 256-thread team of K1 against K2 (wall clock per call)."""
import numpy as np, torch, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine
eng = Engine(0)
for dt, nm, b in ((torch.float32, "f32", 4), (torch.float64, "f64", 8)):
    for G, n, k in ((4000, 2500, 8), (3333, 3000, 8), (2500, 4000, 8), (4000, 2500, 4), (4000, 2500, 6), (4000, 2520, 8), (5000, 2000, 8), (6000, 1500, 8), (8000, 1250, 8),
                    (5000, 2000, 9), (5000, 2000, 10), (3333, 3000, 9), (2500, 4000, 10), (8000, 1250, 9)):
        gen = torch.Generator(device="cuda").manual_seed(3)
        cols = [torch.randn(G * n, generator=gen, device="cuda", dtype=dt) for _ in range(k)]
        y = sum(cols) + 0.1 * torch.randn(G * n, generator=gen, device="cuda", dtype=dt)
        offs = np.arange(G + 1, dtype=np.int64) * n
        if n == 2520:                                                # ragged: 2 300..2 520 rows
            offs = np.concatenate([[0], np.cumsum(np.random.default_rng(0).integers(2300, 2521, size=G))]).astype(np.int64)
            N = int(offs[-1]); cols = [c[:N] for c in cols]; y = y[:N]
        for e in (None, "k2"):
            eng.set_option("STATIC_ENGINE", e)
            plan = eng.plan_least_squares(y, cols, offs, want=("pred",))
            for _ in range(3): plan.run()
            eng.synchronize(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): plan.run()
            eng.synchronize(); torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / 10
            print(f"{nm} {G} x {n} x {k} engine={e}: {ms:7.3f} ms {int(offs[-1])*(k+2)*b/ms/1e9:5.2f} TB/s {eng.last_kernel}", flush=True)
        eng.set_option("STATIC_ENGINE", None)
