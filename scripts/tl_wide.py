"""Phase timelines (POLS_TIMELINE) of the 17-31-column kernels on 1 000-row groups: K1 wide (VALU passes) vs K2w (MFMA, two tiles)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine
eng = Engine(0)
for dt, nm in ((torch.float32, "f32"), (torch.float64, "f64")):
    N = 4_000_000
    gen = torch.Generator(device="cuda").manual_seed(1)
    allc = [torch.randn(N, generator=gen, device="cuda", dtype=dt) for _ in range(31)]
    for k in (20, 24, 31):
        cols = allc[:k]
        y = sum(cols[:4]) + 0.1 * torch.randn(N, generator=gen, device="cuda", dtype=dt)
        for n in (1000, 500):
            G = N // n
            offs = np.arange(G + 1, dtype=np.int64) * n
            for engine in (None, "k2w"):
                eng.set_option("STATIC_ENGINE", engine)
                plan = eng.plan_least_squares(y, cols, offs, want=("pred",))
                for _ in range(3): plan.run()
                eng.synchronize(); torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(5): plan.run()
                eng.synchronize(); torch.cuda.synchronize()
                ms = 1e3 * (time.perf_counter() - t0) / 5
                b = 4 if dt == torch.float32 else 8
                print(f"{nm} k={k} rows={n} engine={engine}: {ms:.3f} ms {N * (k + 2) * b / ms / 1e9:.2f} TB/s {eng.last_kernel}", flush=True)
                eng.set_option("TIMELINE", "1")
                try:
                    plan.run(); eng.synchronize()
                except Exception as exc:
                    print("timeline:", exc)
                eng.set_option("TIMELINE", None)
            eng.set_option("STATIC_ENGINE", None)
