"""f64, 17..23 columns, groups of 300 / 500 rows (a year or two of daily data against twenty factors): the resident multi-pass K1 (256-thread
team; round 6: three waves per SIMD from 18 columns -- shorter passes, the solving wave's rows parked in LDS) next to K2w (MFMA, two tiles) and
the dispatcher's choice.  Wall clock per call."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine
eng = Engine(0)
DT = torch.float32 if os.environ.get("DT") == "f32" else torch.float64          # DT=f32: the same sweep in f32 (16..31 columns with KS)
NM, B = ("f32", 4) if DT == torch.float32 else ("f64", 8)
KS = tuple(int(v) for v in os.environ.get("KS", "17,18,19,20,21,22,23").split(","))
N = 3_840_000          # (a multiple of 500, 300 and 256)
gen = torch.Generator(device="cuda").manual_seed(1)
allc = [torch.randn(N, generator=gen, device="cuda", dtype=DT) for _ in range(max(KS))]
for k in KS:
    cols = allc[:k]
    y = sum(cols[:4]) + 0.1 * torch.randn(N, generator=gen, device="cuda", dtype=DT)
    for n in (500, 300, 256):
        G = N // n
        offs = np.arange(G + 1, dtype=np.int64) * n
        for engine in ((None, "nok2", "k2w") if DT == torch.float64 else (None,)):
            eng.set_option("STATIC_ENGINE", engine)
            try:
                plan = eng.plan_least_squares(y, cols, offs, want=("pred",))
                for _ in range(3): plan.run()
                eng.synchronize(); torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(8): plan.run()
                eng.synchronize(); torch.cuda.synchronize()
                ms = 1e3 * (time.perf_counter() - t0) / 8
                print(f"{NM} k={k} rows={n} engine={engine}: {ms:.3f} ms {G * n * (k + 2) * B / ms / 1e9:.2f} TB/s {eng.last_kernel}", flush=True)
            except Exception as exc:
                print(f"{NM} k={k} rows={n} engine={engine}: {exc}")
        eng.set_option("STATIC_ENGINE", None)
