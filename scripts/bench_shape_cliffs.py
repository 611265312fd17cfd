"""Static OLS (predictions) over odd frame shapes with the same ~10M rows x 8 features f32 / f64: where does the rate fall off a cliff?"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine  # noqa: E402

eng = Engine(0)
time.sleep(2.0)                                               # (a benchmark process that has just exited is still being torn down: its frees disturb the first measurement)
N, k = 10_000_000, 8
SHAPES = [("1 x 10M", [N]), ("10 x 1M", [N // 10] * 10), ("100 x 100k", [100_000] * 100), ("1k x 10k", [10_000] * 1000),
          ("4k x 2.5k", [2_500] * 4000), ("10k x 1k", [1_000] * 10_000), ("100k x 100", [100] * 100_000), ("1M x 10", [10] * 1_000_000),
          ("mixed 1 x 5M + 5k x 1k", [5_000_000] + [1_000] * 5_000)]
# (2 500 000 groups of 4 rows x 8 features -- fewer rows than columns: EVERY group goes through the minimum-norm fix-up pass, 65 s per
#  call; not part of the default sweep.  ONLY=<substring> picks shapes.)
if os.environ.get("SHORT"):                                   # round 5: those groups take K6s (a sub-wave team per group), milliseconds
    SHAPES += [("2.5M x 4 (n < k)", [4] * 2_500_000), ("1.25M x 8 (n = k)", [8] * 1_250_000), ("mixed 5k x 1k + 100k x 6", [1_000] * 5_000 + [6] * 100_000)]
if os.environ.get("ONLY"):
    SHAPES = [sh for sh in SHAPES if os.environ["ONLY"] in sh[0]]
for dt, dname in ((torch.float32, "f32"), (torch.float64, "f64")):
    gen = torch.Generator(device="cuda").manual_seed(0)
    cols = [torch.randn(N, device="cuda", generator=gen, dtype=dt) for _ in range(k)]
    y = sum(cols) + 0.1 * torch.randn(N, device="cuda", generator=gen, dtype=dt)
    for name, sizes in SHAPES:
        offs = np.concatenate([[0], np.cumsum(np.asarray(sizes, dtype=np.int64))])
        n = int(offs[-1])
        try:
            plan = eng.plan_least_squares(y[:n], [c[:n] for c in cols], offs, want=("pred",))
            for _ in range(30 if name == SHAPES[0][0] else 2):     # (the first shape of a dtype also wakes the clocks: 2 runs were 0.3 ms of work)
                plan.run()
            eng.synchronize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                plan.run()
            eng.synchronize(); torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / 5
            tb = n * (k + 2) * (4 if dt == torch.float32 else 8) / ms / 1e9
            print(f"{dname} {name:26s} {ms:9.3f} ms  {tb:5.2f} TB/s  {eng.last_kernel}", flush=True)
        except Exception as exc:  # noqa: BLE001
            print(f"{dname} {name:26s} ERROR {repr(exc)[:120]}", flush=True)
