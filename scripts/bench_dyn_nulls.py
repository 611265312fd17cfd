"""The dynamic entries on a frame with nulls (10 000 sequences x 1 000 rows x 6, f64; 3 % of the targets NaN or a validity mask):
wall clock per call next to the null-free frame -- which engine takes it and what the host-side tables cost."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine  # noqa: E402

eng = Engine(0)
G, n, k = int(os.environ.get("G", 10_000)), 1_000, int(os.environ.get("K", 6))
gen = torch.Generator(device="cuda").manual_seed(3)
cols = [torch.randn(G * n, generator=gen, device="cuda", dtype=torch.float64) for _ in range(k)]
y = sum(cols) + 0.1 * torch.randn(G * n, generator=gen, device="cuda", dtype=torch.float64)
offs = np.arange(G + 1, dtype=np.int64) * n
valid = (torch.rand(G * n, generator=gen, device="cuda") > 0.03).to(torch.uint8)
y_nan = torch.where(valid.bool(), y, torch.full_like(y, float("nan")))


def timed(fn, reps=5):
    fn(); eng.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    eng.synchronize(); torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


for name, kw in (("null-free", dict(y=y, null_free=True)), ("validity bytes", dict(y=y, valid=valid)), ("NaN targets", dict(y=y_nan))):
    yy = kw.pop("y")
    for pol in ("drop", "drop_window"):
        ms = timed(lambda: eng.rolling_least_squares(yy, cols, offs, window_size=252, min_periods=max(6, k), null_policy=pol, **kw))
        print(f"k={k:2d} rolling {name:15s} {pol:12s} {ms:8.3f} ms per call  {eng.last_kernel}")
    ms = timed(lambda: eng.recursive_least_squares(yy, cols, offs, half_life=21.0, **kw))
    print(f"k={k:2d} rls     {name:15s} {'':12s} {ms:8.3f} ms per call  {eng.last_kernel}")
