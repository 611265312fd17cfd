"""Interleaved A/B of the XCD-contiguous workgroup -> group map (POLS_K1_XCD) on the static BASELINE shapes, ONE process, three rotated
frames per shape (1.2 GB: nothing is served from the Infinity Cache), wall clock per call over back-to-back launches."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine  # noqa: E402

eng = Engine(0)
SHAPES = [("cfg2 f32 10k x 1000 x 8", torch.float32, 10_000, 1000, 8, {}, False),
          ("cfg3 f64 ridge + w", torch.float64, 10_000, 1000, 8, dict(alpha=1.0, l1_ratio=0.0), True),
          ("f64 10k x 1000 x 8", torch.float64, 10_000, 1000, 8, {}, False),
          ("f32 ragged 900..1020", torch.float32, 10_000, (900, 1020), 8, {}, False),
          ("f32 50k x 200 x 8", torch.float32, 50_000, 200, 8, {}, False),
          ("f32 10k x 1000 x 6", torch.float32, 10_000, 1000, 6, {}, False)]
if os.environ.get("ONLY"):
    SHAPES = [s for s in SHAPES if os.environ["ONLY"] in s[0]]
for name, dt, G, n, k, kw, weighted in SHAPES:
    if isinstance(n, tuple):
        offs = np.concatenate([[0], np.cumsum(np.random.default_rng(0).integers(n[0], n[1] + 1, size=G))]).astype(np.int64)
    else:
        offs = np.arange(0, (G + 1) * n, n, dtype=np.int64)
    N = int(offs[-1])
    plans = []
    for f in range(3):
        g = torch.Generator(device="cuda").manual_seed(f)
        cols = [torch.randn(N, device="cuda", generator=g, dtype=dt) for _ in range(k)]
        y = sum(cols) + 0.1 * torch.randn(N, device="cuda", generator=g, dtype=dt)
        w = torch.rand(N, device="cuda", generator=g, dtype=dt) + 0.5 if weighted else None
        plans.append(eng.plan_least_squares(y, cols, offs, weights=w, want=("pred", "coef"), null_free=True, **kw))
    for f in range(1, 3):
        for key in ("coef", "pred"):
            plans[f].set_output(key, plans[0].results[key])
    res = {0: [], 1: []}
    names = {}
    ref = None
    for rnd in range(10):
        for xcd in (0, 1):
            eng.set_option("K1_XCD", str(xcd) if xcd else None)
            for i in range(6):
                plans[i % 3].run()
            eng.synchronize()
            if rnd == 0:
                plans[0].run(); eng.synchronize(); torch.cuda.synchronize()
                cur = plans[0].results["pred"].clone()
                if ref is None:
                    ref = cur
                else:
                    assert torch.equal(ref, cur), "the remapped launch must produce the same bits"
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(150):
                plans[i % 3].run()
            e1.record()
            torch.cuda.synchronize()
            res[xcd].append(e0.elapsed_time(e1) * 1e3 / 150)
            names[xcd] = eng.last_kernel
    eng.set_option("K1_XCD", None)
    b = N * (k + 2 + (1 if weighted else 0)) * (4 if dt == torch.float32 else 8)
    for xcd in (0, 1):
        a = np.array(res[xcd][2:])
        print(f"{name:26s} xcd={xcd} median {np.median(a):7.2f} us  min {a.min():7.2f}  max {a.max():7.2f}  {b / np.median(a) / 1e6:5.2f} TB/s  {names[xcd]}", flush=True)
