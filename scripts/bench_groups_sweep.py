"""The headline shape (1 000 rows x 8 f32 columns per group) at growing group counts, kernel time by HIP events, beside pols_stream_probe on
the same buffers: how much of the distance to the stream ceiling is the drain of the last workgroups (a fixed cost per launch)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine  # noqa: E402

eng = Engine(0)
res = {}
for G in (2_500, 5_000, 10_000, 20_000, 40_000, 80_000):
    n, k = 1000, 8
    N = G * n
    offs = np.arange(G + 1, dtype=np.int64) * n
    g = torch.Generator(device="cuda").manual_seed(0)
    frames = []
    for f in range(3 if G <= 20_000 else 2):
        cols = [torch.randn(N, device="cuda", generator=g, dtype=torch.float32) for _ in range(k)]
        y = sum(cols) + 0.1 * torch.randn(N, device="cuda", generator=g, dtype=torch.float32)
        frames.append(eng.plan_least_squares(y, cols, offs, want=("pred", "coef")))
    row = {}
    for what in ("kernel", "probe"):
        fn = (lambda p: p.run()) if what == "kernel" else (lambda p: p.stream_probe())
        for i in range(6):
            fn(frames[i % len(frames)])
        eng.synchronize()
        eng.timing(4)
        for i in range(48):
            fn(frames[i % len(frames)])
        eng.synchronize()
        ms = eng.timing_collect()
        eng.timing(False)
        us = float(np.mean(ms)) * 1e3
        row[what] = {"us": round(us, 1), "TBps": round(N * 40 / us / 1e6, 2)}
        if what == "kernel":
            row["name"] = eng.last_kernel
    row["kernel_over_probe"] = round(row["probe"]["us"] / row["kernel"]["us"], 3)
    res[G] = row
    del frames
    torch.cuda.empty_cache()
print(json.dumps(res))
