"""Per-phase cycle counts (TIMELINE option) of K2w: p0 loads, p1 Gram, p2 reduce, p3 solve, p4 predictions."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine  # noqa: E402

eng = Engine(0)
eng.set_option("STATIC_ENGINE", "k2w")
for dt in (torch.float64, torch.float32):
    for k, G, n in ((31, 5_000, 1000), (20, 5_000, 1000), (31, 20_000, 200)):
        offs = np.arange(0, (G + 1) * n, n, dtype=np.int64)
        g = torch.Generator(device="cuda").manual_seed(0)
        cols = [torch.randn(G * n, device="cuda", generator=g, dtype=dt) for _ in range(k)]
        y = sum(cols) + 0.1 * torch.randn(G * n, device="cuda", generator=g, dtype=dt)
        plan = eng.plan_least_squares(y, cols, offs, want=("pred",))
        plan.run()
        eng.set_option("TIMELINE", "1")
        print(dt, "k", k, G, "x", n, file=sys.stderr, flush=True)
        plan.run()
        eng.synchronize()
        eng.set_option("TIMELINE", None)
        del cols, y, plan
