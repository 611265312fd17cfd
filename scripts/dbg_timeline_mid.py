"""Per-phase cycle counts (POLS TIMELINE option) of the wave-per-group kernel on mid-size groups."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine  # noqa: E402

eng = Engine(0)
rng = np.random.default_rng(0)
for name, lo, hi, G in (("year_130_252", 130, 252, 500_000), ("ragged_100_300", 100, 300, 50_000), ("small_40_120", 40, 120, 500_000)):
    sizes = rng.integers(lo, hi + 1, size=G)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(offs[-1])
    g = torch.Generator(device="cuda").manual_seed(0)
    cols = [torch.randn(n, device="cuda", generator=g) for _ in range(8)]
    y = sum(cols) + 0.1 * torch.randn(n, device="cuda", generator=g)
    plan = eng.plan_least_squares(y, cols, offs, want=("pred",))
    for sub in (os.environ.get("SUBS", "0,16,32,64").split(",")):
        eng.set_option("K1_PERSIST", "0" if sub == "0" else "1")
        eng.set_option("K1_PERSIST_SUB", None if sub == "0" else sub)
        plan.run()
        eng.set_option("TIMELINE", "1")
        print(name, "sub", sub, file=sys.stderr)
        plan.run()
        eng.synchronize()
        eng.set_option("TIMELINE", None)
    del cols, y, plan
