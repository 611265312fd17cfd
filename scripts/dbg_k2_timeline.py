"""K2 phase timeline on the cfg5 shape (2 000 rows x 16 f64 columns, elastic net) with and without the predictions output: what the top of
the persistent loop waits for (POLS_TIMELINE=1 prints the mean cycles per phase: p0 loop top -> first tile, p1 Gram, p2 barrier, p3 solve,
p4 predictions; x7 = the last wave's first tile)."""
import os
import sys

import numpy as np
import torch

os.environ["POLS_TIMELINE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine  # noqa: E402

eng = Engine(0)
G, n, k = int(os.environ.get("G", 50_000)), 2000, 16
g = torch.Generator(device="cuda").manual_seed(0)
cols = [torch.randn(G * n, device="cuda", generator=g, dtype=torch.float64) for _ in range(k)]
y = sum(cols) + 0.1 * torch.randn(G * n, device="cuda", generator=g, dtype=torch.float64)
offs = np.arange(G + 1, dtype=np.int64) * n
for want in (("pred", "coef"), ("coef",)):
    print("want", want, flush=True)
    sys.stderr.flush()
    plan = eng.plan_least_squares(y, cols, offs, alpha=0.001, l1_ratio=0.5, want=want)
    plan.run(); plan.run()
    eng.synchronize()
