"""K1 with a fused null policy next to the plain kernel: BASELINE configs[1] with 5 % null targets, null_policy="drop"."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine  # noqa: E402


def main():
    G, n, k = 10_000, 1_000, 8
    eng = Engine(0)
    g = torch.Generator(device="cuda").manual_seed(0)
    cols = [torch.randn(G * n, device="cuda", generator=g) for _ in range(k)]
    y = sum(cols) + 0.1 * torch.randn(G * n, device="cuda", generator=g)
    yn = y.clone()
    yn[torch.rand(G * n, device="cuda", generator=g) < 0.05] = float("nan")
    offs = np.arange(G + 1, dtype=np.int64) * n
    res = {}
    for name, yy, kw in (("plain", y, {}), ("drop_5pct_null_targets", yn, {"null_policy": "drop"}),
                         ("drop_no_nulls", y, {"null_policy": "drop"}), ("zero", yn, {"null_policy": "zero"})):
        plan = eng.plan_least_squares(yy, cols, offs, want=("pred",), **kw)
        for _ in range(10):
            plan.run()
        eng.timing(1)
        for _ in range(40):
            plan.run()
        ms = eng.timing_collect()
        eng.timing(False)
        res[name] = {"kernel": eng.last_kernel, "us": float(np.mean(ms) * 1e3)}
    w = torch.rand(G * n, device="cuda", generator=g) + 0.5
    offs_r = np.concatenate([[0], np.cumsum(np.random.default_rng(0).integers(900, 1021, size=G))]).astype(np.int64)
    nr = int(offs_r[-1])
    for name, yy, oo, kw in (("weights_drop_5pct_null_targets", yn, offs, {"null_policy": "drop", "weights": w}),
                             ("ragged_900_1020_drop", yn[:nr], offs_r, {"null_policy": "drop"})):
        cc = [c[:len(yy)] for c in cols]
        if "weights" in kw:
            kw = dict(kw, weights=kw["weights"][:len(yy)])
        plan = eng.plan_least_squares(yy, cc, oo, want=("pred",), **kw)
        for _ in range(10):
            plan.run()
        eng.timing(1)
        for _ in range(40):
            plan.run()
        ms = eng.timing_collect()
        eng.timing(False)
        res[name] = {"kernel": eng.last_kernel, "us": float(np.mean(ms) * 1e3)}
    # the same in f64 (BASELINE configs[2]'s shape without the weights)
    cols64 = [c.double() for c in cols]
    y64, yn64 = y.double(), yn.double()
    for name, yy, kw in (("f64_plain", y64, {}), ("f64_drop_5pct_null_targets", yn64, {"null_policy": "drop"}),
                         ("f64_drop_no_nulls", y64, {"null_policy": "drop"})):
        plan = eng.plan_least_squares(yy, cols64, offs, want=("pred",), **kw)
        for _ in range(10):
            plan.run()
        eng.timing(1)
        for _ in range(40):
            plan.run()
        ms = eng.timing_collect()
        eng.timing(False)
        res[name] = {"kernel": eng.last_kernel, "us": float(np.mean(ms) * 1e3)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
