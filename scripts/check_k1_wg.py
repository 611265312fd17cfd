"""POLS_K1_WG=2|4 (several teams per workgroup, k1_kernel_wg) against the shipped 256-thread build on the headline and cfg3 shapes:
the arithmetic is the same instruction stream per team, so the outputs must be bit-identical; also group counts that are not a
multiple of the teams per workgroup."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine  # noqa: E402

eng = Engine(0)
g = torch.Generator(device="cuda").manual_seed(3)
ok = True
for dt, kw in ((torch.float32, {}), (torch.float64, dict(alpha=1.0, weights=True))):
    for G in (10_000, 1_001, 7):
        n, k = 1000, 8
        N = G * n
        offs = np.arange(0, N + 1, n, dtype=np.int64)
        cols = [torch.randn(N, device="cuda", generator=g, dtype=dt) for _ in range(k)]
        y = sum(cols) + 0.1 * torch.randn(N, device="cuda", generator=g, dtype=dt)
        kws = dict(kw)
        if kws.pop("weights", False):
            kws["weights"] = torch.rand(N, device="cuda", generator=g, dtype=dt) + 0.5
        ref = None
        for wg in (None, "2", "4"):
            eng.set_option("K1_WG", wg)
            out = eng.least_squares(y, cols, offs, want=("pred", "coef", "status"), **kws)
            eng.synchronize()
            cur = {key: out[key].clone() for key in ("pred", "coef", "status")}
            name = eng.last_kernel
            if ref is None:
                ref = cur
            same = all(torch.equal(ref[key], cur[key]) for key in cur)
            ok = ok and same and (wg is None or name.endswith("_wg" + wg))
            print(f"{str(dt):14s} G={G:6d} wg={wg} {name:60s} bit-identical={same}")
        eng.set_option("K1_WG", None)
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
