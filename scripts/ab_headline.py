"""Interleaved A/B of kernel variants on BASELINE configs[1] in ONE process: wall clock per call over back-to-back launches."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine  # noqa: E402

eng = Engine(0)
G, n, k = 10_000, 1000, int(os.environ.get("K", "8"))
offs = np.arange(0, (G + 1) * n, n, dtype=np.int64)
if os.environ.get("RAGGED"):                                  # RAGGED=lo,hi: unequal, unaligned groups
    lo, hi = (int(v) for v in os.environ["RAGGED"].split(","))
    offs = np.concatenate([[0], np.cumsum(np.random.default_rng(0).integers(lo, hi + 1, size=G))]).astype(np.int64)
N = int(offs[-1])
g = torch.Generator(device="cuda").manual_seed(0)
F64 = os.environ.get("DTYPE") == "f64"                        # DTYPE=f64: cfg3's shape (f64, ridge alpha = 1, sample weights)
dt = torch.float64 if F64 else torch.float32
FRAMES = int(os.environ.get("FRAMES", "3"))                   # rotated frames: more input than the Infinity Cache holds, like bench.py
plans = []
out = None
for f in range(FRAMES):
    cols = [torch.randn(N, device="cuda", generator=g, dtype=dt) for _ in range(k)]
    y = sum(cols) + 0.1 * torch.randn(N, device="cuda", generator=g, dtype=dt)
    kw = dict(weights=torch.rand(N, device="cuda", generator=g, dtype=dt) + 0.5, alpha=1.0) if F64 else {}
    p = eng.plan_least_squares(y, cols, offs, want=("pred", "coef"), **kw)
    if out is None:
        out = p.results
    else:
        for key in ("pred", "coef"):
            p.set_output(key, out[key])
    plans.append(p)


class _Rot:
    i = 0

    def run(self):
        plans[self.i % FRAMES].run()
        self.i += 1


plan = _Rot()
variants = {"wave_rc4_nt": {"K1_SHAPE": "wave"}, "default": {}, "team256_rc1_nt": {"K1_SHAPE": "team"}, "team256_rc1_p2_nt": {"K1_SHAPE": "team", "K1_PASSES": "2"},
            "team256_rc1_p2": {"K1_SHAPE": "team", "K1_PASSES": "2", "K1_NT_LOADS": "0"}, "team256_rc1_p3_nt": {"K1_SHAPE": "team", "K1_PASSES": "3"},
            # round 6: several teams per workgroup (k1_kernel_wg): 512 / 1 024 threads = 2 / 4 times the groups of the 256-thread block
            "wg2": {"K1_WG": "2"}, "wg4": {"K1_WG": "4"}, "wg4_p3": {"K1_WG": "4", "K1_PASSES": "3"}, "wg2_p3": {"K1_WG": "2", "K1_PASSES": "3"}}
if os.environ.get("ONLY"):
    variants = {k: v for k, v in variants.items() if k in os.environ["ONLY"].split(",")}
res = {v: [] for v in variants}
names = {}
for rnd in range(12):
    for v, opts in variants.items():
        for key in ("K1_SHAPE", "K1_PASSES", "K1_NT_LOADS", "K1_WG"):
            eng.set_option(key, opts.get(key))
        for _ in range(5):
            plan.run()
        eng.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            plan.run()
        e1.record()
        torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) * 1e3 / 200)
        names[v] = eng.last_kernel
for v in variants:
    a = np.array(res[v][2:])
    print(f"{v:20s} median {np.median(a):6.2f} us  min {a.min():6.2f}  max {a.max():6.2f}  {names[v]}")
