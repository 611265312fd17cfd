"""debug aid (round 6): the look-back form of K3c against its own fallback and the scan on the route-boundary frame"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine
eng = Engine(0)
k, half_life = int(os.environ.get("K", 1)), float(os.environ.get("HL", 5.0))
rng = np.random.default_rng(31 * k + int(half_life or 0))
sizes = np.array([20_000, 1, 3, 1020, 1024, 2048, 5, 4099, 700, 2, 3500, 1024 * 3 - 7, 7, 9000], dtype=np.int64)
offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
N = int(offs[-1])
cols = [rng.standard_normal(N) for _ in range(k)]
y = sum(cols) + 0.1 * rng.standard_normal(N)
cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
kw = dict(half_life=half_life, initial_state_covariance=1e3, initial_state_mean=[0.25] * k, null_free=True)
res = {}
for name, opts in (("default", {}), ("slow", {"RLS_SPINS": "0"}), ("halo", {"RLS_ENGINE": "halo"}), ("scan", {"RLS_ENGINE": "scan"})):
    for kk, v in opts.items(): eng.set_option(kk, v)
    out = eng.recursive_least_squares(cu(y), [cu(c) for c in cols], offs, **kw)
    res[name] = out["coef"].double().cpu().numpy(); print(name, eng.last_kernel)
    for kk in opts: eng.set_option(kk, None)
for name in ("default", "slow", "halo"):
    d = np.abs(res[name] - res["scan"]).max(axis=1)
    bad = np.flatnonzero(d > 1e-7)
    print(name, "max diff vs scan", d.max(), "rows > 1e-7:", len(bad), bad[:10], "tiles", np.unique(bad // 1024)[:20])
    for r in bad[:3]:
        g = int(np.searchsorted(offs, r, side="right") - 1)
        print("   row", r, "tile", r // 1024, "pos in tile", r % 1024, "group", g, "group start", offs[g], "row in group", r - offs[g], res[name][r], res["scan"][r])
