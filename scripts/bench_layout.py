"""Times K9 group-key ingestion on BASELINE configs[1]'s frame in arrival order (10 000 interleaved groups x 1 000 rows, 8 f32
features + target): layout build, columns into group order, predictions back to frame order -- next to the solve itself."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_ols_amd.engine import Engine, Layout  # noqa: E402


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    G, n_per, k = 10_000, 1_000, 8
    n = G * n_per
    eng = Engine(0)
    g = torch.Generator(device="cuda").manual_seed(0)
    key = torch.randint(0, G, (n,), device="cuda", generator=g)
    cols = [torch.randn(n, device="cuda", generator=g) for _ in range(k + 1)]
    res = {"rows": n, "groups": G, "columns": k + 1}
    res["layout_create_ms"] = timed(lambda: Layout(eng, key).close())
    lay = Layout(eng, key)
    res["take_9_f32_columns_ms"] = timed(lambda: lay.take(cols))
    for mode in ("gather", "scatter"):
        eng.set_option("K9_TAKE", mode)
        res[f"take_9_f32_columns_{mode}_ms"] = timed(lambda: lay.take(cols))
    eng.set_option("K9_TAKE", None)
    cols64 = [c.double() for c in cols[:3]]
    res["take_3_f64_columns_ms"] = timed(lambda: lay.take(cols64))
    moved = lay.take(cols)
    res["untake_1_f32_column_ms"] = timed(lambda: lay.untake([moved[0]]))
    res["row_groups_ms"] = timed(lambda: lay.row_groups())
    res["torch_sort_stable_ms"] = timed(lambda: torch.sort(key, stable=True))
    order = torch.sort(key, stable=True)[1]
    res["torch_index_9_columns_ms"] = timed(lambda: [c[order] for c in cols])
    plan = eng.plan_least_squares(moved[0], moved[1:], lay.offsets, want=("pred",))
    res["solve_ms"] = timed(lambda: plan.run())
    res["solve_kernel"] = eng.last_kernel
    res["group_rows_min_max"] = [int(np.diff(lay.offsets).min()), int(np.diff(lay.offsets).max())]
    # bytes: take reads the index (4 B) + 9 gathered 4-byte elements (each a 32 B sector at random) and writes 36 B per row
    res["take_useful_GBps"] = n * (4 + 2 * 4 * (k + 1)) / res["take_9_f32_columns_ms"] / 1e6
    print(json.dumps(res))


if __name__ == "__main__":
    main()
