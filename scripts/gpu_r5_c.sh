#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import numpy as np, torch, time, sys
sys.path.insert(0, '.')
from polars_ols_amd.engine import Engine
eng = Engine(0)
for dt, nm in ((torch.float32, "f32"),):
    for G, n, k in ((4000, 2500, 8), (3333, 3000, 8), (2500, 4000, 8), (4000, 2500, 4), (4000, 2500, 6), (4000, 2520, 8)):
        gen = torch.Generator(device="cuda").manual_seed(3)
        cols = [torch.randn(G * n, generator=gen, device="cuda", dtype=dt) for _ in range(k)]
        y = sum(cols) + 0.1 * torch.randn(G * n, generator=gen, device="cuda", dtype=dt)
        offs = np.arange(G + 1, dtype=np.int64) * n
        if n == 2520:
            offs = np.concatenate([[0], np.cumsum(np.random.default_rng(0).integers(2300, 2521, size=G))]).astype(np.int64)
            N = int(offs[-1]); cols = [c[:N] for c in cols]; y = y[:N]
        for e in (None, "k2"):
            eng.set_option("STATIC_ENGINE", e)
            plan = eng.plan_least_squares(y, cols, offs, want=("pred",))
            for _ in range(3): plan.run()
            eng.synchronize(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): plan.run()
            eng.synchronize(); torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / 10
            print(f"{nm} {G} x {n} x {k} engine={e}: {ms:7.3f} ms {int(offs[-1])*(k+2)*4/ms/1e9:5.2f} TB/s {eng.last_kernel}")
        eng.set_option("STATIC_ENGINE", None)
PY
timeout 1500 python -m pytest tests/test_k1_gpu.py tests/test_k2_gpu.py tests/test_routing_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^    \|^$" | tail -8 | cut -c1-300
