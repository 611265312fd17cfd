#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r2b; O=gpurun_out/r2b
echo "== pytest subset"
timeout 900 python -m pytest tests/test_k2_gpu.py tests/test_k5_gpu.py tests/test_k6_gpu.py tests/test_k7_gpu.py -m gpu -q --maxfail=30 --tb=line --deselect tests/test_k5_gpu.py::test_cfg5_full_size 2>&1 | tail -30 | cut -c1-300 | tee $O/pytest_gpu.log
echo "== timeline cfg5 K2"
POLS_TIMELINE=1 timeout 300 python bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep timeline | tail -1 | tee $O/timeline_cfg5.txt
echo "== bench cfg5 K2"
timeout 300 python bench.py --config cfg5 --steps 10 --warmup 3 --no-cpu-baseline 2>$O/cfg5.err | tee $O/bench_cfg5_k2.json | cut -c1-200
echo "== bench cfg2 plain / nt loads"
timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>$O/cfg2.err | tee $O/bench_cfg2.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
POLS_K1_NT_LOADS=1 timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>>$O/cfg2.err | tee $O/bench_cfg2_nt.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>$O/cfg2.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
POLS_K1_NT_LOADS=1 timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>>$O/cfg2.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
echo "== smoke shape: 8 feats + intercept f32, K2 vs K1m"
python - <<'PY'
import numpy as np, torch, time
from polars_ols_amd import Engine
eng = Engine(0)
G, n, k = 10_000, 1_000, 8
for dt in (torch.float32, torch.float64):
    cols = [torch.randn(G*n, device="cuda", dtype=dt) for _ in range(k)]
    y = sum(cols) + 0.1*torch.randn(G*n, device="cuda", dtype=dt)
    offs = np.arange(G+1, dtype=np.int64)*n
    for opt in (None, "mfma"):
        eng.set_option("K1_ENGINE", opt)
        plan = eng.plan_least_squares(y, cols, offs, add_intercept=True, want=("pred","coef"))
        for _ in range(5): plan.run()
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(50): plan.run()
        torch.cuda.synchronize(); dt_ms=(time.perf_counter()-t0)/50*1e3
        b = 4 if dt==torch.float32 else 8
        print(dt, opt, eng.last_kernel, "ms=%.4f TB/s=%.2f" % (dt_ms, b*n*(k+2)*G/dt_ms/1e9))
    eng.set_option("K1_ENGINE", None)
PY
