#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r5v
timeout 900 python -m pytest tests/test_k1_gpu.py tests/test_k2_gpu.py tests/test_routing_gpu.py tests/test_k6_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert|^E " | head -12
ROWS=16,24,32,48 COLS=16 timeout 600 python scripts/bench_rows_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5v/sweep_k1t_16_f64.txt
