#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r5v
timeout 900 python -m pytest tests/test_k1_gpu.py tests/test_k2_gpu.py tests/test_routing_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert|^E " | head -12
ROWS=24,48,64,100,128,200 COLS=11,12,13,15 timeout 600 python scripts/bench_rows_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5v/sweep_k1t_wide.txt
POLS_K1_NOTINY=1 ROWS=24,48,64,100,128,200 COLS=11,12,13,15 timeout 600 python scripts/bench_rows_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5v/sweep_k1t_wide_off.txt
