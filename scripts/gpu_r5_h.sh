#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/r5h; O=$R/gpurun_out/r5h
timeout 1200 python -m pytest tests/test_k5_gpu.py tests/test_routing_gpu.py tests/test_k1_gpu.py tests/test_k2_gpu.py tests/test_k7_gpu.py tests/test_frontend_gpu.py -m gpu -x -q 2>&1 | tail -15
ROWS=4000,2500,2500,6000,10000 COLS=2,8 timeout 600 python scripts/bench_rows_sweep.py 2>&1 | grep -v amdgpu.ids
