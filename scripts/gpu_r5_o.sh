#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/r5o; O=$R/gpurun_out/r5o
timeout 1200 python -m pytest tests/test_k1_gpu.py tests/test_k5_gpu.py tests/test_nulls_gpu.py tests/test_routing_gpu.py tests/test_k7_gpu.py tests/test_k2_gpu.py -m gpu -x -q 2>&1 | tail -12
timeout 600 python scripts/bench_spread.py 2>&1 | grep -v amdgpu.ids | tee $O/bench_spread.txt
