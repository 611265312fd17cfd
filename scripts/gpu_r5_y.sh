#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r5y
timeout 1200 python -m pytest tests/test_k3_gpu.py tests/test_k4_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|^E |assert" | head -6
sleep 2
timeout 900 python scripts/bench_dyn_spread.py 2>&1 | grep -v amdgpu.ids | grep "k=12" | tee gpurun_out/r5y/bench_dyn_spread_sorted.txt
