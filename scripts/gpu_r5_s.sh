#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r5s
timeout 1200 python -m pytest tests/test_k5_gpu.py tests/test_nulls_gpu.py tests/test_predict_policy_gpu.py tests/test_k7_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | head
timeout 600 python scripts/bench_long_nulls.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5s/bench_long_nulls_after.txt
