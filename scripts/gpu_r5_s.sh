#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r5s
timeout 1200 python -m pytest tests/test_k5_gpu.py tests/test_nulls_gpu.py tests/test_predict_policy_gpu.py tests/test_k7_gpu.py tests/test_frontend_gpu.py tests/test_k1_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | head
