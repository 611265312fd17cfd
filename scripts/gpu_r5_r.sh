#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/r5r; O=$R/gpurun_out/r5r
timeout 1200 python -m pytest tests/test_k5_gpu.py tests/test_k1_gpu.py tests/test_routing_gpu.py tests/test_k2_gpu.py tests/test_frontend_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | head
