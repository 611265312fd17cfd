#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r5y
sleep 2
SHORT=1 ONLY="mixed" timeout 300 python scripts/bench_shape_cliffs.py 2>&1 | grep "TB/s"
ONLY="90% 50 + 10% 1000;50% 30 + 50% 1000;lognormal(300)" timeout 600 python scripts/bench_spread.py 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_k1_gpu.py tests/test_k6_gpu.py -m gpu -x -q -k "size_classes or short_groups" 2>&1 | grep -E "passed|failed|^E " | head -5
