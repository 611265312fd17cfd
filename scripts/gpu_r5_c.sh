#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
ONLY="4k" timeout 300 python scripts/bench_shape_cliffs.py 2>&1 | grep -v amdgpu.ids
timeout 1500 python -m pytest tests/test_k1_gpu.py tests/test_k2_gpu.py tests/test_routing_gpu.py tests/test_frontend_gpu.py tests/test_nulls_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^    \|^$" | tail -8 | cut -c1-300
