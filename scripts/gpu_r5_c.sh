#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r5c
timeout 1500 python -m pytest tests/test_k3_gpu.py tests/test_k4_gpu.py tests/test_k6_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^    \|^$" | tail -30 | cut -c1-300
KS=12,16,32 timeout 600 python scripts/bench_dyn_edges.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5c/bench_dyn_edges.txt
