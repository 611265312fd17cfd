#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r5c; rm -f gpurun_out/r5c/bench_dyn_nulls.txt
timeout 1500 python -m pytest tests/test_k3_gpu.py tests/test_k4_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^    \|^$" | tail -30 | cut -c1-300
for K in 6 12 32; do K=$K timeout 600 python scripts/bench_dyn_nulls.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r5c/bench_dyn_nulls.txt; done
