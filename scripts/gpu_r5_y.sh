#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r5y
sleep 2
KS=6,10,12,16,32 timeout 600 python scripts/bench_dyn_edges.py 2>/dev/null | grep -v amdgpu | tee gpurun_out/r5y/bench_dyn_edges.txt
for K in 6 12; do K=$K timeout 600 python scripts/bench_dyn_nulls.py 2>/dev/null | grep -v amdgpu; done | tee gpurun_out/r5y/bench_dyn_nulls.txt
