#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r5c; rm -f gpurun_out/r5c/*
timeout 1500 python -m pytest tests/test_k3_gpu.py tests/test_k4_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^    \|^$" | tail -10 | cut -c1-300
KS=12,16,24,32 timeout 600 python scripts/bench_dyn_edges.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r5c/bench_dyn_edges.txt
POLS_K4P_LPS=32 KS=32 timeout 600 python scripts/bench_dyn_edges.py 2>&1 | grep -v amdgpu.ids | grep rls
POLS_K4P_LPS=64 KS=12 timeout 600 python scripts/bench_dyn_edges.py 2>&1 | grep -v amdgpu.ids | grep rls
