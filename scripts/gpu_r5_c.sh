#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r5c
timeout 1500 python scripts/bench_rows_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5c/bench_rows_sweep.txt
