#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r5y
sleep 2
timeout 600 python scripts/bench_spread.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5y/bench_spread_alpha05.txt
