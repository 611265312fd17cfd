#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r5t
timeout 600 python scripts/ab_wide_f64.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5t/ab_wide_f64.txt
