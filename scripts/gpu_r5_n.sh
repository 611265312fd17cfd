#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r5n
timeout 600 python scripts/tl_wide.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5n/tl_wide.txt
