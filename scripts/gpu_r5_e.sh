#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r5e
hipcc --offload-arch=gfx950 -O3 scripts/probe_reread.hip -o gpurun_out/r5e/probe_reread 2>&1 | tail -3
timeout 300 gpurun_out/r5e/probe_reread 2>&1 | tee gpurun_out/r5e/probe_reread.txt
