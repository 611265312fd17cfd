#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 200 python scripts/dbg_cliffs_first.py a 2>&1 | grep -v amdgpu.ids
timeout 200 python scripts/dbg_cliffs_first.py b 2>&1 | grep -v amdgpu.ids
ONLY="1 x 10M" timeout 200 python scripts/bench_shape_cliffs.py 2>&1 | grep "TB/s"
ONLY="1" timeout 200 python scripts/bench_shape_cliffs.py 2>&1 | grep "TB/s" | head -3
