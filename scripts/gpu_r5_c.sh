#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
KS=12,32 timeout 600 python scripts/bench_dyn_edges.py 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_k4_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^    \|^$" | tail -8 | cut -c1-300
