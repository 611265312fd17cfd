#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
python scripts/dbg_k4p_long.py 2>&1 | grep -v amdgpu.ids
G=10000 N=1000 python scripts/dbg_k4p_long.py 2>&1 | grep -v amdgpu.ids
G=10 N=1000000 python scripts/dbg_k4p_long.py 2>&1 | grep -v amdgpu.ids
timeout 1200 python -m pytest tests/test_k4_gpu.py tests/test_k3_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^    \|^$" | tail -5 | cut -c1-300
