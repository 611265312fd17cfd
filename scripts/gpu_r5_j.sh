#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/r5j; O=$R/gpurun_out/r5j
{
echo "== default routing (K2 rc4 up to 8 columns)"
ROWS=2500,3000,4000,5000,6000,8000 COLS=2,4,6,8 timeout 600 python scripts/bench_rows_sweep.py 2>&1 | grep -v amdgpu.ids
echo "== POLS_STATIC_ENGINE=stream"
POLS_STATIC_ENGINE=stream ROWS=2500,3000,4000,5000,6000,8000 COLS=2,4,6,8 timeout 600 python scripts/bench_rows_sweep.py 2>&1 | grep -v amdgpu.ids
} | tee $O/sweep_k2rc4_ab.txt
timeout 900 python -m pytest tests/test_k2_gpu.py -m gpu -x -q 2>&1 | tail -5
