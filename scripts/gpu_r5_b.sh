#!/bin/bash
# Round 5, visit b: K4p / K3p (11..32 features, wave per chunk, inverse in registers): parity + the 10 000 x 1 000 frame; K6s test rerun.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/r5b; O=$R/gpurun_out/r5b
echo "== K6s test"
timeout 600 python -m pytest tests/test_k6_gpu.py -m gpu -q -x -k "short_groups" 2>&1 | tail -8 | cut -c1-600
echo "== K4p / K3p tests"
timeout 1500 python -m pytest tests/test_k3_gpu.py tests/test_k4_gpu.py -m gpu -q -k "wide or wave_per_chunk or tiles_null_free" 2>&1 | tail -25 | cut -c1-500
echo "== dyn edges"
KS=10,12,16,24,32 timeout 900 python scripts/bench_dyn_edges.py 2>&1 | tee $O/bench_dyn_edges.txt | cut -c1-200
