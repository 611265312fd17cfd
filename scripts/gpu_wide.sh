#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
echo "== wide tests"
timeout 900 python -m pytest tests/test_k3_gpu.py tests/test_k4_gpu.py -q -x -k "wide or non_contiguous" 2>&1 | tail -40 | cut -c1-400 | tee gpurun_out/wide.log
echo "== full gpu suite"
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -8 | cut -c1-300 | tee gpurun_out/pytest_gpu.log
echo "== bench, collective path forced on one GPU"
POLS_BENCH_FORCE_COLLECTIVE=1 timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2> gpurun_out/bench_coll.err | cut -c1-330
