#!/bin/bash
# statistics / multi-target tests first (fast feedback), then the whole GPU suite, then the N > 1 bench code on one GPU.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
echo "== k7 tests"
timeout 600 python -m pytest tests/test_k7_gpu.py -q 2>&1 | tail -40 | cut -c1-400 | tee gpurun_out/k7.log
echo "== full gpu suite"
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -8 | cut -c1-300 | tee gpurun_out/pytest_gpu.log
echo "== bench, collective path forced on one GPU"
POLS_BENCH_FORCE_COLLECTIVE=1 timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2> gpurun_out/bench_coll.err | cut -c1-900
tail -3 gpurun_out/bench_coll.err | cut -c1-300
echo "== bench default"
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2> gpurun_out/bench.err | cut -c1-900
