#!/bin/bash
# Round-2 visit A: full GPU suite (every failure listed), cfg2 / cfg5 bench lines for the K2 kernel against the three-launch path.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r2a; O=gpurun_out/r2a
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 --tb=line --deselect tests/test_k5_gpu.py::test_cfg5_full_size 2>&1 | tail -60 | cut -c1-330 | tee $O/pytest_gpu.log
echo "== timeline cfg5 K2"
POLS_TIMELINE=1 timeout 300 python bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep timeline | tail -2 | tee $O/timeline_cfg5.txt
echo "== bench cfg5: K2 vs stream"
timeout 300 python bench.py --config cfg5 --steps 10 --warmup 3 --no-cpu-baseline 2>$O/cfg5.err | tee $O/bench_cfg5_k2.json | cut -c1-200
POLS_STATIC_ENGINE=stream timeout 300 python bench.py --config cfg5 --steps 10 --warmup 3 --no-cpu-baseline 2>>$O/cfg5.err | tee $O/bench_cfg5_stream.json | cut -c1-200
