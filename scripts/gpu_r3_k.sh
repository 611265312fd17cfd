#!/bin/bash
# Round 3, call k: full suite on the final build, then an extended fuzz (12 seeds + 3 whole-chip seeds), smoke, default bench line.
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --maxfail=20 --tb=short -p no:cacheprovider > gpurun_out/r3k_tests.log 2>&1; echo "pytest exit $?"
grep -E "passed|failed" gpurun_out/r3k_tests.log | tail -2
grep -E "^(FAILED|ERROR)|^E  " gpurun_out/r3k_tests.log | head -20 | cut -c1-250
for seed in 61 62 63 64 65 66 67 68 69 70 71 72; do timeout 300 python scripts/fuzz_gpu.py $seed 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-400; done
for seed in 73 74 75; do timeout 500 python scripts/fuzz_gpu.py $seed big 2>&1 | grep -v amdgpu.ids | grep -v "big frames ran" | tail -4 | cut -c1-400; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 2>/dev/null | cut -c1-600
timeout 200 python scripts/bench_ragged.py 2>/dev/null | tail -1 > gpurun_out/r03_bench_ragged.json; cut -c1-2600 gpurun_out/r03_bench_ragged.json
