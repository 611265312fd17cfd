#!/bin/bash
# Null policies at 16-31 columns in the register-resident kernels: full suite, bench next to the streamed path, fuzz.
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --maxfail=10 --tb=short > gpurun_out/q_tests.log 2>&1; echo "pytest exit $?"
grep -E "passed|failed|Error|assert" gpurun_out/q_tests.log | tail -15 | cut -c1-400
timeout 300 python scripts/bench_nulls_wide.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee gpurun_out/q_bench_nulls_wide.json
for seed in 41 42; do timeout 400 python scripts/fuzz_gpu.py $seed big 2>&1 | grep -v amdgpu.ids | grep -v "big frames ran" | tail -6 | cut -c1-600; done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
