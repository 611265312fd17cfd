#!/bin/bash
# Null policies at 11-15 columns in the register-resident kernels: parity, bench next to the streamed path, fuzz.
mkdir -p gpurun_out
python -m pytest tests/test_nulls_gpu.py tests/test_k7_gpu.py tests/test_k8_gpu.py tests/test_frontend_gpu.py -m gpu -q --maxfail=10 --tb=short > gpurun_out/p_tests.log 2>&1; echo "pytest exit $?"
tail -25 gpurun_out/p_tests.log | cut -c1-400
timeout 300 python scripts/bench_nulls_wide.py 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/p_bench_nulls_wide.json
for seed in 31 32; do timeout 400 python scripts/fuzz_gpu.py $seed big 2>&1 | grep -v amdgpu.ids | grep -v "big frames ran" | tail -6 | cut -c1-600; done
