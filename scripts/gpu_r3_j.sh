#!/bin/bash
# Round 3, call j: after the SVD cut-off change (eps * max(n, k) per group) -- full suite, fuzz.
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --maxfail=20 --tb=short -p no:cacheprovider > gpurun_out/r3j_tests.log 2>&1; echo "pytest exit $?"
grep -E "passed|failed" gpurun_out/r3j_tests.log | tail -2
grep -E "^(FAILED|ERROR)|^E  " gpurun_out/r3j_tests.log | head -20 | cut -c1-250
for seed in 51 52 55 56; do timeout 400 python scripts/fuzz_gpu.py $seed 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-400; done
