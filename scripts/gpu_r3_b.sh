#!/bin/bash
# Round 3, call b: K6 modes after the Cholesky-noise threshold, the new Arrow entries, full suite.
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --maxfail=60 --tb=short -p no:cacheprovider > gpurun_out/r3b_tests.log 2>&1; echo "pytest exit $?"
grep -E "passed|failed" gpurun_out/r3b_tests.log | tail -3
grep -E "^(FAILED|ERROR)" gpurun_out/r3b_tests.log | head -60 | cut -c1-250
