#!/bin/bash
# Round 3, call h: full suite after the pruning (general multi-pass K1 forms gone, tiny frames -> fix-up) + routing audit; bench event stride.
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --maxfail=40 --tb=short -p no:cacheprovider > gpurun_out/r3h_tests.log 2>&1; echo "pytest exit $?"
grep -E "passed|failed" gpurun_out/r3h_tests.log | tail -3
grep -E "^(FAILED|ERROR)|^E  " gpurun_out/r3h_tests.log | head -40 | cut -c1-300

