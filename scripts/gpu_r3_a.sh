#!/bin/bash
# Round 3, call a: rank-deficient parity (K6 modes) + notebook KATs + tightened tolerances -- full suite, every failure listed.
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --maxfail=60 --tb=line -p no:cacheprovider > gpurun_out/r3a_tests.log 2>&1; echo "pytest exit $?"
grep -E "passed|failed" gpurun_out/r3a_tests.log | tail -3
grep -E "^/|Error|^FAILED" gpurun_out/r3a_tests.log | head -70 | cut -c1-330
