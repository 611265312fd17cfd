#!/bin/bash
# Round 3, call c: K8 fix modes (QR basic / Cholesky->LU beyond 31 columns) + the shared fix-up solvers in K6.
mkdir -p gpurun_out
python -m pytest tests/test_k8_gpu.py tests/test_k6_gpu.py tests/test_frontend_gpu.py tests/test_k7_gpu.py -m gpu -q --maxfail=30 --tb=short -p no:cacheprovider > gpurun_out/r3c_tests.log 2>&1; echo "pytest exit $?"
grep -E "passed|failed" gpurun_out/r3c_tests.log | tail -3
grep -E "^(FAILED|ERROR)|^E  " gpurun_out/r3c_tests.log | head -60 | cut -c1-300
