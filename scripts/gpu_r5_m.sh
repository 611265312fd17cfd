#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_k6_gpu.py tests/test_k8_gpu.py tests/test_nulls_gpu.py tests/test_frontend_gpu.py -m gpu -q -x 2>&1 | tail -5
FUZZ_SHORT=120 FUZZ_DYN2=0 FUZZ_DYN3=0 timeout 900 python scripts/fuzz_gpu.py 2>&1 | tail -3
