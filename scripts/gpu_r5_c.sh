#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests/test_k4_gpu.py tests/test_k6_gpu.py -m gpu -q -k "divergence or short_groups or fit_wide" --tb=short 2>&1 | grep -v "^    \|^$" | tail -40 | cut -c1-300
