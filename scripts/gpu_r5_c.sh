#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests/test_k1_gpu.py -m gpu -q --tb=short -k "decade" 2>&1 | grep -v "^    \|^$" | tail -12 | cut -c1-300
