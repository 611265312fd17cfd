#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests/test_k2_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^    \|^$" | tail -12 | cut -c1-300
