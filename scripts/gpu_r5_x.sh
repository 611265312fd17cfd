#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_k5_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert|^E " | head -8
