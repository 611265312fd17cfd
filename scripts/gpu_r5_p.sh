#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_k1_gpu.py -m gpu -q -k "size_classes" 2>&1 | grep -E "Error|error|assert|FAILED|passed|failed|kernel" | head -30
