#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests/test_k1_gpu.py tests/test_k5_gpu.py tests/test_routing_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED" | head -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
