#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_k6_gpu.py tests/test_k8_gpu.py tests/test_frontend_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^    \|^$" | tail -12 | cut -c1-300
for seed in 101 105 106; do echo "== fuzz seed $seed"; FUZZ_DYN2=20 timeout 900 python scripts/fuzz_gpu.py $seed 2>&1 | grep -v amdgpu.ids | grep -v "cases ran" | tail -6 | cut -c1-300; done
