#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
FUZZ_SHORT=0 FUZZ_DYN2=0 FUZZ_DYN3=0 FUZZ_SPREAD=40 timeout 1200 python scripts/fuzz_gpu.py 2>&1 | grep -v amdgpu.ids | tail -12
