#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/r5q; O=$R/gpurun_out/r5q
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
FUZZ_SPREAD=30 timeout 1500 python scripts/fuzz_gpu.py 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/fuzz.txt
SEED=777 FUZZ_SPREAD=30 timeout 1500 python scripts/fuzz_gpu.py 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $O/fuzz.txt
