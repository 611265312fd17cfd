#!/bin/bash
# the last pass of round 5: the whole GPU suite, smoke, fuzz with two seeds, the group-length x columns sweep on the final tree
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/r5w; O=$R/gpurun_out/r5w
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^E |FAILED" | head -8 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
FUZZ_SPREAD=24 timeout 1500 python scripts/fuzz_gpu.py 11 2>&1 | grep -v amdgpu.ids | grep -E "MISMATCH|ERROR|fuzz done|size-class" | tee $O/fuzz.txt
FUZZ_SPREAD=24 timeout 1500 python scripts/fuzz_gpu.py 12 2>&1 | grep -v amdgpu.ids | grep -E "MISMATCH|ERROR|fuzz done|size-class" | tee -a $O/fuzz.txt
sleep 2
timeout 900 python scripts/bench_rows_sweep.py 2>&1 | grep -v amdgpu.ids | tee $O/bench_rows_sweep.txt
timeout 300 python bench.py 2>/dev/null | cut -c1-300
