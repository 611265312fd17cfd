#!/bin/bash
# the last pass of round 5: the whole GPU suite, smoke, fuzz, the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/r5w; O=$R/gpurun_out/r5w
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^E |FAILED" | head -8 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
FUZZ_SPREAD=16 timeout 1500 python scripts/fuzz_gpu.py 21 2>&1 | grep -v amdgpu.ids | grep -E "MISMATCH|ERROR|fuzz done|size-class" | tee $O/fuzz.txt
timeout 300 python bench.py 2>/dev/null | tee $O/bench.json | cut -c1-200
