#!/bin/bash
# Round 3, call f: full suite after K2w + the fix_gram diagonal fix; K2w rates at its routed shapes; cfg5 timeline for the overlap work.
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --maxfail=40 --tb=short -p no:cacheprovider > gpurun_out/r3f_tests.log 2>&1; echo "pytest exit $?"
grep -E "passed|failed" gpurun_out/r3f_tests.log | tail -3
grep -E "^(FAILED|ERROR)|^E  " gpurun_out/r3f_tests.log | head -40 | cut -c1-300
echo "== default routing"; KS=20,24,25,31 timeout 300 python scripts/bench_k16.py 2>&1 | grep -v amdgpu.ids | grep x1000 | tee gpurun_out/r3f_k16_default.txt
