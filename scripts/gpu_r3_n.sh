#!/bin/bash
# Round 3, call n: full GPU suite on the build with the K1t reduce-scatter / eight-lane teams / leaner edge loads, then the side benches.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/n; O=$R/gpurun_out/n
python -m pytest tests -m gpu -q --maxfail=30 --tb=short -p no:cacheprovider -x > $O/tests.log 2>&1; echo "pytest exit $?"
grep -E "passed|failed" $O/tests.log | tail -2
grep -E "^(FAILED|ERROR)|^E  " $O/tests.log | head -20 | cut -c1-250
timeout 400 python scripts/bench_tiny.py 2>&1 | grep -v amdgpu.ids | tail -1 > $O/r03_bench_tiny.json; cut -c1-3000 $O/r03_bench_tiny.json
timeout 200 python scripts/bench_ragged.py 2>/dev/null | tail -1 > $O/r03_bench_ragged.json; cut -c1-2500 $O/r03_bench_ragged.json; echo
timeout 120 python scripts/bench_nulls.py 2>/dev/null | tail -1 > $O/r03_bench_nulls.json; cat $O/r03_bench_nulls.json; echo
timeout 200 python scripts/bench_k9.py 2>/dev/null | grep -v amdgpu > $O/r03_bench_k9.txt; cut -c1-160 $O/r03_bench_k9.txt
timeout 200 python scripts/bench_k16.py 2>/dev/null | grep -v amdgpu > $O/r03_bench_k16.txt; cat $O/r03_bench_k16.txt
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/r03_bench_quick.json; cut -c1-400 $O/r03_bench_quick.json
