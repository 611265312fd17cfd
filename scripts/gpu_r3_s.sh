#!/bin/bash
# Round 3, call s: K2 full-width fast path (no per-column wave-uniform branches at 16 user columns), K2w compile-time guards: parity + benches.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/s; O=$R/gpurun_out/s
python -m pytest tests/test_k2_gpu.py tests/test_k5_gpu.py tests/test_k6_gpu.py tests/test_bench_gpu.py tests/test_routing_gpu.py -m gpu -q --maxfail=10 --tb=short -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest exit $?"
grep -E "passed|failed" $O/tests.log | tail -2
grep -E "^(FAILED|ERROR)|^E  " $O/tests.log | head -20 | cut -c1-250
timeout 300 python bench.py --config cfg5 --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_cfg5.json; python3 -c "
import json; d=json.load(open('$O/bench_cfg5.json')); r=d['roofline']; print('cfg5', round(d['ms_per_step'],4), r['kernel'], round(r['achieved']), round(r['frac'],3), round(r['stream_ceiling']['achieved_over_ceiling'],3))"
timeout 300 python scripts/dbg_timeline_k2w.py 2>&1 | grep -v amdgpu.ids | grep timeline | cut -c1-200
ENGINE= KS=16,17,20,24,25,28,31 timeout 300 python scripts/bench_k16.py 2>/dev/null | grep -v amdgpu | grep -E "k=(16|17|20|24|25|28|31)"
