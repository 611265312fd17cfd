#!/bin/bash
# Round 3, call p: the stream-ceiling probe behind bench.py (cfg2, cfg3, cfg5), its tests, the predict-policy tests.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/p; O=$R/gpurun_out/p
python -m pytest tests/test_bench_gpu.py tests/test_predict_policy_gpu.py tests/test_frontend_gpu.py -m gpu -q --maxfail=10 --tb=short -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest exit $?"
grep -E "passed|failed" $O/tests.log | tail -2
grep -E "^(FAILED|ERROR)|^E  " $O/tests.log | head -20 | cut -c1-250
for cfg in cfg2 cfg3 cfg5; do
  timeout 300 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$cfg.json
  python3 -c "
import json,sys; d=json.load(open('$O/bench_$cfg.json')); r=d['roofline']; print('$cfg', d['ms_per_step'], r['kernel'], round(r['kernel_ms'],4), round(r['achieved']), round(r['frac'],3), {k:(round(v,3) if isinstance(v,float) else v) for k,v in r.get('stream_ceiling',{}).items() if k!='what'})"
done
