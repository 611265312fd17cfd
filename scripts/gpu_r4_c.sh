#!/bin/bash
# Round 4, visit c: K3c (two-level look-back) + K4c (rolling tiles): parity, bench lines, timelines.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/r4c; O=$R/gpurun_out/r4c
line() { python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
d=json.loads(t[-1]); r=d['roofline']
print('$1', 'value=%.4g'%d['value'], 'ms/step=%.4f'%d['ms_per_step'], r['kernel'], 'kernel_ms=%.4f n=%s'%(r['kernel_ms'], r.get('kernel_samples')), 'GB/s=%.0f frac=%.3f'%(r['achieved'], r['frac']))"; }
echo "== tests (dynamic family)"
timeout 1200 python -m pytest tests/test_k3_gpu.py tests/test_k4_gpu.py tests/test_dyn_prep_gpu.py -m gpu -q 2>&1 | tail -25 | cut -c1-400
echo "== bench"
for c in cfg4 cfg4r rlsg rlsgr; do
  timeout 300 python bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline 2>$O/$c.err | tee $O/$c.json | line $c
done
echo "== timelines"
POLS_TIMELINE=1 timeout 300 python bench.py --config cfg4 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep timeline | tail -1 | cut -c1-400
POLS_TIMELINE=1 timeout 300 python bench.py --config rlsg --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep timeline | tail -1 | cut -c1-400
for f in $O/*.err; do echo "--- $f"; tail -n 3 $f | cut -c1-300; done 2>/dev/null | grep -v "^$" | head -30
