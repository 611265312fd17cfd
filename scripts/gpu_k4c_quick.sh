#!/bin/bash
# K4c quick loop: rolling parity + the two rolling bench lines (+ timelines with TL=1)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
line() { python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
d=json.loads(t[-1]); r=d['roofline']
print('$1', 'value=%.4g'%d['value'], 'ms/step=%.4f'%d['ms_per_step'], r['kernel'], 'kernel_ms=%.4f n=%s'%(r['kernel_ms'], r.get('kernel_samples')), 'GB/s=%.0f frac=%.3f'%(r['achieved'], r['frac']))"; }
timeout 900 python -m pytest tests/test_k4_gpu.py -m gpu -q -x 2>&1 | tail -4 | cut -c1-300
for c in ${CFGS:-cfg4r rlsgr}; do
  timeout 300 python bench.py --config $c --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | line $c
  [ -n "$TL" ] && POLS_TIMELINE=1 timeout 300 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep timeline | tail -1 | cut -c1-300
done
