#!/bin/bash
# quick iteration loop: correctness of both engines + timeline + A/B bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_k1_gpu.py -m gpu -x -q 2>&1 | tail -4 | cut -c1-300
for dt in f32 f64; do
POLS_TIMELINE=1 POLS_K1_ENGINE=valu timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --dtype $dt 2>&1 | grep -E "timeline" | tail -1 | cut -c1-200
done
for eng in valu; do for dt in f32 f64; do
  POLS_K1_ENGINE=$eng timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --dtype $dt 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$eng $dt', 'value=%.3e'%d['value'], 'ms/step=%.4f'%d['ms_per_step'], r['kernel'], 'kernel_ms=%.4f'%r['kernel_ms'], 'GB/s=%.0f frac=%.3f'%(r['achieved'], r['frac']))"
done; done
POLS_K1_NOFAST=1 POLS_K1_ENGINE=valu timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('valu-nofast f32', 'value=%.3e'%d['value'], r['kernel'], 'kernel_ms=%.4f'%r['kernel_ms'], 'GB/s=%.0f'%r['achieved'])"
