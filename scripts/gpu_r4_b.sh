#!/bin/bash
# Round 4, visit b: K3c with tagged-granule records + workgroup-wide look-back windows; timelines of K3c and K2 (cfg5).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/r4b; O=$R/gpurun_out/r4b
line() { python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
d=json.loads(t[-1]); r=d['roofline']
print('$1', 'value=%.4g'%d['value'], 'ms/step=%.4f'%d['ms_per_step'], r['kernel'], 'kernel_ms=%.4f n=%s'%(r['kernel_ms'], r.get('kernel_samples')), 'GB/s=%.0f frac=%.3f'%(r['achieved'], r['frac']))"; }
echo "== tests (RLS family)"
timeout 900 python -m pytest tests/test_k3_gpu.py tests/test_dyn_prep_gpu.py -m gpu -x -q 2>&1 | tail -8 | cut -c1-300
echo "== cfg4 / rlsg"
timeout 300 python bench.py --config cfg4 --steps 50 --warmup 10 --no-cpu-baseline 2>$O/cfg4.err | tee $O/cfg4.json | line cfg4
timeout 300 python bench.py --config rlsg --steps 20 --warmup 5 --no-cpu-baseline 2>$O/rlsg.err | tee $O/rlsg.json | line rlsg
echo "== timelines"
POLS_TIMELINE=1 timeout 300 python bench.py --config cfg4 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep timeline | tail -2 | cut -c1-400
POLS_TIMELINE=1 timeout 300 python bench.py --config rlsg --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep timeline | tail -1 | cut -c1-400
POLS_TIMELINE=1 timeout 300 python bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep timeline | tail -1 | cut -c1-400
tail -3 $O/*.err | cut -c1-300 | grep -v "^$" | head -20
