#!/bin/bash
# Round 4, visit d: full GPU suite after the dynamic-kernel rewrite + null-weight fill + sharded rendezvous; what the empty fix-up
# dispatch costs on the stream (event sampling off).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/r4d; O=$R/gpurun_out/r4d
echo "== full gpu suite"
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | cut -c1-300
echo "== fix-up dispatch cost (cfg2, 200 steps, no event sampling inside the timed region)"
for i in 1 2 3; do
for skip in 0 1; do
  if [ $skip = 1 ]; then export POLS_DEBUG_SKIP_FIXUP=1; else unset POLS_DEBUG_SKIP_FIXUP; fi
  POLS_BENCH_EVENT_STRIDE=100000 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip_fixup=$skip ms_per_step=%.5f kernel_ms=%.5f'%(d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done
unset POLS_DEBUG_SKIP_FIXUP
