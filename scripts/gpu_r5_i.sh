#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/r5i; O=$R/gpurun_out/r5i
echo "== default"; timeout 600 python scripts/bench_ragged.py 2>/dev/null | tee $O/bench_ragged.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d.items(): print(f'{k:22s} {v[\"us\"]:8.1f} us {v[\"TBps\"]:5.2f} TB/s  {v[\"kernel\"]}')"
echo "== POLS_K1T_SUB8=1 (no two/three-chunk eight-lane teams)"; POLS_K1T_SUB8=1 timeout 600 python scripts/bench_ragged.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d.items():
    if 'tiny' in k or 'small' in k: print(f'{k:22s} {v[\"us\"]:8.1f} us {v[\"TBps\"]:5.2f} TB/s  {v[\"kernel\"]}')"
timeout 900 python -m pytest tests/test_k1_gpu.py -m gpu -x -q 2>&1 | tail -3
