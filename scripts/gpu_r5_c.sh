#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r5c; rm -f gpurun_out/r5c/*
timeout 1500 python -m pytest tests/test_k3_gpu.py tests/test_k7_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^    \|^$" | tail -10 | cut -c1-300
for i in 1 2 3; do timeout 300 python bench.py --config cfg4 --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 ms_per_step=%.5f kernel_ms=%.5f frac=%.3f'%(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"; done
