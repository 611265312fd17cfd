#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for dt in f32 f64; do
POLS_TIMELINE=1 POLS_K1_ENGINE=mfma timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --dtype $dt 2>&1 | grep -E "timeline" | tail -2
done
