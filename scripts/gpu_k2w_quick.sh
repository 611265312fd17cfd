#!/bin/bash
# K2w quick loop: parity of the 9..31-column solvers + the 17..31-column bench lines with / without the LDS prefetch
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1200 python -m pytest tests/test_k2_gpu.py -m gpu -q -x 2>&1 | tail -4 | cut -c1-300
echo "== prefetch on"
KS=${KS:-20,24,31} timeout 600 python scripts/bench_k16.py 2>&1 | grep "us/call" | grep -E "${ONLY:-f64}"
echo "== prefetch off"
POLS_K2_NOPREFETCH=1 KS=${KS:-20,24,31} timeout 600 python scripts/bench_k16.py 2>&1 | grep "us/call" | grep -E "${ONLY:-f64}"
if [ -n "$TL" ]; then
  KS=${KS:-20,24,31} POLS_TIMELINE=1 timeout 500 python scripts/bench_k16.py 2>&1 | grep -E "timeline|us/call" | awk '/timeline/{last=$0} /us.call/{print last; print $0}' | grep -A1 "k2w.*f64" | cut -c1-300
fi
