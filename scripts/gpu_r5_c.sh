#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
ONLY="1 x 10M" timeout 300 python scripts/bench_shape_cliffs.py 2>&1 | grep -v amdgpu.ids
ONLY="1 x 10M" timeout 300 python scripts/bench_shape_cliffs.py 2>&1 | grep -v amdgpu.ids
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt; ONLY="1 x 10M" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o k -- python $R/scripts/bench_shape_cliffs.py > /dev/null 2>&1
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-200
