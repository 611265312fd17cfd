#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && rm -rf $R/gpurun_out/kt5 && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt5 -o k5 -- python $R/bench.py --config cfg5 --steps 8 --warmup 2 > /dev/null 2> $R/gpurun_out/kt5.err; f=$(find $R/gpurun_out/kt5 -name "*kernel_stats.csv" | head -1); head -3 $f | cut -c1-150
