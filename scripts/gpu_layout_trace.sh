#!/bin/bash
# kernel trace of scripts/bench_layout.py (K9 ingestion): which kernels the 0.8 ms of pols_layout_create is made of
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/layout; mkdir -p $O; rm -rf $O/kt
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o l -- python $R/scripts/bench_layout.py > $O/bench_layout.json 2> $O/kt.err
f=$(find $O/kt -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $O/layout_kernel_stats.csv && cut -d, -f1-4,8 $f | head -30
tail -1 $O/bench_layout.json | cut -c1-300
