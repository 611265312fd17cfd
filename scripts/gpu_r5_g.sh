#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/r5g; O=$R/gpurun_out/r5g
run() { echo "== $*"; env "$@" ONLY="$SH" timeout 300 python scripts/bench_shape_cliffs.py 2>&1 | grep "TB/s"; }
{
SH="1 x 10M"
run POLS_KG_SINGLE_BUFFER=1 POLS_PREDICT_LOOP=1
run A=1
for st in 2048 8192; do run POLS_SEG_TARGET=$st; done
SH="1k x 10k"
run POLS_KG_SINGLE_BUFFER=1 POLS_PREDICT_LOOP=1
run A=1
SH="100 x 100k"
run A=1
SH="4k x 2.5k"
run A=1
} | tee $O/long_ab.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kt; ONLY="1 x 10M" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python $R/scripts/bench_shape_cliffs.py 2>&1 | grep "TB/s"
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_1x10M.csv && grep pols $O/kernel_stats_1x10M.csv | cut -c1-180
rm -rf $O/kt
cd $R; timeout 900 python -m pytest tests/test_k5_gpu.py tests/test_routing_gpu.py tests/test_nulls_gpu.py tests/test_predict_policy_gpu.py tests/test_k7_gpu.py -m gpu -x -q 2>&1 | tail -5
