#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/r5f; O=$R/gpurun_out/r5f
cd /tmp && export TMPDIR=/tmp
for only in "1 x 10M" "100 x 100k" "1k x 10k"; do
  tag=$(echo $only | tr -d ' ')
  rm -rf $O/kt_$tag
  ONLY="$only" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$tag -o k -- python $R/scripts/bench_shape_cliffs.py 2>&1 | grep -v amdgpu.ids | grep "TB/s"
  f=$(find $O/kt_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_$tag.csv && head -12 $O/kernel_stats_$tag.csv | cut -c1-200
  rm -rf $O/kt_$tag
done
