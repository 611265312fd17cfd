#!/bin/bash
# round-5 closing pass: the whole GPU suite, smoke, the default bench line, and the shape sweeps on the final kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/r5k; O=$R/gpurun_out/r5k
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 600 python bench.py 2>/dev/null | tee $O/bench.json | cut -c1-600
SHORT=1 timeout 900 python scripts/bench_shape_cliffs.py 2>&1 | grep "TB/s\|ERROR" | tee $O/bench_shape_cliffs.txt
timeout 900 python scripts/bench_rows_sweep.py 2>&1 | grep -v amdgpu.ids | tee $O/bench_rows_sweep.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kt; ONLY="1 x 10M" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python $R/scripts/bench_shape_cliffs.py > /dev/null 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep "Name\|pols" "$f" > $O/kernel_stats_long_groups.csv; rm -rf $O/kt
cat $O/kernel_stats_long_groups.csv | cut -c1-160
