#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprofv3 kernel trace.  Outputs under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
R=$PWD
echo "== rocminfo" ; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -8 | tee gpurun_out/smoke.log
echo "== bench"
timeout 600 python bench.py --steps 50 --warmup 10 2> gpurun_out/bench.err | tee gpurun_out/bench.json
tail -5 gpurun_out/bench.err
echo "== rocprofv3 kernel trace"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o k1 -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
cat $R/gpurun_out/prof_bench.json
find $R/gpurun_out/prof -name "*kernel_stats*" | head -3
f=$(find $R/gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f"
