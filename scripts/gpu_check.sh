#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench (A/B of both engines), rocprofv3 kernel trace.  Outputs under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
R=$PWD
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | cut -c1-300 | tee gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -8 | tee gpurun_out/smoke.log
for eng in valu mfma; do
  echo "== bench engine=$eng"
  POLS_K1_ENGINE=$eng timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2> gpurun_out/bench_$eng.err > gpurun_out/bench_$eng.json
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_$eng.json")); r=d["roofline"]
print("$eng", "value=%.3e"%d["value"], "ms/step=%.4f"%d["ms_per_step"], r["kernel"], "kernel_ms=%.4f"%r["kernel_ms"], "GB/s=%.0f frac=%.3f"%(r["achieved"], r["frac"]))
PY
done
echo "== bench default (with cpu baseline)"
timeout 900 python bench.py 2> gpurun_out/bench.err > gpurun_out/bench.json; cut -c1-1500 gpurun_out/bench.json
echo "== rocprofv3 kernel trace"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o k1 -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
f=$(find $R/gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 "$f" | cut -c1-200
