#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/profile
for cfg in cfg3 cfg4 cfg5; do
  steps=20; [ $cfg = cfg5 ] && steps=5; [ $cfg = cfg4 ] && steps=5
  timeout 900 python bench.py --config $cfg --steps $steps --warmup 2 --no-cpu-baseline 2> gpurun_out/profile/$cfg.err > gpurun_out/profile/r01_bench_$cfg.json
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/profile/r01_bench_$cfg.json")); r=d["roofline"]
    print("$cfg", "value=%.4e %s"%(d["value"], d["unit"]), "ms/step=%.3f"%d["ms_per_step"], r["kernel"], "kernel_ms=%.3f"%r["kernel_ms"], "GB/s=%.0f"%r["achieved"])
except Exception as e:
    print("$cfg failed", e); print(open("gpurun_out/profile/$cfg.err").read()[-1500:])
PY
done
