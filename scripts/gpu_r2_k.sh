#!/bin/bash
# full GPU suite + default-rule ragged bench
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --maxfail=20 --tb=short -x -n 0 > gpurun_out/k_tests.log 2>&1
tail -25 gpurun_out/k_tests.log
python scripts/bench_ragged.py > gpurun_out/k_ragged.json 2>gpurun_out/k_ragged.err
python - <<'PY'
import json
a=json.load(open("gpurun_out/k_ragged.json"))
for k,v in a.items(): print(f"{k:22s} {v['us']:8.1f} us {v['TBps']:5.2f} TB/s {v['kernel']}")
PY
python bench.py > gpurun_out/k_bench.json 2> gpurun_out/k_bench.err; cat gpurun_out/k_bench.json
