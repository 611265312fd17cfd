#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_k1_gpu.py -m gpu -q --maxfail=10 --tb=short -k "persistent" > gpurun_out/i_tests.log 2>&1
tail -30 gpurun_out/i_tests.log
POLS_K1_PERSIST=0 python scripts/bench_ragged.py > gpurun_out/i_ragged_off.json 2>gpurun_out/i_ragged_off.err
for v in 16 32 64; do POLS_K1_PERSIST=1 POLS_K1_PERSIST_SUB=$v python scripts/bench_ragged.py > gpurun_out/i_ragged_$v.json 2>gpurun_out/i_ragged_$v.err; done
python - <<'PY'
import json
a=json.load(open("gpurun_out/i_ragged_off.json"))
bs={v:json.load(open(f"gpurun_out/i_ragged_{v}.json")) for v in (16,32,64)}
for k in a:
    line=f"{k:20s} off {a[k]['us']:7.1f}us {a[k]['TBps']:5.2f} {a[k]['kernel'][:34]:34s}"
    for v,b in bs.items(): line+=f" | {v}: {b[k]['us']:7.1f}us {b[k]['TBps']:5.2f} {b[k]['kernel'][-12:] if b[k]['kernel'].startswith('k1p') else '-'}"
    print(line)
PY
