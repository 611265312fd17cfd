#!/bin/bash
# Round 3, call g: what the bench's own event sampling costs the step time (stride 4 / 8 / 16).
for st in 4 8 16; do POLS_BENCH_EVENT_STRIDE=$st python bench.py --no-cpu-baseline --steps 48 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stride $st steps 48 ms_per_step',round(d['ms_per_step'],5),'kernel_ms',round(d['roofline']['kernel_ms'],5))"; done
for st in 4 8; do POLS_BENCH_EVENT_STRIDE=$st python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stride $st steps 20 ms_per_step',round(d['ms_per_step'],5),'kernel_ms',round(d['roofline']['kernel_ms'],5))"; done
