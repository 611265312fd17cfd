#!/bin/bash
# Round 3, call r: fuzz with the tiny-group frames (K1t team shapes) on the build with the K1t reduce-scatter / predict policies / stream probe.
mkdir -p gpurun_out
for seed in 81 82 83 84 85 86 87 88 89 90; do timeout 300 python scripts/fuzz_gpu.py $seed 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-400; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
