#!/bin/bash
# Whole-chip differential fuzz (thousands of groups per frame) on three seeds.
mkdir -p gpurun_out
for seed in 21 22 23; do s=$(date +%s); timeout 500 python scripts/fuzz_gpu.py $seed big 2>&1 | grep -v amdgpu.ids | tail -12 | cut -c1-1500; echo "seed $seed: $(( $(date +%s) - s )) s"; done | tee gpurun_out/o_fuzz.log
