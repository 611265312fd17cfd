#!/bin/bash
# Round 3, call e: K2w (two-tile MFMA resident kernel for 17..31 columns): parity, phase timeline, rates.
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_k2_gpu.py -m gpu -q --maxfail=20 --tb=short -p no:cacheprovider -k "k2w" > gpurun_out/r3e_tests.log 2>&1; echo "pytest exit $?"
grep -E "passed|failed" gpurun_out/r3e_tests.log | tail -3
grep -E "^(FAILED|ERROR)|^E  " gpurun_out/r3e_tests.log | head -30 | cut -c1-300
timeout 300 python scripts/dbg_timeline_k2w.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3e_timeline.txt
echo "== ENGINE=k2w"; KS=20,31 ENGINE=k2w timeout 300 python scripts/bench_k16.py 2>&1 | grep -v amdgpu.ids | grep x1000 | tee gpurun_out/r3e_k16_k2w.txt
