#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_k1_gpu.py -m gpu -q --maxfail=10 --tb=short -k "persistent" > gpurun_out/j_tests.log 2>&1
tail -12 gpurun_out/j_tests.log
python scripts/dbg_timeline_mid.py 2>&1 | grep -v amdgpu.ids | tail -40
