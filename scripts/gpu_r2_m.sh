#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_k1_gpu.py -m gpu -q --maxfail=10 --tb=short -k "sixteen" > gpurun_out/m_tests.log 2>&1
tail -25 gpurun_out/m_tests.log
python scripts/bench_k16.py 2>&1 | grep -v amdgpu.ids
