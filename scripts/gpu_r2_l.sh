#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_k1_gpu.py tests/test_k2_gpu.py tests/test_nulls_gpu.py tests/test_k7_gpu.py tests/test_frontend_gpu.py -m gpu -q --maxfail=10 --tb=short > gpurun_out/l_tests.log 2>&1
tail -25 gpurun_out/l_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
python scripts/bench_k9.py 2>&1 | grep -v amdgpu.ids | cut -c1-150
