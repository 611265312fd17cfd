#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_k7_gpu.py tests/test_dyn_prep_gpu.py tests/test_frontend_gpu.py tests/test_nulls_gpu.py -m gpu -q --maxfail=15 --tb=short > gpurun_out/h_tests.log 2>&1
tail -60 gpurun_out/h_tests.log
