#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_k4_gpu.py tests/test_dyn_prep_gpu.py tests/test_frontend_gpu.py tests/test_arrow_gpu.py tests/test_nulls_gpu.py -m gpu -q --maxfail=15 --tb=short > gpurun_out/g_tests.log 2>&1
tail -60 gpurun_out/g_tests.log
for c in cfg4 cfg4r; do python bench.py --config $c --no-cpu-baseline > gpurun_out/g_bench_$c.json 2> gpurun_out/g_bench_$c.err; grep -o '"ms_per_step": [0-9.]*' gpurun_out/g_bench_$c.json; done
