#!/bin/bash
# round 2, visit f: dynamic entries on raw columns (dyn_prep) + the front-end / dynamic suites + cfg4 / cfg4r bench lines
mkdir -p gpurun_out
python -m pytest tests/test_dyn_prep_gpu.py tests/test_frontend_gpu.py tests/test_k3_gpu.py tests/test_k4_gpu.py -m gpu -q --maxfail=15 --tb=short > gpurun_out/f_tests.log 2>&1
tail -40 gpurun_out/f_tests.log
for c in cfg4 cfg4r; do python bench.py --config $c --no-cpu-baseline > gpurun_out/f_bench_$c.json 2> gpurun_out/f_bench_$c.err; cat gpurun_out/f_bench_$c.json; done
