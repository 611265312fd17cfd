#!/bin/bash
# Round-end sanity: the full GPU suite, smoke(), the default bench line, two more fuzz seeds.
mkdir -p gpurun_out
s=$(date +%s); python -m pytest tests -m gpu -q --maxfail=10 --tb=short > gpurun_out/n_tests.log 2>&1; echo "pytest exit $? in $(( $(date +%s) - s )) s"
tail -8 gpurun_out/n_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
python bench.py 2>/dev/null | tee gpurun_out/n_bench.json | cut -c1-900
for seed in 11 12; do timeout 400 python scripts/fuzz_gpu.py $seed 2>&1 | grep -v amdgpu.ids | tail -4; done
