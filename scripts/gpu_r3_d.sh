#!/bin/bash
# Round 3, call d: sharded entry, bench (rotation, --gather pred), headline A/B baseline.
mkdir -p gpurun_out
python -m pytest tests/test_sharded_gpu.py tests/test_bench_gpu.py tests/test_arrow_plugins_gpu.py -m gpu -q --maxfail=30 --tb=short -p no:cacheprovider > gpurun_out/r3d_tests.log 2>&1; echo "pytest exit $?"
grep -E "passed|failed" gpurun_out/r3d_tests.log | tail -3
grep -E "^(FAILED|ERROR)|^E  " gpurun_out/r3d_tests.log | head -40 | cut -c1-300
for f in 1 3; do python bench.py --no-cpu-baseline --frames $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('frames',d['config']['frames_rotated'],'ms_per_step',round(d['ms_per_step'],5),'kernel_ms',round(d['roofline']['kernel_ms'],5),'frac',round(d['roofline']['frac'],4), d['roofline']['kernel'])"; done
python bench.py --no-cpu-baseline --config cfg3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3 frames',d['config']['frames_rotated'],'ms_per_step',round(d['ms_per_step'],5),'kernel_ms',round(d['roofline']['kernel_ms'],5),'frac',round(d['roofline']['frac'],4), d['roofline']['kernel'])"
ONLY=default,team256_rc1_p2_nt,team256_rc1_p3_nt,team256_rc1_p4_nt python scripts/ab_headline.py 2>&1 | grep -v amdgpu.ids | tail -12
