#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_k5_gpu.py tests/test_k7_gpu.py tests/test_nulls_gpu.py tests/test_frontend_gpu.py -q 2>&1 | tail -3
timeout 600 python bench.py --config cfg5 --steps 8 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms/step=%.3f gram_ms=%.3f GB/s=%.0f %s' % (d['ms_per_step'], r['kernel_ms'], r['achieved'], r['kernel']))"
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && rm -rf $R/gpurun_out/kt5 && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt5 -o k5 -- python $R/bench.py --config cfg5 --steps 8 --warmup 2 > /dev/null 2> $R/gpurun_out/kt5.err; f=$(find $R/gpurun_out/kt5 -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/r01_kernel_stats_cfg5.csv; head -6 $f | cut -c1-150
