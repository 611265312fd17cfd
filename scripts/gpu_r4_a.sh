#!/bin/bash
# Round 4, visit a: the row-parallel RLS kernel (K3c) -- parity, cfg4 A/B against the lane-per-chunk K3s, the many-sequence shape,
# both stream-probe modes beside the headline kernel.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/r4a; O=$R/gpurun_out/r4a
line() { python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
d=json.loads(t[-1]); r=d['roofline']
sc=r.get('stream_ceiling') or {}
print('$1', 'value=%.4g'%d['value'], 'ms/step=%.4f'%d['ms_per_step'], r['kernel'], 'kernel_ms=%.4f n=%s'%(r['kernel_ms'], r.get('kernel_samples')), 'GB/s=%.0f frac=%.3f'%(r['achieved'], r['frac']), 'probe0=%.0f'%sc.get('GBps',0), 'probe1=%.0f'%(sc.get('persistent') or {}).get('GBps',0))"; }
echo "== tests (RLS family)"
timeout 900 python -m pytest tests/test_k3_gpu.py tests/test_dyn_prep_gpu.py -m gpu -x -q 2>&1 | tail -15 | cut -c1-300
timeout 600 python -m pytest tests/test_frontend_gpu.py tests/test_notebook_gpu.py tests/test_arrow_plugins_gpu.py -m gpu -x -q -k "rls or recursive or cells" 2>&1 | tail -5 | cut -c1-300
echo "== cfg4 A/B"
for e in scan chunk; do
  POLS_RLS_ENGINE=$e timeout 300 python bench.py --config cfg4 --steps 50 --warmup 10 --no-cpu-baseline 2>$O/cfg4_$e.err | tee $O/cfg4_$e.json | line "cfg4/$e"
done
timeout 300 python bench.py --config cfg4r --steps 50 --warmup 10 --no-cpu-baseline 2>$O/cfg4r.err | tee $O/cfg4r.json | line cfg4r
echo "== rlsg"
for e in scan seq; do
  POLS_RLS_ENGINE=$e timeout 300 python bench.py --config rlsg --steps 20 --warmup 5 --no-cpu-baseline 2>$O/rlsg_$e.err | tee $O/rlsg_$e.json | line "rlsg/$e"
done
timeout 300 python bench.py --config rlsgr --steps 10 --warmup 3 --no-cpu-baseline 2>$O/rlsgr.err | tee $O/rlsgr.json | line rlsgr
echo "== headline + probes"
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>$O/cfg2.err | tee $O/cfg2.json | line cfg2
timeout 300 python bench.py --config cfg3 --steps 50 --warmup 10 --no-cpu-baseline 2>$O/cfg3.err | tee $O/cfg3.json | line cfg3
timeout 300 python bench.py --config cfg5 --steps 10 --warmup 3 --no-cpu-baseline 2>$O/cfg5.err | tee $O/cfg5.json | line cfg5
echo "== rocprofv3 cfg4"
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kt_cfg4; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_cfg4 -o k -- python $R/bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu-baseline > $O/cfg4_under_rocprof.json 2> $O/kt_cfg4.err
f=$(find $O/kt_cfg4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r04_kernel_stats_cfg4.csv && head -6 $O/r04_kernel_stats_cfg4.csv | cut -c1-260
tail -3 $O/*.err | cut -c1-300 | grep -v "^$" | head -40
