#!/bin/bash
# Round-5 closing profile on the final tree: bench lines of every BASELINE config, rocprofv3 kernel stats + PMC traffic of the static configs (their
# kernels were rebuilt with the size-class argument), and the side sweeps.  The dynamic kernels did not change after scripts/gpu_profile_r5.sh.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/profile_r5f; O=$R/gpurun_out/profile_r5f
TAG=r05
echo "== bench lines"
timeout 600 python bench.py 2> $O/bench.err > $O/${TAG}_bench.json; cut -c1-400 $O/${TAG}_bench.json
for cfg in cfg1 cfg3 cfg4 cfg4r rlsg rlsgr cfg5 ref100 rls100 roll100; do
  timeout 600 python bench.py --config $cfg --steps 20 --warmup 5 2>/dev/null > $O/${TAG}_bench_$cfg.json; cut -c1-260 $O/${TAG}_bench_$cfg.json; echo
done
for g in 12500; do
  timeout 600 python bench.py --config cfg5 --groups $g --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null > $O/${TAG}_bench_cfg5_g$g.json; cut -c1-260 $O/${TAG}_bench_cfg5_g$g.json; echo
done
timeout 300 python bench.py --dtype f64 --no-cpu-baseline 2>/dev/null > $O/${TAG}_bench_f64.json; cut -c1-260 $O/${TAG}_bench_f64.json; echo
cd /tmp && export TMPDIR=/tmp
for cfg in cfg2 cfg3 cfg5; do
  echo "== rocprofv3 --kernel-trace --stats $cfg"
  rm -rf $O/kt_$cfg; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$cfg -o k -- python $R/bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench_under_rocprof_$cfg.json 2> $O/kt_$cfg.err
  f=$(find $O/kt_$cfg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${TAG}_kernel_stats_$cfg.csv && head -5 $O/${TAG}_kernel_stats_$cfg.csv | cut -c1-220
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_${ctr}_$cfg; timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_${ctr}_$cfg -o p -- python $R/bench.py --config $cfg --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/pmc_${ctr}_$cfg.err
    f=$(find $O/pmc_${ctr}_$cfg -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python3 - "$f" $ctr <<'PY' | tee $O/${TAG}_pmc_${ctr}_$cfg.txt
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name']
    if r.get('Counter_Name')==sys.argv[2] and 'pols::' in k and 'probe' not in k and 'start_kernel' not in k:
        acc[k[:100]].append(float(r['Counter_Value']))
for k,v in acc.items(): print(sys.argv[2], 'dispatches', len(v), 'mean', sum(v)/len(v), 'kernel', k)
PY
    else tail -3 $O/pmc_${ctr}_$cfg.err; fi
  done
done
cd $R
echo "== side benches"
timeout 200 python scripts/bench_ragged.py 2>/dev/null | tail -1 > $O/${TAG}_bench_ragged.json; cut -c1-600 $O/${TAG}_bench_ragged.json; echo
timeout 120 python scripts/bench_nulls.py 2>/dev/null | tail -1 > $O/${TAG}_bench_nulls.json; cut -c1-600 $O/${TAG}_bench_nulls.json; echo
timeout 200 python scripts/bench_k9.py 2>/dev/null | grep -v amdgpu > $O/${TAG}_bench_k9.txt; cut -c1-160 $O/${TAG}_bench_k9.txt
KS=15,16,17,20,24,28,31 timeout 300 python scripts/bench_k16.py 2>/dev/null | grep -v amdgpu > $O/${TAG}_bench_k16.txt; cat $O/${TAG}_bench_k16.txt
timeout 300 python scripts/bench_decade.py 2>/dev/null | grep -v amdgpu > $O/${TAG}_bench_decade.txt; cat $O/${TAG}_bench_decade.txt
SHORT=1 timeout 900 python scripts/bench_shape_cliffs.py 2>/dev/null | grep -v amdgpu > $O/${TAG}_bench_shape_cliffs.txt; cat $O/${TAG}_bench_shape_cliffs.txt
timeout 900 python scripts/bench_rows_sweep.py 2>/dev/null | grep -v amdgpu > $O/${TAG}_bench_rows_sweep.txt; cat $O/${TAG}_bench_rows_sweep.txt
rm -rf $O/kt_* $O/pmc_FETCH* $O/pmc_WRITE*
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed" | tee $O/${TAG}_pytest_gpu.txt
ls $O
