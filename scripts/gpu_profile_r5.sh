#!/bin/bash
# Round-5 profile: bench lines for every BASELINE config (+ the many-sequence dynamic shapes, the cfg5 shard sizes), rocprofv3
# kernel-trace stats, PMC traffic counters in their own passes (FETCH_SIZE / WRITE_SIZE), SQ wait / MFMA counters for cfg5.
# -> gpurun_out/profile_r5/ (copied to profiles/)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/profile_r5; O=$R/gpurun_out/profile_r5
TAG=r05
echo "== bench lines"
timeout 600 python bench.py 2> $O/bench.err > $O/${TAG}_bench.json; cut -c1-400 $O/${TAG}_bench.json
for cfg in cfg1 cfg3 cfg4 cfg4r rlsg rlsgr cfg5 ref100 rls100 roll100; do
  timeout 600 python bench.py --config $cfg --steps 20 --warmup 5 2>/dev/null > $O/${TAG}_bench_$cfg.json; cut -c1-300 $O/${TAG}_bench_$cfg.json; echo
done
for g in 12500 25000 50000; do
  timeout 600 python bench.py --config cfg5 --groups $g --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null > $O/${TAG}_bench_cfg5_g$g.json; cut -c1-260 $O/${TAG}_bench_cfg5_g$g.json; echo
done
timeout 300 python bench.py --dtype f64 --no-cpu-baseline 2>/dev/null > $O/${TAG}_bench_f64.json; cut -c1-300 $O/${TAG}_bench_f64.json; echo
timeout 300 python bench.py --mem host --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null > $O/${TAG}_bench_host.json; cut -c1-300 $O/${TAG}_bench_host.json; echo
cd /tmp && export TMPDIR=/tmp
for cfg in cfg2 cfg3 cfg5 cfg4 cfg4r rlsg rlsgr; do
  echo "== rocprofv3 --kernel-trace --stats $cfg"
  rm -rf $O/kt_$cfg; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$cfg -o k -- python $R/bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench_under_rocprof_$cfg.json 2> $O/kt_$cfg.err
  f=$(find $O/kt_$cfg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${TAG}_kernel_stats_$cfg.csv && head -5 $O/${TAG}_kernel_stats_$cfg.csv | cut -c1-220
  for ctr in FETCH_SIZE WRITE_SIZE; do
    echo "== rocprofv3 --pmc $ctr $cfg"
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
echo "== rocprofv3 --pmc SQ wait / MFMA busy, cfg5 fused kernel"
rm -rf $O/pmc_mfma; timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_mfma -o p -- python $R/bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_mfma.err
f=$(find $O/pmc_mfma -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then python3 - "$f" <<'PY' | tee $O/${TAG}_pmc_mfma_cfg5.txt
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if 'k2_' in r['Kernel_Name']: acc[r['Kernel_Name'][:80]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items(): print(k, {c: sum(x)/len(x) for c,x in v.items()})
PY
else tail -5 $O/pmc_mfma.err; fi
echo "== SQ counters of the dynamic kernels (cfg4, cfg4r)"
for cfg in cfg4 cfg4r; do
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM"; do
  rm -rf /tmp/pmc; timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc -o p -- python $R/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/pmc.err
  f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name']
    if 'k3c_' in k or 'k4c_' in k: acc[k[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items(): print(k, ' '.join('%s=%.4g'%(c, sum(x)/len(x)) for c,x in v.items()))
PY
done; done 2>&1 | tee $O/${TAG}_pmc_sq_dynamic.txt
echo "== K2w (f64, 31 columns x 1 000 rows): HBM traffic and matrix-core / wait counters"
for ctr in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  rm -rf /tmp/pmc; KS=31 ONLY_F64=1 timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc -o p -- python $R/scripts/bench_k16.py > /dev/null 2> /tmp/pmc.err
  f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name']
    if 'k2w_kernel<double, 8' in k: acc[k[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items(): print(k, ' '.join('%s=%.6g (n=%d)'%(c, sum(x)/len(x), len(x)) for c,x in v.items()))
PY
done 2>&1 | tee $O/${TAG}_pmc_k2w.txt

echo "== K4p / K3p (12 and 32 features, 10 000 sequences x 1 000 rows): kernel stats, HBM traffic, SQ counters"
cd /tmp
for KS in 12 32; do
  rm -rf $O/kt_dyn; KS=$KS timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_dyn -o k -- python $R/scripts/bench_dyn_edges.py > $O/${TAG}_bench_dyn_edges_under_rocprof_k$KS.txt 2> $O/kt_dyn.err
  f=$(find $O/kt_dyn -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${TAG}_kernel_stats_dyn_k$KS.csv && head -6 $O/${TAG}_kernel_stats_dyn_k$KS.csv | cut -c1-200
  for ctr in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM"; do
    rm -rf /tmp/pmc; KS=$KS timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc -o p -- python $R/scripts/bench_dyn_edges.py > /dev/null 2> /tmp/pmc.err
    f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 - "$f" $KS <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name']
    if 'kp_' in k: acc[k[:90]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items(): print('k=%s'%sys.argv[2], k, ' '.join('%s=%.6g (n=%d)'%(c, sum(x)/len(x), len(x)) for c,x in v.items()))
PY
  done
done 2>&1 | tee $O/${TAG}_pmc_k4p.txt
cd $R
echo "== dynamic side benches"
KS=6,8,10,12,16,24,32 timeout 600 python scripts/bench_dyn_edges.py 2>/dev/null | grep -v amdgpu > $O/${TAG}_bench_dyn_edges.txt; cat $O/${TAG}_bench_dyn_edges.txt
for K in 6 12 32; do K=$K timeout 600 python scripts/bench_dyn_nulls.py 2>/dev/null | grep -v amdgpu; done > $O/${TAG}_bench_dyn_nulls.txt; cat $O/${TAG}_bench_dyn_nulls.txt
SHORT=1 timeout 900 python scripts/bench_shape_cliffs.py 2>/dev/null | grep -v amdgpu > $O/${TAG}_bench_shape_cliffs.txt; cat $O/${TAG}_bench_shape_cliffs.txt
echo "== side benches"
timeout 200 python scripts/bench_ragged.py 2>/dev/null | tail -1 > $O/${TAG}_bench_ragged.json; cut -c1-1200 $O/${TAG}_bench_ragged.json; echo
timeout 120 python scripts/bench_nulls.py 2>/dev/null | tail -1 > $O/${TAG}_bench_nulls.json; cat $O/${TAG}_bench_nulls.json; echo
timeout 200 python scripts/bench_k9.py 2>/dev/null | grep -v amdgpu > $O/${TAG}_bench_k9.txt; cut -c1-160 $O/${TAG}_bench_k9.txt
KS=15,16,17,20,24,28,31 timeout 300 python scripts/bench_k16.py 2>/dev/null | grep -v amdgpu > $O/${TAG}_bench_k16.txt; cat $O/${TAG}_bench_k16.txt
rm -rf $O/kt_* $O/pmc_FETCH* $O/pmc_WRITE* $O/pmc_mfma
ls $O
