#!/bin/bash
# Round profile: bench line, rocprofv3 kernel-trace stats, PMC traffic counters (own passes), bandwidth probe.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/profile; O=$R/gpurun_out/profile
TAG=${1:-r01}
echo "== bench (default, with cpu baseline)"
timeout 300 python bench.py 2> $O/bench.err > $O/${TAG}_bench.json; cut -c1-1800 $O/${TAG}_bench.json
echo "== bench f64"
timeout 240 python bench.py --dtype f64 --no-cpu-baseline 2>/dev/null > $O/${TAG}_bench_f64.json; cut -c1-400 $O/${TAG}_bench_f64.json
echo "== bandwidth probe (same access pattern, no math)"
./scripts/bw_probe.bin | tee $O/${TAG}_bw_probe.txt
cd /tmp && export TMPDIR=/tmp
echo "== rocprofv3 --kernel-trace --stats"
rm -rf $O/kt; timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k1 -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/${TAG}_bench_under_rocprof.json 2> $O/kt.err
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${TAG}_kernel_stats.csv && head -4 $O/${TAG}_kernel_stats.csv | cut -c1-220
for ctr in FETCH_SIZE WRITE_SIZE; do
  echo "== rocprofv3 --pmc $ctr"
  rm -rf $O/pmc_$ctr; timeout 240 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$ctr -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/pmc_$ctr.err
  f=$(find $O/pmc_$ctr -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" $ctr <<'PY' | tee $O/${TAG}_pmc_$2$ctr.txt
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'k1' in r.get('Kernel_Name','') ]
vals=[float(r['Counter_Value']) for r in rows if r.get('Counter_Name')==sys.argv[2]]
print(sys.argv[2], 'dispatches', len(vals), 'mean', sum(vals)/max(1,len(vals)), 'kernel', rows[0]['Kernel_Name'][:60] if rows else None)
PY
  else tail -3 $O/pmc_$ctr.err; fi
done
ls $O | head -30
cd $R
echo "== other BASELINE configs"
for cfg in cfg3 cfg4 cfg5; do
  timeout 300 python bench.py --config $cfg --steps 20 --warmup 5 2>/dev/null > $O/${TAG}_bench_$cfg.json; cut -c1-700 $O/${TAG}_bench_$cfg.json; echo
done
cd /tmp
echo "== rocprofv3 --kernel-trace --stats, cfg5 (streamed Gram on the matrix cores + coordinate descent + predict)"
rm -rf $O/kt5; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt5 -o k5 -- python $R/bench.py --config cfg5 --steps 10 --warmup 3 > /dev/null 2> $O/kt5.err
f=$(find $O/kt5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${TAG}_kernel_stats_cfg5.csv && head -6 $O/${TAG}_kernel_stats_cfg5.csv | cut -c1-220
echo "== MFMA counters available"
timeout 60 rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u | tr '\n' ' ' | tee $O/${TAG}_mfma_counters_available.txt; echo
echo "== rocprofv3 --pmc MFMA busy, cfg5 Gram kernel"
rm -rf $O/pmc_mfma; timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -o p -- python $R/bench.py --config cfg5 --steps 3 --warmup 1 > /dev/null 2> $O/pmc_mfma.err
f=$(find $O/pmc_mfma -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then python3 - "$f" <<'PY' | tee $O/${TAG}_pmc_mfma_cfg5.txt
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    print(k, {c: sum(x)/len(x) for c,x in v.items()})
PY
else tail -5 $O/pmc_mfma.err; fi
ls $O | head -40
echo "== K9 group-key ingestion: timings + kernel trace"
timeout 120 python $R/scripts/bench_layout.py 2>/dev/null | tail -1 > $O/${TAG}_bench_layout.json; cut -c1-700 $O/${TAG}_bench_layout.json; echo
rm -rf $O/kt9; timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt9 -o l -- python $R/scripts/bench_layout.py > /dev/null 2> $O/kt9.err
f=$(find $O/kt9 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -v "at::native\|ROCPRIM_400001" "$f" > $O/${TAG}_kernel_stats_layout.csv && head -12 $O/${TAG}_kernel_stats_layout.csv | cut -c1-160
timeout 120 python $R/scripts/bench_nulls.py 2>/dev/null | tail -1 > $O/${TAG}_bench_nulls.json; cat $O/${TAG}_bench_nulls.json
timeout 120 python $R/scripts/bench_ragged.py 2>/dev/null | tail -1 > $O/${TAG}_bench_ragged.json; cat $O/${TAG}_bench_ragged.json
