#!/bin/bash
# Round profile: bench line, rocprofv3 kernel-trace stats, PMC traffic counters (own passes), bandwidth probe.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/profile; O=$R/gpurun_out/profile
TAG=${1:-r01}
echo "== bench (default, with cpu baseline)"
timeout 900 python bench.py 2> $O/bench.err > $O/${TAG}_bench.json; cut -c1-1800 $O/${TAG}_bench.json
echo "== bench f64"
timeout 600 python bench.py --dtype f64 --no-cpu-baseline 2>/dev/null > $O/${TAG}_bench_f64.json; cut -c1-400 $O/${TAG}_bench_f64.json
echo "== bandwidth probe (same access pattern, no math)"
./scripts/bw_probe.bin | tee $O/${TAG}_bw_probe.txt
cd /tmp && export TMPDIR=/tmp
echo "== rocprofv3 --kernel-trace --stats"
rm -rf $O/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k1 -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/${TAG}_bench_under_rocprof.json 2> $O/kt.err
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${TAG}_kernel_stats.csv && head -4 $O/${TAG}_kernel_stats.csv | cut -c1-220
for ctr in FETCH_SIZE WRITE_SIZE; do
  echo "== rocprofv3 --pmc $ctr"
  rm -rf $O/pmc_$ctr; timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$ctr -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/pmc_$ctr.err
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
