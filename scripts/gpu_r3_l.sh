#!/bin/bash
# Round 3, call l: K1t with the team reduce-scatter and eight-lane teams: parity, the A/B of the team shapes, counters of the new default.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/tiny; O=$R/gpurun_out/tiny
python -m pytest tests/test_k1_gpu.py tests/test_routing_gpu.py tests/test_k6_gpu.py tests/test_nulls_gpu.py tests/test_frontend_gpu.py -m gpu -q --maxfail=20 --tb=short -p no:cacheprovider > gpurun_out/r3l_tests.log 2>&1; echo "pytest exit $?"
grep -E "passed|failed" gpurun_out/r3l_tests.log | tail -2
grep -E "^(FAILED|ERROR)|^E  " gpurun_out/r3l_tests.log | head -20 | cut -c1-250
timeout 400 python scripts/bench_tiny.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee gpurun_out/r03_bench_tiny.json
cd /tmp && export TMPDIR=/tmp
pmc() { # counters...
  rm -rf $O/pmc; MODE=default timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc -o p -- python $R/scripts/prof_tiny.py > /dev/null 2> $O/pmc.err
  f=$(find $O/pmc -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" <<'PY' | tee -a $O/r03_pmc_tiny.txt
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if 'k1t_' in r['Kernel_Name']: acc[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items(): print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
  else tail -3 $O/pmc.err; fi
}
rm -f $O/r03_pmc_tiny.txt
pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
pmc FETCH_SIZE
pmc WRITE_SIZE
rm -rf $O/kt; MODE=default timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python $R/scripts/prof_tiny.py > /dev/null 2> $O/kt.err
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "Name|k1t" "$f" | cut -c1-220 | tee $O/r03_kernel_stats_tiny.csv
rm -rf $O/kt $O/pmc
