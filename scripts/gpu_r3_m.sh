#!/bin/bash
# Round 3, call m: counters and kernel-trace of the K1t kernel on 500 000 groups of 12..40 rows (8 f32 columns), default routing and the 16-lane form.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/tiny; O=$R/gpurun_out/tiny
cd /tmp && export TMPDIR=/tmp
pmc() { # mode, counters...
  m=$1; shift
  rm -rf $O/pmc; MODE=$m timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc -o p -- python $R/scripts/prof_tiny.py > /dev/null 2> $O/pmc.err
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
for m in default sub16; do
  pmc $m SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
  pmc $m SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
  pmc $m FETCH_SIZE
  pmc $m WRITE_SIZE
  rm -rf $O/kt; MODE=$m timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python $R/scripts/prof_tiny.py > /dev/null 2> $O/kt.err
  f=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "Name|k1t" "$f" | cut -c1-220 | tee $O/r03_kernel_stats_tiny_$m.csv
done
rm -rf $O/kt $O/pmc
