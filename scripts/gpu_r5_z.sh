#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/r5z; O=$R/gpurun_out/r5z
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc; ONLY="1 x 10M" timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc -o p -- python $R/scripts/bench_shape_cliffs.py > /dev/null 2> /tmp/pmc.err
  f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" $ctr <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name']
    if r.get('Counter_Name')==sys.argv[2] and 'pols::' in k: acc[k[:90]].append(float(r['Counter_Value']))
for k,v in acc.items(): print(sys.argv[2], 'dispatches', len(v), 'mean KiB', round(sum(v)/len(v),1), 'kernel', k)
PY
done 2>&1 | tee $O/pmc_long_groups.txt
