#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc4
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc4 -o p -- python $R/bench.py --config cfg4 --steps 3 --warmup 1 > /dev/null 2> $R/gpurun_out/pmc4.err
f=$(find $R/gpurun_out/pmc4 -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if 'pols' in r['Kernel_Name']: acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    print(k)
    for c,x in sorted(v.items()): print("   %-24s %.4g" % (c, sum(x)/len(x)))
PY
else tail -5 $R/gpurun_out/pmc4.err; fi
