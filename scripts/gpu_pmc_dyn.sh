#!/bin/bash
# SQ counters of the dynamic row-parallel kernels (separate --pmc passes, kernel-trace only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/pmc_dyn; O=$R/gpurun_out/pmc_dyn
cd /tmp && export TMPDIR=/tmp
CFG=${1:-cfg4}
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_F64"; do
  rm -rf /tmp/pmc; timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc -o p -- python $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/pmc.err
  f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name']
    if 'k3c_' in k or 'k4c_' in k or 'k2_kernel' in k: acc[k[:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    print(k, ' '.join('%s=%.4g'%(c, sum(x)/len(x)) for c,x in v.items()))
PY
  else tail -3 /tmp/pmc.err; fi
done 2>&1 | tee $O/pmc_$CFG.txt
