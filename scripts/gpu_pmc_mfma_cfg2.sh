#!/bin/bash
# MFMA counters of the LDS-tile engine (K1m) on BASELINE configs[1] -- the shape the product runs on the VALU kernel; the evidence
# behind that choice.  -> gpurun_out/r02_pmc_mfma_cfg2_k1m.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_k1m; POLS_K1_ENGINE=mfma timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_k1m -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/pmc_k1m_bench.json 2> $O/pmc_k1m.err
f=$(find $O/pmc_k1m -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY' | tee $O/r02_pmc_mfma_cfg2_k1m.txt
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if 'pols::' in r['Kernel_Name']: acc[r['Kernel_Name'][:80]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items(): print(k, {c: sum(x)/len(x) for c,x in v.items()}, 'dispatches', len(next(iter(v.values()))))
PY
grep -o '"kernel": "[^"]*", "kernel_ms": [0-9.]*' $O/pmc_k1m_bench.json | tee -a $O/r02_pmc_mfma_cfg2_k1m.txt
rm -rf $O/pmc_k1m
