#!/bin/bash
# Round 3, call o: instruction / wave-life counters of the shapes furthest below the roofline (one rocprofv3 pass set per shape).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/survey; O=$R/gpurun_out/survey
cd /tmp && export TMPDIR=/tmp
rm -f $O/r03_pmc_survey.txt
shape() { # tag, env...
  tag=$1; shift
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64" "FETCH_SIZE"; do
    rm -rf $O/pmc; env "$@" timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc -o p -- python $R/scripts/prof_shape.py > $O/out.txt 2> $O/pmc.err
    f=$(find $O/pmc -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python3 - "$f" "$tag" <<'PY' | tee -a $O/r03_pmc_survey.txt
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    kn=r['Kernel_Name']
    if 'pols::k' in kn and 'k6_' not in kn and 'mark' not in kn: acc[kn[:90]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items(): print(sys.argv[2], k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
    else tail -2 $O/pmc.err; fi
  done
  rm -rf $O/kt; env "$@" timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python $R/scripts/prof_shape.py > /dev/null 2> $O/kt.err
  f=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "pols::k" "$f" | grep -v "k6_\|mark" | cut -c1-160 | sed "s/^/$tag /" | tee -a $O/r03_pmc_survey.txt
}
shape f32_k15_200 G=50000 LO=200 K=15
shape f32_k16_200 G=50000 LO=200 K=16
shape f64_k12_200 G=50000 LO=200 K=12 DT=f64
shape f32_k8_100_300 G=50000 LO=100 HI=300 K=8
shape f32_k8_drop G=10000 LO=1000 K=8 POLICY=drop
shape f64_k8_12_40 G=500000 LO=12 HI=40 K=8 DT=f64
shape f32_k24_1000 G=10000 LO=1000 K=24
shape f32_k8_w_200 G=50000 LO=200 K=8 W=1
rm -rf $O/kt $O/pmc
