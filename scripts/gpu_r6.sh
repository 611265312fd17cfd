#!/bin/bash
# Round-6 GPU lease driver: ONE script, stages picked by name (scripts/gpu_r6.sh stage [stage ...]); outputs under gpurun_out/r6/.
#   k3h      halo-form RLS: its parity tests, the cfg4 A/B against the scan (bench lines + timeline), rlsg for regressions
#   tests    the whole GPU suite
#   prof     rocprofv3 kernel stats + PMC traffic of the BASELINE configs (copied to profiles/ by hand afterwards)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; O=$R/gpurun_out/r6; mkdir -p $O
TAG=r06
bench() { # name, extra env..., -- args
  local name=$1; shift
  timeout 300 env "$@" 2>$O/$name.err > $O/$name.json; cut -c1-330 $O/$name.json; echo
}
pmc() { # cfg, env...
  local cfg=$1; shift
  ( cd /tmp && export TMPDIR=/tmp
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_${ctr}_$cfg; timeout 300 env "$@" rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_${ctr}_$cfg -o p -- python $R/bench.py --config $cfg --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/pmc_${ctr}_$cfg.err
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
    rm -rf $O/pmc_${ctr}_$cfg
  done )
}
kstats() { # cfg, env...
  local cfg=$1; shift
  ( cd /tmp && export TMPDIR=/tmp
  rm -rf $O/kt_$cfg; timeout 300 env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$cfg -o k -- python $R/bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench_under_rocprof_$cfg.json 2> $O/kt_$cfg.err
  f=$(find $O/kt_$cfg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${TAG}_kernel_stats_$cfg.csv && head -6 $O/${TAG}_kernel_stats_$cfg.csv | cut -c1-200
  rm -rf $O/kt_$cfg )
}
for stage in "$@"; do
case $stage in
k3h)
  echo "== halo-form RLS: parity"
  timeout 1500 python -m pytest tests/test_k3_gpu.py -m gpu -x -q -k "halo or cfg4 or many_sequences or lookback or packed" 2>&1 | tail -15 | tee $O/${TAG}_pytest_k3h.txt
  echo "== cfg4 A/B: look-back (default) vs halo vs scan"
  for i in 1 2 3; do
    bench ${TAG}_bench_cfg4_lookback_$i A=1 python bench.py --config cfg4 --steps 50 --warmup 10 --no-cpu-baseline
    bench ${TAG}_bench_cfg4_halo_$i POLS_RLS_ENGINE=halo python bench.py --config cfg4 --steps 50 --warmup 10 --no-cpu-baseline
    bench ${TAG}_bench_cfg4_scan_$i POLS_RLS_ENGINE=scan python bench.py --config cfg4 --steps 50 --warmup 10 --no-cpu-baseline
  done
  bench ${TAG}_bench_cfg4_lookback_slow POLS_RLS_SPINS=0 python bench.py --config cfg4 --steps 50 --warmup 10 --no-cpu-baseline
  bench ${TAG}_bench_rlsg A=1 python bench.py --config rlsg --steps 20 --warmup 5 --no-cpu-baseline
  echo "== timeline (halo)"
  POLS_TIMELINE=1 timeout 120 python bench.py --config cfg4 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep -v amdgpu | tail -20 | tee $O/${TAG}_timeline_cfg4.txt
  kstats cfg4 A=1
  pmc cfg4 A=1
  ;;
k1wg)
  echo "== headline A/B: teams per workgroup (scripts/ab_headline.py, three rotated frames)"
  ONLY=default,wg2,wg4,wg4_p3,wg2_p3,team256_rc1_p3_nt timeout 600 python scripts/ab_headline.py 2>&1 | grep -v amdgpu | tee $O/${TAG}_ab_headline_wg.txt
  echo "== cfg3 shape (f64, weights, ridge)"
  DTYPE=f64 ONLY=default,wg2,wg4 timeout 600 python scripts/ab_headline.py 2>&1 | grep -v amdgpu | tee $O/${TAG}_ab_cfg3_wg.txt
  echo "== parity of the wg builds (bit-identical outputs expected)"
  timeout 300 python scripts/check_k1_wg.py 2>&1 | grep -v amdgpu | tee $O/${TAG}_check_k1_wg.txt
  echo "== timeline of the headline kernel"
  POLS_TIMELINE=1 timeout 120 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep timeline | tail -3 | tee $O/${TAG}_timeline_cfg2.txt
  echo "== SQ wait share: K1 and the stream probe in one trace"
  ( cd /tmp && export TMPDIR=/tmp; rm -rf $O/pmc_sq; timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_sq -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/pmc_sq.err
    f=$(find $O/pmc_sq -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python3 - "$f" <<'PY' | tee $O/${TAG}_pmc_sq_cfg2.txt
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name']
    if 'pols::' in k: acc[k[:90]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    m={c: sum(x)/len(x) for c,x in v.items()}
    print(k, {c: round(x) for c,x in m.items()}, 'wait share', round(m.get('SQ_WAIT_ANY',0)/max(1,m.get('SQ_WAVE_CYCLES',1)),3))
PY
    else tail -3 $O/pmc_sq.err; fi; rm -rf $O/pmc_sq )
  ;;
k4cm)
  echo "== masked tile kernel: parity"
  timeout 1500 python -m pytest tests/test_k4_gpu.py -m gpu -x -q -k "many_groups or tile_kernel or divergence" 2>&1 | tail -15 | tee $O/${TAG}_pytest_k4cm.txt
  echo "== dynamic entries on frames with nulls"
  for K in 6 8 10; do K=$K timeout 600 python scripts/bench_dyn_nulls.py 2>&1 | grep -v amdgpu; done | tee $O/${TAG}_bench_dyn_nulls.txt
  ( cd /tmp && export TMPDIR=/tmp; rm -rf $O/kt_dn; K=6 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_dn -o k -- python $R/scripts/bench_dyn_nulls.py > /dev/null 2> $O/kt_dn.err
    f=$(find $O/kt_dn -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${TAG}_kernel_stats_dyn_nulls_k6.csv && head -30 $O/${TAG}_kernel_stats_dyn_nulls_k6.csv | cut -c1-160; rm -rf $O/kt_dn )
  ;;
k4plu)
  echo "== K4p: LU on rows without an inverse"
  timeout 1500 python -m pytest tests/test_k4_gpu.py -m gpu -x -q 2>&1 | tail -15 | tee $O/${TAG}_pytest_k4plu.txt
  ;;
k4self)
  echo "== K4c without a halo wave: parity (rolling tests), then cfg4r A/B against the halo-wave form"
  timeout 1500 python -m pytest tests/test_k4_gpu.py tests/test_dyn_prep_gpu.py -m gpu -x -q 2>&1 | tail -15 | tee $O/${TAG}_pytest_k4self.txt
  for i in 1 2 3; do
    bench ${TAG}_bench_cfg4r_self_$i A=1 python bench.py --config cfg4r --steps 50 --warmup 10 --no-cpu-baseline
    bench ${TAG}_bench_cfg4r_halowave_$i POLS_ROLLING_ENGINE=halowave python bench.py --config cfg4r --steps 50 --warmup 10 --no-cpu-baseline
  done
  POLS_TIMELINE=1 timeout 120 python bench.py --config cfg4r --steps 2 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep timeline | tail -3 | tee $O/${TAG}_timeline_cfg4r.txt
  for K in 6 8; do K=$K timeout 600 python scripts/bench_dyn_nulls.py 2>&1 | grep -v amdgpu; done | tee $O/${TAG}_bench_dyn_nulls_self.txt
  kstats cfg4r A=1
  pmc cfg4r A=1
  ;;
nulls)
  echo "== null-policy cost on the headline shape: nt loads in the null-policy build (default) vs plain loads, one and two passes"
  for v in "A=1" "POLS_K1_NT_LOADS=0" "POLS_K1_PASSES=2"; do
    echo "-- $v"; env $v timeout 600 python scripts/bench_nulls.py 2>/dev/null | python -c "
import sys, json
for k, v in json.loads(sys.stdin.read()).items(): print(f'{k:36s} {v[\"us\"]:8.2f} us  {v[\"kernel\"]}')"
  done | tee $O/${TAG}_bench_nulls.txt
  timeout 900 python -m pytest tests/test_nulls_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $O/${TAG}_pytest_nulls.txt
  ;;
gather)
  echo "== drop family with nulls: the gathered tile kernel; RLS without the zero-fill rewrite"
  timeout 1500 python -m pytest tests/test_k4_gpu.py tests/test_k3_gpu.py tests/test_dyn_prep_gpu.py tests/test_frontend_gpu.py -m gpu -x -q 2>&1 | tail -15 | tee $O/${TAG}_pytest_gather.txt
  for K in 6 8 10; do K=$K timeout 600 python scripts/bench_dyn_nulls.py 2>&1 | grep -v amdgpu; done | tee $O/${TAG}_bench_dyn_nulls.txt
  echo "-- POLS_ROLLING_ENGINE=scatter (the three-pass form)" | tee -a $O/${TAG}_bench_dyn_nulls.txt
  K=6 POLS_ROLLING_ENGINE=scatter timeout 600 python scripts/bench_dyn_nulls.py 2>&1 | grep -v amdgpu | grep "drop " | tee -a $O/${TAG}_bench_dyn_nulls.txt
  ( cd /tmp && export TMPDIR=/tmp; rm -rf $O/kt_dn; K=6 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_dn -o k -- python $R/scripts/bench_dyn_nulls.py > /dev/null 2> $O/kt_dn.err
    f=$(find $O/kt_dn -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${TAG}_kernel_stats_dyn_nulls_k6.csv && grep pols:: $O/${TAG}_kernel_stats_dyn_nulls_k6.csv | cut -c1-150 | head -24; rm -rf $O/kt_dn )
  ;;
pmcinst)
  echo "== dynamic instruction mix of the row-parallel dynamic kernels (per dispatch; SQ counters in their own runs)"
  for cfg in cfg4r cfg4; do
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  ( cd /tmp && export TMPDIR=/tmp; rm -rf $O/pmc_i; timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_i -o p -- python $R/bench.py --config $cfg --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/pmc_i.err
    f=$(find $O/pmc_i -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python3 - "$f" $cfg <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name']
    if ('k4c_kernel' in k or 'k3c_kernel' in k): acc[k[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items(): print(sys.argv[2], k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
    else tail -3 $O/pmc_i.err; fi; rm -rf $O/pmc_i )
  done; done | tee $O/${TAG}_pmc_inst_dyn.txt
  ;;
tests)
  timeout 3000 python -m pytest tests -m gpu -q --maxfail=25 2>&1 | tail -40 | tee $O/${TAG}_pytest_gpu.txt
  ;;
prof)
  timeout 600 python bench.py 2> $O/bench.err > $O/${TAG}_bench.json; cut -c1-400 $O/${TAG}_bench.json
  for cfg in cfg3 cfg4 cfg4r rlsg rlsgr cfg5; do bench ${TAG}_bench_$cfg A=1 python bench.py --config $cfg --steps 20 --warmup 5; done
  for cfg in cfg2 cfg3 cfg4 cfg4r cfg5; do kstats $cfg A=1; pmc $cfg A=1; done
  ;;
*) echo "unknown stage $stage";;
esac
done
ls $O | head -50
