for env in "POLS_K1_SHAPE=team POLS_K1_NT_LOADS=0" "POLS_K1_SHAPE=team" "POLS_K1_SHAPE=team POLS_K1_PASSES=2" "POLS_K1_SHAPE=team POLS_K1_PASSES=3" "POLS_K1_SHAPE=team POLS_K1_PASSES=2 POLS_K1_NT_LOADS=0" ""; do
  for i in 1 2; do env $env python bench.py --no-cpu-baseline --steps 100 --warmup 10 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$env', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel'])"; done
done
