#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r2d; O=gpurun_out/r2d
echo "== pytest bench + distributed pieces"
timeout 1200 python -m pytest tests/test_bench_gpu.py tests/test_k1_gpu.py tests/test_k2_gpu.py -m gpu -q --maxfail=20 --tb=short 2>&1 | tail -40 | cut -c1-300 | tee $O/pytest_gpu.log
echo "== bench cfg2 plain / nt"
for i in 1 2; do
timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>$O/cfg2.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['kernel'])"
POLS_K1_NT_LOADS=1 timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>>$O/cfg2.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nt   ', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['kernel'])"
done
echo "== bench with cpu baseline (cfg2)"
timeout 600 python bench.py 2>$O/bench.err | tee $O/bench_cfg2.json | cut -c1-2500
echo "== host mode"
timeout 300 python bench.py --mem host --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tee $O/bench_cfg2_host.json | cut -c1-400
