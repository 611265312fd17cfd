#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/profile_r2; O=gpurun_out/profile_r2
echo "== pytest -m gpu (all)"
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 --tb=short 2>&1 | tail -30 | cut -c1-300 | tee gpurun_out/pytest_gpu_all.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4
echo "== side benches"
timeout 200 python scripts/bench_ragged.py 2>/dev/null | tail -1 > $O/r02_bench_ragged.json; cut -c1-1800 $O/r02_bench_ragged.json; echo
timeout 120 python scripts/bench_nulls.py 2>/dev/null | tail -1 > $O/r02_bench_nulls.json; cat $O/r02_bench_nulls.json; echo
timeout 120 python scripts/bench_layout.py 2>/dev/null | tail -1 > $O/r02_bench_layout.json; cut -c1-300 $O/r02_bench_layout.json; echo
