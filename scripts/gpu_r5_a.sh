#!/bin/bash
# Round 5, visit a: the probe matrix (XCD map x piece size x mix x load / store kind), K1 with the XCD-contiguous map A/B, K6s (short
# groups: n < k) tests + the shape-cliff bench, the rolling divergence-band tests.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/r5a; O=$R/gpurun_out/r5a
echo "== probe matrix"
timeout 600 scripts/probe_matrix.bin 40 > $O/probe_matrix.txt 2>&1; tail -4 $O/probe_matrix.txt
echo "== new tests"
timeout 1200 python -m pytest tests/test_k6_gpu.py tests/test_k4_gpu.py -m gpu -q -x -k "short_groups or divergence or clamped or degenerate or fit_wide" 2>&1 | tail -15 | cut -c1-400
echo "== K1 XCD A/B"
timeout 600 python scripts/ab_xcd.py 2>&1 | tee $O/ab_xcd.txt | cut -c1-200
echo "== shape cliffs (short groups)"
SHORT=1 ONLY="(n" timeout 900 python scripts/bench_shape_cliffs.py 2>&1 | tee $O/bench_shape_cliffs_short.txt | cut -c1-200
SHORT=1 ONLY="mixed 5k" timeout 900 python scripts/bench_shape_cliffs.py 2>&1 | tee -a $O/bench_shape_cliffs_short.txt | cut -c1-200
