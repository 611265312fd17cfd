#!/bin/bash
# Round 5, visit d: the full GPU suite; smoke; the default bench line; the reference's published dynamic rows.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/r5d; O=$R/gpurun_out/r5d
echo "== full gpu suite"
timeout 2400 python -m pytest tests -m gpu -q -x --tb=short 2>&1 | grep -v "^    " | tail -15 | cut -c1-300
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== bench default"
timeout 600 python bench.py 2>/dev/null | tee $O/bench.json | cut -c1-600
for c in rls100 roll100; do timeout 600 python bench.py --config $c 2>/dev/null | tee $O/bench_$c.json | cut -c1-700; done
