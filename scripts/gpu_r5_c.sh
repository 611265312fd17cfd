#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r5c
for skew in 0 4096 33024 1048576 1056768 69888; do timeout 300 scripts/probe_matrix.bin 60 $skew quick 2>&1 | grep -v "^# best\|^# probe\|^# *mix" ; done | tee gpurun_out/r5c/probe_skew.txt
