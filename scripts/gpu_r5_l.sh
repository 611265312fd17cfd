#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 300 python scripts/dbg_first_calls.py 2>&1 | grep -v amdgpu.ids
timeout 300 python scripts/dbg_first_calls.py 2>&1 | grep -v amdgpu.ids
