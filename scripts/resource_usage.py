#!/usr/bin/env python3
"""Kernel resource usage of one translation unit: hipcc -Rpass-analysis=kernel-resource-usage, one line per kernel.
usage: scripts/resource_usage.py polars_ols_amd/csrc/k4c_rolling.hip [name filter]"""
import re, subprocess, sys, os
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize", "-Wno-pass-failed",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
err = subprocess.run(cmd, capture_output=True, text=True, cwd=os.getcwd()).stderr if not src.endswith(".txt") else open(src).read()
cur = None; rows = {}
for line in err.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip(); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1); rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt and flt not in dem: continue
    print(f"{dem[:90]:90s} VGPR {r.get('VGPRs','?'):>4} AGPR {r.get('AGPRs','?'):>4} spill {r.get('VGPRs Spill','?'):>4} scratch {r.get('ScratchSize [bytes/lane]','?'):>5} occ {r.get('Occupancy [waves/SIMD]','?')} LDS {r.get('LDS Size [bytes/block]','?')}")
