#!/usr/bin/env python3
"""Compile one .hip translation unit for gfx950 with -Rpass-analysis=kernel-resource-usage and print one line per kernel:
VGPRs / AGPRs / SGPRs / spills / scratch / occupancy / LDS.  Runs without a GPU.  usage: kernel_resources.py k2_f64.hip [filter]"""
import re
import subprocess
import sys
from pathlib import Path

csrc = Path(__file__).resolve().parent.parent / "polars_ols_amd" / "csrc"
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only"]
out = subprocess.run(cmd, cwd=csrc, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
        continue
    m = re.search(r"remark: [^:]+:\d+:\d+:\s+([A-Za-z \[\]/]+): (\d+)", line) or re.search(r"\s{4}([A-Za-z \[\]/]+): (\d+) \[-Rpass", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for name, r in rows.items():
    if flt and flt not in name:
        continue
    short = re.sub(r"\(pols::\w+\)$", "", name.replace("pols::", ""))
    print(f"{short[:80]:80s} vgpr={r.get('VGPRs')} agpr={r.get('AGPRs')} sgpr={r.get('SGPRs')} spill={r.get('VGPRs Spill')}/{r.get('SGPRs Spill')} "
          f"scratch={r.get('ScratchSize [bytes/lane]')} occ={r.get('Occupancy [waves/SIMD]')} lds={r.get('LDS Size [bytes/block]')}")
