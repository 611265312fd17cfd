#!/usr/bin/env python3
"""Per-kernel resource table of a gfx950 assembly file (hipcc -save-temps): VGPRs, AGPRs, SGPRs, scratch, LDS, occupancy.
usage: kstats.py file.s [name-filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt and flt not in dem: continue
    g = lambda k: (re.search(r"\.amdhsa_" + k + r"\s+(\S+)", body) or [None, "?"])[1]
    # the comment block after the kernel carries the real counts
    c = re.search(re.escape(name) + r"\n.*?; NumVgprs: (\d+)\n; NumAgprs: (\d+)\n; TotalNumVgprs: (\d+)\n; ScratchSize: (\d+)\n; MemoryBound: \d+\n(?:.*?\n)*?; LDSByteSize: (\d+).*?\n(?:.*?\n)*?; Occupancy: (\d+)", txt)
    if c: print(f"{dem[:110]:110s} v={c.group(1)} a={c.group(2)} tot={c.group(3)} scratch={c.group(4)} lds={c.group(5)} occ={c.group(6)}")
    else: print(dem[:110], "next_free_vgpr", g("next_free_vgpr"), "lds", g("group_segment_fixed_size"), "scratch", g("private_segment_fixed_size"))
