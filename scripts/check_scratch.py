#!/usr/bin/env python3
"""Every gfx950 kernel of libpols_mi355x.so with its private-segment (scratch) size, read from the code objects' metadata -- no GPU,
no recompilation.  A kernel that keeps an array in scratch memory instead of registers is several times slower than it should be
(a four-line lambda around the K1 loads once cost the ragged kernels 736 bytes per lane and 4x); tests/test_abi_cpu.py asserts
the allow-list below.  usage: check_scratch.py [path/to/libpols_mi355x.so]"""
import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")


def kernel_scratch(lib: Path):
    """{demangled kernel name: (scratch bytes per lane, vgprs, agprs)}"""
    out = {}
    with tempfile.TemporaryDirectory() as td:
        tmp = Path(td) / lib.name
        shutil.copy(lib, tmp)
        subprocess.run([str(LLVM / "llvm-objdump"), "--offloading", str(tmp)], capture_output=True, text=True, check=True)
        for co in sorted(Path(td).glob("*gfx950*")):
            notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], capture_output=True, text=True).stdout
            for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk)
                priv = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
                vgpr = re.search(r"\.vgpr_count:\s+(\d+)", blk)
                agpr = re.match(r"\s*(\d+)", blk)              # the block starts right behind ".agpr_count:"
                if name and priv:
                    out[name.group(1)] = (int(priv.group(1)), int(vgpr.group(1)) if vgpr else -1, int(agpr.group(1)) if agpr else -1)
    if out:
        dem = subprocess.run(["c++filt"], input="\n".join(out), capture_output=True, text=True).stdout.splitlines()
        out = {d: v for d, v in zip(dem, out.values())}
    return out


if __name__ == "__main__":
    lib = Path(sys.argv[1]) if len(sys.argv) > 1 else Path(__file__).resolve().parent.parent / "polars_ols_amd" / "libpols_mi355x.so"
    ks = kernel_scratch(lib)
    bad = {k: v for k, v in ks.items() if v[0] > 0}
    print(f"{len(ks)} kernels, {len(bad)} with scratch")
    for k, (p, v, _a) in sorted(bad.items(), key=lambda kv: -kv[1][0]):
        print(f"  {p:6d} B/lane  vgpr={v:3d}  {k[:150]}")
