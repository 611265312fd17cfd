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


def _code_objects(lib: Path, td: Path):
    """the gfx950 code objects of the fat binary as ELF files under td.  Plain bundles: llvm-objdump --offloading.  Compressed bundles
    (--offload-compress, what the Makefile builds: a CCOB record per translation unit in .hip_fatbin): the section is cut at the
    records' own sizes and every record goes through clang-offload-bundler, which unpacks it."""
    tmp = td / lib.name
    shutil.copy(lib, tmp)
    subprocess.run([str(LLVM / "llvm-objdump"), "--offloading", str(tmp)], capture_output=True, text=True, check=True)
    cos = sorted(td.glob("*gfx950*"))
    if cos and all(c.read_bytes()[:4] == b"\x7fELF" for c in cos):
        return cos
    import struct

    fat = td / "fatbin.bin"
    subprocess.run([str(LLVM / "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", str(tmp), str(td / "scratch.so")],
                   capture_output=True, text=True, check=True)
    data, out, pos, n = fat.read_bytes(), [], 0, 0
    while True:
        pos = data.find(b"CCOB", pos)
        if pos < 0:
            break
        _ver, _method = struct.unpack_from("<HH", data, pos + 4)
        total = struct.unpack_from("<Q", data, pos + 8)[0] if _ver >= 3 else struct.unpack_from("<I", data, pos + 8)[0]
        rec, co = td / f"rec{n}.bin", td / f"co{n}.elf"
        rec.write_bytes(data[pos:pos + total])
        r = subprocess.run([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            f"--input={rec}", f"--output={co}"], capture_output=True, text=True)
        if r.returncode == 0 and co.exists() and co.read_bytes()[:4] == b"\x7fELF":
            out.append(co)
        pos += max(total, 4)
        n += 1
    return out


def kernel_scratch(lib: Path):
    """{demangled kernel name: (scratch bytes per lane, vgprs, agprs)}"""
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for co in _code_objects(lib, Path(td)):
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
