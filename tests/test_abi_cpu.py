"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/*.h declares,
its parameter defaults equal the reference's Python dataclass defaults, and it fails LOUDLY without a GPU
(no CPU fallback).  No compute calls here."""
import ctypes as C
import re
from pathlib import Path

import pytest

from polars_ols_amd import _lib

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def L():
    if not _lib.LIB_PATH.exists():
        _lib.build()
    return _lib.lib()


def test_exports_every_declared_symbol(L):
    header = (ROOT / "include" / "pols_mi355x.h").read_text()
    declared = set(re.findall(r"\b(pols_[a-z_0-9]+)\s*\(", header))
    assert declared, "no prototypes found"
    missing = sorted(s for s in declared if not hasattr(L, s))
    assert not missing, missing
    assert declared == set(_lib.EXPORTS)
    # the measurement aids live in their own header: nothing of them is declared by the reference interface
    dbg = (ROOT / "include" / "pols_mi355x_debug.h").read_text()
    dbg_declared = set(re.findall(r"\b(pols_[a-z_0-9]+)\s*\(", dbg))
    assert dbg_declared == set(_lib.DEBUG_EXPORTS)
    assert not (dbg_declared & declared)
    assert not [s for s in dbg_declared if not hasattr(L, s)]


def test_header_cites_reference_interfaces():
    header = (ROOT / "include" / "pols_mi355x.h").read_text()
    for cite in ("src/expressions.rs:351-388", "src/least_squares.rs:568-598", "src/least_squares.rs:848-1032",
                 "src/expressions.rs:706-741", "src/expressions.rs:15-18"):
        assert cite in header


def test_ols_defaults_match_reference_dataclass(L):
    # polars_ols/least_squares.py:101-107
    p = _lib.OlsParams()
    L.pols_ols_params_default(C.byref(p))
    assert (p.alpha, p.has_l1_ratio, p.max_iter, p.tol, p.positive, p.solve_method, p.has_rcond, p.null_policy) == \
        (0.0, 0, 1000, 1e-5, 0, 0, 0, _lib.NULL_POLICIES["ignore"])


def test_rls_and_rolling_defaults_match_reference_dataclass(L):
    # polars_ols/least_squares.py:137-140 and :156-160
    p = _lib.RlsParams()
    L.pols_rls_params_default(C.byref(p))
    assert (p.has_half_life, p.initial_state_covariance, bool(p.initial_state_mean), p.null_policy) == \
        (0, 10.0, False, _lib.NULL_POLICIES["drop"])
    r = _lib.RollingParams()
    L.pols_rolling_params_default(C.byref(r))
    assert (r.window_size, r.min_periods, r.use_woodbury, r.alpha, r.null_policy) == \
        (1_000_000, -1, -1, 0.0, _lib.NULL_POLICIES["drop_window"])


def test_enum_values_match_header():
    header = (ROOT / "include" / "pols_mi355x.h").read_text()
    for name, val in {"POLS_SOLVE_QR": 1, "POLS_SOLVE_SVD": 2, "POLS_SOLVE_CHOL": 3, "POLS_SOLVE_LU": 4,
                      "POLS_SOLVE_CD": 5, "POLS_SOLVE_CD_ACTIVE_SET": 6, "POLS_NULL_DROP_WINDOW": 5,
                      "POLS_ERR_PANIC": -4, "POLS_ERR_NO_DEVICE": -5}.items():
        assert re.search(rf"{name}\s*=\s*{val}\b", header), name


def test_fails_loudly_without_gpu(L):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = L.pols_create(0, C.byref(h))
    assert rc == -5 and not h.value
    assert b"no CPU fallback" in L.pols_last_error()
    from polars_ols_amd import Engine, PolsError

    with pytest.raises(PolsError):
        Engine(0)


def test_product_never_imports_the_oracle():
    for f in (ROOT / "polars_ols_amd").rglob("*"):
        if f.suffix in {".py", ".hip", ".hpp", ".inl", ".cpp"}:
            txt = f.read_text()
            assert "oracle" not in txt.replace("the CPU oracle under ``oracle/`` is test infrastructure", "") \
                .replace("and is never imported from here", "") or f.name == "_lib.py", f


def test_header_is_plain_c_and_links(tmp_path):
    """The boundary is a C ABI: the header compiles as strict C99 and a C translation unit links against the library with
    nothing but the header (what a cgo / Rust-bindgen / ctypes consumer relies on)."""
    import shutil
    import subprocess

    if not _lib.LIB_PATH.exists():
        _lib.build()
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    src = tmp_path / "consumer.c"
    src.write_text(
        '#include <stdio.h>\n#include "pols_mi355x.h"\n'
        "int main(void) {\n"
        "    pols_ols_params p; pols_rls_params r; pols_rolling_params w; pols_stats_out s = {0};\n"
        "    pols_ols_params_default(&p); pols_rls_params_default(&r); pols_rolling_params_default(&w);\n"
        "    (void)s;\n"
        "    pols_layout *lay = 0;   /* NULL handles are answered, not dereferenced */\n"
        "    if (pols_layout_n_rows(lay) != -1 || pols_layout_n_groups(lay) != -1 || pols_layout_group_offsets(lay)) return 1;\n"
        "    if (pols_layout_create(0, 0, 0, POLS_MEM_HOST, &lay) != POLS_ERR_INVALID) return 2;\n"
        "    pols_layout_destroy(lay);\n"
        '    printf("%s %d %g %lld %d\\n", pols_version(), p.null_policy == POLS_NULL_IGNORE, r.initial_state_covariance,\n'
        "           (long long)w.window_size, POLS_MAX_FEATURES_STATIC);\n"
        "    return 0;\n}\n")
    exe = tmp_path / "consumer"
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", f"-I{ROOT / 'include'}", str(src), "-o", str(exe),
                    f"-L{_lib.LIB_PATH.parent}", "-lpols_mi355x", f"-Wl,-rpath,{_lib.LIB_PATH.parent}"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert out[-4:] == ["1", "10", "1000000", "1024"], out


def test_hot_kernels_keep_their_arrays_in_registers():
    """The code objects' own metadata (no GPU, no recompilation): none of the HBM-bound static kernels may hold an array in scratch
    memory -- private_segment_fixed_size 0.  A lambda around K1's loads once put the ragged kernels' chunks in scratch (736 bytes per
    lane, 4x slower) without a single compiler warning; this pins it."""
    import sys

    sys.path.insert(0, str(ROOT / "scripts"))
    from check_scratch import LLVM, kernel_scratch

    if not (LLVM / "llvm-objdump").exists():
        pytest.skip("ROCm LLVM tools not present")
    if not _lib.LIB_PATH.exists():
        _lib.build()
    ks = kernel_scratch(_lib.LIB_PATH)
    assert len(ks) > 500, len(ks)
    hot = ("pols::k1_kernel<", "pols::k1t_kernel<", "pols::k1p_kernel<", "pols::k1m_kernel<", "pols::k2_kernel<", "pols::gram_stream_kernel<",
           "pols::predict_kernel<", "pols::gram_solve_kernel<", "pols::take_kernel<", "pols::arrow_ingest_kernel<",
           "pols::k2w_kernel<", "pols::k3c_kernel<", "pols::k4c_kernel<",      # round 4: the 17..31-column and the row-parallel dynamic kernels
           "pols::kp_rls_walk_kernel<", "pols::kp_rolling_walk_kernel<", "pols::kp_totals_kernel<", "pols::k6s_kernel<",
           "pols::gram_valu_kernel<", "pols::predict_groups_kernel<")   # round 5 (three nested lambdas
    # around K4p's chunk-start sums once parked its state in 864 bytes of scratch per lane: 2.6 -> 17.8 ms, again without a warning)
    # (round 6, deliberate: the 7-feature tile kernel on packed tiles is held to 256 registers for a second workgroup per CU -- 25-42 spilled
    #  REGISTERS, up to 172 bytes per lane, no array: 0.60 -> 0.40 ms; anything beyond 256 bytes there would be an array again)
    def spills_on_purpose(name, scratch):
        return "pols::k4c_kernel<" in name and ", 7, 0, 4, " in name and scratch <= 256

    bad = {k: v for k, v in ks.items() if any(h in k for h in hot) and v[0] > 0 and not spills_on_purpose(k, v[0])}
    assert not bad, sorted(bad.items())[:5]


def test_bench_kernels_keep_two_waves_per_simd():
    """The kernels BASELINE's configs are measured on sit at the edge of the register file: one value more and the allocator takes
    an AGPR, which halves the occupancy of gfx950's unified 512-entry file (the null-policy wave kernel went 80.7 -> 132 us that
    way, unnoticed until a profile run).  Read straight from the built code objects."""
    import sys

    sys.path.insert(0, str(ROOT / "scripts"))
    from check_scratch import LLVM, kernel_scratch

    if not (LLVM / "llvm-objdump").exists():
        pytest.skip("ROCm LLVM tools not present")
    if not _lib.LIB_PATH.exists():
        _lib.build()
    ks = kernel_scratch(_lib.LIB_PATH)
    want = {
        "pols::k1_kernel<float, 8, false, 256, 1, true, 2, false, true, false>(": 96,       # configs[1]: 256-thread team, two passes, nt loads
        "pols::k1_kernel<float, 8, false, 64, 4, true, 1, false, true, false>(": 256,       # ... and the wave-per-group form (ragged frames)
        "pols::k1_kernel<float, 8, false, 64, 4, true, 1, false, false, false>(": 256,
        "pols::k1_kernel_occ2<float, 8, false, 64, 4, true, 1, true>(": 256,         # configs[1] under a null policy
        "pols::k1_kernel<double, 8, true, 128, 4, true, 2, false, true, false>(": 256,      # configs[2]
        "pols::k1_kernel<float, 9, false, 64, 4, true, 3, false, false, false>(": 256,
        "pols::k1_kernel<float, 8, false, 64, 1, true, 1, false, false, true>(": 128,    # ragged year-sized groups: the EDGE wave kernel      # smoke(): 8 features + intercept
        "pols::k2_kernel<double, 16, 8, 2, true, false>(": 256,                             # configs[4]
        "pols::k3c_kernel<double, 6, 4, 4, 1>(": 168,                                       # configs[3] as RLS: three tiles (12 waves) per CU
        "pols::k3c_kernel<double, 6, 4, 4, 0>(": 128,                                       # ... its first pass: four waves per SIMD
        "pols::k3c_kernel<double, 6, 4, 4, 3>(": 168,                                       # ... the look-back-one form (round 6: what configs[3] runs)
        "pols::k4c_kernel<double, 6, 0, 4, false, true, false>(": 256,                      # configs[3] as rolling OLS: two four-wave workgroups per CU (own-halo form)
        "pols::k4c_kernel<double, 6, 1, 4, false, false, false>(": 256,                     # ... with a halo wave (POLS_ROLLING_ENGINE=halowave)
        "pols::k4c_kernel<double, 6, 0, 4, false, false, false>(": 256,                     # ... on packed tiles (many sequences)
        "pols::k4c_kernel<double, 6, 0, 4, true, false, false>(": 256,                      # ... "drop_window" with nulls (masked)
        "pols::k4c_kernel<double, 6, 0, 4, false, false, true>(": 256,                      # ... "drop" with nulls (gathered through the source map)
        "pols::k4c_kernel<double, 7, 0, 4, false, false, false>(": 256,                     # 7 features on packed tiles: held to two workgroups per CU (a few spills) since round 6
    }
    for key, cap in want.items():
        hits = [(k, v) for k, v in ks.items() if key in k]
        assert len(hits) == 1, (key, [k for k, _ in hits][:3])
        _scratch, vgpr, agpr = hits[0][1]
        assert agpr == 0 and 0 < vgpr <= cap, (hits[0][0], vgpr, agpr)
