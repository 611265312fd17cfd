"""bench.py honours the driver's contract: ONE JSON line with the metric, roofline and (default run) cpu_baseline objects; the
N > 1 code path (ring-buffered coefficient gather over RCCL on a side stream) is exercised on one GPU with a 1-rank group."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline")


def _run(extra_env, *args):
    env = dict(os.environ, **extra_env)
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "12", "--warmup", "3", *args], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_line_contract():
    d = _run({}, "--no-cpu-baseline")
    for key in REQUIRED:
        assert key in d, key
    assert d["metric"] == "group_regressions_per_sec" and d["unit"] == "regressions/s" and d["dtype"] == "f32"
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 3 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert "10000 groups x 1000 rows x 8 feats" in d["config"]["workload"] and d["config"]["collective"] == "none"
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["kernel"].startswith("k1_gram_chol_f32_k8")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.3 < r["frac"] < 1.0
    assert r["algorithmic_bytes_per_launch"] == 400_000_000 and (r["traffic"] is None or 0.9 < r["traffic"] / 4e8 < 1.3)
    assert 0.5 < d["value"] * d["ms_per_step"] * 1e-3 / 10_000 < 1.5       # value == groups / step time


def test_bench_collective_path_on_one_gpu():
    d = _run({"POLS_BENCH_FORCE_COLLECTIVE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"}, "--no-cpu-baseline")
    assert d["config"]["collective"].startswith("all_gather(coefficient tables of 8 steps)"), d["config"]
    assert d["value"] > 1e7


def test_native_cabi_harness(tmp_path):
    """examples/cabi_bench.cpp: the hot path from native code through the C-ABI alone (no Python / PyTorch in the process); the
    harness checks group 0 against a double-precision normal-equation solve itself and exits non-zero on a mismatch."""
    import shutil

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "cabi_bench"
    lib = ROOT / "polars_ols_amd"
    subprocess.run([hipcc, "-O2", "-std=c++17", "-o", str(exe), str(ROOT / "examples" / "cabi_bench.cpp"), f"-L{lib}", "-lpols_mi355x",
                    f"-Wl,-rpath,{lib}"], check=True, timeout=600)
    out = subprocess.run([str(exe), "2000", "1000", "8", "20"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["kernel"].startswith("k1_gram_chol_f32_k8") and d["groups_not_ok"] == 0
    assert d["max_abs_dcoef_group0"] < 1e-4 and d["max_abs_dpred_group0"] < 1e-4 and d["regressions_per_s"] > 1e6
