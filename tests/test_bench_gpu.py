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
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout       # ONE JSON line and nothing else (RCCL's banner goes to stderr)
    return json.loads(lines[0])


def test_bench_line_contract():
    d = _run({}, "--no-cpu-baseline")
    for key in REQUIRED:
        assert key in d, key
    assert d["metric"] == "group_regressions_per_sec" and d["unit"] == "regressions/s" and d["dtype"] == "f32"
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 3 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert "10000 groups x 1000 rows x 8 feats" in d["config"]["workload"] and d["config"]["collective"]["kind"] == "none"
    assert d["config"]["world_size"] == 1
    # N = 1 says what a world-of-one communicator saw, through the same entries the N > 1 line uses (pols_comm_query)
    col = d["config"]["collective"]
    assert col.get("nranks_seen") == 1 and len(col["ranks"]) == 1 and col["ranks"][0]["rank"] == 0 and col["ranks"][0]["pci_bus_id"], col
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["kernel"].startswith("k1_gram_chol_f32_k8")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.3 < r["frac"] < 1.0
    assert r["algorithmic_bytes_per_launch"] == 400_000_000 and (r["traffic"] is None or 0.9 < r["traffic"] / 4e8 < 1.3)
    assert 0.5 < d["value"] * d["ms_per_step"] * 1e-3 / 10_000 < 1.5       # value == groups / step time
    assert d["config"]["frames_rotated"] >= 3                              # every step streams its input from HBM, not from a cache
    c = r["stream_ceiling"]                                                # the copy ceiling of the same traffic mix, same frames
    assert 0.5 < c["frac_of_peak"] < 1.0 and abs(c["achieved_over_ceiling"] - r["achieved"] / c["GBps"]) < 1e-9
    assert 0.7 < c["achieved_over_ceiling"] < 1.15, c


def test_stream_probe_reads_every_column_and_writes_their_sum():
    import numpy as np
    import torch

    from polars_ols_amd import Engine

    eng = Engine(0)
    try:
        for dt in (torch.float32, torch.float64):
            n = 100_003
            g = torch.Generator(device="cuda").manual_seed(1)
            cols = [torch.randn(n, device="cuda", generator=g, dtype=dt) for _ in range(11)]
            y = torch.randn(n, device="cuda", generator=g, dtype=dt)
            w = torch.rand(n, device="cuda", generator=g, dtype=dt) + 0.5
            plan = eng.plan_least_squares(y, cols, np.array([0, n], dtype=np.int64), weights=w, want=("pred",))
            plan.stream_probe()
            eng.synchronize()
            assert eng.last_kernel.startswith("stream_probe_")
            exp = sum(c.double() for c in cols) + y.double() + w.double()
            assert torch.allclose(plan.results["pred"].double(), exp, rtol=1e-5, atol=1e-5)
    finally:
        eng.close()


def test_bench_collective_path_on_one_gpu():
    d = _run({"POLS_BENCH_FORCE_COLLECTIVE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"}, "--no-cpu-baseline")
    c = d["config"]["collective"]
    assert c["kind"].startswith("pols_comm_allgather_rows") and c["backend"] == "nccl", d["config"]
    assert d["value"] > 1e7
    # the line proves its collective: what RCCL reports for the communicator, and the gathered table checked against its parts
    assert c["nranks_seen"] == 1 and c["library"].startswith("rccl") and c["rccl_version"] > 0 and c["distinct_devices"] == 1, c
    assert c["gathered_equals_concatenation"] is True


def test_bench_predictions_gather_on_one_gpu():
    """--gather pred: the predictions column re-assembled on rank 0 every step (pols_comm_gather_rows), here as a world of one."""
    d = _run({"POLS_BENCH_FORCE_COLLECTIVE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29534"}, "--no-cpu-baseline", "--gather", "pred")
    c = d["config"]["collective"]
    assert c["kind"].startswith("pols_comm_gather_rows") and c["backend"] == "nccl" and c["bytes_per_step_per_rank"] == 40_000_000, d["config"]
    assert d["value"] > 1e6
    assert c["nranks_seen"] == 1 and c["gathered_equals_concatenation"] is True, c


@pytest.mark.parametrize("cfg,metric,kernel", [("cfg1", "single_problems_per_sec", "k5_gram_stream"), ("cfg5", "group_regressions_per_sec", "k2_gram_mfma_resident_f64_k16yv_w8_rc2_cd")])
def test_bench_other_configs(cfg, metric, kernel):
    """--config cfg1 (BASELINE configs[0]: the per-plugin-call shape) and cfg5 (configs[4]: the whole fused path's bytes)."""
    d = _run({}, "--no-cpu-baseline", "--config", cfg)
    assert d["metric"] == metric and d["dtype"] == "f64" and d["roofline"]["kernel"].startswith(kernel), d["roofline"]
    if cfg == "cfg5":
        assert d["roofline"]["algorithmic_bytes_per_launch"] == 100_000 * (2_000 * 17 * 8 + 2_000 * 8)   # X, y in; predictions out
        assert d["ms_per_step"] < 9.0


def test_bench_host_memory_mode():
    """--mem host: the same workload with host numpy columns through POLS_MEM_HOST -- the PCIe-inclusive rate, labelled as such."""
    d = _run({}, "--no-cpu-baseline", "--mem", "host")
    assert "host-resident" in d["data"] and "host memory" in d["config"]["workload"] and d["value"] > 1e5


def test_comm_world_of_one(tmp_path):
    """pols_comm_* on a 1-rank world (all a single GPU can run): create from a unique id, all-gather and gather are copies."""
    import numpy as np
    import torch

    from polars_ols_amd import Engine
    from polars_ols_amd.engine import Comm

    eng = Engine(0)
    comm = Comm(eng, 1, 0, Comm.unique_id())
    assert comm.world == 1 and comm.rank == 0
    local = torch.randn(1000, 8, device="cuda", dtype=torch.float64)
    out = comm.allgather_rows(local, [1000])
    root = comm.gather_rows(local[:, 0].contiguous(), [1000], root=0)
    eng.synchronize()
    assert torch.equal(out, local) and torch.equal(root, local[:, 0])
    comm.close()
    eng.close()


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
