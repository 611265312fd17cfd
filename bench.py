#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: BASELINE.json configs[1] --
10 000 groups x 1 000 rows x 8 features, f32, OLS, mode="predictions" -- with the input columns already
resident in HBM when the timed region starts.  One process per GPU; groups are independent, so every rank owns
its own 10 000-group shard (weak scaling, no data-path collective); for N > 1 each step also all-gathers the
per-group coefficient table over RCCL/xGMI (the "reassemble the coefficients column" step of north_star),
issued asynchronously so it overlaps the next step's kernel.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

GROUPS, ROWS, FEATS = 10_000, 1_000, 8
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes_per_group(rows: int, feats: int, itemsize: int, weights: bool = False) -> int:
    """read X (n*k) + y (n) [+ w (n)], write predictions (n)."""
    return itemsize * rows * (feats + 1 + (1 if weights else 0)) + itemsize * rows


def cpu_baseline(rows: int, feats: int, target_seconds: float = 12.0) -> dict:
    """Reference-equivalent CPU path (oracle/pols_oracle.c: per group the column->row-major marshal of
    src/expressions.rs:22-63, pivoted-QR solve_ols of src/least_squares.rs:195-240 and X.beta), OpenMP over
    groups on all host cores like Polars' rayon pool.  Bounded sample; f64 because the reference casts to f64."""
    from oracle import orc
    from refdata import synthetic_groups

    cores = orc.max_threads()
    sample_groups = max(2_000, 64 * cores)
    d = synthetic_groups(sample_groups, rows, feats, seed=1, dtype=np.float64)
    orc.batched_least_squares(d["y"], d["cols"], d["offsets"], n_threads=cores, want=("pred",))  # warm-up
    t0 = time.perf_counter()
    reps = 0
    while True:
        orc.batched_least_squares(d["y"], d["cols"], d["offsets"], n_threads=cores, want=("pred",))
        reps += 1
        if time.perf_counter() - t0 > target_seconds:
            break
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    orc.batched_least_squares(d["y"], d["cols"], d["offsets"], n_threads=1, want=("pred",))
    dt1 = time.perf_counter() - t1
    return {"value": sample_groups * reps / dt, "unit": "regressions/s", "cores": cores, "kind": "port",
            "sample": f"{reps} passes over {sample_groups} groups x {rows} rows x {feats} feats f64 "
                      f"(marshal + pivoted-QR + predictions), OpenMP {cores} threads; "
                      f"1-thread rate {sample_groups / dt1:.0f}/s"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        dist = dist_mod

    from polars_ols_amd import Engine

    tdt = torch.float32 if args.dtype == "f32" else torch.float64
    itemsize = 4 if args.dtype == "f32" else 8
    N = GROUPS * ROWS
    gen = torch.Generator(device="cuda").manual_seed(1234 + rank)
    cols = [torch.randn(N, generator=gen, device="cuda", dtype=tdt) for _ in range(FEATS)]
    y = sum(cols) + 0.1 * torch.randn(N, generator=gen, device="cuda", dtype=tdt)
    offsets = np.arange(GROUPS + 1, dtype=np.int64) * ROWS

    eng = Engine(local_rank)
    out = {"pred": torch.empty(N, device="cuda", dtype=tdt), "coef": torch.empty(GROUPS, FEATS, device="cuda", dtype=tdt)}
    gathered = torch.empty(world * GROUPS, FEATS, device="cuda", dtype=tdt) if world > 1 else None
    side = torch.cuda.Stream() if world > 1 else None
    plan = eng.plan_least_squares(y, cols, offsets, want=("pred", "coef"), out=out)   # marshal once
    torch.cuda.synchronize()                                                           # inputs are resident

    def step():
        plan.run()
        if world > 1:
            # hand the coefficient table to the collective stream; the next step's kernel overlaps the gather
            eng_stream_event.record(eng_stream)
            side.wait_event(eng_stream_event)
            with torch.cuda.stream(side):
                dist.all_gather_into_tensor(gathered, out["coef"])

    # run the engine on a torch-visible stream so torch events / RCCL can order against it
    eng_stream = torch.cuda.Stream()
    eng.set_stream(eng_stream.cuda_stream)
    eng_stream_event = torch.cuda.Event()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    eng.timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = eng.timing_collect()
    eng.timing(False)

    if dist:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * GROUPS * args.steps / elapsed
        k_ms = float(np.mean(kernel_ms)) if len(kernel_ms) else float("nan")
        bytes_per_launch = algorithmic_bytes_per_group(ROWS, FEATS, itemsize) * GROUPS
        achieved = bytes_per_launch / (k_ms * 1e-3) / 1e9
        # HBM bytes per launch measured with rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in their own runs,
        # gfx950 correction applied) for this exact kernel + workload; committed under profiles/.  null if absent.
        traffic = None
        try:
            pmc = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text())
            traffic = pmc.get(eng.last_kernel, {}).get("traffic_bytes")
        except Exception:
            traffic = None
        line = {
            "metric": "group_regressions_per_sec", "value": value, "unit": "regressions/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: {GROUPS} groups x {ROWS} rows x {FEATS} feats {args.dtype} OLS "
                                   f"mode=predictions (+coefficients), inputs resident in HBM, per GPU",
                       "groups_per_gpu": GROUPS, "rows_per_group": ROWS, "features": FEATS,
                       "sharding": "groups" if world > 1 else "none",
                       "collective": "all_gather(coefficients) overlapped" if world > 1 else "none"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": eng.last_kernel,
                         "kernel_ms": k_ms, "algorithmic_bytes_per_launch": bytes_per_launch},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(ROWS, FEATS)
        print(json.dumps(line), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
