#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg3|cfg4|cfg5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input, with the input columns already resident
in HBM when the timed region starts.  Default workload = BASELINE.json configs[1] (the configuration the metric is
quoted on): 10 000 groups x 1 000 rows x 8 features, f32, OLS, mode="predictions".  One process per GPU; groups are
independent, so every rank owns its own shard of groups (weak scaling, no data-path collective); for N > 1 the per-group
coefficient tables are all-gathered over RCCL/xGMI (the "reassemble the coefficients column" step of north_star), eight
steps' tables per collective, on a side stream so it overlaps the following kernels.  Prints ONE JSON line on rank 0.

--config selects the other BASELINE configs for the numbers quoted in DESIGN.md (same JSON shape):
  cfg3  10 000 x 1 000 x 8, f64, ridge alpha = 1 + sample_weights, predictions
  cfg4  1 000 000-row RLS, 6 features, half_life = 21, f64 (ONE sequence: a dependency chain, replicas only)
  cfg4r the other reading of configs[3]: 1 000 000-row rolling OLS, window = 252, 6 features, f64
  ref100 the reference's own benchmark shape: ONE 10 000 x 100 f64 OLS problem (published: 17.6 ms per call, M2 Max)
  cfg5  100 000 groups x 2 000 rows x 16 feats elastic net alpha = 0.001 l1_ratio = 0.5, f64; the groups are SPLIT
        across the ranks (strong scaling: 100 000 / N per GPU)
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

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


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


def make_columns(n: int, feats: int, tdt, seed: int, weights: bool = False):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    cols = [torch.randn(n, generator=gen, device="cuda", dtype=tdt) for _ in range(feats)]
    y = torch.zeros(n, device="cuda", dtype=tdt)
    for c in cols:
        y += c
    y += 0.1 * torch.randn(n, generator=gen, device="cuda", dtype=tdt)
    w = None
    if weights:
        w = torch.rand(n, generator=gen, device="cuda", dtype=tdt)
        w /= w.mean()
    return y, cols, w


def build_workload(cfg: str, eng, rank: int, world: int, dtype_flag: str):
    """Returns (plan, units_per_step_per_rank, unit_name, algorithmic_bytes_per_launch, workload_text, dtype, coef, scaling)."""
    if cfg == "cfg2":
        G, n, k = 10_000, 1_000, 8
        tdt = torch.float32 if dtype_flag == "f32" else torch.float64
        b = 4 if dtype_flag == "f32" else 8
        y, cols, _ = make_columns(G * n, k, tdt, 1234 + rank)
        out = {"pred": torch.empty(G * n, device="cuda", dtype=tdt), "coef": torch.empty(G, k, device="cuda", dtype=tdt)}
        plan = eng.plan_least_squares(y, cols, np.arange(G + 1, dtype=np.int64) * n, want=("pred", "coef"), out=out)
        text = (f"BASELINE configs[1]: {G} groups x {n} rows x {k} feats {dtype_flag} OLS mode=predictions "
                f"(+coefficients), inputs resident in HBM, per GPU")
        return plan, G, "regressions/s", b * n * (k + 1) * G + b * n * G, text, dtype_flag, out["coef"], "weak"
    if cfg == "cfg3":
        G, n, k = 10_000, 1_000, 8
        y, cols, w = make_columns(G * n, k, torch.float64, 1234 + rank, weights=True)
        out = {"pred": torch.empty(G * n, device="cuda", dtype=torch.float64),
               "coef": torch.empty(G, k, device="cuda", dtype=torch.float64)}
        plan = eng.plan_least_squares(y, cols, np.arange(G + 1, dtype=np.int64) * n, weights=w, alpha=1.0, l1_ratio=0.0,
                                      want=("pred", "coef"), out=out)
        text = f"BASELINE configs[2]: {G} groups x {n} rows x {k} feats f64 ridge alpha=1.0 + sample_weights, predictions, per GPU"
        return plan, G, "regressions/s", 8 * n * (k + 2) * G + 8 * n * G, text, "f64", out["coef"], "weak"
    if cfg == "cfg4":
        n, k = 1_000_000, 6
        y, cols, _ = make_columns(n, k, torch.float64, 1234 + rank)
        out = {"pred": torch.empty(n, device="cuda", dtype=torch.float64),
               "coef": torch.empty(n, k, device="cuda", dtype=torch.float64)}
        plan = eng.plan_recursive_least_squares(y, cols, np.array([0, n], dtype=np.int64), half_life=21.0, out=out)
        text = f"BASELINE configs[3]: ONE {n}-row sequence, {k} feats f64 RLS half_life=21 (coefficients + predictions); replicas only"
        return plan, n, "rows/s", 8 * n * (k + 1) + 8 * n * (k + 1), text, "f64", None, "weak"
    if cfg == "cfg4r":
        n, k = 1_000_000, 6
        y, cols, _ = make_columns(n, k, torch.float64, 1234 + rank)
        out = {"pred": torch.empty(n, device="cuda", dtype=torch.float64),
               "coef": torch.empty(n, k, device="cuda", dtype=torch.float64)}
        plan = eng.plan_rolling_least_squares(y, cols, np.array([0, n], dtype=np.int64), window_size=252, min_periods=6,
                                              null_policy="drop", out=out)
        text = f"BASELINE configs[3], second reading: ONE {n}-row sequence, {k} feats f64 rolling OLS window=252 (coefficients + predictions); replicas only"
        return plan, n, "rows/s", 8 * n * (k + 1) + 8 * n * (k + 1), text, "f64", None, "weak"
    if cfg == "ref100":
        n, k = 10_000, 100
        y, cols, _ = make_columns(n, k, torch.float64, 1234 + rank)
        out = {"pred": torch.empty(n, device="cuda", dtype=torch.float64), "coef": torch.empty(1, k, device="cuda", dtype=torch.float64)}
        plan = eng.plan_least_squares(y, cols, np.array([0, n], dtype=np.int64), want=("pred", "coef"), out=out)
        text = (f"the reference's own benchmark shape (tests/benchmark.py:219, README.md:229): ONE problem, {n} rows x {k} feats f64 OLS, "
                f"predictions; published 17.6 ms per call on an M2 Max incl. Polars overhead")
        return plan, 1, "problems/s", 8 * n * (k + 1) + 8 * n, text, "f64", None, "weak"
    if cfg == "cfg5":
        Gtot, n, k = 100_000, 2_000, 16
        G = Gtot // world
        y, cols, _ = make_columns(G * n, k, torch.float64, 1234 + rank)
        out = {"coef": torch.empty(G, k, device="cuda", dtype=torch.float64),
               "pred": torch.empty(G * n, device="cuda", dtype=torch.float64)}
        plan = eng.plan_least_squares(y, cols, np.arange(G + 1, dtype=np.int64) * n, alpha=0.001, l1_ratio=0.5,
                                      want=("coef", "pred"), out=out)
        text = (f"BASELINE configs[4]: {Gtot} groups x {n} rows x {k} feats f64 elastic net alpha=0.001 l1_ratio=0.5, "
                f"predictions (+coefficients), groups split over {world} GPU(s): {G} per GPU")
        # whole path, one launch: X and y read once (8 n (k + 1) bytes per group), predictions written (8 n)
        return plan, G, "regressions/s", 8 * n * (k + 1) * G + 8 * n * G, text, "f64", out["coef"], "strong"
    raise SystemExit(f"unknown --config {cfg}")


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg4r", "cfg5", "ref100"])
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("POLS_BENCH_FORCE_COLLECTIVE") == "1":   # the env knob exercises the N > 1 code on one GPU
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend_note = None
        try:
            dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        except Exception as exc:  # RCCL unavailable: keep the ranks in step over gloo (barrier + max), skip the coefficient gather
            backend_note = f"nccl init failed ({type(exc).__name__}); gloo used for barriers only"
            dist_mod.init_process_group("gloo", rank=rank, world_size=world)
        dist = dist_mod

    from polars_ols_amd import Engine

    eng = Engine(local_rank)
    # run the engine on a torch-visible stream so torch events / RCCL can order against it
    eng_stream = torch.cuda.Stream()
    eng.set_stream(eng_stream.cuda_stream)
    plan, units, unit_name, alg_bytes, text, dtype_name, coef, scaling = build_workload(args.config, eng, rank, world, args.dtype)
    torch.cuda.synchronize()                                                           # inputs are resident
    on_gloo = dist is not None and dist.get_backend() == "gloo"
    gather = dist is not None and coef is not None and not on_gloo
    collective_note = "none" if not on_gloo else backend_note
    RING = 8
    if gather:
        # Reassembling the coefficient column is the one exchange step of the path (north_star).  Fewer, larger collectives:
        # the coefficient tables of RING consecutive steps go into one ring buffer and ONE all-gather moves the whole ring
        # (RING x 320 KB per rank at cfg2) on a side stream while the next steps' kernels run.  Two rings alternate; a ring
        # is rewritten only after the gather that read it has finished -- checked on the HOST (the event is long complete in
        # steady state), not with a barrier packet on the engine stream, which would open a bubble per step.
        rings = [torch.empty((RING,) + tuple(coef.shape), device="cuda", dtype=coef.dtype) for _ in range(2)]
        slots = [[ring[i] for i in range(RING)] for ring in rings]      # views made once, not per step
        gathered = torch.empty((world * RING,) + tuple(coef.shape), device="cuda", dtype=coef.dtype)
        side = torch.cuda.Stream()
        produced = [torch.cuda.Event(), torch.cuda.Event()]
        consumed = [torch.cuda.Event(), torch.cuda.Event()]
        for ev in consumed:
            ev.record(side)
        collective_note = f"all_gather(coefficient tables of {RING} steps) on a side stream, overlapped"
    step_no = [0]

    def exchange(r, used=RING):
        nonlocal collective_note, gather
        try:
            produced[r].record(eng_stream)
            side.wait_event(produced[r])
            with torch.cuda.stream(side):
                # a partly filled ring (the flush after the last step) moves only the slots that were written
                dist.all_gather_into_tensor(gathered[: world * used], rings[r][:used])
                consumed[r].record(side)
        except Exception as exc:  # keep the benchmark alive: report the failure instead of dying
            gather = False
            collective_note = f"all_gather failed: {type(exc).__name__}: {exc}"[:200]

    def step():
        i = step_no[0]
        step_no[0] += 1
        if not gather:
            plan.run()
            return
        slot, r = i % RING, (i // RING) & 1
        if slot == 0:
            consumed[r].synchronize()
        plan.set_output("coef", slots[r][slot])
        plan.run()
        if slot == RING - 1:
            exchange(r)

    def flush():
        """gather the partly filled ring so that every timed step's coefficients have been reassembled"""
        i = step_no[0]
        if gather and i % RING != 0:
            exchange((i // RING) & 1, i % RING)
        step_no[0] = ((i + RING - 1) // RING) * RING

    for _ in range(args.warmup):
        step()
    flush()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    # per-launch HIP events inside the timed region, stamped by the kernel's own dispatch packet (hipExtLaunchKernelGGL); every
    # 4th launch is sampled: an event pair costs ~5 us on the stream's timeline, 6 % of this kernel
    eng.timing(4)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    flush()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = eng.timing_collect()
    eng.timing(False)

    if dist:
        t = torch.tensor([elapsed], device="cpu" if on_gloo else "cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * units * args.steps / elapsed
        k_ms = float(np.mean(kernel_ms)) if len(kernel_ms) else float("nan")
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        # HBM bytes per launch measured with rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in their own runs,
        # gfx950 correction applied) for this exact kernel + workload; committed under profiles/.  null if absent.
        traffic = None
        try:
            pmc = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text())
            traffic = pmc.get(eng.last_kernel, {}).get("traffic_bytes") if args.config == "cfg2" else None
        except Exception:
            traffic = None
        line = {
            "metric": "group_regressions_per_sec" if unit_name == "regressions/s" else ("rolling_rows_per_sec" if args.config == "cfg4r" else ("single_problems_per_sec" if args.config == "ref100" else "rls_rows_per_sec")),
            "value": value, "unit": unit_name, "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
            "config": {"workload": text, "units_per_gpu_per_step": units,
                       "sharding": "groups" if world > 1 else "none",
                       "collective": collective_note},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": eng.last_kernel,
                         "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg_bytes},
        }
        if args.config == "ref100":
            # BASELINE.md section 2 holds a published number for exactly this shape: 17.6 ms per call (OLS QR, 10 000 x 100, M2 Max,
            # through Polars + pyo3) = 56.8 problems/s.  Different hardware and it includes the Polars overhead: context, not a target.
            line["vs_baseline"] = value / (1.0 / 17.6e-3)
        if args.config in ("cfg4", "cfg4r"):
            line["roofline"]["note"] = ("single sequence: bound by the serial rank-1 update chain, not by HBM; "
                                        "achieved/peak only shows how far from memory-bound it is")
        if not args.no_cpu_baseline and world == 1 and args.config == "cfg2":
            line["cpu_baseline"] = cpu_baseline(1_000, 8)
        print(json.dumps(line), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
