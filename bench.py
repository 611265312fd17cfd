#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg1|cfg2|cfg3|cfg4|cfg4r|rlsg|rlsgr|cfg5|ref100|rls100|roll100] [--mem device|host]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input, with the input columns already resident in HBM when the
timed region starts.  Default workload = BASELINE.json configs[1] (the configuration the metric is quoted on): 10 000 groups x
1 000 rows x 8 features, f32, OLS, mode="predictions".  Prints ONE JSON line on rank 0.

N > 1: one process per GPU.  The frame's groups are partitioned with the product's own partitioner
(polars_ols_amd.distributed.shard_for_rank: contiguous group ranges balanced by rows); every rank solves its shard -- no data-path
collective -- and the per-group coefficient tables are re-assembled on every rank through the product's communicator
(pols_comm_allgather_rows: RCCL over xGMI behind the C-ABI), eight steps' tables per collective, on a side stream so that it
overlaps the following kernels.  torch.distributed carries the rendezvous (the communicator's 128-byte id), the barriers and the
max-over-ranks of the elapsed time; if its RCCL backend or the product's communicator cannot be built, or a gather fails, the run
FAILS -- a multi-GPU line is never printed without its collective.

--config selects the other BASELINE configs (same JSON shape):
  cfg1  configs[0]: ONE group, 10 000 rows x 4 f64 features, mode="coefficients" (the reference's own per-plugin-call shape)
  cfg3  10 000 x 1 000 x 8, f64, ridge alpha = 1 + sample_weights, predictions
  cfg4  1 000 000-row RLS, 6 features, half_life = 21, f64 (ONE sequence: a dependency chain, replicas only)
  cfg4r the other reading of configs[3]: 1 000 000-row rolling OLS, window = 252, 6 features, f64
  rlsg  the dynamic models the way the reference is used (README.md:119-137, `.rls(...).over("group")`): 10 000 sequences x
        1 000 rows x 6 features, f64, RLS half_life = 21, coefficients + predictions; sequences shard across ranks
  rlsgr the same frame through rolling OLS, window = 252
  ref100 the reference's own benchmark shape: ONE 10 000 x 100 f64 OLS problem (published: 17.6 ms per call, M2 Max)
  rls100 / roll100 the reference's published dynamic rows (README.md:235-236): ONE 10 000-row sequence x 100 features, RLS half_life = 252
        (270 ms) / rolling OLS window = 252 (371 ms)
  cfg5  100 000 groups x 2 000 rows x 16 feats elastic net alpha = 0.001 l1_ratio = 0.5, f64; the groups are SPLIT
        across the ranks (strong scaling: 100 000 / N per GPU)
--frames F (default: as many as it takes for the inputs to exceed 3 x the 256 MB Infinity Cache, at least 3 for cfg2 / cfg3): the
  steps rotate over F independent synthetic frames of the same shape, so that no step can find its input in a cache -- the number is
  an HBM-stream number.  (cfg5's one frame is 27 GB.)
--gather coef|pred|none (N > 1; default coef): what is re-assembled every step.  coef: the per-group coefficient tables, all-gathered
  eight steps at a time (above).  pred: the PREDICTIONS column gathered to rank 0 with pols_comm_gather_rows every step (the root
  pulls 40 MB per peer per step over its xGMI links; double-buffered on a side stream) -- the exchange north_star names; it is
  bound by the links, which is why the product leaves predictions sharded unless asked.
--mem host (cfg1 / cfg2 / cfg3): the columns are host numpy arrays handed to the C-ABI as POLS_MEM_HOST -- the PCIe-inclusive
  rate ("data": "synthetic, host-resident"); never the headline value.
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


# ------------------------------------------------------------------------------------------------ CPU baseline (rank 0, N = 1)

def _thread_ladder(cores: int):
    return sorted({t for t in (1, 8, 32, cores) if t <= cores})


def cpu_baseline(cfg: str, target_seconds: float = 10.0) -> dict:
    """The reference-equivalent CPU path for this config (oracle/pols_oracle.c restates src/least_squares.rs + the marshalling of
    src/expressions.rs:22-63), timed INSIDE liborc on a bounded sample: buffers pre-allocated and first-touched outside the clock,
    groups dealt to OpenMP threads in static ranges (Polars' rayon pool analogue), f64 like the reference.  Reports the all-core
    rate with the column -> row-major marshalling copy (what a plugin call really does) plus, in `sample`, the rates at 1 / 8 / 32
    threads and the solve-only rate (pre-marshalled matrices: the variant that flatters the reference)."""
    from oracle import orc
    from refdata import synthetic_groups

    cores = orc.max_threads()
    if cfg in ("rls100", "roll100"):
        n, k = 10_000, 100
        rng = np.random.default_rng(1)
        cols = [rng.standard_normal(n) for _ in range(k)]
        y = np.sum(cols, axis=0) + 0.1 * rng.standard_normal(n)
        kind = "rls" if cfg == "rls100" else "rolling"
        one = orc.bench_dynamic(kind, y, cols, half_life=252.0, window=252, min_periods=k, passes=1)
        passes = int(max(1, min(20, target_seconds / max(one, 1e-3))))
        sec = orc.bench_dynamic(kind, y, cols, half_life=252.0, window=252, min_periods=k, passes=passes)
        return {"value": passes * n / sec, "unit": "rows/s", "cores": 1, "kind": "port",
                "sample": f"{passes} passes over ONE {n}-row sequence x {k} feats f64, "
                          f"{'solve_recursive_least_squares half_life=252' if kind == 'rls' else 'solve_rolling_ols window=252 min_periods=100 (Woodbury: k > 60)'} + "
                          f"dynamic predictions, timed inside liborc; a sequence is a dependency chain: one core"}
    if cfg in ("cfg4", "cfg4r"):
        n, k = 200_000, 6
        rng = np.random.default_rng(1)
        cols = [rng.standard_normal(n) for _ in range(k)]
        y = np.sum(cols, axis=0) + 0.1 * rng.standard_normal(n)
        kind = "rls" if cfg == "cfg4" else "rolling"
        orc.bench_dynamic(kind, y, cols, passes=1)
        sec = orc.bench_dynamic(kind, y, cols, passes=3)
        return {"value": 3 * n / sec, "unit": "rows/s", "cores": 1, "kind": "port",
                "sample": f"3 passes over ONE {n}-row sequence x {k} feats f64, "
                          f"{'solve_recursive_least_squares half_life=21' if kind == 'rls' else 'solve_rolling_ols window=252'} + "
                          f"dynamic predictions, timed inside liborc; a sequence is a dependency chain: one core"}
    if cfg in ("rlsg", "rlsgr"):
        G, n, k = max(512, 8 * cores), 1_000, 6
        rng = np.random.default_rng(1)
        cols = [rng.standard_normal(G * n) for _ in range(k)]
        y = np.sum(cols, axis=0) + 0.1 * rng.standard_normal(G * n)
        offs = np.arange(G + 1, dtype=np.int64) * n

        def run(threads):
            if cfg == "rlsg":
                orc.batched_rls(y, cols, offs, half_life=21.0, n_threads=threads)
            else:
                orc.batched_rolling(y, cols, offs, window_size=252, min_periods=6, null_policy="drop", n_threads=threads)

        rates = {}
        for t in _thread_ladder(cores):
            run(t)
            t0 = time.perf_counter()
            reps = 0
            while time.perf_counter() - t0 < target_seconds / 5:
                run(t)
                reps += 1
            rates[t] = reps * G * n / (time.perf_counter() - t0)
        best = max(rates, key=lambda t: rates[t])
        scal = ", ".join(f"{t} thr {rates[t]:.3g} rows/s" for t in sorted(rates))
        return {"value": rates[best], "unit": "rows/s", "cores": best, "kind": "port",
                "sample": f"{G} sequences x {n} rows x {k} feats f64, "
                          f"{'solve_recursive_least_squares half_life=21' if cfg == 'rlsg' else 'solve_rolling_ols window=252 (drop)'} + "
                          f"dynamic predictions (marshal ex.rs:22-63 included), one sequence per OpenMP task, wall clock around "
                          f"orc.batched_* (output arrays allocated inside); best of the thread ladder: {scal}"}
    shapes = {"cfg1": (1, 10_000, 4, {}, False), "cfg2": (max(2_048, 32 * cores), 1_000, 8, {}, False),
              "cfg3": (max(2_048, 32 * cores), 1_000, 8, dict(alpha=1.0, l1_ratio=0.0), True),
              "cfg5": (max(1_024, 8 * cores), 2_000, 16, dict(alpha=0.001, l1_ratio=0.5), False),
              "ref100": (1, 10_000, 100, {}, False)}
    G, n, k, kw, weighted = shapes[cfg]
    d = synthetic_groups(G, n, k, seed=1, dtype=np.float64, with_weights=weighted)
    w = d.get("w")
    method = {"cfg1": "pivoted-QR solve_ols", "cfg2": "pivoted-QR solve_ols", "ref100": "pivoted-QR solve_ols",
              "cfg3": "solve_ridge (X'X + aI, Cholesky)", "cfg5": "solve_elastic_net (residual-form cyclic CD)"}[cfg]

    def rate(threads, solve_only=False, budget=target_seconds / 6):
        t = min(threads, G)
        one = orc.bench_static(d["y"], d["cols"], d["offsets"], weights=w, passes=1, n_threads=t, solve_only=solve_only, **kw)
        passes = int(max(1, min(200, budget / max(one, 1e-6))))
        sec = orc.bench_static(d["y"], d["cols"], d["offsets"], weights=w, passes=passes, n_threads=t, solve_only=solve_only, **kw)
        return G * passes / sec, passes

    ladder = _thread_ladder(cores if G > 1 else 1)
    rates = {t: rate(t)[0] for t in ladder}
    best = max(ladder, key=lambda t: rates[t])               # past the socket's memory-bandwidth knee more threads are slower:
    full, passes = rate(best, budget=target_seconds / 3)     # the baseline is the BEST thread count, stated in `cores`
    solve_only, _ = rate(best, solve_only=True)
    scal = ", ".join(f"{t} thr {rates[t]:.0f}/s" for t in ladder)
    return {"value": full, "unit": "regressions/s" if G > 1 else "problems/s", "cores": best, "kind": "port",
            "sample": f"{passes} passes over {G} groups x {n} rows x {k} feats f64, marshal (ex.rs:22-63) + {method} + predictions, "
                      f"timed inside liborc, static OpenMP ranges, best of the thread ladder ({orc.max_threads()} hardware threads); "
                      f"scaling: {scal}; solve-only (pre-marshalled) {solve_only:.0f}/s at {best} threads"}


# ------------------------------------------------------------------------------------------------ workloads

def make_columns(n: int, feats: int, tdt, seed: int, weights: bool = False):
    """The synthetic frame of SURVEY.md section 8d, generated ON the device by the counter-based generator (synth.py: Philox-4x32-10 keyed by
    (seed; row, column)): x ~ N(0, 1), beta = 1, y = x.beta + 0.1 N(0, 1), w ~ U(0, 1) / mean.  A host regenerates any row range of it
    bit for bit (synth.frame_columns(seed, feats, lo, hi)) -- no device-to-host copy is needed to check a group."""
    import synth

    y, cols, w = synth.frame_columns(seed, feats, 0, n, dtype=tdt, device="cuda", weights=weights)
    if w is not None:
        w /= w.mean()
    return y, cols, w


def build_workload(cfg: str, eng, rank: int, world: int, dtype_flag: str, mem: str, frame: int = 0, groups: int = 0):
    """Returns dict(plan, units, unit, alg_bytes, text, dtype, coef, scaling, shard).  `frame`: which of the rotated synthetic frames
    (another seed, same shape)."""
    from polars_ols_amd.distributed import shard_for_rank

    host = mem == "host"

    def to_mem(t):
        return t.cpu().numpy() if host else t

    def grouped(G_total, n, k, tdt, b, **kw):
        # the GLOBAL frame's offsets, partitioned by the product's partitioner; this rank generates only its shard
        shard = shard_for_rank(np.arange(G_total + 1, dtype=np.int64) * n, world, rank)
        G = shard.group_hi - shard.group_lo
        y, cols, w = make_columns(G * n, k, tdt, 1234 + rank + 1000 * frame, weights=kw.pop("weights", False))
        want = kw.pop("want", ("pred", "coef"))
        out = None
        if not host:
            out = {"coef": torch.empty(G, k, device="cuda", dtype=tdt)}
            if "pred" in want:
                out["pred"] = torch.empty(G * n, device="cuda", dtype=tdt)
        # null_free: the synthetic columns hold no nulls, which a Polars caller reads off null_count (with sample weights the entry
        # would otherwise spend a pass over the weights column on the null-weight fill of least_squares.py:193)
        plan = eng.plan_least_squares(to_mem(y), [to_mem(c) for c in cols], shard.offsets, weights=None if w is None else to_mem(w),
                                      want=want, out=out, null_free=True, **kw)
        nbytes = b * n * (k + 1 + (1 if w is not None else 0)) * G + (b * n * G if "pred" in want else 0)
        return plan, G, nbytes, (out or plan.results).get("coef"), shard

    if cfg == "cfg1":
        plan, G, nbytes, coef, shard = grouped(1, 10_000, 4, torch.float64, 8, want=("coef",))
        text = "BASELINE configs[0]: ONE group, 10000 rows x 4 feats f64 OLS mode=coefficients (the per-plugin-call shape)"
        return dict(plan=plan, units=1, unit="problems/s", alg_bytes=nbytes, text=text, dtype="f64", coef=None, scaling="weak", shard=shard)
    if cfg == "cfg2":
        G_per, n, k = (groups or 10_000), 1_000, 8
        tdt = torch.float32 if dtype_flag == "f32" else torch.float64
        b = 4 if dtype_flag == "f32" else 8
        plan, G, nbytes, coef, shard = grouped(G_per * world, n, k, tdt, b)
        text = (f"BASELINE configs[1]: {G} groups x {n} rows x {k} feats {dtype_flag} OLS mode=predictions "
                f"(+coefficients), inputs resident in {'host memory (POLS_MEM_HOST)' if host else 'HBM'}, per GPU")
        return dict(plan=plan, units=G, unit="regressions/s", alg_bytes=nbytes, text=text, dtype=dtype_flag, coef=coef, scaling="weak", shard=shard)
    if cfg == "cfg3":
        G_per, n, k = (groups or 10_000), 1_000, 8
        plan, G, nbytes, coef, shard = grouped(G_per * world, n, k, torch.float64, 8, weights=True, alpha=1.0, l1_ratio=0.0)
        text = f"BASELINE configs[2]: {G} groups x {n} rows x {k} feats f64 ridge alpha=1.0 + sample_weights, predictions, per GPU"
        return dict(plan=plan, units=G, unit="regressions/s", alg_bytes=nbytes, text=text, dtype="f64", coef=coef, scaling="weak", shard=shard)
    if cfg in ("cfg4", "cfg4r"):
        n, k = 1_000_000, 6
        y, cols, _ = make_columns(n, k, torch.float64, 1234 + rank)
        out = {"pred": torch.empty(n, device="cuda", dtype=torch.float64), "coef": torch.empty(n, k, device="cuda", dtype=torch.float64)}
        if cfg == "cfg4":
            # null_free: the synthetic columns hold no nulls, which a Polars caller reads off null_count (the reference's own
            # validity mask is then an all-true bitmap, src/expressions.rs:201-228); without it the entry scans the columns
            plan = eng.plan_recursive_least_squares(y, cols, np.array([0, n], dtype=np.int64), half_life=21.0, out=out, null_free=True)
            text = f"BASELINE configs[3]: ONE {n}-row sequence, {k} feats f64 RLS half_life=21 (coefficients + predictions); replicas only"
        else:
            plan = eng.plan_rolling_least_squares(y, cols, np.array([0, n], dtype=np.int64), window_size=252, min_periods=6,
                                                  null_policy="drop", out=out, null_free=True)
            text = f"BASELINE configs[3], second reading: ONE {n}-row sequence, {k} feats f64 rolling OLS window=252 (coefficients + predictions); replicas only"
        return dict(plan=plan, units=n, unit="rows/s", alg_bytes=8 * n * (k + 1) + 8 * n * (k + 1), text=text, dtype="f64", coef=None,
                    scaling="weak", shard=None)
    if cfg in ("rlsg", "rlsgr"):
        G_per, n, k = 10_000, 1_000, 6
        shard = shard_for_rank(np.arange(G_per * world + 1, dtype=np.int64) * n, world, rank)
        G = shard.group_hi - shard.group_lo
        y, cols, _ = make_columns(G * n, k, torch.float64, 1234 + rank + 1000 * frame)
        out = {"pred": torch.empty(G * n, device="cuda", dtype=torch.float64), "coef": torch.empty(G * n, k, device="cuda", dtype=torch.float64)}
        if cfg == "rlsg":
            plan = eng.plan_recursive_least_squares(y, cols, shard.offsets, half_life=21.0, out=out, null_free=True)
            what = "RLS half_life=21"
        else:
            plan = eng.plan_rolling_least_squares(y, cols, shard.offsets, window_size=252, min_periods=6, null_policy="drop", out=out,
                                                  null_free=True)
            what = "rolling OLS window=252"
        text = (f"the dynamic models over groups (README.md:119-137): {G} sequences x {n} rows x {k} feats f64 {what} "
                f"(coefficients + predictions), per GPU")
        return dict(plan=plan, units=G * n, unit="rows/s", alg_bytes=16 * G * n * (k + 1), text=text, dtype="f64", coef=None,
                    scaling="weak", shard=None)
    if cfg == "ref100":
        n, k = 10_000, 100
        y, cols, _ = make_columns(n, k, torch.float64, 1234 + rank)
        out = {"pred": torch.empty(n, device="cuda", dtype=torch.float64), "coef": torch.empty(1, k, device="cuda", dtype=torch.float64)}
        plan = eng.plan_least_squares(y, cols, np.array([0, n], dtype=np.int64), want=("pred", "coef"), out=out)
        text = (f"the reference's own benchmark shape (tests/benchmark.py:219, README.md:229): ONE problem, {n} rows x {k} feats f64 OLS, "
                f"predictions; published 17.6 ms per call on an M2 Max incl. Polars overhead")
        return dict(plan=plan, units=1, unit="problems/s", alg_bytes=8 * n * (k + 1) + 8 * n, text=text, dtype="f64", coef=None, scaling="weak", shard=None)
    if cfg in ("rls100", "roll100"):
        # the reference's own published dynamic rows (README.md:235-236; tests/benchmark.py:146-172): ONE 10 000-row sequence x 100
        # features, f64 -- RLS half_life = 252 / rolling OLS window = 252, min_periods = 100, null_policy = "drop_window"; mode = predictions
        # (the plugin functions `recursive_least_squares` / `rolling_least_squares`), coefficients are written as well
        n, k = 10_000, 100
        y, cols, _ = make_columns(n, k, torch.float64, 1234 + rank)
        out = {"pred": torch.empty(n, device="cuda", dtype=torch.float64), "coef": torch.empty(n, k, device="cuda", dtype=torch.float64)}
        offs1 = np.array([0, n], dtype=np.int64)
        if cfg == "rls100":
            plan = eng.plan_recursive_least_squares(y, cols, offs1, half_life=252.0, out=out, null_free=True)
            text = (f"the reference's published RLS row (README.md:235, tests/benchmark.py:146-157): ONE {n}-row sequence x {k} feats f64, "
                    f"half_life=252; published 270 ms per call on an M2 Max incl. Polars overhead")
        else:
            plan = eng.plan_rolling_least_squares(y, cols, offs1, window_size=252, min_periods=k, null_policy="drop_window", out=out, null_free=True)
            text = (f"the reference's published rolling row (README.md:236, tests/benchmark.py:159-172): ONE {n}-row sequence x {k} feats f64, "
                    f"window=252 min_periods={k} drop_window; published 371 ms per call on an M2 Max incl. Polars overhead")
        return dict(plan=plan, units=n, unit="rows/s", alg_bytes=8 * n * (k + 1) + 8 * n * (k + 1), text=text, dtype="f64", coef=None,
                    scaling="weak", shard=None)
    if cfg == "cfg5":
        Gtot, n, k = (groups or 100_000), 2_000, 16
        plan, G, nbytes, coef, shard = grouped(Gtot, n, k, torch.float64, 8, alpha=0.001, l1_ratio=0.5, want=("coef", "pred"))
        text = (f"BASELINE configs[4]: {Gtot} groups x {n} rows x {k} feats f64 elastic net alpha=0.001 l1_ratio=0.5, "
                f"predictions (+coefficients), groups split over {world} GPU(s): {G} on this one")
        # whole path, one launch: X and y read once (8 n (k + 1) bytes per group), predictions written (8 n)
        return dict(plan=plan, units=G, unit="regressions/s", alg_bytes=nbytes, text=text, dtype="f64", coef=coef, scaling="strong", shard=shard)
    raise SystemExit(f"unknown --config {cfg}")


def self_launch_command(n_gpus: int, argv: list) -> list:
    """argv of the launcher `bench.py --gpus N` turns itself into when nothing has launched it (no WORLD_SIZE): the driver's own
    command form, on 127.0.0.1 (the container's hostname may not resolve)."""
    port = os.environ.get("MASTER_PORT", "29531")
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), str(Path(__file__).resolve())] + list(argv)


def self_launch_env() -> dict:
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return env


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--config", default="cfg2", choices=["cfg1", "cfg2", "cfg3", "cfg4", "cfg4r", "rlsg", "rlsgr", "cfg5", "ref100", "rls100", "roll100"])
    ap.add_argument("--mem", default="device", choices=["device", "host"])
    ap.add_argument("--frames", type=int, default=0, help="rotate the steps over this many independent frames (0: automatic)")
    ap.add_argument("--groups", type=int, default=0, help="cfg2 / cfg3 / cfg5: this many groups (per GPU for cfg2 / cfg3, in total for cfg5) "
                                                         "instead of the config's own count -- e.g. 12500 = the 8-GPU shard of cfg5 on one GPU")
    ap.add_argument("--gather", default="coef", choices=["coef", "pred", "none"], help="N > 1: what is re-assembled every step")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher.  One rank per GPU under torch.distributed.run with the
        # same arguments; rank 0 still prints the ONE JSON line (the children inherit stdout).  A box with fewer than N GPUs fails
        # INSIDE the ranks (torch.cuda.set_device: invalid device ordinal), never here.
        os.execvpe(sys.executable, self_launch_command(args.gpus, sys.argv[1:]), self_launch_env())
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    if args.mem == "host" and (world > 1 or args.config not in ("cfg1", "cfg2", "cfg3")):
        raise SystemExit("--mem host: cfg1 / cfg2 / cfg3 on one GPU")
    # stdout carries ONE JSON line and nothing else: libraries that write to file descriptor 1 on their own (RCCL prints a version banner
    # when a communicator is created) are pointed at stderr for the whole run; the line goes out through the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    torch.cuda.set_device(local_rank)
    dist = None
    force_collective = os.environ.get("POLS_BENCH_FORCE_COLLECTIVE") == "1"   # exercises the N > 1 code path on one GPU
    if world > 1 or force_collective:
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # no fallback: if RCCL cannot initialise, the multi-GPU run fails here
        dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        dist = dist_mod

    from polars_ols_amd import Engine
    from polars_ols_amd.distributed import CoefficientRing, check_gathered_table, collective_identity, create_comm

    eng = Engine(local_rank)
    # run the engine on a torch-visible stream so torch events can order the collective's stream against it
    eng_stream = torch.cuda.Stream()
    eng.set_stream(eng_stream.cuda_stream)
    wl = build_workload(args.config, eng, rank, world, args.dtype, args.mem, groups=args.groups)
    plan, coef, shard = wl["plan"], wl["coef"], wl["shard"]
    # Frame rotation: a 360 MB frame re-read every step could live in part in the 256 MB Infinity Cache; rotating over frames whose
    # inputs add up to more than three times that makes every step stream its input from HBM.  Outputs go to the first frame's
    # buffers (written, never read).
    in_bytes = wl["alg_bytes"]
    n_frames = args.frames if args.frames > 0 else (1 if (args.config in ("cfg1", "cfg4", "cfg4r", "rlsg", "rlsgr", "ref100", "rls100", "roll100") or args.mem == "host")
                                                    else max(1, min(8, -(-3 * 256 * 2 ** 20 // max(1, in_bytes)))))
    if args.frames == 0 and args.config in ("cfg2", "cfg3") and args.mem == "device":
        n_frames = max(3, n_frames)
    plans = [plan]
    for f in range(1, n_frames):
        w2 = build_workload(args.config, eng, rank, world, args.dtype, args.mem, frame=f, groups=args.groups)
        for key in ("coef", "pred", "resid"):
            if key in plan.results and key in w2["plan"].results:
                w2["plan"].set_output(key, plan.results[key])
        plans.append(w2["plan"])
    torch.cuda.synchronize()                                                           # inputs are resident
    gather = dist is not None and coef is not None and args.gather == "coef"
    gather_pred = dist is not None and args.gather == "pred" and "pred" in plan.results and shard is not None
    collective = {"kind": "none", "backend": None, "bytes_per_step_per_rank": 0}
    RING = 8
    ring = None
    if gather:
        # the product's communicator, on its own context / side stream so that a gather overlaps the next steps' kernels
        side = torch.cuda.Stream()
        eng_comm = Engine(local_rank)
        eng_comm.set_stream(side.cuda_stream)
        comm = create_comm(eng_comm)                                                   # raises if RCCL / the communicator fails
        counts = shard.group_counts                                                    # groups per rank (from the partitioner)
        k = coef.shape[1]
        gathered = torch.empty((RING * sum(counts), k), device="cuda", dtype=coef.dtype)
        produced = [torch.cuda.Event(), torch.cuda.Event()]
        consumed = [torch.cuda.Event(), torch.cuda.Event()]
        for ev in consumed:
            ev.record(side)

        def do_gather(ring_view, used):
            # ONE collective for `used` steps' tables: every rank contributes used * (its groups) rows of k coefficients
            comm.allgather_rows(ring_view.reshape(used * counts[rank], k), [used * c for c in counts], out=gathered[: used * sum(counts)])

        def mark_produced(r):
            produced[r].record(eng_stream)
            side.wait_event(produced[r])

        ring = CoefficientRing(lambda n: torch.empty((n,) + tuple(coef.shape), device="cuda", dtype=coef.dtype), RING, do_gather,
                               produced=mark_produced, wait_consumed=lambda r: consumed[r].synchronize(),
                               consumed=lambda r: consumed[r].record(side))
        collective = {"kind": f"pols_comm_allgather_rows (RCCL behind the C-ABI) of {RING} steps' coefficient tables, side stream",
                      "backend": dist.get_backend(), "bytes_per_step_per_rank": int(coef.numel() * coef.element_size()) * (world - 1)}
        collective.update(collective_identity(comm))          # what RCCL itself saw: rank count, every rank's device + PCI bus id

    pred_state = None
    if gather_pred:
        # predictions to rank 0 every step (pols_comm_gather_rows: grouped ncclSend / ncclRecv, the root pulls from its peers over
        # distinct xGMI links), on a side stream, double-buffered: step i + 2 rewrites buffer i % 2 only after gather i has read it
        side = torch.cuda.Stream()
        eng_comm = Engine(local_rank)
        eng_comm.set_stream(side.cuda_stream)
        comm = create_comm(eng_comm)
        pred0 = plan.results["pred"]
        bufs = [pred0, torch.empty_like(pred0)]
        rows = shard.row_counts
        gathered_pred = torch.empty(sum(rows), device="cuda", dtype=pred0.dtype) if rank == 0 else None
        produced = [torch.cuda.Event(), torch.cuda.Event()]
        consumed = [torch.cuda.Event(), torch.cuda.Event()]
        for ev in consumed:
            ev.record(side)
        pred_state = dict(i=0)
        collective = {"kind": "pols_comm_gather_rows (RCCL behind the C-ABI) of the predictions column to rank 0, every step, side stream",
                      "backend": dist.get_backend(), "bytes_per_step_per_rank": int(pred0.numel() * pred0.element_size())}
        collective.update(collective_identity(comm))

    step_no = [0]

    def step():
        pl = plans[step_no[0] % len(plans)]
        step_no[0] += 1
        if pred_state is not None:
            i = pred_state["i"]; pred_state["i"] = i + 1
            consumed[i & 1].synchronize()
            pl.set_output("pred", bufs[i & 1])
            pl.run()
            produced[i & 1].record(eng_stream)
            side.wait_event(produced[i & 1])
            comm.gather_rows(bufs[i & 1], rows, root=0, out=gathered_pred)
            consumed[i & 1].record(side)
            return
        if ring is None:
            pl.run()
            return
        pl.set_output("coef", ring.begin_step())
        pl.run()
        ring.end_step()

    def flush():
        if ring is not None:
            ring.flush()

    for _ in range(args.warmup):
        step()
    flush()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    # per-launch HIP events inside the timed region, stamped by the kernel's own dispatch packet (hipExtLaunchKernelGGL); every
    # 8th launch is sampled (launches 0, 8, 16, ...): an event-bracketed launch cannot overlap its neighbours' ramp-up / tail and costs
    # ~5 us on the stream's timeline, 6 % of the headline kernel
    stride = int(os.environ.get("POLS_BENCH_EVENT_STRIDE", "8"))
    eng.timing(stride)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    flush()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # the re-assembled table against its parts, outside the timed region: one more gather of the current step's table, then every rank's
    # checksum of its own rows against the checksum of its slice of what arrived (collective: all ranks take part, rank 0 asserts)
    if gather:
        local_tab = plan.results["coef"]
        side.wait_stream(eng_stream)
        whole = comm.allgather_rows(local_tab, counts)
        side.synchronize()
        chk = check_gathered_table(whole, local_tab, counts)
        collective["gathered_equals_concatenation"] = chk["ok"]
        assert chk["ok"], f"rank {rank}: the gathered coefficient table is not the concatenation of the shards: {chk}"
    elif gather_pred:
        side.synchronize()
        last = bufs[(pred_state["i"] - 1) & 1]
        chk = check_gathered_table(gathered_pred, last, rows)
        collective["gathered_equals_concatenation"] = chk["ok"]
        assert chk["ok"] is not False, f"rank {rank}: the gathered predictions are not the concatenation of the shards: {chk}"
    elif world == 1 and not os.environ.get("POLS_BENCH_NO_COMM_PROBE"):
        # N = 1: a world of one through the same entries, so that the line says what the communicator saw here too
        try:
            from polars_ols_amd.engine import Comm

            c1 = Comm(eng, 1, 0, Comm.unique_id())
            collective.update(collective_identity(c1))
            c1.close()
        except Exception as exc:                                                       # RCCL missing on a one-GPU box is not a bench failure
            collective["nranks_seen"] = None
            collective["identity_error"] = str(exc)[:200]
    kernel_ms = list(eng.timing_collect())
    eng.timing(False)
    kernel_name = eng.last_kernel
    # at least 16 kernel-duration samples whatever --steps is: a sampled pass over the same frames right behind the timed region
    # (outside it) with the SAME stride -- an event-bracketed launch cannot overlap its neighbours' ramp-up / tail, and the fewer
    # un-bracketed launches sit between two bracketed ones the longer a bracketed one measures (stride 2: 76.4 us, stride 8: 74 us,
    # rocprofv3 on every launch: 72.2 us for the headline kernel)
    MIN_SAMPLES = 16
    samples_in_region = len(kernel_ms)
    if len(kernel_ms) < MIN_SAMPLES and pred_state is None and ring is None:
        eng.timing(stride)
        for _ in range(stride * (MIN_SAMPLES - len(kernel_ms)) + 1):
            step()
        torch.cuda.synchronize()
        kernel_ms += list(eng.timing_collect())
        eng.timing(False)

    # the bandwidth ceiling of this workload's traffic mix, measured on the same frames right behind the timed region (outside it):
    # pols_stream_probe reads every input column and writes the predictions column with the arithmetic removed, sampled by the
    # same per-launch HIP events.  Static device-resident configs at N = 1 only.
    probe_ms = None
    probe_persistent_ms = None
    if world == 1 and args.mem == "device" and args.config in ("cfg2", "cfg3", "cfg5") and all("pred" in p.results for p in plans):
        def probe(mode):
            for p in plans:
                p.stream_probe(mode)
            torch.cuda.synchronize()
            eng.timing(stride)
            for i in range(max(args.steps, stride * MIN_SAMPLES, 3 * len(plans))):
                plans[i % len(plans)].stream_probe(mode)
            torch.cuda.synchronize()
            pm = eng.timing_collect()
            eng.timing(False)
            return float(np.mean(pm)) if len(pm) else None

        probe_ms = probe(0)
        probe_persistent_ms = probe(1)

    total_units = float(wl["units"])
    if dist:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        u = torch.tensor([float(wl["units"])], device="cuda", dtype=torch.float64)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        total_units = float(u.item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = total_units * args.steps / elapsed
        k_ms = float(np.mean(kernel_ms)) if len(kernel_ms) else float("nan")
        achieved = wl["alg_bytes"] / (k_ms * 1e-3) / 1e9
        # HBM bytes per launch measured with rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in their own runs, gfx950 correction
        # applied) for this exact kernel + workload; committed under profiles/ (pmc_traffic.json).  null if absent.
        traffic = None
        traffic_source = None
        try:
            pmc = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text())
            entry = pmc.get(f"{kernel_name}|{args.config}") or pmc.get(kernel_name, {})
            if entry.get("config", "cfg2") == args.config and args.mem == "device" and world == 1 and not args.groups:
                traffic = entry.get("traffic_bytes")
                traffic_source = {"file": "profiles/pmc_traffic.json", "kernel": kernel_name, "round": entry.get("round"),
                                  "from": entry.get("source")}
        except Exception:
            traffic = None
        if traffic is None and args.mem == "device" and world == 1:
            print(f"bench.py: no PMC traffic entry for kernel {kernel_name!r} / {args.config} in profiles/pmc_traffic.json -- "
                  f"roofline.traffic is null", file=sys.stderr)
        unit_name = wl["unit"]
        line = {
            "metric": {"regressions/s": "group_regressions_per_sec", "problems/s": "single_problems_per_sec"}.get(
                unit_name, "rolling_rows_per_sec" if args.config in ("cfg4r", "rlsgr", "roll100") else "rls_rows_per_sec"),
            "value": value, "unit": unit_name, "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": wl["scaling"], "vs_baseline": None, "dtype": wl["dtype"],
            "data": "synthetic" if args.mem == "device" else "synthetic, host-resident (PCIe-inclusive)",
            "config": {"workload": wl["text"], "units_per_gpu_per_step": wl["units"], "frames_rotated": n_frames,
                       "sharding": "groups (shard_for_rank: contiguous ranges balanced by rows)" if world > 1 else "none",
                       "world_size": world, "collective": collective},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source, "kernel": kernel_name,
                         "kernel_ms": k_ms, "kernel_samples": len(kernel_ms), "kernel_samples_in_timed_region": samples_in_region,
                         "algorithmic_bytes_per_launch": wl["alg_bytes"]},
        }
        if probe_ms:
            ceiling = wl["alg_bytes"] / (probe_ms * 1e-3) / 1e9
            line["roofline"]["stream_ceiling"] = {
                "GBps": ceiling, "kernel_ms": probe_ms, "frac_of_peak": ceiling / HBM_PEAK_GBS, "achieved_over_ceiling": achieved / ceiling,
                "what": "pols_stream_probe_ex mode 0: the same columns read and the predictions column written with streaming 16-byte "
                        "accesses, no arithmetic, in the launch shape of the resident kernels (one piece per lane, exits); same frames, "
                        "event-sampled, measured right behind the timed region"}
            if probe_persistent_ms:
                pc = wl["alg_bytes"] / (probe_persistent_ms * 1e-3) / 1e9
                line["roofline"]["stream_ceiling"]["persistent"] = {
                    "GBps": pc, "kernel_ms": probe_persistent_ms, "frac_of_peak": pc / HBM_PEAK_GBS, "achieved_over_ceiling": achieved / pc,
                    "what": "mode 1: the same traffic as a persistent grid-stride stream (4 workgroups per CU, the next piece's loads "
                            "issued before the current piece is stored): no dispatch ramp, no tail"}
        if args.config == "ref100":
            # BASELINE.md section 2 holds a published number for exactly this shape: 17.6 ms per call (OLS QR, 10 000 x 100, M2 Max,
            # through Polars + pyo3) = 56.8 problems/s.  Different hardware and it includes the Polars overhead: context, not a target.
            line["vs_baseline"] = value / (1.0 / 17.6e-3)
        if args.config in ("rls100", "roll100"):
            published_ms = 270.0 if args.config == "rls100" else 371.0     # BASELINE.md section 2 / README.md:235-236 (M2 Max, through Polars)
            line["vs_baseline"] = value / (10_000 / (published_ms * 1e-3))
            line["roofline"]["note"] = ("ONE sequence of 100 features: 157 chunks of 64 rows, one workgroup each, the 100 x 100 inverse propagated "
                                        "in LDS (O(k^2) per row, a dependency chain inside a chunk) -- latency-bound, nowhere near HBM; "
                                        "achieved/peak only shows how far")
        if args.config in ("cfg4", "cfg4r", "rlsg", "rlsgr"):
            line["roofline"]["note"] = ("dynamic models: algorithmic bytes = every input column read once + coefficients and predictions "
                                        "written once (16 (k + 1) bytes per row, f64); the sequence is a scan, not a chain -- rows are "
                                        "processed in parallel and the bound is HBM")
        if args.config in ("cfg1", "ref100") or args.mem == "host":
            line["roofline"]["note"] = ("one small problem / host-resident data: launch- or PCIe-bound; achieved/peak only shows how far "
                                        "from HBM-bound it is")
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.config)
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    os.close(json_fd)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
