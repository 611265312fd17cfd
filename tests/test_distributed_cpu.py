"""N > 1 host path on CPU: world_size-2 (and 3) `gloo` process groups exercising the group sharding and the
ragged gathers of polars_ols_amd/distributed.py.  The per-shard compute is stood in for by the CPU oracle (tests may
use it; the product never does) so the check is end-to-end: sharded result == single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from polars_ols_amd.distributed import partition_groups, shard_for_rank


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _frame(seed=0, n_groups=41, k=3):
    rng = np.random.default_rng(seed)
    sizes = rng.integers(0, 60, size=n_groups)
    sizes[5] = 0
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(offs[-1])
    cols = [rng.standard_normal(N) for _ in range(k)]
    y = sum(cols) + 0.1 * rng.standard_normal(N)
    return y, cols, offs


def _worker(rank, world, port, q):
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    sys.path[:0] = [str(root), str(root / "tests")]
    from oracle import orc
    from polars_ols_amd.distributed import gather_coefficients, gather_rows, shard_for_rank, slice_columns

    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    y, cols, offs = _frame()
    sh = shard_for_rank(offs, world, rank)
    ly, *lcols = slice_columns([y] + cols, sh)
    out = orc.batched_least_squares(ly, lcols, sh.offsets, alpha=0.5)        # stand-in for Engine.least_squares on this rank's GPU
    coef = gather_coefficients(torch.from_numpy(out["coef"]), sh)
    pred = gather_rows(torch.from_numpy(out["pred"]), sh, dst=0)
    if rank == 0:
        q.put((coef.numpy(), pred.numpy()))
    else:
        assert pred is None
        q.put((coef.numpy(), None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_equals_single_process(world):
    from oracle import orc

    y, cols, offs = _frame()
    ref = orc.batched_least_squares(y, cols, offs, alpha=0.5)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for coef, pred in results:
        assert np.array_equal(coef, ref["coef"])                 # every rank holds the full table, in group order
        if pred is not None:
            assert np.array_equal(pred, ref["pred"])             # root holds the full prediction column, in row order


def test_partition_is_contiguous_balanced_and_deterministic():
    offs = np.arange(10_001, dtype=np.int64) * 1_000            # BASELINE configs[1] frame
    for world in (1, 2, 4, 8):
        b = partition_groups(offs, world)
        assert b[0] == 0 and b[-1] == 10_000 and all(b[i] <= b[i + 1] for i in range(world))
        rows = [offs[b[r + 1]] - offs[b[r]] for r in range(world)]
        assert max(rows) - min(rows) <= 1_000
    rng = np.random.default_rng(1)
    offs = np.concatenate([[0], np.cumsum(rng.integers(0, 5000, 777))])
    shards = [shard_for_rank(offs, 8, r) for r in range(8)]
    assert shards[0].group_lo == 0 and shards[-1].group_hi == 777
    assert all(shards[r].group_hi == shards[r + 1].group_lo for r in range(7))
    assert sum(s.row_hi - s.row_lo for s in shards) == offs[-1]
    assert all(s.offsets[0] == 0 and s.offsets[-1] == s.row_hi - s.row_lo for s in shards)
    assert max(shards[0].row_counts) <= offs[-1] / 8 + 5000      # balanced to within one group


def test_more_ranks_than_groups():
    offs = np.array([0, 10, 20], dtype=np.int64)
    shards = [shard_for_rank(offs, 4, r) for r in range(4)]
    assert sum(s.group_hi - s.group_lo for s in shards) == 2


def _ring_worker(rank, world, port, q, steps):
    """bench.py's step / exchange / flush logic (CoefficientRing), with gloo's all_gather standing in for
    pols_comm_allgather_rows and no streams: every step writes a recognisable table; every gathered block is checked."""
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    sys.path[:0] = [str(root), str(root / "tests")]
    from polars_ols_amd.distributed import CoefficientRing, _all_gather_ragged, shard_for_rank

    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    offs = np.concatenate([[0], np.cumsum(np.random.default_rng(4).integers(1, 50, size=23))]).astype(np.int64)
    sh = shard_for_rank(offs, world, rank)
    G, k, RING = sh.group_hi - sh.group_lo, 3, 4
    seen, order = [], []

    def gather(view, used):                                   # view: [used, G, k]
        full = _all_gather_ragged(view.reshape(used * G, k), [used * c for c in sh.group_counts])
        seen.append((used, full.clone()))

    ring = CoefficientRing(lambda n: torch.zeros((n, G, k), dtype=torch.float64), RING, gather,
                           produced=lambda r: order.append(("produced", r)), wait_consumed=lambda r: order.append(("wait", r)),
                           consumed=lambda r: order.append(("consumed", r)))
    for step in range(steps):
        buf = ring.begin_step()
        buf[:] = 1000.0 * step + 10.0 * rank + torch.arange(G * k, dtype=torch.float64).reshape(G, k) / 1000.0   # "plan.run()"
        ring.end_step()
    ring.flush()
    # every step's table of every rank arrived, in step order within a gather and rank order within the world
    got_steps = 0
    for used, full in seen:
        pos = 0
        for r in range(world):
            Gr = sh.group_counts[r]
            block = full[pos:pos + used * Gr].reshape(used, Gr, k)
            for u in range(used):
                exp = 1000.0 * (got_steps + u) + 10.0 * r + torch.arange(Gr * k, dtype=torch.float64).reshape(Gr, k) / 1000.0
                assert torch.equal(block[u], exp), (rank, r, got_steps + u)
            pos += used * Gr
        got_steps += used
    assert got_steps == steps and ring.exchanges == len(seen) == -(-steps // RING)
    # a ring is waited for before it is rewritten, and marked consumed right behind its gather
    assert order[0] == ("wait", 0) and all(order[i + 1] == ("consumed", order[i][1]) for i, o in enumerate(order) if o[0] == "produced")
    assert ring.step_no % RING == 0                            # flush leaves the next step on a fresh ring
    q.put(len(seen))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("steps", [4, 11])
def test_bench_ring_logic_world2(steps):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ring_worker, args=(r, 2, port, q, steps)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1] == -(-steps // 4)


def _identity_worker(rank, world, port, q):
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    sys.path[:0] = [str(root), str(root / "tests")]
    from polars_ols_amd.distributed import check_gathered_table, collective_identity, gather_coefficients, shard_for_rank

    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ident = collective_identity(None)                              # the gloo twin: the group itself stands in for the communicator
    _, _, offs = _frame()
    sh = shard_for_rank(offs, world, rank)
    g = sh.group_hi - sh.group_lo
    local = torch.arange(g * 3, dtype=torch.float64).reshape(g, 3) + 1000.0 * rank
    whole = gather_coefficients(local, sh)
    good = check_gathered_table(whole, local, sh.group_counts)
    broken = whole.clone()
    broken[sh.group_counts[0]] += 1.0                              # first row of rank 1's slice
    bad = check_gathered_table(broken, local, sh.group_counts)
    q.put((rank, ident, good, bad))
    dist.barrier()
    dist.destroy_process_group()


def test_collective_identity_and_table_check_world2():
    """What bench.py puts into config.collective for N > 1 (nranks_seen, one entry per rank) and the gathered-table-equals-concatenation
    check, on the world-2 gloo twin: the fields are there, the rank count is the group's, a corrupted slice is pinned to its rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_identity_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(2)), key=lambda e: e[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ident, good, bad in res:
        assert ident["nranks_seen"] == 2 and ident["library"] == "gloo"
        assert [e["rank"] for e in ident["ranks"]] == [0, 1]
        assert set(ident["ranks"][0]) == {"rank", "device", "pci_bus_id"} and "rccl_version" in ident and "distinct_devices" in ident
        assert good == {"ok": True, "per_rank": [True, True]}
        assert bad == {"ok": False, "per_rank": [True, False]}


def test_collective_identity_single_process():
    from polars_ols_amd.distributed import check_gathered_table, collective_identity

    ident = collective_identity(None)
    assert ident["nranks_seen"] == 1 and len(ident["ranks"]) == 1
    t = torch.arange(12, dtype=torch.float64).reshape(4, 3)
    assert check_gathered_table(t, t, [4])["ok"] is True
    assert check_gathered_table(None, t, [4])["ok"] is None


def test_native_partition_matches_python():
    """pols_partition_groups (the C-ABI's partitioner, what a non-Python host calls) == distributed.partition_groups."""
    from polars_ols_amd.engine import partition_groups_native

    rng = np.random.default_rng(0)
    for trial in range(20):
        sizes = rng.integers(0, 2_000, size=int(rng.integers(1, 300)))
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        for world in (1, 2, 3, 8):
            assert partition_groups_native(offs, world) == partition_groups(offs, world)
    offs = np.arange(100_001, dtype=np.int64) * 2_000           # BASELINE configs[4]
    assert partition_groups_native(offs, 8) == [12_500 * r for r in range(9)]


def test_comm_entries_fail_cleanly_without_a_gpu():
    import ctypes as C

    from polars_ols_amd import _lib

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = _lib.lib()
    h = C.c_void_p()
    assert L.pols_comm_create(None, None, 2, 0, C.byref(h)) == -1 and not h.value    # POLS_ERR_INVALID: no context without a GPU
    assert L.pols_comm_world_size(None) == -1 and L.pols_comm_rank(None) == -1
    L.pols_comm_destroy(None)


def test_bench_self_launch_command():
    """`python bench.py --gpus N` with no launcher around it re-executes itself under torch.distributed.run in the driver's own command
    form (it used to exit at argument handling)."""
    import importlib.util
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    spec = importlib.util.spec_from_file_location("bench_for_test", root / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cmd = bench.self_launch_command(8, ["--gpus", "8", "--steps", "5", "--warmup", "2"])
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(str(root / "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    env = bench.self_launch_env()
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_gpus2_fails_inside_the_ranks_not_at_argument_handling():
    """No GPU here: `python bench.py --gpus 2` must get as far as the ranks (torch.distributed.run starts two of them, each dies
    selecting its device) instead of refusing the command line."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env["MASTER_PORT"] = str(_free_port())
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=str(root))
    assert r.returncode != 0
    err = r.stderr + r.stdout
    assert "launch with torch.distributed.run" not in err
    # the launcher ran and its children failed: torch.distributed.run reports the failed ranks
    assert "ChildFailedError" in err or "local_rank" in err or "exitcode" in err, err[-2000:]
