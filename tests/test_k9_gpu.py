"""K9 group-key ingestion (``pols_layout_*``): `.over(key)` partitioning on the device, against numpy's stable argsort.

Integer / index work: the bar is bit-exact (order, offsets, keys, moved columns)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def engine():
    from polars_ols_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _keys(kind: str, n: int, rng) -> np.ndarray:
    if kind == "dense":
        return rng.integers(0, max(n // 50, 1), size=n).astype(np.int64)
    if kind == "negative":
        return rng.integers(-1000, 1000, size=n).astype(np.int64)
    if kind == "wide":                    # more than 32 bits of range: the 64-bit radix keys
        return rng.choice(np.array([-(1 << 62), -5, 0, 7, (1 << 40) + 3, (1 << 62) + 11], dtype=np.int64), size=n)
    if kind == "extremes":                # range == 2^64 - 1
        return rng.choice(np.array([np.iinfo(np.int64).min, -1, 0, np.iinfo(np.int64).max], dtype=np.int64), size=n)
    if kind == "sorted":
        return np.sort(rng.integers(0, 37, size=n)).astype(np.int64)
    if kind == "one":
        return np.full(n, 42, dtype=np.int64)
    if kind == "distinct":
        return rng.permutation(n).astype(np.int64) * 3 - 17
    raise AssertionError(kind)


def _expect(k: np.ndarray):
    order = np.argsort(k, kind="stable")
    keys, counts = np.unique(k, return_counts=True)
    return order, np.concatenate([[0], np.cumsum(counts)]).astype(np.int64), keys


@pytest.mark.parametrize("kind", ["dense", "negative", "wide", "extremes", "sorted", "one", "distinct"])
@pytest.mark.parametrize("n", [1, 2, 63, 1000, 100_003])
@pytest.mark.parametrize("where", ["device", "host"])
def test_layout_matches_stable_argsort(engine, kind, n, where):
    from polars_ols_amd.engine import Layout
    rng = np.random.default_rng(hash((kind, n)) & 0xffff)
    k = _keys(kind, n, rng)
    order, offs, keys = _expect(k)
    lay = Layout(engine, torch.as_tensor(k, device="cuda") if where == "device" else k)
    assert lay.n_groups == len(keys)
    np.testing.assert_array_equal(lay.offsets, offs)
    np.testing.assert_array_equal(lay.keys, keys)
    assert lay.identity == bool(np.all(k[1:] >= k[:-1]))
    iota = np.arange(n, dtype=np.int64)
    f32, u8 = rng.standard_normal(n).astype(np.float32), rng.integers(0, 2, size=n).astype(np.uint8)
    tab = rng.standard_normal((n, 3))
    cols = [iota, f32, u8, tab, None]
    if where == "device":
        cols = [None if c is None else torch.as_tensor(c, device="cuda") for c in cols]
    moved = lay.take(cols)
    assert moved[4] is None
    got = [m.cpu().numpy() if where == "device" else m for m in moved[:4]]
    np.testing.assert_array_equal(got[0], order)                     # stable: rows of a group keep their frame order
    np.testing.assert_array_equal(got[1], f32[order])
    np.testing.assert_array_equal(got[2], u8[order])
    np.testing.assert_array_equal(got[3], tab[order])
    back = lay.untake(moved[:4])
    for b, c in zip(back, [iota, f32, u8, tab]):
        np.testing.assert_array_equal(b.cpu().numpy() if where == "device" else b, c)
    gid = lay.row_groups()
    gid = gid.cpu().numpy() if where == "device" else gid
    np.testing.assert_array_equal(keys[gid], k)
    lay.close()


def test_layout_empty_frame(engine):
    from polars_ols_amd.engine import Layout
    lay = Layout(engine, np.zeros(0, dtype=np.int64))
    assert lay.n_groups == 0 and lay.identity
    np.testing.assert_array_equal(lay.offsets, [0])


def test_layout_rejects_mismatched_columns(engine):
    from polars_ols_amd.engine import Layout
    lay = Layout(engine, np.array([3, 1, 2], dtype=np.int64))
    with pytest.raises(ValueError):
        lay.take([np.zeros(4, dtype=np.float32)])
    with pytest.raises(TypeError):
        lay.take([np.zeros(3, dtype=np.int16)])


def test_layout_ten_million_rows(engine):
    """BASELINE configs[1] as a frame in arrival order: 10 000 interleaved groups x 1 000 rows; size-independent checks."""
    from polars_ols_amd.engine import Layout
    n, G = 10_000_000, 10_000
    g = torch.Generator(device="cuda").manual_seed(0)
    key = torch.randint(0, G, (n,), device="cuda", generator=g)
    lay = Layout(engine, key)
    assert lay.n_groups == G and lay.offsets[-1] == n and not lay.identity
    np.testing.assert_array_equal(np.diff(lay.offsets), torch.bincount(key, minlength=G).cpu().numpy())
    iota = torch.arange(n, device="cuda")
    ks, order = lay.take([key, iota])
    assert bool((ks[1:] >= ks[:-1]).all())                                          # sortedness
    assert bool(((ks[1:] > ks[:-1]) | (order[1:] > order[:-1])).all())              # stability
    assert bool((lay.untake([order])[0] == iota).all())                             # round trip
    assert engine.last_kernel.startswith("k9_group_layout_sort_u32")


@pytest.mark.parametrize("where", ["device", "host"])
def test_over_float_and_string_keys(engine, where):
    """non-integer keys are dictionary-encoded, like Polars' own group-by (README.md:50-57 groups on a string column)"""
    rng = np.random.default_rng(3)
    n = 600
    x1, x2 = rng.standard_normal(n), rng.standard_normal(n)
    names = np.array(["a", "b", "c"])[rng.integers(0, 3, size=n)]
    slope = {"a": 1.0, "b": -2.0, "c": 0.5}
    y = np.array([slope[s] for s in names]) * x1 + 0.3 * x2
    fkey = np.array([{"a": 0.5, "b": -1.25, "c": 7.0}[s] for s in names])
    from polars_ols_amd.least_squares import Frame, col
    e = col("y").least_squares.ols(col("x1"), col("x2")).over("g").alias("p")
    if where == "device":
        frame = Frame(y=torch.as_tensor(y, device="cuda"), x1=torch.as_tensor(x1, device="cuda"),
                      x2=torch.as_tensor(x2, device="cuda"), g=torch.as_tensor(fkey, device="cuda"))
        p = frame.select(e, engine=engine)["p"].cpu().numpy()
    else:
        p = Frame(y=y, x1=x1, x2=x2, g=names).select(e, engine=engine)["p"]
    np.testing.assert_allclose(p, y, rtol=1e-6, atol=1e-9)
