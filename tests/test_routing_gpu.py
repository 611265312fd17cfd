"""A routing audit of the static dispatcher: walk the shape grid -- dtype x columns x weights x null policy x group
length x aligned / ragged --, record the kernel every shape is routed to, and hold EVERY launched register-resident kernel to the
code-object metadata of the built library: no scratch (an array parked in private memory costs several x), no AGPRs (gfx950's
unified register file: one accumulator register and the occupancy halves), at most 256 VGPRs (two waves per SIMD).  That is how the
f64 9-column-with-weights tip of round 2 (274 registers, one wave per SIMD, 258 us instead of 177) would have been caught before a
profile run.  Results are also checked against the oracle on the sampled shapes -- a route that exists must be a route that is right."""
import re
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "scripts"))


@pytest.fixture(scope="module")
def eng():
    from polars_ols_amd import Engine

    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def metadata():
    from check_scratch import LLVM, kernel_scratch
    from polars_ols_amd import _lib

    if not (LLVM / "llvm-objdump").exists():
        pytest.skip("ROCm LLVM tools not present")
    return kernel_scratch(_lib.LIB_PATH)


def _symbol_pattern(name: str):
    """launcher name -> regex over the demangled kernel symbols it can stand for (k1_gram_chol_* only)"""
    m = re.match(r"k1_gram_chol_(f32|f64)_k(\d+)(_w)?_team(\d+)_rc(\d+)(_fast|_edge)?(_p(\d+))?(_nulls)?(_nt)?$", name)
    if not m:
        return None
    t, kt, w, team, rc, kind, _p, npass, nulls, nt = m.groups()
    T = "float" if t == "f32" else "double"
    fast = "true" if kind in ("_fast", "_edge") else "false"
    edge = "true" if kind == "_edge" else "false"
    args = f"{T}, {kt}, {'true' if w else 'false'}, {team}, {rc}, {fast}, {npass or 1}, {'true' if nulls else 'false'}"
    # k1_kernel<..., NT, EDGE> or the occupancy-pinned wrappers k1_kernel_occ4 / _occ2<...> (no NT / EDGE parameters)
    return re.compile(r"pols::k1_kernel<" + re.escape(args) + rf", {'true' if nt else 'false'}, {edge}>\(|pols::k1_kernel_occ[24]<" + re.escape(args) + r">\(")


SHAPES = [(dt, kt, w, pol) for dt in (np.float32, np.float64) for kt in (1, 3, 6, 8, 9, 10, 12, 15, 16, 17, 20, 24, 25, 31)
          for w in (False, True) for pol in ("ignore", "drop")]


@pytest.mark.parametrize("rows,ragged", [(40, True), (200, False), (200, True), (500, True), (1000, False), (1000, True), (1900, True), (3000, True)])
def test_every_route_of_the_grid_keeps_its_registers(eng, metadata, rows, ragged):
    from test_nulls_gpu import _expected

    rng = np.random.default_rng(rows + int(ragged))
    seen = {}
    for dt, kt, w, pol in SHAPES:
        sizes = rng.integers(max(kt + 3, rows - rows // 8), rows + 1, size=5) if ragged else np.full(5, rows)
        if not ragged and rows % 4:
            continue
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        n = int(offs[-1])
        cols = [rng.standard_normal(n).astype(dt) for _ in range(kt)]
        y = (sum(c.astype(np.float64) for c in cols) + 0.1 * rng.standard_normal(n)).astype(dt)
        if pol == "drop":
            y[rng.random(n) < 0.02] = np.nan
        ww = rng.uniform(0.5, 2.0, n).astype(dt) if w else None
        out = eng.least_squares(y, cols, offs, weights=ww, null_policy=pol, want=("coef", "pred"))
        name = eng.last_kernel
        seen.setdefault(name, (dt, kt, w, pol))
        if (kt, w) in ((8, False), (17, True), (31, False)):          # sampled parity: a route must also be right
            coef, pred, _ = _expected(y, cols, offs, ww, False, pol)      # the oracle on the rows the policy keeps
            tol = 1e-4 if dt == np.float32 else 1e-6
            assert np.allclose(out["coef"], coef, rtol=tol, atol=tol), (name, kt)
            assert np.allclose(out["pred"], pred, rtol=tol, atol=tol, equal_nan=True), (name, kt)
    assert len(seen) >= 10, sorted(seen)
    unknown = []
    for name, shape in sorted(seen.items()):
        pat = _symbol_pattern(name)
        if pat is None:
            fam = re.match(r"(k1t|k1p|k1m|k2w|k2|k5|k6|k8)_", name)
            assert fam, name                                           # every route carries a known family name
            continue
        hits = [(k, v) for k, v in metadata.items() if pat.search(k)]
        if not hits:
            unknown.append(name)
            continue
        for sym, (scratch, vgpr, agpr) in hits:
            assert scratch == 0, (name, sym, scratch)
            assert agpr == 0 and 0 < vgpr <= 256, (name, sym, vgpr, agpr)
    assert not unknown, unknown                                        # a launched kernel the metadata does not know: the name map drifted


def test_families_without_a_name_map_have_no_scratch(metadata):
    hot = ("pols::k1t_kernel<", "pols::k1p_kernel<", "pols::k1m_kernel<", "pols::k2_kernel<", "pols::k2w_kernel<", "pols::gram_stream_kernel<",
           "pols::predict_kernel<", "pols::gram_solve_kernel<")
    # (the MFMA families keep their accumulator tiles in AGPRs by design: only scratch is held to zero there)
    bad = {k: v for k, v in metadata.items() if any(h in k for h in hot) and v[0] > 0}
    assert not bad, sorted(bad.items())[:5]
