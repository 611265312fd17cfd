"""Randomised differential check of the C-ABI against the oracle (not part of pytest; run on a GPU box)."""
import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from polars_ols_amd import Engine
from oracle import orc

eng = Engine(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
def frame(G, lo, hi, k, dtype):
    sizes = rng.integers(lo, hi + 1, size=G)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(offs[-1])
    cols = [rng.normal(size=n).astype(dtype) for _ in range(k)]
    y = (sum(c.astype(np.float64) for c in cols) * rng.normal() + 0.3 * rng.normal(size=n) + rng.normal()).astype(dtype)
    w = rng.uniform(0.1, 3.0, size=n).astype(dtype)
    return y, cols, offs, w
N_STATIC, N_DYN = 150, 60
for it in range(N_STATIC):
    dtype = np.float64 if rng.random() < 0.5 else np.float32
    k = int(rng.integers(1, 41)); G = int(rng.integers(1, 30))
    lo = int(rng.integers(0, 3)) * k; hi = lo + int(rng.integers(2 * k + 5, 2500))
    icpt = bool(rng.random() < 0.4); wts = bool(rng.random() < 0.4)
    kind = rng.choice(["ols", "ridge", "enet"])
    kw = {} if kind == "ols" else ({"alpha": float(rng.uniform(0.01, 2.0))} if kind == "ridge" else
                                   {"alpha": float(rng.uniform(0.001, 0.2)), "l1_ratio": float(rng.uniform(0.1, 1.0)), "tol": 1e-10, "max_iter": 50_000,
                                    "positive": bool(rng.random() < 0.3)})
    y, cols, offs, w = frame(G, max(lo, 3 * (k + 1)), hi + 3 * (k + 1), k, dtype)
    w = w if wts else None
    try:
        out = eng.least_squares(y, cols, offs, weights=w, add_intercept=icpt, want=("coef", "pred", "resid", "status"), **kw)
        ref = orc.batched_least_squares(y, cols, offs, weights=w, add_intercept=icpt, **kw)
        tol = 1e-6 if dtype == np.float64 else 1e-3
        for key in ("coef", "pred", "resid"):
            if not np.allclose(out[key], ref[key], rtol=tol, atol=tol, equal_nan=True):
                bad += 1
                print("STATIC MISMATCH", it, key, dtype.__name__, "k", k, "G", G, "icpt", icpt, "w", wts, kind, kw, eng.last_kernel,
                      float(np.nanmax(np.abs(np.asarray(out[key], dtype=np.float64) - ref[key]))), "status", out["status"][:8])
                break
    except Exception as exc:
        bad += 1
        print("STATIC ERROR", it, dtype.__name__, k, G, kind, kw, repr(exc)[:200])
for it in range(N_DYN):
    dtype = np.float64
    k = int(rng.integers(1, 45)); G = int(rng.integers(1, 6))
    y, cols, offs, _ = frame(G, 0, int(rng.integers(4 * k + 10, 4000)), k, dtype)
    valid = (rng.random(len(y)) > rng.choice([0.0, 0.1])).astype(np.uint8)
    try:
        if rng.random() < 0.5:
            hl = None if rng.random() < 0.5 else float(rng.uniform(30 + 8 * k, 500 + 8 * k))
            out = eng.recursive_least_squares(y, cols, offs, valid=valid, half_life=hl, initial_state_covariance=10.0)
            ref = orc.batched_rls(y, cols, offs, half_life=hl, initial_state_covariance=10.0, is_valid=valid)
            ok = np.allclose(out["coef"], ref["coef"], rtol=2e-6, atol=2e-6) and np.allclose(out["pred"], ref["pred"], rtol=2e-6, atol=2e-6)
            what = ("rls", hl)
        else:
            win = int(rng.integers(3 * k + 5, 6 * k + 300)); pol = rng.choice(["drop", "drop_window"])
            out = eng.rolling_least_squares(y, cols, offs, valid=valid, window_size=win, min_periods=None, null_policy=pol)
            ref = orc.batched_rolling(y, cols, offs, win, null_policy=pol, is_valid=valid)
            sane = np.isfinite(ref["coef"]).all(axis=1) & (np.abs(ref["coef"]).max(axis=1) < 1e2)
            # windows with fewer than 2k observations are too ill-conditioned for a 1e-5 comparison
            v = valid.astype(np.int64); nobs = np.zeros(len(y), dtype=np.int64)
            for g in range(G):
                s, e = offs[g], offs[g + 1]; c = np.cumsum(v[s:e])
                if pol == "drop": nobs[s:e] = np.minimum(c, win)
                else: nobs[s:e] = c - np.concatenate([np.zeros(min(win, e - s), dtype=np.int64), c[: max(0, e - s - win)]])
            m = sane & (nobs >= 2 * k + 4)
            ok = np.allclose(out["coef"][m], ref["coef"][m], rtol=2e-5, atol=2e-5) and np.allclose(out["pred"][m], ref["pred"][m], rtol=2e-5, atol=2e-5)
            what = ("rolling", win, pol)
        if not ok:
            bad += 1
            print("DYNAMIC MISMATCH", it, "k", k, "G", G, what, eng.last_kernel)
    except Exception as exc:
        bad += 1
        print("DYNAMIC ERROR", it, k, G, repr(exc)[:200])
print("fuzz done: bad =", bad)
