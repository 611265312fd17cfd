"""Randomised differential check of the C-ABI against the oracle (not part of pytest; run on a GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
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
    if kind != "enet" and k <= 10 and rng.random() < 0.3:              # many tiny groups: K1t's team shapes (eight / four groups per wave)
        vec = 2 if dtype == np.float64 else 4
        G = int(rng.integers(4100, 9000)); slots = int(rng.choice([8, 16, 32, 64]))
        y, cols, offs, w = frame(G, 3 * (k + 1), max(3 * (k + 1) + 1, slots * vec - vec), k, dtype)
    w = w if wts else None
    if k >= 2 and kind != "enet" and rng.random() < 0.25:            # a rank-deficient group: every solve_method has its own answer for it
        gg = int(rng.integers(0, G)); a, b2 = rng.choice(k, size=2, replace=False)
        cols[a][offs[gg]:offs[gg + 1]] = cols[b2][offs[gg]:offs[gg + 1]]
        if kind == "ols":                                            # ("svd" on EXACT dependence is a knife edge beyond a few columns, in LAPACK too)
            kw = {"solve_method": rng.choice([None, "qr", "svd"] if k <= 8 else [None, "qr"])}
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
            vm = valid.astype(bool)                                  # masked rows: predictions are nulls (ex.rs:640-645)
            ok = (np.allclose(out["coef"], ref["coef"], rtol=2e-6, atol=2e-6) and np.allclose(out["pred"][vm], ref["pred"][vm], rtol=2e-6, atol=2e-6)
                  and np.isnan(out["pred"][~vm]).all())
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
            vm = valid.astype(bool)
            ok = (np.allclose(out["coef"][m], ref["coef"][m], rtol=2e-5, atol=2e-5) and np.allclose(out["pred"][m & vm], ref["pred"][m & vm], rtol=2e-5, atol=2e-5)
                  and np.isnan(out["pred"][~vm]).all())
            what = ("rolling", win, pol)
        if not ok:
            bad += 1
            print("DYNAMIC MISMATCH", it, "k", k, "G", G, what, eng.last_kernel)
    except Exception as exc:
        bad += 1
        print("DYNAMIC ERROR", it, k, G, repr(exc)[:200])
# ---- the row-parallel dynamic kernels (K3c / K4c): many sequences around the tile sizes (packed tiles, halo tiles, two-pass RLS), f32 and f64
N_DYN2 = int(os.environ.get("FUZZ_DYN2", "120")); seen2 = {}
for it in range(N_DYN2):
    dtype = np.float64 if rng.random() < 0.6 else np.float32
    k = int(rng.integers(1, 9)); G = int(rng.integers(1, 400))
    top = int(rng.choice([3, 40, 300, 509, 1021, 1024, 1100, 2600]))
    y, cols, offs, _ = frame(G, int(rng.choice([0, 1, top // 2])), top, k, dtype)
    if len(y) < 8:
        continue
    valid = None if rng.random() < 0.6 else (rng.random(len(y)) > 0.08).astype(np.uint8)
    tol = 2e-6 if dtype == np.float64 else 2e-3
    try:
        if rng.random() < 0.5:
            hl = None if rng.random() < 0.3 else float(rng.uniform(5, 300))
            p0 = float(rng.choice([1.0, 10.0, 1e4]))
            out = eng.recursive_least_squares(y, cols, offs, valid=valid, half_life=hl, initial_state_covariance=p0, null_free=valid is None)
            ref = orc.batched_rls(y, cols, offs, half_life=hl, initial_state_covariance=p0, is_valid=valid)
            vm = np.ones(len(y), dtype=bool) if valid is None else valid.astype(bool)
            ok = (np.allclose(out["coef"], ref["coef"], rtol=tol, atol=tol) and np.allclose(out["pred"][vm], ref["pred"][vm], rtol=tol, atol=tol)
                  and np.isnan(out["pred"][~vm]).all())
            what = ("rls", hl, p0)
        else:
            win = int(rng.integers(max(2, k), 509)); pol = str(rng.choice(["drop", "drop_window"]))
            mp = int(rng.integers(1, win + 1)) if rng.random() < 0.5 else None
            out = eng.rolling_least_squares(y, cols, offs, valid=valid, window_size=win, min_periods=mp, null_policy=pol, null_free=valid is None)
            ref = orc.batched_rolling(y, cols, offs, win, min_periods=mp, null_policy=pol, is_valid=valid)
            sane = np.isfinite(ref["coef"]).all(axis=1) & (np.abs(ref["coef"]).max(axis=1) < 1e2)
            v = np.ones(len(y), dtype=np.int64) if valid is None else valid.astype(np.int64); nobs = np.zeros(len(y), dtype=np.int64)
            for g in range(G):
                s_, e_ = offs[g], offs[g + 1]; c = np.cumsum(v[s_:e_])
                if pol == "drop": nobs[s_:e_] = np.minimum(c, win)
                else: nobs[s_:e_] = c - np.concatenate([np.zeros(min(win, e_ - s_), dtype=np.int64), c[: max(0, e_ - s_ - win)]])
            m = sane & (nobs >= 2 * k + 4)
            vm = v.astype(bool)
            rt = 10 * tol
            ok = (np.allclose(out["coef"][m], ref["coef"][m], rtol=rt, atol=rt) and np.allclose(out["pred"][m & vm], ref["pred"][m & vm], rtol=rt, atol=rt)
                  and np.isnan(out["pred"][~vm]).all())
            what = ("rolling", win, mp, pol)
        seen2[eng.last_kernel] = seen2.get(eng.last_kernel, 0) + 1
        if not ok:
            bad += 1
            print("DYN2 MISMATCH", it, dtype.__name__, "k", k, "G", G, "top", top, "valid", valid is not None, what, eng.last_kernel)
    except Exception as exc:
        bad += 1
        print("DYN2 ERROR", it, dtype.__name__, k, G, top, repr(exc)[:200])
print("row-parallel dynamic cases ran:", dict(sorted(seen2.items())))
# ---- round 5: K4p / K3p (9..32 features, wave per chunk, forced sub-wave packings), DEVICE columns and validity bytes (the device-built
# validity prefix), and K6s (groups with no more rows than columns)
import torch
N_DYN3 = int(os.environ.get("FUZZ_DYN3", "60")); seen3 = {}
for it in range(N_DYN3):
    dtype = np.float64 if rng.random() < 0.7 else np.float32
    k = int(rng.integers(9, 33)); G = int(rng.integers(1, 60))
    top = int(rng.choice([40, 300, 1024, 1100, 2600]))
    y, cols, offs, _ = frame(G, int(rng.choice([0, 1, top // 2])), top, k, dtype)
    if len(y) < 8:
        continue
    valid = None if rng.random() < 0.5 else (rng.random(len(y)) > float(rng.choice([0.05, 0.3]))).astype(np.uint8)
    tol = 2e-6 if dtype == np.float64 else 2e-3
    lps = str(rng.choice(["64", "16" if k <= 16 else "32", ""]))
    eng.set_option("K4P_LPS", lps or None)
    yd = torch.from_numpy(y).cuda(); cd = [torch.from_numpy(c).cuda() for c in cols]; vd = None if valid is None else torch.from_numpy(valid).cuda()
    try:
        if rng.random() < 0.45:
            hl = None if rng.random() < 0.3 else float(rng.uniform(5, 300))
            p0 = float(rng.choice([1.0, 10.0, 1e4]))
            out = eng.recursive_least_squares(yd, cd, offs, valid=vd, half_life=hl, initial_state_covariance=p0, null_free=valid is None)
            ref = orc.batched_rls(y, cols, offs, half_life=hl, initial_state_covariance=p0, is_valid=valid)
            vm = np.ones(len(y), dtype=bool) if valid is None else valid.astype(bool)
            oc, op = out["coef"].double().cpu().numpy(), out["pred"].double().cpu().numpy()
            ok = np.allclose(oc, ref["coef"], rtol=tol, atol=tol) and np.allclose(op[vm], ref["pred"][vm], rtol=tol, atol=tol) and np.isnan(op[~vm]).all()
            what = ("rls", hl, p0, lps)
        else:
            win = int(rng.integers(max(2, k), 1300)); pol = str(rng.choice(["drop", "drop_window"]))
            mp = int(rng.integers(1, win + 1)) if rng.random() < 0.4 else None
            alpha = None if rng.random() < 0.7 else float(rng.uniform(0.01, 1.0))
            out = eng.rolling_least_squares(yd, cd, offs, valid=vd, window_size=win, min_periods=mp, alpha=alpha, null_policy=pol, null_free=valid is None)
            ref = orc.batched_rolling(y, cols, offs, win, min_periods=mp, alpha=alpha, null_policy=pol, is_valid=valid)
            sane = np.isfinite(ref["coef"]).all(axis=1) & (np.abs(ref["coef"]).max(axis=1) < 1e2)
            v = np.ones(len(y), dtype=np.int64) if valid is None else valid.astype(np.int64); nobs = np.zeros(len(y), dtype=np.int64)
            for g in range(G):
                s_, e_ = offs[g], offs[g + 1]; c = np.cumsum(v[s_:e_])
                if pol == "drop": nobs[s_:e_] = np.minimum(c, win)
                else: nobs[s_:e_] = c - np.concatenate([np.zeros(min(win, e_ - s_), dtype=np.int64), c[: max(0, e_ - s_ - win)]])
            m = sane & ((nobs >= 2 * k + 4) | (alpha is not None))
            vm = v.astype(bool)
            rt = 10 * tol
            oc, op = out["coef"].double().cpu().numpy(), out["pred"].double().cpu().numpy()
            ok = (np.allclose(oc[m], ref["coef"][m], rtol=rt, atol=rt) and np.allclose(op[m & vm], ref["pred"][m & vm], rtol=rt, atol=rt)
                  and np.isnan(op[~vm]).all() and np.isnan(oc[np.isnan(ref["coef"]).all(axis=1)]).all())
            what = ("rolling", win, mp, alpha, pol, lps)
        seen3[eng.last_kernel] = seen3.get(eng.last_kernel, 0) + 1
        if not ok:
            bad += 1
            print("DYN3 MISMATCH", it, dtype.__name__, "k", k, "G", G, "top", top, "valid", valid is not None, what, eng.last_kernel)
    except Exception as exc:
        bad += 1
        print("DYN3 ERROR", it, dtype.__name__, k, G, top, repr(exc)[:200])
eng.set_option("K4P_LPS", None)
print("wave-per-chunk dynamic cases ran:", dict(sorted(seen3.items())))
for it in range(int(os.environ.get("FUZZ_SHORT", "30"))):
    dtype = np.float64 if rng.random() < 0.5 else np.float32
    k = int(rng.integers(2, 32)); G = int(rng.integers(1, 4000)); icpt = bool(rng.random() < 0.3); wts = bool(rng.random() < 0.3)
    kt = k + int(icpt)
    y, cols, offs, w = frame(G, int(rng.choice([0, 1])), int(rng.choice([max(1, kt - 1), kt, min(32, kt + 3)])), k, dtype)
    if len(y) < 4:
        continue
    method = rng.choice([None, "svd"])
    w = w if wts else None
    try:
        out = eng.least_squares(y, cols, offs, weights=w, add_intercept=icpt, solve_method=method, want=("coef", "pred", "status"))
        ref = orc.batched_least_squares(y, cols, offs, weights=w, add_intercept=icpt, solve_method=method)
        sizes = np.diff(offs); under = sizes < kt                      # under-determined groups: the minimum-norm solution is unique
        tol = 2e-6 if dtype == np.float64 else 2e-3
        rows = np.repeat(under, sizes)
        ok = np.allclose(out["coef"][under], ref["coef"][under], rtol=tol, atol=tol) and np.allclose(out["pred"][rows], ref["pred"][rows], rtol=tol, atol=tol)
        if not ok:
            bad += 1
            print("SHORT MISMATCH", it, dtype.__name__, "k", k, "G", G, "icpt", icpt, "w", wts, method, eng.last_kernel,
                  float(np.nanmax(np.abs(out["coef"][under] - ref["coef"][under]))))
    except Exception as exc:
        bad += 1
        print("SHORT ERROR", it, dtype.__name__, k, G, method, repr(exc)[:200])
# ---- size classes (round 5): frames whose group sizes spread widely, the per-class launches against the one-launch form of the same call
seen_c = {}
for it in range(int(os.environ.get("FUZZ_SPREAD", "24"))):
    dtype = np.float64 if rng.random() < 0.5 else np.float32
    k = int(rng.integers(1, 16)); G = int(rng.integers(2048, 30000)); icpt = bool(rng.random() < 0.3); wts = bool(rng.random() < 0.3)
    cap = int(rng.choice([300, 1000, 1900 if dtype == np.float64 else 3900])) if k <= 10 else int(rng.choice([200, 500]))
    kind = int(rng.integers(0, 3))
    if kind == 0:
        sizes = np.clip(rng.lognormal(np.log(rng.uniform(20, 300)), rng.uniform(0.5, 1.2), size=G).astype(np.int64), 0, cap)
    elif kind == 1:
        sizes = np.where(rng.random(G) < rng.uniform(0.5, 0.98), rng.integers(1, 60, size=G), rng.integers(cap // 2, cap + 1, size=G))
    else:
        sizes = np.where(rng.random(G) < 0.5, rng.integers(0, 20, size=G), np.where(rng.random(G) < 0.8, rng.integers(60, 200, size=G), cap))
    sizes[int(rng.integers(0, G))] = cap
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(offs[-1])
    cols = [rng.standard_normal(N).astype(dtype) for _ in range(k)]
    y = (sum(c.astype(np.float64) for c in cols) + 0.1 * rng.standard_normal(N)).astype(dtype)
    w = rng.uniform(0.2, 2.0, N).astype(dtype) if wts else None
    policy = str(rng.choice(["ignore", "ignore", "drop", "zero"]))
    if policy != "ignore":
        y[rng.random(N) < 0.01] = np.nan
    kt = k + int(icpt)
    try:
        kw = dict(weights=w, add_intercept=icpt, want=("coef", "pred", "status"), null_policy=policy)
        out = eng.least_squares(y, cols, offs, **kw)
        name = eng.last_kernel
        seen_c[name.count(" | ") + 1] = seen_c.get(name.count(" | ") + 1, 0) + 1
        eng.set_option("NO_CLASSES", "1")
        one = eng.least_squares(y, cols, offs, **kw)
        eng.set_option("NO_CLASSES", None)
        full = sizes > 3 * kt
        rows = np.repeat(full, sizes)
        tol = 2e-5 if dtype == np.float64 else 5e-3
        ok = np.array_equal(out["status"], one["status"]) and np.allclose(out["coef"][full], one["coef"][full], rtol=tol, atol=tol, equal_nan=True) and \
            np.allclose(out["pred"][rows], one["pred"][rows], rtol=tol, atol=tol, equal_nan=True)
        if not ok:
            bad += 1
            print("SPREAD MISMATCH", it, dtype.__name__, "k", k, "G", G, "cap", cap, "kind", kind, "icpt", icpt, "w", wts, policy, name)
    except Exception as exc:
        eng.set_option("NO_CLASSES", None)
        bad += 1
        print("SPREAD ERROR", it, dtype.__name__, k, G, cap, kind, repr(exc)[:200])
print("size-class cases by number of launches:", dict(sorted(seen_c.items())))
# ---- null policies (static models; expected values composed like the reference composes them: tests/test_nulls_gpu.py::_expected)
from test_nulls_gpu import _expected
for it in range(80):
    dtype = np.float64 if rng.random() < 0.6 else np.float32
    k = int(rng.integers(1, 36)); G = int(rng.integers(1, 25))
    icpt = bool(rng.random() < 0.4); wts = bool(rng.random() < 0.4)
    policy = str(rng.choice(["zero", "drop", "drop_zero", "drop_y_zero_x", "drop_window"]))
    kind = rng.choice(["ols", "ridge", "enet"])
    kw = {} if kind == "ols" else ({"alpha": float(rng.uniform(0.05, 2.0))} if kind == "ridge" else
                                   {"alpha": float(rng.uniform(0.005, 0.2)), "l1_ratio": float(rng.uniform(0.1, 1.0)), "tol": 1e-10, "max_iter": 50_000})
    y, cols, offs, w = frame(G, 4 * (k + 2), 4 * (k + 2) + int(rng.integers(10, 1500)), k, dtype)
    frac = float(rng.choice([0.0, 0.01, 0.1]))
    for c in [y] + cols[: int(rng.integers(0, k + 1))]:
        c[rng.random(len(y)) < frac] = np.nan
    if G > 2 and rng.random() < 0.3:
        y[offs[1]:offs[2]] = np.nan                                  # a group with nothing left to fit
    w = w if wts else None
    try:
        out = eng.least_squares(y, cols, offs, weights=w, add_intercept=icpt, want=("coef", "pred", "resid"), null_policy=policy, **kw)
        coef, pred, resid = _expected(y, cols, offs, w, icpt, "drop_zero" if policy == "drop_window" else policy, **kw)
        tol = 2e-6 if dtype == np.float64 else 2e-3
        # groups left with fewer fit rows than columns are min-norm problems: compare predictions on fit rows only there
        ok = np.array_equal(np.isnan(out["pred"]), np.isnan(pred)) and np.allclose(out["pred"], pred, rtol=tol, atol=tol, equal_nan=True)
        ok = ok and np.allclose(out["coef"], coef, rtol=tol, atol=tol)
        if not ok:
            bad += 1
            print("NULLS MISMATCH", it, dtype.__name__, "k", k, "G", G, "icpt", icpt, "w", wts, policy, kind, kw, "frac", frac, eng.last_kernel,
                  float(np.nanmax(np.abs(out["coef"] - coef))))
    except Exception as exc:
        bad += 1
        print("NULLS ERROR", it, dtype.__name__, k, G, policy, kind, repr(exc)[:200])
# ---- statistics
for it in range(40):
    dtype = np.float64
    k = int(rng.integers(1, 60)); G = int(rng.integers(1, 12))
    icpt = bool(rng.random() < 0.5); alpha = float(rng.choice([0.0, 0.0, 0.3]))
    y, cols, offs, w = frame(G, 3 * (k + 2), 3 * (k + 2) + int(rng.integers(10, 1200)), k, dtype)
    try:
        out = eng.least_squares_statistics(y, cols, offs, add_intercept=icpt, alpha=alpha)
        for g in range(G):
            s, e = offs[g], offs[g + 1]
            X = np.column_stack([c[s:e] for c in cols] + ([np.ones(e - s)] if icpt else [])).astype(np.float64)
            ref = orc.statistics(y[s:e].astype(np.float64), X, alpha=alpha)
            tol = 1e-6 if dtype == np.float64 else 5e-3
            for a, b in (("r2", "r2"), ("mae", "mae"), ("mse", "mse"), ("std_err", "standard_errors"), ("t_values", "t_values"), ("p_values", "p_values")):
                got, exp = np.asarray(out[a][g], dtype=np.float64), np.asarray(ref[b], dtype=np.float64)
                if not np.allclose(got, exp, rtol=tol, atol=tol, equal_nan=True):
                    bad += 1
                    print("STATS MISMATCH", it, dtype.__name__, "k", k, "icpt", icpt, "alpha", alpha, a, eng.last_kernel, float(np.nanmax(np.abs(got - exp))))
                    break
    except Exception as exc:
        bad += 1
        print("STATS ERROR", it, dtype.__name__, k, G, repr(exc)[:300])
# ---- .over(key): a frame in arrival order through the native layout == the same frame pre-sorted, bit for bit
import torch
from polars_ols_amd.least_squares import Frame, col
for it in range(30):
    k = int(rng.integers(1, 12)); G = int(rng.integers(1, 400)); n = int(rng.integers(G * (k + 3), G * (k + 3) + 20000))
    keys = rng.integers(-G, G, size=n) * int(rng.choice([1, 1, 1 << 33]))
    cols = {f"x{j}": rng.normal(size=n) for j in range(k)}
    y = sum(cols.values()) + 0.1 * rng.normal(size=n)
    dev = bool(rng.random() < 0.5)
    conv = (lambda a: torch.as_tensor(a, device="cuda")) if dev else (lambda a: a)
    order = np.argsort(keys, kind="stable")
    try:
        res = []
        for idx in (np.arange(n), order):
            f = Frame({name: conv(c[idx]) for name, c in cols.items()})
            f["y"], f["g"] = conv(y[idx]), conv(keys[idx])
            mode = str(rng.choice(["predictions", "residuals"])) if idx is not order else mode
            p = f.select(col("y").least_squares.ols(*[col(c) for c in cols], mode=mode).over("g").alias("p"), engine=eng)["p"]
            res.append(p.cpu().numpy() if dev else p)
        if not np.array_equal(res[0][order], res[1], equal_nan=True):
            bad += 1
            print("OVER MISMATCH", it, "k", k, "G", G, "n", n, "dev", dev, float(np.nanmax(np.abs(res[0][order] - res[1]))))
    except Exception as exc:
        bad += 1
        print("OVER ERROR", it, k, G, n, dev, repr(exc)[:300])
# ---- whole-chip frames: thousands of groups, so every workgroup shape runs with a full grid, a ragged tail and the chunk that
# crosses the end of the columns (the cases above use a few dozen groups)
N_BIG = 40 if len(sys.argv) > 2 and sys.argv[2] == "big" else 0
seen = {}
for it in range(N_BIG):
    dtype = np.float64 if rng.random() < 0.4 else np.float32
    k = int(rng.integers(1, 16)); icpt = bool(rng.random() < 0.4); wts = bool(rng.random() < 0.3)
    shape = str(rng.choice(["equal", "tight", "wide", "tiny", "small"]))
    base = int(rng.choice([64, 130, 250, 500, 1000, 1100]))
    lo, hi = {"equal": (base, base), "tight": (int(base * 0.9), int(base * 1.02)), "wide": (max(k + 3, base // 4), base),
              "tiny": (3 * (k + 1), 5 * (k + 1)), "small": (4 * (k + 1), 10 * (k + 1))}[shape]
    lo = max(lo, k + 3); hi = max(hi, lo)
    G = int(min(rng.integers(2000, 40000), 6_000_000 // hi))
    nulls = None if rng.random() < 0.7 else str(rng.choice(["drop", "zero", "drop_zero"]))
    kw = {} if rng.random() < 0.7 else {"alpha": float(rng.uniform(0.05, 2.0))}
    if nulls: G = min(G, 6000)                                       # the expected values of the null policies are a Python loop
    y, cols, offs, w = frame(G, lo, hi, k, dtype)
    w = w if wts else None
    if nulls:
        for c in [y] + cols[: int(rng.integers(0, k + 1))]:
            c[rng.random(len(y)) < 0.02] = np.nan
    try:
        extra = {"null_policy": nulls} if nulls else {}
        out = eng.least_squares(y, cols, offs, weights=w, add_intercept=icpt, want=("coef", "pred"), **extra, **kw)
        if nulls: coef, pred, _ = _expected(y, cols, offs, w, icpt, nulls, **kw)
        else:
            ref = orc.batched_least_squares(y, cols, offs, weights=w, add_intercept=icpt, **kw); coef, pred = ref["coef"], ref["pred"]
        thin = shape in ("tiny", "small") or lo < 4 * (k + 1)         # groups with fewer than ~4 rows per column ("wide" at base 64 and 15 columns: 18 rows)
        tol = 1e-6 if dtype == np.float64 else (1e-3 if thin else 1e-4)
        seen[eng.last_kernel] = seen.get(eng.last_kernel, 0) + 1
        ok = np.allclose(out["coef"], coef, rtol=tol, atol=tol, equal_nan=True) and np.allclose(out["pred"], pred, rtol=tol, atol=tol, equal_nan=True)
        if not ok and dtype == np.float32 and thin:
            # a few of 10^4 random groups with ~4 rows per column are ill-conditioned beyond f32: hold 99.9% of the groups to
            # the tolerance and every group to 0.05 (a wrong row range or a mixed-up group is an O(1) error)
            err = np.nanmax(np.abs(np.asarray(out["coef"], dtype=np.float64) - coef) / (1.0 + np.abs(coef)), axis=1)
            ok = np.nanquantile(err, 0.999) <= tol and np.nanmax(err) <= 0.05
        if not ok:
            bad += 1
            d = np.abs(np.asarray(out["coef"], dtype=np.float64) - coef); g = int(np.nanargmax(d.max(axis=1)))
            print("BIG MISMATCH", it, dtype.__name__, "k", k, "G", G, "rows", (lo, hi), shape, "icpt", icpt, "w", wts, nulls, kw, eng.last_kernel,
                  "max|dcoef|", float(np.nanmax(d)), "at group", g, "rows", int(offs[g + 1] - offs[g]))
    except Exception as exc:
        bad += 1
        print("BIG ERROR", it, dtype.__name__, k, G, shape, nulls, repr(exc)[:300])
if N_BIG: print("big frames ran:", dict(sorted(seen.items())))
print(f"fuzz done: {N_STATIC} static + {N_DYN} dynamic + 80 null-policy + 40 statistics + 30 over(key) + {N_BIG} whole-chip cases, bad =", bad)
