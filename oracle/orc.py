"""ctypes binding of ``oracle/liborc.so`` (the plain-C restatement of the reference
solve path).  TEST INFRASTRUCTURE ONLY -- see ``oracle/pols_oracle.h``.

Every wrapper takes/returns float64 numpy arrays; matrices are row-major n x k
exactly like the ndarray the reference builds in src/expressions.rs:26.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path
from typing import Optional, Sequence

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "liborc.so"

METHODS = {None: 0, "qr": 1, "svd": 2, "chol": 3, "lu": 4, "cd": 5, "cd_active_set": 6}
NULL_POLICIES = {"ignore": 0, "zero": 1, "drop": 2, "drop_zero": 3, "drop_y_zero_x": 4, "drop_window": 5}


class OlsParams(C.Structure):
    _fields_ = [
        ("alpha", C.c_double),
        ("l1_ratio", C.c_double),
        ("has_l1_ratio", C.c_int32),
        ("max_iter", C.c_int64),
        ("tol", C.c_double),
        ("positive", C.c_int32),
        ("solve_method", C.c_int32),
        ("rcond", C.c_double),
        ("has_rcond", C.c_int32),
    ]


def build(force: bool = False) -> Path:
    """Compile liborc.so with gcc (oracle/Makefile).  Building the checker is not using it."""
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < (_HERE / "pols_oracle.c").stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "-B" if force else "-s"], check=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            build()
        _lib = C.CDLL(str(_LIB_PATH))
        _lib.orc_student_t_two_sided_p.restype = C.c_double
        _lib.orc_student_t_two_sided_p.argtypes = [C.c_double, C.c_double]
    return _lib


def _p(a: Optional[np.ndarray], ctype=C.c_double):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(ctype))


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def make_params(alpha=0.0, l1_ratio=None, max_iter=1000, tol=1e-5, positive=False, solve_method=None,
                rcond=None, **_ignored) -> OlsParams:
    return OlsParams(
        alpha=float(alpha if alpha is not None else 0.0),
        l1_ratio=float(l1_ratio) if l1_ratio is not None else 0.0,
        has_l1_ratio=int(l1_ratio is not None),
        max_iter=int(max_iter if max_iter is not None else 1000),
        tol=float(tol if tol is not None else 1e-5),
        positive=int(bool(positive)),
        solve_method=METHODS[solve_method],
        rcond=float(rcond) if rcond is not None else 0.0,
        has_rcond=int(rcond is not None),
    )


def _col_ptrs(cols: Sequence[np.ndarray]):
    cols = [_f64(c) for c in cols]
    arr = (C.POINTER(C.c_double) * len(cols))(*[_p(c) for c in cols])
    return cols, arr


# ------------------------------------------------------------------ single problem

def get_coefficients(y, x, **kwargs) -> np.ndarray:
    """src/expressions.rs:351-388 dispatcher on a single (y, x)."""
    y, x = _f64(y), _f64(x)
    n, k = x.shape
    beta = np.empty(k)
    p = make_params(**kwargs)
    rc = lib().orc_get_coefficients(_p(y), _p(x), C.c_int64(n), C.c_int(k), C.byref(p), _p(beta))
    if rc < 0:
        raise RuntimeError(f"reference would panic (code {rc})")
    return beta


def solve_ols_qr(y, x) -> np.ndarray:
    y, x = _f64(y), _f64(x)
    n, k = x.shape
    beta = np.empty(k)
    lib().orc_solve_ols_qr(_p(y), _p(x), C.c_int64(n), C.c_int(k), _p(beta))
    return beta


def solve_ols_svd(y, x) -> np.ndarray:
    y, x = _f64(y), _f64(x)
    n, k = x.shape
    m = 1 if y.ndim == 1 else y.shape[1]
    beta = np.empty((k, m))
    lib().orc_solve_ols_svd(_p(y), _p(x), C.c_int64(n), C.c_int(k), C.c_int(m), _p(beta))
    return beta[:, 0] if y.ndim == 1 else beta


def solve_ridge_svd(y, x, alpha, rcond=None) -> np.ndarray:
    y, x = _f64(y), _f64(x)
    n, k = x.shape
    m = 1 if y.ndim == 1 else y.shape[1]
    beta = np.empty((k, m))
    lib().orc_solve_ridge_svd(_p(y), _p(x), C.c_int64(n), C.c_int(k), C.c_int(m), C.c_double(alpha),
                              C.c_int(rcond is not None), C.c_double(rcond or 0.0), _p(beta))
    return beta[:, 0] if y.ndim == 1 else beta


def solve_multi_target(y, x, alpha=0.0, rcond=None) -> np.ndarray:
    """src/least_squares.rs:243-260: y is n x m, returns k x m."""
    y, x = _f64(y), _f64(x)
    n, k = x.shape
    m = y.shape[1]
    beta = np.empty((k, m))
    lib().orc_solve_multi_target(_p(y), _p(x), C.c_int64(n), C.c_int(k), C.c_int(m), C.c_double(alpha),
                                 C.c_int(rcond is not None), C.c_double(rcond or 0.0), _p(beta))
    return beta


def solve_elastic_net(y, x, alpha, l1_ratio=None, max_iter=1000, tol=1e-5, positive=False, solve_method=None):
    y, x = _f64(y), _f64(x)
    n, k = x.shape
    w = np.empty(k)
    n_iter = C.c_int64(0)
    rc = lib().orc_solve_elastic_net(_p(y), _p(x), C.c_int64(n), C.c_int(k), C.c_double(alpha),
                                     C.c_int(l1_ratio is not None), C.c_double(l1_ratio or 0.0),
                                     C.c_int64(max_iter), C.c_double(tol), C.c_int(bool(positive)),
                                     C.c_int(METHODS[solve_method]), _p(w), C.byref(n_iter))
    if rc < 0:
        raise RuntimeError(f"reference would panic (code {rc})")
    return w, n_iter.value


def solve_rls(y, x, half_life=None, initial_state_covariance=10.0, initial_state_mean=None, is_valid=None):
    y, x = _f64(y), _f64(x)
    n, k = x.shape
    out = np.empty((n, k))
    mean0 = _f64(initial_state_mean) if initial_state_mean is not None else None
    valid = np.ascontiguousarray(is_valid, dtype=np.uint8) if is_valid is not None else None
    lib().orc_solve_rls(_p(y), _p(x), C.c_int64(n), C.c_int(k), C.c_int(half_life is not None),
                        C.c_double(half_life or 0.0), C.c_double(initial_state_covariance),
                        _p(mean0), _p(valid, C.c_uint8), _p(out))
    return out


def solve_rolling_ols(y, x, window_size, min_periods=None, use_woodbury=None, alpha=None, is_valid=None,
                      null_policy="drop"):
    y, x = _f64(y), _f64(x)
    n, k = x.shape
    out = np.empty((n, k))
    valid = np.ascontiguousarray(is_valid, dtype=np.uint8) if is_valid is not None else None
    lib().orc_solve_rolling_ols(_p(y), _p(x), C.c_int64(n), C.c_int(k), C.c_int64(window_size),
                                C.c_int64(-1 if min_periods is None else min_periods),
                                C.c_int(-1 if use_woodbury is None else int(use_woodbury)),
                                C.c_double(alpha or 0.0), _p(valid, C.c_uint8),
                                C.c_int(NULL_POLICIES[null_policy]), _p(out))
    return out


def inv(a, use_cholesky=True):
    a = _f64(a)
    k = a.shape[0]
    out = np.empty((k, k))
    lib().orc_inv(_p(a), C.c_int(k), C.c_int(int(use_cholesky)), _p(out))
    return out


def woodbury_update(a_inv, u, c, v, c_is_diag=True):
    a_inv, u, c, v = _f64(a_inv), _f64(u), _f64(c), _f64(v)
    k, r = u.shape
    out = np.empty((k, k))
    lib().orc_woodbury_update(_p(a_inv), _p(u), _p(c), _p(v), C.c_int(k), C.c_int(r), C.c_int(int(c_is_diag)), _p(out))
    return out


def update_xtx_inv(xtx_inv, x_update, c=None):
    xtx_inv, x_update = _f64(xtx_inv), _f64(x_update)
    r, k = x_update.shape
    cc = _f64(c) if c is not None else None
    out = np.empty((k, k))
    lib().orc_update_xtx_inv(_p(xtx_inv), _p(x_update), _p(cc), C.c_int(k), C.c_int(r), _p(out))
    return out


def statistics(y, x, alpha=0.0):
    """src/statistics.rs: returns dict(r2, mae, mse, standard_errors, t_values, p_values) for coefficients
    from the default dispatcher."""
    y, x = _f64(y), _f64(x)
    n, k = x.shape
    coef = get_coefficients(y, x, alpha=alpha)
    pred = x @ coef

    class RM(C.Structure):
        _fields_ = [("r2", C.c_double), ("mae", C.c_double), ("mse", C.c_double)]

    rm = RM()
    lib().orc_residual_metrics_compute(_p(y), _p(pred), C.c_int64(n), C.byref(rm))
    se, tv, pv = np.empty(k), np.empty(k), np.empty(k)
    lib().orc_feature_metrics(_p(x), _p(y), C.c_int64(n), C.c_int(k), C.c_double(alpha), _p(se), _p(tv), _p(pv))
    return dict(r2=rm.r2, mae=rm.mae, mse=rm.mse, coefficients=coef, standard_errors=se, t_values=tv, p_values=pv)


# ------------------------------------------------------------------ batched (grouped frame)

def batched_least_squares(y, x_cols, group_offsets, weights=None, add_intercept=False, n_threads=0,
                          want=("coef", "pred", "resid"), **kwargs):
    """pl.col(y).least_squares.<model>(*x, sample_weights=w, add_intercept=..).over(group) with rows
    sorted by group; returns dict with the requested outputs."""
    y = _f64(y)
    cols, colp = _col_ptrs(x_cols)
    k = len(cols)
    offs = np.ascontiguousarray(group_offsets, dtype=np.int64)
    G = len(offs) - 1
    N = len(y)
    kt = k + int(bool(add_intercept))
    w = _f64(weights) if weights is not None else None
    coef = np.empty((G, kt)) if "coef" in want else None
    pred = np.empty(N) if "pred" in want else None
    resid = np.empty(N) if "resid" in want else None
    p = make_params(**kwargs)
    rc = lib().orc_batched_least_squares(_p(y), colp, _p(w), C.c_int64(N), C.c_int(k), _p(offs, C.c_int64),
                                         C.c_int64(G), C.c_int(int(bool(add_intercept))), C.byref(p),
                                         _p(coef), _p(pred), _p(resid), C.c_int(n_threads))
    if rc < 0:
        raise RuntimeError(f"reference would panic (code {rc})")
    return dict(coef=coef, pred=pred, resid=resid)


def batched_rls(y, x_cols, group_offsets, half_life=None, initial_state_covariance=10.0, initial_state_mean=None,
                is_valid=None, n_threads=0, want=("coef", "pred")):
    y = _f64(y)
    cols, colp = _col_ptrs(x_cols)
    k = len(cols)
    offs = np.ascontiguousarray(group_offsets, dtype=np.int64)
    G, N = len(offs) - 1, len(y)
    coef = np.empty((N, k)) if "coef" in want else None
    pred = np.empty(N) if "pred" in want else None
    mean0 = _f64(initial_state_mean) if initial_state_mean is not None else None
    valid = np.ascontiguousarray(is_valid, dtype=np.uint8) if is_valid is not None else None
    lib().orc_batched_rls(_p(y), colp, C.c_int64(N), C.c_int(k), _p(offs, C.c_int64), C.c_int64(G),
                          C.c_int(half_life is not None), C.c_double(half_life or 0.0),
                          C.c_double(initial_state_covariance), _p(mean0), _p(valid, C.c_uint8),
                          _p(coef), _p(pred), C.c_int(n_threads))
    return dict(coef=coef, pred=pred)


def batched_rolling(y, x_cols, group_offsets, window_size, min_periods=None, use_woodbury=None, alpha=None,
                    null_policy="drop", is_valid=None, n_threads=0, want=("coef", "pred")):
    y = _f64(y)
    cols, colp = _col_ptrs(x_cols)
    k = len(cols)
    offs = np.ascontiguousarray(group_offsets, dtype=np.int64)
    G, N = len(offs) - 1, len(y)
    coef = np.empty((N, k)) if "coef" in want else None
    pred = np.empty(N) if "pred" in want else None
    valid = np.ascontiguousarray(is_valid, dtype=np.uint8) if is_valid is not None else None
    lib().orc_batched_rolling(_p(y), colp, C.c_int64(N), C.c_int(k), _p(offs, C.c_int64), C.c_int64(G),
                              C.c_int64(window_size), C.c_int64(-1 if min_periods is None else min_periods),
                              C.c_int(-1 if use_woodbury is None else int(use_woodbury)),
                              C.c_double(alpha or 0.0), C.c_int(NULL_POLICIES[null_policy]),
                              _p(valid, C.c_uint8), _p(coef), _p(pred), C.c_int(n_threads))
    return dict(coef=coef, pred=pred)


def max_threads() -> int:
    return int(lib().orc_max_threads())


# ------------------------------------------------------------------ CPU baseline harness (timed inside C)

def bench_static(y, x_cols, group_offsets, weights=None, add_intercept=False, solve_only=False, passes=1, n_threads=0, **kwargs) -> float:
    """Wall seconds of ``passes`` repetitions of the grouped static path, timed INSIDE liborc with pre-allocated, first-touched
    buffers (orc_bench_static)."""
    y = _f64(y)
    cols, colp = _col_ptrs(x_cols)
    offs = np.ascontiguousarray(group_offsets, dtype=np.int64)
    w = _f64(weights) if weights is not None else None
    p = make_params(**kwargs)
    fn = lib().orc_bench_static
    fn.restype = C.c_double
    sec = fn(_p(y), colp, _p(w), C.c_int64(len(y)), C.c_int(len(cols)), _p(offs, C.c_int64), C.c_int64(len(offs) - 1),
             C.c_int(int(bool(add_intercept))), C.byref(p), C.c_int(int(bool(solve_only))), C.c_int(int(passes)), C.c_int(int(n_threads)))
    if sec < 0:
        raise RuntimeError("reference would panic")
    return float(sec)


def bench_dynamic(kind: str, y, x_cols, half_life=21.0, window=252, min_periods=6, passes=1) -> float:
    """Wall seconds of ``passes`` runs of ONE sequence: kind "rls" or "rolling" (orc_bench_dynamic)."""
    y = _f64(y)
    cols, colp = _col_ptrs(x_cols)
    fn = lib().orc_bench_dynamic
    fn.restype = C.c_double
    return float(fn(C.c_int(0 if kind == "rls" else 1), _p(y), colp, C.c_int64(len(y)), C.c_int(len(cols)), C.c_double(half_life),
                    C.c_int64(window), C.c_int64(min_periods), C.c_int(int(passes))))
