"""Engine: a ``pols_ctx`` plus buffer marshalling for numpy (host) and torch (device) columns.

PyTorch appears here only as a device-memory owner and stream provider (``tensor.data_ptr()``,
``torch.cuda.current_stream().cuda_stream``); every FLOP of the hot path runs in the hand-written gfx950
kernels behind the C-ABI.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Dict, Optional, Sequence

import numpy as np

from . import _lib as L

try:  # torch is optional for the host-buffer path
    import torch
except Exception:  # pragma: no cover
    torch = None


def _is_torch(a) -> bool:
    return torch is not None and isinstance(a, torch.Tensor)


class Engine:
    """One context (device, stream, scratch).  Not thread-safe: create one per thread."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self._lib = L.lib()
        h = C.c_void_p()
        L.check(self._lib.pols_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)
        # torch inputs: launch on torch's CURRENT stream (same-stream ordering with the ops that produced the
        # inputs / consume the outputs) unless the caller pinned a stream with set_stream().
        self._follow_torch = stream is None
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pols_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ stream / timing
    def set_stream(self, stream_ptr: Optional[int]):
        """Pin the HIP stream every launch goes to (0 = HIP's null stream); None: back to following torch's
        current stream for torch inputs (host-buffer calls use the context's private stream)."""
        self._follow_torch = stream_ptr is None
        if stream_ptr is None:
            L.check(self._lib.pols_use_private_stream(self._h))
        else:
            L.check(self._lib.pols_set_stream(self._h, C.c_void_p(stream_ptr)))

    def use_private_stream(self):
        """Every launch (torch inputs included) goes to the context's own non-blocking stream: what a host thread of its
        own wants.  The caller orders it against the producers / consumers of the buffers (``synchronize()``)."""
        self._follow_torch = False
        L.check(self._lib.pols_use_private_stream(self._h))

    def use_torch_stream(self):
        self.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def synchronize(self):
        L.check(self._lib.pols_synchronize(self._h))

    def timing(self, enable):
        """``True`` / ``False``, or an int n > 1 to time every n-th call only."""
        L.check(self._lib.pols_timing_enable(self._h, int(enable)))

    def timing_collect(self, max_n: int = 4096) -> np.ndarray:
        buf = (C.c_float * max_n)()
        n = self._lib.pols_timing_collect(self._h, buf, max_n)
        L.check(n)
        return np.frombuffer(buf, dtype=np.float32, count=n).copy()

    def set_option(self, key: str, value: Optional[str]) -> None:
        """Tuning / diagnostic knob (``pols_set_option``): the name of a POLS_* variable and its value, ``None`` = default."""
        L.check(self._lib.pols_set_option(self._h, key.encode(), None if value is None else str(value).encode()))

    @property
    def last_kernel(self) -> str:
        return self._lib.pols_last_kernel_name(self._h).decode()

    # ------------------------------------------------------------------ marshalling
    def _batch(self, y, x_cols: Sequence, offsets, weights, valid, add_intercept: bool, null_free: bool = False):
        dev = _is_torch(y)
        cols = list(x_cols)
        if len(cols) == 0:
            raise ValueError("must pass at least 2 series")  # src/expressions.rs:72
        if dev:
            if not y.is_cuda:
                raise ValueError("torch inputs must live on the GPU; pass numpy arrays for host data")
            dt = y.dtype
            if dt not in (torch.float32, torch.float64):
                raise TypeError("dtype must be float32 or float64")
            if self._follow_torch:
                L.check(self._lib.pols_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(y.device).cuda_stream)))
            keep = [y.contiguous()] + [c.to(dt).contiguous() for c in cols]
            w = weights.to(dt).contiguous() if weights is not None else None
            v = valid.to(torch.uint8).contiguous() if valid is not None else None
            ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
            dtype = L.POLS_F32 if dt == torch.float32 else L.POLS_F64
            n = y.numel()
        else:
            y = np.asarray(y)
            dt = np.float32 if y.dtype == np.float32 else np.float64
            keep = [np.ascontiguousarray(y, dtype=dt)] + [np.ascontiguousarray(c, dtype=dt) for c in cols]
            w = np.ascontiguousarray(weights, dtype=dt) if weights is not None else None
            v = np.ascontiguousarray(valid, dtype=np.uint8) if valid is not None else None
            ptr = lambda a: a.ctypes.data if a is not None else None  # noqa: E731
            dtype = L.POLS_F32 if dt == np.float32 else L.POLS_F64
            n = keep[0].shape[0]
        for c in keep[1:]:
            if (c.numel() if dev else c.shape[0]) != n:
                raise ValueError("all input series passed must be of equal length")  # src/expressions.rs:96-100
        # a private COPY: every Plan promises the library (offsets_generation) that its offsets never change, which must not
        # depend on what the caller does to the array it passed in afterwards
        offs = np.array(offsets, dtype=np.int64, copy=True, order="C")
        colp = (C.c_void_p * len(cols))(*[ptr(c) for c in keep[1:]])
        b = L.Batch(dtype=dtype, mem=L.POLS_MEM_DEVICE if dev else L.POLS_MEM_HOST, n_rows=n,
                    n_groups=len(offs) - 1, group_offsets=offs.ctypes.data_as(C.POINTER(C.c_int64)),
                    n_features=len(cols), y=ptr(keep[0]), x_cols=colp, weights=ptr(w), valid=ptr(v),
                    add_intercept=int(bool(add_intercept)), null_free=int(bool(null_free)))
        return b, (keep, w, v, offs, colp), dev, dt

    def _alloc(self, dev: bool, dt, shape, like=None):
        if dev:
            return torch.empty(shape, dtype=dt, device=like.device)
        return np.empty(shape, dtype=dt)

    @staticmethod
    def _ptr(a):
        if a is None:
            return None
        return a.data_ptr() if _is_torch(a) else a.ctypes.data

    # ------------------------------------------------------------------ compute
    def plan_least_squares(self, y, x_cols: Sequence, offsets, *, weights=None, valid=None,
                           add_intercept: bool = False, want: Sequence[str] = ("pred",), out: Optional[Dict] = None,
                           alpha: float = 0.0, l1_ratio: Optional[float] = None, max_iter: int = 1000,
                           tol: float = 1e-5, positive: bool = False, solve_method: Optional[str] = None,
                           rcond: Optional[float] = None, null_policy: str = "ignore", null_free: bool = False) -> "Plan":
        """Marshal once, launch many times (``plan.run()``): the buffers are borrowed, nothing is copied."""
        b, keep, dev, dt = self._batch(y, x_cols, offsets, weights, valid, add_intercept, null_free)
        kt = b.n_features + b.add_intercept
        res: Dict = dict(out or {})
        yy = keep[0][0]
        if "coef" in want and "coef" not in res:
            res["coef"] = self._alloc(dev, dt, (b.n_groups, kt), yy)
        if "pred" in want and "pred" not in res:
            res["pred"] = self._alloc(dev, dt, (b.n_rows,), yy)
        if "resid" in want and "resid" not in res:
            res["resid"] = self._alloc(dev, dt, (b.n_rows,), yy)
        if "status" in want and "status" not in res:
            res["status"] = self._alloc(dev, torch.int32 if dev else np.int32, (b.n_groups,), yy)
        o = L.Out(coef=self._ptr(res.get("coef")), pred=self._ptr(res.get("pred")), resid=self._ptr(res.get("resid")),
                  status=self._ptr(res.get("status")))
        p = L.OlsParams()
        self._lib.pols_ols_params_default(C.byref(p))
        p.alpha = float(alpha if alpha is not None else 0.0)
        p.has_l1_ratio = int(l1_ratio is not None)
        p.l1_ratio = float(l1_ratio) if l1_ratio is not None else 0.0
        p.max_iter = int(max_iter if max_iter is not None else 1000)
        p.tol = float(tol if tol is not None else 1e-5)
        p.positive = int(bool(positive))
        p.solve_method = L.SOLVE_METHODS[solve_method]
        p.has_rcond = int(rcond is not None)
        p.rcond = float(rcond) if rcond is not None else 0.0
        p.null_policy = L.NULL_POLICIES[null_policy]
        return Plan(self, self._lib.pols_least_squares, b, p, o, res, keep)

    def multi_target_least_squares(self, y_cols: Sequence, x_cols: Sequence, offsets, *, weights=None, add_intercept: bool = False,
                                   alpha: float = 0.0, solve_method: Optional[str] = None, rcond: Optional[float] = None,
                                   null_policy: str = "ignore", valid=None, null_free: bool = False,
                                   want: Sequence[str] = ("pred", "coef")) -> Dict:
        """solve_multi_target (src/least_squares.rs:243-260) for every group: ONE Gram pass and ONE factorisation shared by all
        targets.  Returns ``pred`` (list of n_targets columns), ``coef`` [n_groups, n_targets, k], ``status`` [n_groups].
        Null policies like the plugin body (src/expressions.rs:521-591): joint mask, fit on the rows it leaves, every row predicted;
        "ignore" zero-fills the nulls it leaves in place (construct_features_array(.., true), :546-547).  ``null_free=True`` promises
        that no target / feature / weight is null and skips the device-side null pass."""
        ys = list(y_cols)
        plan = self.plan_least_squares(ys[0], x_cols, offsets, weights=weights, add_intercept=add_intercept, alpha=alpha,
                                       solve_method=solve_method, rcond=rcond, null_policy=null_policy, valid=valid, null_free=null_free,
                                       want=())
        b = plan._b
        dev = b.mem == L.POLS_MEM_DEVICE
        like = plan._keep[0][0]
        dt = like.dtype
        m, kt = len(ys), b.n_features + b.add_intercept
        ys_k = [(y.to(dt).contiguous() if dev else np.ascontiguousarray(y, dtype=dt)) for y in ys]
        res: Dict = {"status": self._alloc(dev, torch.int32 if dev else np.int32, (b.n_groups,), like)}
        if "coef" in want:
            res["coef"] = self._alloc(dev, dt, (b.n_groups, m, kt), like)
        preds = [self._alloc(dev, dt, (b.n_rows,), like) for _ in range(m)] if "pred" in want else None
        yp = (C.c_void_p * m)(*[self._ptr(y) for y in ys_k])
        pp = (C.c_void_p * m)(*[self._ptr(q) for q in preds]) if preds is not None else None
        L.check(self._lib.pols_multi_target_least_squares(self._h, C.byref(b), yp, C.c_int32(m), C.byref(plan._p), pp,
                                                          C.c_void_p(self._ptr(res.get("coef"))), C.c_void_p(self._ptr(res["status"]))))
        if preds is not None:
            res["pred"] = preds
        return res

    def least_squares_statistics(self, y, x_cols: Sequence, offsets, **kwargs) -> Dict:
        """mode="statistics" (src/expressions.rs:468-509) for every group: returns ``coef`` (batch dtype), ``status`` and
        the f64 arrays ``r2 mae mse`` [n_groups] and ``std_err t_values p_values`` [n_groups, k]."""
        kwargs.setdefault("want", ("coef", "status"))
        plan = self.plan_least_squares(y, x_cols, offsets, **kwargs)
        b = plan._b
        kt = b.n_features + b.add_intercept
        dev = b.mem == L.POLS_MEM_DEVICE
        like = plan._keep[0][0]
        f64 = torch.float64 if dev else np.float64
        res = plan.results
        for key in ("r2", "mae", "mse"):
            res[key] = self._alloc(dev, f64, (b.n_groups,), like)
        for key in ("std_err", "t_values", "p_values"):
            res[key] = self._alloc(dev, f64, (b.n_groups, kt), like)
        so = L.StatsOut(**{k: self._ptr(res[k]) for k in ("r2", "mae", "mse", "std_err", "t_values", "p_values")})
        L.check(self._lib.pols_least_squares_statistics(self._h, C.byref(plan._b), C.byref(plan._p), C.byref(plan._o),
                                                        C.byref(so)))
        return res

    def least_squares(self, y, x_cols: Sequence, offsets, **kwargs) -> Dict:
        """All groups of a (group-sorted) frame in one launch.  ``want`` subset of {"coef","pred","resid","status"};
        ``out`` may carry pre-allocated buffers under the same keys."""
        return self.plan_least_squares(y, x_cols, offsets, **kwargs).run()

    def _dynamic_outputs(self, b, keep, dev, dt, want, out):
        res: Dict = dict(out or {})
        yy = keep[0][0]
        if "coef" in want and "coef" not in res:
            res["coef"] = self._alloc(dev, dt, (b.n_rows, b.n_features + b.add_intercept), yy)
        if "pred" in want and "pred" not in res:
            res["pred"] = self._alloc(dev, dt, (b.n_rows,), yy)
        o = L.Out(coef=self._ptr(res.get("coef")), pred=self._ptr(res.get("pred")), resid=None, status=None)
        return res, o

    def plan_recursive_least_squares(self, y, x_cols: Sequence, offsets, *, weights=None, valid=None, add_intercept: bool = False,
                                     null_free: bool = False, want: Sequence[str] = ("coef", "pred"),
                                     out: Optional[Dict] = None, half_life: Optional[float] = None,
                                     initial_state_covariance: Optional[float] = 10.0, initial_state_mean=None,
                                     null_policy: str = "drop") -> "Plan":
        """solve_recursive_least_squares (src/least_squares.rs:568-598) for every group; ``coef`` is n_rows x kt.  Raw columns
        go in: sqrt(w) scaling, the ones column, the null policy's validity mask (NaN = null unless ``valid`` is given), the
        zero-filling and the 1/sqrt(w) un-scaling of the predictions all happen on the device behind the C-ABI.
        ``null_free=True`` promises that no value is null / NaN (a Polars caller knows from null_count): no validity scan."""
        b, keep, dev, dt = self._batch(y, x_cols, offsets, weights, valid, add_intercept, null_free)
        res, o = self._dynamic_outputs(b, keep, dev, dt, want, out)
        p = L.RlsParams()
        self._lib.pols_rls_params_default(C.byref(p))
        p.has_half_life = int(half_life is not None)
        p.half_life = float(half_life) if half_life is not None else 0.0
        p.initial_state_covariance = float(10.0 if initial_state_covariance is None else initial_state_covariance)
        mean = None
        if initial_state_mean is not None:
            mean = np.ascontiguousarray(np.broadcast_to(np.asarray(initial_state_mean, dtype=np.float64), (b.n_features + b.add_intercept,)))
            p.initial_state_mean = mean.ctypes.data_as(C.POINTER(C.c_double))
        p.null_policy = L.NULL_POLICIES[null_policy]
        return Plan(self, self._lib.pols_recursive_least_squares, b, p, o, res, (keep, mean))

    def recursive_least_squares(self, y, x_cols: Sequence, offsets, **kwargs) -> Dict:
        return self.plan_recursive_least_squares(y, x_cols, offsets, **kwargs).run()

    def plan_rolling_least_squares(self, y, x_cols: Sequence, offsets, *, window_size: int, weights=None, valid=None,
                                   add_intercept: bool = False, null_free: bool = False, want: Sequence[str] = ("coef", "pred"),
                                   out: Optional[Dict] = None,
                                   min_periods: Optional[int] = None, use_woodbury: Optional[bool] = None,
                                   alpha: Optional[float] = None, null_policy: str = "drop_window") -> "Plan":
        """solve_rolling_ols (src/least_squares.rs:848-1032) for every group; ``coef`` is n_rows x kt, NaN where undefined.
        Raw columns in, like plan_recursive_least_squares.

        Two places where this differs from the reference in KIND, both documented in include/pols_mi355x.h: a window whose X'X has no
        Cholesky factorisation is solved by LU like the reference up to 10 features, but yields NaN at 11 to 32 features (the reference's
        LU returns inf / NaN / huge numbers there; ``Engine.set_option("ROLLING_ENGINE", "chunk")`` selects kernels that run it); and
        ``use_woodbury`` is accepted without selecting a code path (the inverse is propagated from 9 features on whatever it says)."""
        b, keep, dev, dt = self._batch(y, x_cols, offsets, weights, valid, add_intercept, null_free)
        res, o = self._dynamic_outputs(b, keep, dev, dt, want, out)
        p = L.RollingParams()
        self._lib.pols_rolling_params_default(C.byref(p))
        p.window_size = int(window_size)
        p.min_periods = -1 if min_periods is None else int(min_periods)
        p.use_woodbury = -1 if use_woodbury is None else int(bool(use_woodbury))
        p.alpha = float(alpha) if alpha is not None else 0.0
        p.null_policy = L.NULL_POLICIES[null_policy]
        return Plan(self, self._lib.pols_rolling_least_squares, b, p, o, res, keep)

    def rolling_least_squares(self, y, x_cols: Sequence, offsets, **kwargs) -> Dict:
        return self.plan_rolling_least_squares(y, x_cols, offsets, **kwargs).run()


class Layout:
    """`.over(key)` ingestion (``pols_layout_*``): the stable permutation that sorts a frame's rows by an int64 key column,
    the groups' offsets / keys, and the column movers either way.  Device keys stay on the device."""

    def __init__(self, eng: Engine, keys):
        self._eng, self._lib = eng, eng._lib
        dev = _is_torch(keys)
        if dev:
            if not keys.is_cuda:
                raise ValueError("torch keys must live on the GPU; pass a numpy array for host keys")
            if eng._follow_torch:
                L.check(self._lib.pols_set_stream(eng._h, C.c_void_p(torch.cuda.current_stream(keys.device).cuda_stream)))
            k = keys.to(torch.int64).contiguous()
            ptr, n = k.data_ptr(), k.numel()
        else:
            k = np.ascontiguousarray(keys, dtype=np.int64)
            ptr, n = k.ctypes.data, k.shape[0]
        self.on_device, self._like = dev, (keys if dev else None)
        h = C.c_void_p()
        L.check(self._lib.pols_layout_create(eng._h, C.c_void_p(ptr), n, L.POLS_MEM_DEVICE if dev else L.POLS_MEM_HOST, C.byref(h)))
        self._h = h
        self.n_rows, self.n_groups = n, int(self._lib.pols_layout_n_groups(h))
        self.identity = bool(self._lib.pols_layout_is_identity(h))
        self.offsets = np.ctypeslib.as_array(self._lib.pols_layout_group_offsets(h), shape=(self.n_groups + 1,)).copy()
        self.keys = (np.ctypeslib.as_array(self._lib.pols_layout_group_keys(h), shape=(self.n_groups,)).copy()
                     if self.n_groups else np.zeros(0, dtype=np.int64))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pols_layout_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _move(self, fn, cols: Sequence):
        cols = list(cols)
        if self.identity or not cols:
            return cols
        dev = _is_torch(cols[0])
        if dev and self._eng._follow_torch:
            L.check(self._lib.pols_set_stream(self._eng._h, C.c_void_p(torch.cuda.current_stream(cols[0].device).cuda_stream)))
        out, by_size = [None] * len(cols), {}
        src = [c.contiguous() if dev else np.ascontiguousarray(c) for c in cols]
        for i, c in enumerate(src):
            if c.shape[0] != self.n_rows:
                raise ValueError("all input series passed must be of equal length")
            width = c.element_size() if dev else c.itemsize
            if c.ndim > 1:                                       # an [n, k] table moves a row (k elements) at a time
                width *= int(np.prod(c.shape[1:]))
            if width != 1 and width % 4:
                raise TypeError(f"unsupported element size {width}")
            by_size.setdefault(width, []).append(i)
        for width, idxs in by_size.items():
            dst = [torch.empty_like(src[i]) if dev else np.empty_like(src[i]) for i in idxs]
            sp = (C.c_void_p * len(idxs))(*[Engine._ptr(src[i]) for i in idxs])
            dp = (C.c_void_p * len(idxs))(*[Engine._ptr(d) for d in dst])
            L.check(fn(self._eng._h, self._h, width, sp, dp, len(idxs), L.POLS_MEM_DEVICE if dev else L.POLS_MEM_HOST))
            for i, d in zip(idxs, dst):
                out[i] = d
        return out

    def take(self, cols: Sequence):
        """frame order -> group order (``None`` entries pass through)."""
        idx = [i for i, c in enumerate(cols) if c is not None]
        moved = self._move(self._lib.pols_layout_take, [cols[i] for i in idx])
        out = list(cols)
        for i, m in zip(idx, moved):
            out[i] = m
        return out

    def untake(self, cols: Sequence):
        """group order -> frame order."""
        return self._move(self._lib.pols_layout_untake, cols)

    def row_groups(self):
        """group index of every frame row (int64, where the keys live)."""
        if self.on_device:
            out = torch.empty(self.n_rows, dtype=torch.int64, device=self._like.device)
        else:
            out = np.empty(self.n_rows, dtype=np.int64)
        L.check(self._lib.pols_layout_row_groups(self._eng._h, self._h, C.c_void_p(Engine._ptr(out) or 0),
                                                 L.POLS_MEM_DEVICE if self.on_device else L.POLS_MEM_HOST))
        return out


class _ArrowExport:
    """One column (``pyarrow`` Array / ChunkedArray) exported through the Arrow C Data Interface as a ``pols_arrow_column``: a
    named schema plus one ArrowArray struct per chunk.  The structs are re-imported (which releases them) on ``close()``."""

    def __init__(self, name: str, col):
        import pyarrow as pa

        self._pa = pa
        chunks = list(col.chunks) if isinstance(col, pa.ChunkedArray) else [col]
        typ = col.type
        self._schema = (C.c_byte * 72)()
        pa.field(name, typ)._export_to_c(C.addressof(self._schema))
        self._arrays = []
        for ch in chunks:
            a = (C.c_byte * 80)()
            ch._export_to_c(C.addressof(a))
            self._arrays.append(a)
        self._ptrs = (C.c_void_p * max(1, len(chunks)))(*[C.addressof(a) for a in self._arrays])
        self._type = typ
        self.column = L.ArrowColumn(schema=C.addressof(self._schema), chunks=self._ptrs, n_chunks=len(chunks))

    def close(self):
        pa = self._pa
        for a in self._arrays:                               # importing takes ownership back and releases on collection
            pa.Array._import_from_c(C.addressof(a), self._type)
        self._arrays = []
        pa.Field._import_from_c(C.addressof(self._schema))


def _least_squares_arrow(self, target, features, *, target_name: str = "y", weights=None, offsets=None, add_intercept: bool = False,
                         mode: str = "predictions", alpha: float = 0.0, l1_ratio: Optional[float] = None, max_iter: int = 1000,
                         tol: float = 1e-5, positive: bool = False, solve_method: Optional[str] = None, rcond: Optional[float] = None,
                         null_policy: str = "ignore"):
    """``pols_least_squares_arrow``: the plugin bodies of src/expressions.rs:390-446 on Arrow columns as Polars holds them.
    ``target`` / ``weights``: pyarrow Array or ChunkedArray; ``features``: dict name -> column (or a list of columns, named by
    index).  Returns a pyarrow Array: predictions / residuals (nullable) or the ``coefficients`` struct."""
    import pyarrow as pa

    feats = list(features.items()) if isinstance(features, dict) else [("", f) for f in features]
    ex_t = _ArrowExport(target_name, target)
    ex_f = [_ArrowExport(n, f) for n, f in feats]
    ex_w = _ArrowExport("sample_weights", weights) if weights is not None else None
    try:
        fcols = (L.ArrowColumn * len(ex_f))(*[e.column for e in ex_f])
        p = L.OlsParams()
        self._lib.pols_ols_params_default(C.byref(p))
        p.alpha = float(alpha if alpha is not None else 0.0)
        p.has_l1_ratio, p.l1_ratio = int(l1_ratio is not None), float(l1_ratio) if l1_ratio is not None else 0.0
        p.max_iter, p.tol, p.positive = int(max_iter), float(tol), int(bool(positive))
        p.solve_method = L.SOLVE_METHODS[solve_method]
        p.has_rcond, p.rcond = int(rcond is not None), float(rcond) if rcond is not None else 0.0
        p.null_policy = L.NULL_POLICIES[null_policy]
        offs = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.int64)
        out_a, out_s = (C.c_byte * 80)(), (C.c_byte * 72)()
        rc = self._lib.pols_least_squares_arrow(
            self._h, C.byref(ex_t.column), fcols, len(ex_f), C.byref(ex_w.column) if ex_w else None,
            offs.ctypes.data_as(C.POINTER(C.c_int64)) if offs is not None else None, 0 if offs is None else len(offs) - 1,
            int(bool(add_intercept)), C.byref(p), {"predictions": 0, "residuals": 1, "coefficients": 2}[mode],
            C.addressof(out_a), C.addressof(out_s))
        L.check(rc)
        return pa.Array._import_from_c(C.addressof(out_a), C.addressof(out_s))
    finally:
        for e in [ex_t] + ex_f + ([ex_w] if ex_w else []):
            e.close()


Engine.least_squares_arrow = _least_squares_arrow


def _ols_params(lib, alpha=0.0, l1_ratio=None, max_iter=1000, tol=1e-5, positive=False, solve_method=None, rcond=None,
                null_policy="ignore"):
    p = L.OlsParams()
    lib.pols_ols_params_default(C.byref(p))
    p.alpha = float(alpha if alpha is not None else 0.0)
    p.has_l1_ratio, p.l1_ratio = int(l1_ratio is not None), float(l1_ratio) if l1_ratio is not None else 0.0
    p.max_iter, p.tol, p.positive = int(max_iter), float(tol), int(bool(positive))
    p.solve_method = L.SOLVE_METHODS[solve_method]
    p.has_rcond, p.rcond = int(rcond is not None), float(rcond) if rcond is not None else 0.0
    p.null_policy = L.NULL_POLICIES[null_policy]
    return p


def _arrow_call(self, fn, first, first_name: str, features, weights, offsets, add_intercept: bool, tail_args):
    """Shared marshalling of the pols_*_arrow entries: (ctx, first column, features, n, weights, offsets, n_groups, intercept,
    *tail_args, out array, out schema) -> pyarrow Array."""
    import pyarrow as pa

    feats = list(features.items()) if isinstance(features, dict) else [("", f) for f in features]
    ex_t = _ArrowExport(first_name, first)
    ex_f = [_ArrowExport(n, f) for n, f in feats]
    ex_w = _ArrowExport("sample_weights", weights) if weights is not None else None
    try:
        fcols = (L.ArrowColumn * len(ex_f))(*[e.column for e in ex_f])
        offs = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.int64)
        out_a, out_s = (C.c_byte * 80)(), (C.c_byte * 72)()
        rc = fn(self._h, C.byref(ex_t.column), fcols, len(ex_f), C.byref(ex_w.column) if ex_w else None,
                offs.ctypes.data_as(C.POINTER(C.c_int64)) if offs is not None else None, 0 if offs is None else len(offs) - 1,
                int(bool(add_intercept)), *tail_args, C.addressof(out_a), C.addressof(out_s))
        L.check(rc)
        return pa.Array._import_from_c(C.addressof(out_a), C.addressof(out_s))
    finally:
        for e in [ex_t] + ex_f + ([ex_w] if ex_w else []):
            e.close()


def _statistics_arrow(self, target, features, *, target_name: str = "y", weights=None, offsets=None, add_intercept: bool = False, **kw):
    """``pols_least_squares_statistics_arrow`` (plugin least_squares_statistics, src/expressions.rs:448-509): the ``statistics``
    struct, one row per group."""
    p = _ols_params(self._lib, **kw)
    return _arrow_call(self, self._lib.pols_least_squares_statistics_arrow, target, target_name, features, weights, offsets, add_intercept,
                       (C.byref(p),))


def _multi_target_arrow(self, targets, features, *, weights=None, offsets=None, add_intercept: bool = False, **kw):
    """``pols_multi_target_least_squares_arrow`` (src/expressions.rs:511-591): ``targets`` is a pyarrow StructArray (or a chunked
    one); returns the ``predictions`` struct with the same field names."""
    p = _ols_params(self._lib, **kw)
    return _arrow_call(self, self._lib.pols_multi_target_least_squares_arrow, targets, "y", features, weights, offsets, add_intercept,
                       (C.byref(p),))


def _rls_arrow(self, target, features, *, target_name: str = "y", weights=None, offsets=None, add_intercept: bool = False,
               mode: str = "predictions", half_life: Optional[float] = None, initial_state_covariance: Optional[float] = 10.0,
               initial_state_mean=None, null_policy: str = "drop"):
    """``pols_recursive_least_squares_arrow`` (src/expressions.rs:593-646)."""
    p = L.RlsParams()
    self._lib.pols_rls_params_default(C.byref(p))
    p.has_half_life, p.half_life = int(half_life is not None), float(half_life) if half_life is not None else 0.0
    p.initial_state_covariance = float(10.0 if initial_state_covariance is None else initial_state_covariance)
    mean = None
    if initial_state_mean is not None and mode == "coefficients":          # ex.rs:636: the prediction form passes None
        kt = len(features) + int(bool(add_intercept))
        mean = np.ascontiguousarray(np.broadcast_to(np.asarray(initial_state_mean, dtype=np.float64), (kt,)))
        p.initial_state_mean = mean.ctypes.data_as(C.POINTER(C.c_double))
    p.null_policy = L.NULL_POLICIES[null_policy]
    return _arrow_call(self, self._lib.pols_recursive_least_squares_arrow, target, target_name, features, weights, offsets, add_intercept,
                       (C.byref(p), {"predictions": 0, "coefficients": 2}[mode]))


def _rolling_arrow(self, target, features, *, window_size: int, target_name: str = "y", weights=None, offsets=None,
                   add_intercept: bool = False, mode: str = "predictions", min_periods: Optional[int] = None,
                   use_woodbury: Optional[bool] = None, alpha: Optional[float] = None, null_policy: str = "drop_window"):
    """``pols_rolling_least_squares_arrow`` (src/expressions.rs:648-701)."""
    p = L.RollingParams()
    self._lib.pols_rolling_params_default(C.byref(p))
    p.window_size = int(window_size)
    p.min_periods = -1 if min_periods is None else int(min_periods)
    p.use_woodbury = -1 if use_woodbury is None else int(bool(use_woodbury))
    p.alpha = float(alpha) if alpha is not None else 0.0
    p.null_policy = L.NULL_POLICIES[null_policy]
    return _arrow_call(self, self._lib.pols_rolling_least_squares_arrow, target, target_name, features, weights, offsets, add_intercept,
                       (C.byref(p), {"predictions": 0, "coefficients": 2}[mode]))


def _predict_arrow(self, coefficients, features, *, add_intercept: bool = False, null_policy: str = "zero", name: Optional[str] = None):
    """``pols_predict_arrow`` (src/expressions.rs:706-741): ``coefficients`` is a pyarrow StructArray with one row per input row."""
    import pyarrow as pa

    feats = list(features.items()) if isinstance(features, dict) else [("", f) for f in features]
    ex_c = _ArrowExport("coefficients", coefficients)
    ex_f = [_ArrowExport(n, f) for n, f in feats]
    try:
        fcols = (L.ArrowColumn * len(ex_f))(*[e.column for e in ex_f])
        out_a, out_s = (C.c_byte * 80)(), (C.c_byte * 72)()
        rc = self._lib.pols_predict_arrow(self._h, C.byref(ex_c.column), fcols, len(ex_f), int(bool(add_intercept)),
                                          L.NULL_POLICIES[null_policy], name.encode() if name else None, C.addressof(out_a),
                                          C.addressof(out_s))
        L.check(rc)
        return pa.Array._import_from_c(C.addressof(out_a), C.addressof(out_s))
    finally:
        for e in [ex_c] + ex_f:
            e.close()


Engine.least_squares_statistics_arrow = _statistics_arrow
Engine.multi_target_least_squares_arrow = _multi_target_arrow
Engine.recursive_least_squares_arrow = _rls_arrow
Engine.rolling_least_squares_arrow = _rolling_arrow
Engine.predict_arrow = _predict_arrow


class Comm:
    """``pols_comm``: the exchange step of the multi-GPU path (re-assembling an output column over RCCL / xGMI) behind the C-ABI.
    One process per GPU: rank 0 makes ``Comm.unique_id()``, ships the 128 bytes to the other ranks (any channel), every rank builds
    ``Comm(engine, world, rank, uid)``.  Collectives run on the engine's stream."""

    def __init__(self, eng: Engine, world: int, rank: int, unique_id: bytes):
        self._eng, self._lib = eng, eng._lib
        assert len(unique_id) == L.POLS_COMM_ID_BYTES
        buf = (C.c_char * L.POLS_COMM_ID_BYTES).from_buffer_copy(unique_id)
        h = C.c_void_p()
        L.check(self._lib.pols_comm_create(eng._h, buf, int(world), int(rank), C.byref(h)))
        self._h, self.world, self.rank = h, int(world), int(rank)

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_char * L.POLS_COMM_ID_BYTES)()
        L.check(L.lib().pols_comm_unique_id(buf))
        return bytes(buf.raw)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pols_comm_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def info(self) -> Dict:
        """``pols_comm_query``: what RCCL itself reports for this communicator (rank count, rank, device, PCI bus id, version code)."""
        ci = L.CommInfo()
        L.check(self._lib.pols_comm_query(self._h, C.byref(ci)))
        return {"nranks_seen": int(ci.nranks_seen), "rank_seen": int(ci.rank_seen), "device": int(ci.device),
                "pci_bus_id": ci.pci_bus_id.decode(), "rccl_version": int(ci.rccl_version)}

    def _counts(self, counts):
        if len(counts) != self.world:
            raise ValueError("one count per rank")
        return (C.c_int64 * self.world)(*[int(c) for c in counts])

    def allgather_rows(self, local, counts, out=None):
        """Every rank receives all rows in rank order; ``local`` is this rank's [counts[rank], ...] CUDA tensor."""
        local = local.contiguous()
        row_bytes = local.element_size() * int(np.prod(local.shape[1:])) if local.ndim > 1 else local.element_size()
        total = int(sum(int(c) for c in counts))
        if out is None:
            out = torch.empty((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        L.check(self._lib.pols_comm_allgather_rows(self._h, C.c_void_p(local.data_ptr()), self._counts(counts), C.c_int64(row_bytes),
                                                   C.c_void_p(out.data_ptr())))
        return out

    def gather_rows(self, local, counts, root: int = 0, out=None):
        """Rows of every rank on ``root`` (rank order); ``None`` elsewhere."""
        local = local.contiguous()
        row_bytes = local.element_size() * int(np.prod(local.shape[1:])) if local.ndim > 1 else local.element_size()
        if self.rank == root and out is None:
            out = torch.empty((int(sum(int(c) for c in counts)),) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        L.check(self._lib.pols_comm_gather_rows(self._h, C.c_void_p(local.data_ptr()), self._counts(counts), C.c_int64(row_bytes),
                                                C.c_int(root), C.c_void_p(out.data_ptr() if self.rank == root else 0)))
        return out if self.rank == root else None


def comm_create_all(engines: Sequence[Engine]) -> List["Comm"]:
    """``pols_comm_create_all``: one process driving several devices -- rank i of the returned communicators lives on
    ``engines[i]``'s device (ncclCommInitAll)."""
    n = len(engines)
    hs = (C.c_void_p * n)(*[e._h for e in engines])
    out = (C.c_void_p * n)()
    L.check(L.lib().pols_comm_create_all(hs, n, out))
    comms = []
    for i, e in enumerate(engines):
        c = Comm.__new__(Comm)
        c._eng, c._lib, c._h, c.world, c.rank = e, e._lib, C.c_void_p(out[i]), n, i
        comms.append(c)
    return comms


def least_squares_sharded(engines: Sequence[Engine], y, x_cols: Sequence, offsets, *, comms: Optional[Sequence["Comm"]] = None,
                          out: str = "host", weights=None, valid=None, add_intercept: bool = False, want: Sequence[str] = ("pred",),
                          **kwargs) -> Dict:
    """``pols_least_squares_sharded``: ONE process, ``len(engines)`` devices.  ``y`` / ``x_cols`` / ``weights`` are HOST (numpy) columns
    of the whole group-sorted frame; the groups are cut into contiguous ranges balanced by rows, range r runs on ``engines[r]``'s
    device from its own host thread.  ``out="host"``: numpy outputs for the whole frame (each device copies its slice home, no
    collective).  ``out="device"``: torch tensors on ``engines[0]``'s device, assembled over RCCL (``comms`` from
    :func:`comm_create_all`): coefficients all-gathered, predictions / residuals / status gathered to device 0."""
    e0 = engines[0]
    plan = e0.plan_least_squares(y, x_cols, offsets, weights=weights, valid=valid, add_intercept=add_intercept, want=(), **kwargs)
    b = plan._b
    if b.mem != L.POLS_MEM_HOST:
        raise ValueError("the sharded entry takes host (numpy) columns")
    dt = plan._keep[0][0].dtype
    kt = b.n_features + b.add_intercept
    res: Dict = {}
    if out == "host":
        mk = lambda shape, d: np.empty(shape, dtype=d)  # noqa: E731
    else:
        tdt = torch.float32 if dt == np.float32 else torch.float64
        dev = torch.device("cuda", e0.device)
        mk = lambda shape, d: torch.empty(shape, dtype=(torch.int32 if d == np.int32 else tdt), device=dev)  # noqa: E731
    if "coef" in want:
        res["coef"] = mk((b.n_groups, kt), dt)
    if "pred" in want:
        res["pred"] = mk((b.n_rows,), dt)
    if "resid" in want:
        res["resid"] = mk((b.n_rows,), dt)
    if "status" in want:
        res["status"] = mk((b.n_groups,), np.int32)
    o = L.Out(coef=Engine._ptr(res.get("coef")), pred=Engine._ptr(res.get("pred")), resid=Engine._ptr(res.get("resid")),
              status=Engine._ptr(res.get("status")))
    n = len(engines)
    hs = (C.c_void_p * n)(*[e._h for e in engines])
    cs = (C.c_void_p * n)(*[c._h for c in comms]) if comms is not None else None
    b.offsets_generation = 0
    L.check(L.lib().pols_least_squares_sharded(hs, cs, n, C.byref(b), C.byref(plan._p), C.byref(o),
                                               L.POLS_MEM_HOST if out == "host" else L.POLS_MEM_DEVICE))
    return res


def partition_groups_native(offsets, world: int):
    """``pols_partition_groups``: boundaries b[0..world] of contiguous group ranges with near-equal row counts."""
    offs = np.ascontiguousarray(offsets, dtype=np.int64)
    b = (C.c_int64 * (world + 1))()
    L.check(L.lib().pols_partition_groups(offs.ctypes.data_as(C.POINTER(C.c_int64)), len(offs) - 1, int(world), b))
    return [int(v) for v in b]


class Plan:
    """A marshalled call: ctypes structs + references that keep every borrowed buffer alive."""

    _generation = 0

    def __init__(self, eng: Engine, fn, batch, params, out, results: Dict, keep):
        self._eng, self._fn, self._b, self._p, self._o, self.results, self._keep = eng, fn, batch, params, out, results, keep
        # the plan keeps its offsets array alive and never rewrites it: promise that to the library (O(1) offsets check per run)
        Plan._generation += 1
        batch.offsets_generation = Plan._generation
        self._args = (eng._h, C.byref(batch), C.byref(params), C.byref(out))

    def set_output(self, key: str, buf) -> None:
        """Re-point one output (``coef`` / ``pred`` / ``resid`` / ``status``) at another pre-allocated buffer."""
        setattr(self._o, key, Engine._ptr(buf))
        self.results[key] = buf

    def run(self) -> Dict:
        rc = self._fn(*self._args)
        if rc < 0:
            L.check(rc)
        return self.results

    def stream_probe(self, mode: int = 0) -> None:
        """``pols_stream_probe_ex`` on this plan's batch: every input column read, their sum written over the ``pred`` output -- the
        bandwidth ceiling of the plan's traffic mix (a measurement aid; the predictions are garbage afterwards).  mode 0: the launch
        shape of the resident static kernels; mode 1: a persistent grid-stride stream (include/pols_mi355x_debug.h)."""
        L.check(self._eng._lib.pols_stream_probe_ex(self._eng._h, C.byref(self._b), C.c_void_p(Engine._ptr(self.results["pred"])), int(mode)))


_default: Dict[int, Engine] = {}


def default_engine(device: int = 0) -> Engine:
    if device not in _default:
        _default[device] = Engine(device)
    return _default[device]


def _engine_predict(self, x_cols: Sequence, coef, offsets=None, *, add_intercept: bool = False, null_policy: str = "ignore"):
    """``predict`` plugin body (src/expressions.rs:706-741): row-wise sum_j x[t, j] * coef[t, j];
    ``coef`` is n_rows x (k + add_intercept) in the columns' dtype (numpy -> host path, torch CUDA -> device path).  ``null_policy``
    is the plugin's kwarg: "zero" counts null (NaN) features as 0, "drop" / "ignore" leave the rows with a null anywhere null."""
    cols = list(x_cols)
    n = cols[0].numel() if _is_torch(cols[0]) else len(cols[0])
    offs = np.asarray([0, n] if offsets is None else offsets, dtype=np.int64)
    b, keep, dev, dt = self._batch(cols[0], cols, offs, None, None, add_intercept)
    if dev:
        coef_k = coef.to(dt).contiguous()
        out = torch.empty(n, dtype=dt, device=coef_k.device)
    else:
        coef_k = np.ascontiguousarray(coef, dtype=dt)
        out = np.empty(n, dtype=dt)
    rc = self._lib.pols_predict_policy(self._h, C.byref(b), C.c_void_p(self._ptr(coef_k)), C.c_int64(coef_k.shape[0]),
                                       C.c_int32(L.NULL_POLICIES[null_policy]), C.c_void_p(self._ptr(out)))
    L.check(rc)
    return out


Engine.predict = _engine_predict
