"""Host-side mirror of the reference's Python operator interface for the hot path.

Same names, argument meaning, defaults and error behaviour as ``polars_ols/least_squares.py`` and the
``least_squares`` namespace of ``polars_ols/__init__.py`` (reference file:line cited per item), re-stated over a
dict-of-columns frame because Polars is not available in the build image: columns are 1-D numpy arrays (host path)
or CUDA torch tensors (device path), a *null* is a NaN, and ``.over(key)`` is done here by a stable sort-by-key
(what Polars' ``.over`` gather does on the host) followed by ONE batched call into libpols_mi355x for all groups.

    from polars_ols_amd import Frame, col
    df = Frame({"y": y, "x1": x1, "x2": x2, "group": g})
    pred = df.select(col("y").least_squares.ols("x1", "x2", mode="predictions").over("group"))["y"]
    coef = df.select(col("y").least_squares.from_formula("x1 + x2", mode="coefficients").over("group"))["coefficients"]

What runs where: null-policy row filtering, sqrt(w) handling for null policies, group sort / scatter are data
marshalling (the reference does them with Polars ops in src/expressions.rs:201-296 and least_squares.py:163-239);
every solve and prediction runs in the HIP kernels behind the C-ABI.  There is no CPU compute path.
"""
from __future__ import annotations

import logging
import re
from dataclasses import asdict, dataclass
from typing import Any, Dict, List, Literal, Optional, Sequence, Set, Tuple, Union, get_args

import numpy as np

from ._lib import PolsPanic
from .engine import Engine, Layout, _is_torch, default_engine

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

logger = logging.getLogger(__name__)

__all__ = [
    "compute_least_squares", "compute_recursive_least_squares", "compute_rolling_least_squares",
    "compute_least_squares_from_formula", "compute_multi_target_least_squares", "predict",
    "OLSKwargs", "RLSKwargs", "RollingKwargs", "NullPolicy", "OutputMode", "SolveMethod",
    "Frame", "Expr", "col", "struct", "Coefficients", "Statistics", "LeastSquares",
]

# ---- polars_ols/least_squares.py:47-63 --------------------------------------------------------------------------
NullPolicy = Literal["zero", "drop", "ignore", "drop_zero", "drop_y_zero_x", "drop_window"]
OutputMode = Literal["predictions", "residuals", "coefficients", "statistics"]
SolveMethod = Literal["qr", "svd", "chol", "lu", "cd", "cd_active_set"]

_VALID_NULL_POLICIES: Set[str] = set(get_args(NullPolicy))
_VALID_OUTPUT_MODES: Set[str] = set(get_args(OutputMode))
_VALID_SOLVE_METHODS: Set[Any] = set(get_args(SolveMethod)).union({None})
_EPSILON: float = 1.0e-12


@dataclass
class Kwargs:  # least_squares.py:66-77
    null_policy: str = "ignore"

    def to_dict(self) -> Dict[str, Any]:
        return asdict(self)

    def __post_init__(self):
        assert self.null_policy in _VALID_NULL_POLICIES, \
            f"'null_policy' must be one of {_VALID_NULL_POLICIES}. You passed: {self.null_policy}"


@dataclass
class OLSKwargs(Kwargs):  # least_squares.py:80-118
    alpha: Optional[float] = 0.0
    l1_ratio: Optional[float] = None
    max_iter: Optional[int] = 1_000
    tol: Optional[float] = 1.0e-5
    positive: Optional[bool] = False
    solve_method: Optional[str] = None
    rcond: Optional[float] = None

    def __post_init__(self):
        valid_ols_policies = _VALID_NULL_POLICIES - {"drop_window"}
        assert self.null_policy in valid_ols_policies, \
            f"'null_policy' must be one of {valid_ols_policies}. You passed: {self.null_policy}"
        assert self.solve_method in _VALID_SOLVE_METHODS, \
            f"'solve_method' must be one of {_VALID_SOLVE_METHODS}. You passed: {self.solve_method}"


@dataclass
class RLSKwargs(Kwargs):  # least_squares.py:121-140
    half_life: Optional[float] = None
    initial_state_covariance: Optional[float] = 10.0
    initial_state_mean: Union[Optional[List[float]], float] = None
    null_policy: str = "drop"


@dataclass
class RollingKwargs(Kwargs):  # least_squares.py:143-160
    # (use_woodbury is accepted and does not select a code path; at 11 to 32 features a window without a Cholesky factorisation gives NaN
    #  where the reference's LU gives inf / NaN / huge numbers -- Engine.plan_rolling_least_squares' docstring, include/pols_mi355x.h)
    window_size: int = 1_000_000
    min_periods: Optional[int] = None
    use_woodbury: Optional[bool] = None
    alpha: Optional[float] = None
    null_policy: str = "drop_window"


# ---- frame / expression plumbing ---------------------------------------------------------------------------------

class Coefficients:
    """The coefficients struct of the reference (src/expressions.rs:114-143): one field per feature, NaN = null.

    ``values`` is [n_groups x k] for static models (``keys`` = the group keys in sorted order; Polars broadcasts the
    struct to every row of the group -- ``to_rows()``), [n_rows x k] for rls / rolling."""

    def __init__(self, names: Sequence[str], values, keys=None, row_group=None):
        self.names, self.values, self.keys, self._row_group = list(names), values, keys, row_group

    def unnest(self) -> Dict[str, Any]:
        return {n: self.values[:, j] for j, n in enumerate(self.names)}

    def to_rows(self):
        return self.values if self._row_group is None else self.values[self._row_group]

    def __repr__(self):
        return f"Coefficients(names={self.names}, shape={tuple(self.values.shape)})"


class Expr:
    """A column reference (optionally scaled) or a deferred least-squares expression."""

    def __init__(self, name: Optional[str] = None, scale: float = 1.0, fn=None, over=None, alias: Optional[str] = None,
                 factors: Tuple[str, ...] = ()):
        self._name, self._scale, self._fn, self._over, self._alias = name, scale, fn, over, alias
        self._factors = tuple(factors)                         # further columns multiplied in (formula interactions ``a:b``)

    # column arithmetic that the reference's own tests use on features (e.g. ``-pl.col("x2")``, test_ols.py:615) and that
    # its formula front-end builds for interaction terms (utils.py:104-106)
    def __neg__(self):
        return Expr(self._name, -self._scale, alias=self._alias, factors=self._factors)

    def __mul__(self, c):
        if isinstance(c, Expr):
            if c._fn is not None or self._fn is not None:
                raise TypeError("cannot multiply deferred least-squares expressions")
            return Expr(self._name, self._scale * c._scale, alias=self._alias, factors=self._factors + (c._name,) + c._factors)
        return Expr(self._name, self._scale * float(c), alias=self._alias, factors=self._factors)

    __rmul__ = __mul__

    def alias(self, name: str) -> "Expr":
        return Expr(self._name, self._scale, self._fn, self._over, name, self._factors)

    def over(self, key) -> "Expr":
        return Expr(self._name, self._scale, self._fn, key, self._alias, self._factors)

    @property
    def least_squares(self) -> "LeastSquares":
        return LeastSquares(self)

    @property
    def output_name(self) -> str:
        return self._alias or self._name

    def _column(self, frame: "Frame"):
        c = frame[self._name]
        for f in self._factors:
            c = c * frame[f]
        return c if self._scale == 1.0 else c * self._scale


def col(name: str) -> Expr:
    return Expr(name)


def struct(*cols) -> Expr:
    """pl.struct(...) stand-in: bundles the target columns of a multi-target regression."""
    fields = [parse_into_expr(c) for c in cols]
    e = Expr(fields[0]._name)
    e._fields = fields
    return e


def parse_into_expr(e) -> Expr:  # polars_ols/utils.py:21-58 (strings are column names)
    if isinstance(e, Expr):
        return e
    if isinstance(e, str):
        return Expr(e)
    raise TypeError(f"cannot parse {type(e)} into a column expression")


class Frame(dict):
    """dict of equally long 1-D columns (numpy or CUDA torch)."""

    def select(self, *exprs: Expr, engine: Optional[Engine] = None) -> Dict[str, Any]:
        out: Dict[str, Any] = {}
        for e in exprs:
            if e._fn is None:
                out[e.output_name] = e._column(self)
            else:
                name, val = e._fn(self, e._over, engine)
                out[e._alias or name] = val
        return out

    def with_columns(self, *exprs: Expr, engine: Optional[Engine] = None) -> "Frame":
        f = Frame(self)
        f.update(self.select(*exprs, engine=engine))
        return f


# ---- helpers over numpy / torch columns ---------------------------------------------------------------------------

def _xp(a):
    return torch if _is_torch(a) else np


def _take(a, idx):
    return a[idx]


def _to_index(idx, like):
    if _is_torch(idx):
        return idx if _is_torch(like) else idx.cpu().numpy()
    if _is_torch(like):
        return torch.as_tensor(idx, device=like.device)
    return idx


class _Groups:
    """What ``.over(key)`` needs (README.md:19, :57): the groups' offsets / keys and the row movers between frame order and the
    group-sorted order the batched entries take.  Built by the engine's native ingestion (``pols_layout_*``: stable radix sort,
    run lengths, gather kernels) wherever the key column lives; non-integer keys are dictionary-encoded first, like Polars'
    own group-by.  ``key is None`` is the whole-frame fit: one group, nothing moves."""

    def __init__(self, eng: Engine, key, n: int):
        self._lay, self.n = None, n
        if key is None:
            self.offsets, self.keys, self.identity = np.array([0, n], dtype=np.int64), None, True
            return
        codes, uniq = key, None
        if _is_torch(key) and not key.is_cuda:
            key = codes = key.numpy()
        if _is_torch(key):
            if key.dtype.is_floating_point or key.dtype.is_complex:
                uniq, codes = torch.unique(key, return_inverse=True)
                uniq = uniq.cpu().numpy()
        else:
            codes = np.asarray(key)
            if codes.dtype.kind not in "iub":
                uniq, codes = np.unique(codes, return_inverse=True)
        self._lay = Layout(eng, codes)
        self.offsets, self.identity = self._lay.offsets, self._lay.identity
        self.keys = self._lay.keys if uniq is None else uniq[self._lay.keys]

    def take(self, cols):
        """frame order -> group order; ``None`` entries pass through."""
        return list(cols) if self.identity else self._lay.take(cols)

    def untake(self, a):
        """group order -> frame order (a column or an [n, k] table)."""
        return a if self.identity else self._lay.untake([a])[0]

    def gid_frame(self, like):
        """index of its group for every frame row, where ``like`` lives."""
        if self._lay is None:
            g = np.zeros(self.n, dtype=np.int64)
        else:
            g = self._lay.row_groups()
        return _to_index(g, like)

    def gid_sorted(self, like):
        """the same for every group-sorted row."""
        if self._lay is not None and self._lay.on_device and _is_torch(like):
            return self._lay.take([self._lay.row_groups()])[0]
        return _to_index(np.repeat(np.arange(len(self.offsets) - 1, dtype=np.int64), np.diff(self.offsets)), like)


def _pre_process_data(frame: Frame, target: Expr, features: Sequence[Expr], sample_weights, add_intercept: bool):
    """least_squares.py:163-196 up to (not including) the sqrt_w multiplications, which the kernels fuse:
    returns (y, x columns, feature names, add_intercept flag for the engine, weights or None)."""
    names = [f.output_name for f in features]
    icpt = False
    if add_intercept:
        if any(n == "const" for n in names):
            logger.info("feature named 'const' already detected, assuming it is an intercept")  # :185-186
        else:
            names = names + ["const"]                                                          # appended LAST (:188)
            icpt = True
    w = None
    if sample_weights is not None:
        w = parse_into_expr(sample_weights)._column(frame)
        # sqrt_w = w.sqrt().fill_null(1e-12)  (:193): a null weight acts as weight 1e-24.  That fill happens BEHIND the C-ABI, on the
        # device (pols_least_squares: one pass over the weights column unless the batch promises null_free; the dynamic and Arrow
        # entries in dyn_prep.hip / arrow.hip) -- nothing to do here.
    return target._column(frame), [f._column(frame) for f in features], names, icpt, w


def _has_nan(a) -> bool:
    return bool(torch.isnan(a).any()) if _is_torch(a) else bool(np.isnan(a).any())


def _ones_like(a):
    return torch.ones_like(a) if _is_torch(a) else np.ones_like(a)


def _static_fit(eng: Engine, y, xs, offs, w, icpt: bool, want, kw: OLSKwargs):
    d = kw.to_dict()
    d.pop("null_policy")
    return eng.least_squares(y, xs, offs, weights=w, add_intercept=icpt, want=want, **d)


class Statistics(dict):
    """mode="statistics": the fields of the reference's struct (src/expressions.rs:448-466), one entry per group:
    ``r2 mae mse`` [G], ``feature_names``, ``coefficients standard_errors t_values p_values`` [G, k]; ``keys`` holds the
    group keys of an ``.over`` (None for a whole-frame fit, where G == 1)."""

    def __init__(self, names, out, keys):
        super().__init__(r2=out["r2"], mae=out["mae"], mse=out["mse"], feature_names=list(names),
                         coefficients=out["coef"], standard_errors=out["std_err"], t_values=out["t_values"],
                         p_values=out["p_values"])
        self.keys_ = keys


def _static_statistics(eng: Engine, y, xs, offs, w, icpt: bool, kw: OLSKwargs, names, keys) -> Statistics:
    d = kw.to_dict()
    out = eng.least_squares_statistics(y, xs, offs, weights=w, add_intercept=icpt, **d)
    st = out["status"]
    bad = bool((st == 4).any())
    if bad:  # the reference asserts df > 0 and panics the whole query (src/statistics.rs:131-134)
        raise PolsPanic(-4, "Degrees of freedom <= 0. Cannot compute standard errors.")
    return Statistics(names, out, keys)


def _apply_static(frame: Frame, over, eng: Optional[Engine], target: Expr, features: Sequence[Expr], sample_weights,
                  add_intercept: bool, mode: str, kw: OLSKwargs):
    """compute_least_squares body: least_squares.py:199-239 + src/expressions.rs:390-446 (null policies :201-296)."""
    y, xs, names, icpt, w = _pre_process_data(frame, target, features, sample_weights, add_intercept)
    n = y.shape[0]
    eng = eng or default_engine(y.device.index or 0 if _is_torch(y) else 0)
    # ---- group layout (.over)
    grp = _Groups(eng, None if over is None else (frame[over] if isinstance(over, str) else over), n)
    offs, keys = grp.offsets, grp.keys
    moved = grp.take([y, w] + list(xs))
    y_s, w_s, xs_s = moved[0], moved[1], moved[2:]

    policy = kw.null_policy
    if mode != "statistics":
        # Null policies are fused into the kernels (staging pass: dropped rows get weight 0, surviving nulls become 0;
        # prediction pass: zero-filled features, "drop" masks the rows that were not fitted) -- ex.rs:201-296, 398-427.
        want = ("coef",) if mode == "coefficients" else (("pred",) if mode == "predictions" else ("resid",))
        d = kw.to_dict()
        out = eng.least_squares(y_s, xs_s, offs, weights=w_s, add_intercept=icpt, want=want, **d)
        coef, pred = out.get("coef"), out.get("pred") if mode == "predictions" else out.get("resid")
    else:
        # handle_nulls ahead of the statistics code (ex.rs:469-471): the entry filters / zero-fills on the device itself
        return "statistics", _static_statistics(eng, y_s, xs_s, offs, w_s, icpt, kw, names, keys)
    if mode == "coefficients":
        # without .over the single struct broadcasts to every row of the frame, like a Polars scalar (gid is all zeros)
        return "coefficients", Coefficients(names, coef, keys, grp.gid_frame(coef))
    return target.output_name, grp.untake(pred)                # back to the frame's row order


def _apply_dynamic(frame: Frame, over, eng: Optional[Engine], target: Expr, features: Sequence[Expr], sample_weights,
                   add_intercept: bool, mode: str, kind: str, kw):
    """compute_recursive_least_squares / compute_rolling_least_squares bodies (ls.py:332-409 around
    src/expressions.rs:593-701): the plugin gets sqrt_w-scaled, intercept-extended columns, a validity mask from the
    null policy, and zero-filled data (NullPolicy::Zero conversion, ex.rs:603,629,656,683).  All of that is done by the
    C-ABI entries themselves (csrc/dyn_prep.hip); this function only lays the groups out and hands the raw columns over."""
    y, xs, names, icpt, w = _pre_process_data(frame, target, features, sample_weights, add_intercept)
    n = y.shape[0]
    eng = eng or default_engine(y.device.index or 0 if _is_torch(y) else 0)
    policy = kw.null_policy
    grp = _Groups(eng, None if over is None else (frame[over] if isinstance(over, str) else over), n)
    moved = grp.take([y, w] + list(xs))                        # raw columns: everything else happens behind the C-ABI
    ys, ws, xss = moved[0], moved[1], moved[2:]
    want = ("coef",) if mode == "coefficients" else ("pred",)
    if kind == "rls":
        mean = kw.initial_state_mean if mode == "coefficients" else None      # quirk: ex.rs:636 passes None for predictions
        out = eng.recursive_least_squares(ys, xss, grp.offsets, weights=ws, add_intercept=icpt, want=want, half_life=kw.half_life,
                                          initial_state_covariance=kw.initial_state_covariance,
                                          initial_state_mean=mean, null_policy=policy)
    else:
        out = eng.rolling_least_squares(ys, xss, grp.offsets, weights=ws, add_intercept=icpt, want=want,
                                        window_size=kw.window_size, min_periods=kw.min_periods, use_woodbury=kw.use_woodbury,
                                        alpha=kw.alpha, null_policy=policy)
    res = grp.untake(out["coef"] if mode == "coefficients" else out["pred"])
    if mode == "coefficients":
        return "coefficients", Coefficients(names, res)
    return target.output_name, (y - res if mode == "residuals" else res)


# ---- the reference's module-level functions (least_squares.py:242-491) -------------------------------------------

def compute_least_squares(target, *features, sample_weights=None, add_intercept: bool = False,
                          mode: str = "predictions", ols_kwargs: Optional[OLSKwargs] = None) -> Expr:
    assert mode in _VALID_OUTPUT_MODES, f"'mode' must be one of {_VALID_OUTPUT_MODES}"
    kw = ols_kwargs or OLSKwargs()
    t, fs = parse_into_expr(target), [parse_into_expr(f) for f in features]
    return Expr(t._name, fn=lambda frame, over, eng: _apply_static(frame, over, eng, t, fs, sample_weights, add_intercept, mode, kw))


def compute_multi_target_least_squares(targets, *features, sample_weights=None, add_intercept: bool = False,
                                       mode: str = "predictions", ols_kwargs: Optional[OLSKwargs] = None) -> Expr:
    """Several targets regressed on the same features (least_squares.py:282-328, src/expressions.rs:521-591).  ``targets``
    is the list of target columns (the fields of the reference's struct); the result is a dict target -> column.

    A row is valid only if EVERY target (and, for the drop policies, every feature) is non-null (compute_is_valid_mask
    with m targets, ex.rs:539), so the joint mask is applied to all targets first and each one is then solved by the
    engine's multi-target entry: one Gram pass over [X | targets] and one factorisation shared by all targets -- what
    solve_multi_target (ls.rs:243-260) does with one SVD."""
    kw = ols_kwargs or OLSKwargs()
    msg = "Consider running multiple independent regressions on a multi-expression target!"
    assert not kw.positive and (kw.l1_ratio is None or kw.l1_ratio == 0.0), (
        "Multi-target regression is only supported for unconstrained OLS & Ridge problems." + msg)
    assert kw.solve_method in {"svd", None}, "only solve_method='svd' is supported for multi-target regressions"
    if mode not in ("predictions", "residuals"):
        raise NotImplementedError("Only mode={'predictions', 'residuals'} is currently supported. " + msg)
    if isinstance(targets, Expr) and getattr(targets, "_fields", None):
        targets = targets._fields
    elif isinstance(targets, (str, Expr)):
        targets = [targets]
    ts, fs = [parse_into_expr(t) for t in targets], [parse_into_expr(f) for f in features]

    def run(frame: Frame, over, eng):
        y0, xs, names, icpt, w = _pre_process_data(frame, ts[0], fs, sample_weights, add_intercept)
        ys = [y0] + [t._column(frame) for t in ts[1:]]
        n = y0.shape[0]
        eng = eng or default_engine(y0.device.index or 0 if _is_torch(y0) else 0)
        grp = _Groups(eng, None if over is None else (frame[over] if isinstance(over, str) else over), n)
        offs = grp.offsets
        moved = grp.take([w] + list(ys) + list(xs))
        w_s, ys_s, xs_s = moved[0], moved[1:1 + len(ys)], moved[1 + len(ys):]
        # joint validity mask, fit on the rows it leaves, every row predicted from zero-filled features, "drop" masked
        # (ex.rs:539-585): all of it inside the entry (csrc/dyn_prep.hip).  A Polars caller reads null_count off its Series; here the
        # columns are arrays, so the null count is one reduction per column -- cheap next to the compaction pass (every column read
        # AND rewritten) that the entry runs whenever it cannot be promised a null-free frame.
        null_free = not any(_has_nan(c) for c in list(ys_s) + list(xs_s) + ([w_s] if w_s is not None else []))
        preds = eng.multi_target_least_squares(ys_s, xs_s, offs, weights=w_s, add_intercept=icpt, want=("pred",), alpha=kw.alpha,
                                               solve_method=kw.solve_method, rcond=kw.rcond, null_policy=kw.null_policy,
                                               null_free=null_free)["pred"]
        out = {}
        for t, y_s, pr in zip(ts, ys_s, preds):
            val = pr if mode == "predictions" else y_s - pr
            out[t.output_name] = grp.untake(val)
        return "predictions", out

    return Expr(ts[0]._name, fn=run)


def compute_recursive_least_squares(target, *features, sample_weights=None, add_intercept: bool = False,
                                    mode: str = "predictions", rls_kwargs: Optional[RLSKwargs] = None) -> Expr:
    valid_output_modes = _VALID_OUTPUT_MODES - {"statistics"}
    assert mode in valid_output_modes, f"'mode' must be one of {valid_output_modes}"
    kw = rls_kwargs or RLSKwargs()
    t, fs = parse_into_expr(target), [parse_into_expr(f) for f in features]
    return Expr(t._name, fn=lambda frame, over, eng: _apply_dynamic(frame, over, eng, t, fs, sample_weights, add_intercept, mode, "rls", kw))


def compute_rolling_least_squares(target, *features, sample_weights=None, add_intercept: bool = False,
                                  mode: str = "predictions", rolling_kwargs: Optional[RollingKwargs] = None) -> Expr:
    valid_output_modes = _VALID_OUTPUT_MODES - {"statistics"}
    assert mode in valid_output_modes, f"'mode' must be one of {valid_output_modes}"
    kw = rolling_kwargs or RollingKwargs()
    t, fs = parse_into_expr(target), [parse_into_expr(f) for f in features]
    return Expr(t._name, fn=lambda frame, over, eng: _apply_dynamic(frame, over, eng, t, fs, sample_weights, add_intercept, mode, "rolling", kw))


_FORMULA_TOKEN = re.compile(r"\s*(?:(?P<name>[A-Za-z_][A-Za-z_0-9.]*)|(?P<num>\d+)|(?P<op>[+\-:*]))")


def _formula_terms(side: str) -> List[Tuple[str, ...]]:
    """One side of a patsy formula -> its term list in written order, each term a tuple of factor (column) names; the
    intercept is the empty tuple.  Grammar (patsy's operators on plain columns): ``+`` adds terms, ``-`` removes them,
    ``a:b`` is the interaction (product column), ``a*b`` = ``a + b + a:b``, ``1`` / ``0`` add / remove the intercept.
    Anything else patsy understands (function calls like ``log(x)`` / ``C(g)`` / ``I(..)``, parentheses, ``**``, ``/``,
    ``%in%``) raises NotImplementedError: the reference supports column-only formulas too (utils.py:66-72, 96-100)."""
    toks: List[Tuple[str, str]] = []
    pos = 0
    side = side.rstrip()
    while pos < len(side):
        m = _FORMULA_TOKEN.match(side, pos)
        if not m:
            raise NotImplementedError(f"formula syntax not supported at {side[pos:].strip()[:12]!r}: only column names joined by + - : * (and 0 / 1)")
        kind = m.lastgroup
        toks.append((kind, m.group(kind)))
        pos = m.end()
    i = 0

    def atom() -> List[Tuple[str, ...]]:
        nonlocal i
        if i >= len(toks) or toks[i][0] == "op":
            raise ValueError("formula: expected a column name")
        kind, v = toks[i]
        i += 1
        if kind == "num":
            if v not in ("0", "1"):
                raise NotImplementedError(f"formula: numeric term {v!r} (only 0 and 1 carry a meaning)")
            return [("<0>",)] if v == "0" else [()]
        return [(v,)]

    def interaction() -> List[Tuple[str, ...]]:
        nonlocal i
        left = atom()
        while i < len(toks) and toks[i] == ("op", ":"):
            i += 1
            right = atom()
            if left[0] in ((), ("<0>",)) or right[0] in ((), ("<0>",)):
                raise ValueError("formula: the intercept cannot take part in an interaction")
            left = [left[0] + tuple(f for f in right[0] if f not in left[0])]
        return left

    def product() -> List[Tuple[str, ...]]:
        nonlocal i
        left = interaction()
        while i < len(toks) and toks[i] == ("op", "*"):
            i += 1
            right = interaction()
            if any(t in ((), ("<0>",)) for t in left + right):
                raise ValueError("formula: the intercept cannot take part in an interaction")
            left = left + right + [a + tuple(f for f in b if f not in a) for a in left for b in right]
        return left

    terms: List[Tuple[str, ...]] = [()]                        # patsy starts every RHS with the intercept
    sign = "+"
    if toks and toks[0] in (("op", "+"), ("op", "-")):
        sign = toks[0][1]
        i = 1
    while True:
        for t in product():
            if t == ("<0>",):                                  # "+ 0" removes the intercept, "- 0" adds it
                t, add = (), sign == "-"
            else:
                add = sign == "+"
            if add and t not in terms:
                terms.append(t)
            elif not add and t in terms:
                terms.remove(t)
        if i >= len(toks):
            break
        if toks[i] not in (("op", "+"), ("op", "-")):
            raise ValueError(f"formula: unexpected {toks[i][1]!r}")
        sign = toks[i][1]
        i += 1
    return terms


def _term_expr(term: Tuple[str, ...]) -> Expr:
    """utils.py:101-106: one factor -> the column; several -> their product, named ``a:b``."""
    return col(term[0]) if len(term) == 1 else Expr(term[0], factors=term[1:], alias=":".join(term))


def _parse_formula(formula: str, include_dependent_variable: bool) -> Tuple[List[Expr], bool]:
    """build_expressions_from_patsy_formula (utils.py:61-108) without patsy (absent here): same term lists for formulas over
    plain columns (``y ~ x1 + x2:x3 - 1``), NotImplementedError for the rest.  The intercept flag follows the reference's
    rule TO THE LETTER: ``add_intercept = "-1" not in formula`` (utils.py:94) -- a literal substring test, so ``"x1 + x2 -1"``
    drops the intercept while ``"x1 + x2 - 1"`` (with a space) and ``"x1 + x2 + 0"`` keep it, whatever patsy's own reading."""
    if "(" in formula or ")" in formula:
        raise NotImplementedError("formula: function calls / categories / parentheses are not supported (utils.py:66-72, 96-100)")
    if "**" in formula:
        raise NotImplementedError("formula: '**' is not supported")
    parts = formula.split("~")
    if len(parts) > 2:
        raise ValueError("formula: more than one '~'")
    lhs, rhs = (parts[0], parts[1]) if len(parts) == 2 else ("", parts[0])
    lhs_terms = [t for t in _formula_terms(lhs) if t != ()] if lhs.strip() else []
    rhs_terms = [t for t in _formula_terms(rhs) if t != ()]    # the intercept term has no factors: skipped (utils.py:101-106)
    if include_dependent_variable:
        assert len(lhs_terms) == 1, "must provide exactly one LHS variable"
    else:
        assert len(lhs_terms) == 0, "can not provide LHS variables in this context"
    add_intercept = "-1" not in formula
    return [_term_expr(t) for t in lhs_terms + rhs_terms], add_intercept


def compute_least_squares_from_formula(formula: str, sample_weights=None, mode: str = "predictions", **kwargs) -> Expr:
    exprs, add_intercept = _parse_formula(formula, include_dependent_variable=True)   # ls.py:432-452
    if kwargs.get("half_life"):
        return compute_recursive_least_squares(exprs[0], *exprs[1:], add_intercept=add_intercept, sample_weights=sample_weights,
                                               mode=mode, rls_kwargs=RLSKwargs(**kwargs))
    if kwargs.get("window_size"):
        return compute_rolling_least_squares(exprs[0], *exprs[1:], add_intercept=add_intercept, sample_weights=sample_weights,
                                             mode=mode, rolling_kwargs=RollingKwargs(**kwargs))
    return compute_least_squares(exprs[0], *exprs[1:], add_intercept=add_intercept, sample_weights=sample_weights, mode=mode,
                                 ols_kwargs=OLSKwargs(**kwargs))


def predict(coefficients: Coefficients, *features, frame: Frame, null_policy: str = "zero", add_intercept: bool = False,
            name: Optional[str] = None, engine: Optional[Engine] = None):
    """least_squares.py:455-491 + src/expressions.rs:706-741: row-wise features . coefficients."""
    assert null_policy in _VALID_NULL_POLICIES, "'null_policy' must be one of {drop, ignore, zero}"
    fs = [parse_into_expr(f) for f in features]
    xs = [f._column(frame) for f in fs]
    if add_intercept and any(f.output_name == "const" for f in fs):
        logger.warning("feature named 'const' already detected, assuming it is the intercept")
        add_intercept = False
    rows = coefficients.to_rows()
    assert rows.shape[1] == len(xs) + int(add_intercept), "number of coefficients must match number of features!"  # ex.rs:717-721
    eng = engine or default_engine((xs[0].device.index or 0) if _is_torch(xs[0]) else 0)   # the engine of the columns' device
    # the zero fill of "zero" and the masking of "drop" are the plugin body's (ex.rs:725, :732-738): done by the kernel
    return eng.predict(xs, rows, add_intercept=add_intercept, null_policy=null_policy)


# ---- the `least_squares` namespace (polars_ols/__init__.py:35-295) -----------------------------------------------

class LeastSquares:
    def __init__(self, expr: Expr):
        self._expr = expr

    def least_squares(self, *features, sample_weights=None, add_intercept: bool = False, mode: str = "predictions",
                      null_policy: str = "ignore", solve_method: Optional[str] = None, multi_target: bool = False,
                      **ols_kwargs) -> Expr:
        fn = compute_least_squares if not multi_target else compute_multi_target_least_squares
        return fn(self._expr, *features, sample_weights=sample_weights, add_intercept=add_intercept, mode=mode,
                  ols_kwargs=OLSKwargs(null_policy=null_policy, solve_method=solve_method, **ols_kwargs))

    def ols(self, *features, **kwargs) -> Expr:
        return self.least_squares(*features, **kwargs)

    def multi_target_ols(self, *features, **kwargs) -> Expr:
        return self.least_squares(*features, multi_target=True, **kwargs)

    def wls(self, *features, sample_weights, **kwargs) -> Expr:
        return self.least_squares(*features, sample_weights=sample_weights, **kwargs)

    def ridge(self, *features, alpha: float, **kwargs) -> Expr:
        return self.least_squares(*features, alpha=alpha, l1_ratio=0.0, **kwargs)

    def lasso(self, *features, alpha: float, **kwargs) -> Expr:
        return self.least_squares(*features, alpha=alpha, l1_ratio=1.0, **kwargs)

    def elastic_net(self, *features, alpha: float, l1_ratio: float = 0.5, positive: bool = False, **kwargs) -> Expr:
        return self.least_squares(*features, alpha=alpha, l1_ratio=l1_ratio, positive=positive, **kwargs)

    def rls(self, *features, sample_weights=None, add_intercept: bool = False, mode: str = "predictions",
            null_policy: str = "drop", half_life: Optional[float] = None, initial_state_covariance: Optional[float] = 10.0,
            initial_state_mean=None) -> Expr:
        return compute_recursive_least_squares(
            self._expr, *features, sample_weights=sample_weights, add_intercept=add_intercept, mode=mode,
            rls_kwargs=RLSKwargs(null_policy=null_policy, half_life=half_life, initial_state_mean=initial_state_mean,
                                 initial_state_covariance=initial_state_covariance))

    def rolling_ols(self, *features, window_size: int, sample_weights=None, add_intercept: bool = False,
                    mode: str = "predictions", null_policy: str = "drop", min_periods: Optional[int] = None,
                    use_woodbury: Optional[bool] = None, alpha: Optional[float] = None) -> Expr:
        return compute_rolling_least_squares(
            self._expr, *features, sample_weights=sample_weights, add_intercept=add_intercept, mode=mode,
            rolling_kwargs=RollingKwargs(window_size=window_size, min_periods=min_periods, use_woodbury=use_woodbury,
                                         alpha=alpha, null_policy=null_policy))

    def expanding_ols(self, *features, **kwargs) -> Expr:
        return self.rls(*features, half_life=None, **kwargs)

    def from_formula(self, formula: str, **kwargs) -> Expr:
        features, add_intercept = _parse_formula(formula, include_dependent_variable=False)
        if kwargs.get("half_life"):
            return self.rls(*features, add_intercept=add_intercept, **kwargs)
        if kwargs.get("window_size"):
            return self.rolling_ols(*features, add_intercept=add_intercept, **kwargs)
        return self.least_squares(*features, add_intercept=add_intercept, **kwargs)

    def predict(self, *features, name: Optional[str] = None, add_intercept: bool = False, null_policy: str = "zero") -> Expr:
        """``pl.col("coefficients").least_squares.predict(x1, x2)`` (__init__.py:274-287): the namespace's column is a
        coefficients struct (a ``Coefficients`` held in the frame, e.g. by ``with_columns(... mode="coefficients")``)."""
        assert null_policy in _VALID_NULL_POLICIES, "'null_policy' must be one of {drop, ignore, zero}"
        src = self._expr

        def run(frame: Frame, over, eng):
            coefficients = frame[src._name]
            if not isinstance(coefficients, Coefficients):
                raise TypeError(f"column '{src._name}' does not hold a coefficients struct")
            return name or "predictions", predict(coefficients, *features, frame=frame, null_policy=null_policy,
                                                  add_intercept=add_intercept, engine=eng)

        return Expr(src._name, fn=run, alias=name or "predictions")

    def predict_from_formula(self, formula: str, name: Optional[str] = None) -> Expr:  # __init__.py:289-295
        features, add_intercept = _parse_formula(formula, include_dependent_variable=False)
        has_const = any(f.output_name == "const" for f in features)
        return self.predict(*features, name=name, add_intercept=add_intercept and not has_const)
