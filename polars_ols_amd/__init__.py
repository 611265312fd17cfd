"""polars_ols_amd -- MI355X-native batched least-squares engine behind the ``least_squares`` plugin surface
of azmyrajab/polars_ols.

Only the hot path lives here: ``csrc/`` (hand-written gfx950 HIP kernels + the C-ABI of
``include/pols_mi355x.h``) and the host-side mirror of the reference's operator interface
(``least_squares.py``).  Importing the package does not need a GPU; computing does, and there is no CPU fallback.
"""
from ._lib import LIB_PATH, PolsError, PolsPanic, build  # noqa: F401
from .engine import Engine, comm_create_all, default_engine, least_squares_sharded  # noqa: F401
from .least_squares import (  # noqa: F401
    Coefficients, Statistics, Expr, Frame, LeastSquares, OLSKwargs, RLSKwargs, RollingKwargs, col, struct, compute_least_squares,
    compute_multi_target_least_squares,
    compute_least_squares_from_formula, compute_recursive_least_squares, compute_rolling_least_squares, predict,
)

__version__ = "0.1.0"
