"""ctypes binding of ``libpols_mi355x.so`` (C-ABI declared in ``include/pols_mi355x.h``).

The library is the product's only compute path: if it is missing, or no gfx950 device is usable, every
entry raises -- there is no CPU / torch fallback (the CPU oracle under ``oracle/`` is test infrastructure
and is never imported from here).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path
from typing import Optional

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libpols_mi355x.so"
CSRC = _PKG / "csrc"

POLS_F32, POLS_F64 = 0, 1
POLS_MEM_HOST, POLS_MEM_DEVICE = 0, 1
POLS_MAX_FEATURES = 32

SOLVE_METHODS = {None: 0, "qr": 1, "svd": 2, "chol": 3, "lu": 4, "cd": 5, "cd_active_set": 6}
NULL_POLICIES = {"ignore": 0, "zero": 1, "drop": 2, "drop_zero": 3, "drop_y_zero_x": 4, "drop_window": 5}
ERRORS = {-1: "POLS_ERR_INVALID", -2: "POLS_ERR_UNSUPPORTED", -3: "POLS_ERR_HIP", -4: "POLS_ERR_PANIC",
          -5: "POLS_ERR_NO_DEVICE"}


class PolsError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{ERRORS.get(code, code)}: {msg}")
        self.code = code


class PolsPanic(PolsError):
    """The reference would ``panic!`` / ``assert!`` on these arguments (surfaced by pyo3-polars as an exception)."""


class OlsParams(C.Structure):
    _fields_ = [("alpha", C.c_double), ("l1_ratio", C.c_double), ("has_l1_ratio", C.c_int32),
                ("max_iter", C.c_int64), ("tol", C.c_double), ("positive", C.c_int32),
                ("solve_method", C.c_int32), ("rcond", C.c_double), ("has_rcond", C.c_int32),
                ("null_policy", C.c_int32)]


class RlsParams(C.Structure):
    _fields_ = [("half_life", C.c_double), ("has_half_life", C.c_int32),
                ("initial_state_covariance", C.c_double), ("initial_state_mean", C.POINTER(C.c_double)),
                ("null_policy", C.c_int32)]


class RollingParams(C.Structure):
    _fields_ = [("window_size", C.c_int64), ("min_periods", C.c_int64), ("use_woodbury", C.c_int32),
                ("alpha", C.c_double), ("null_policy", C.c_int32)]


class Batch(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("mem", C.c_int32), ("n_rows", C.c_int64), ("n_groups", C.c_int64),
                ("group_offsets", C.POINTER(C.c_int64)), ("n_features", C.c_int32), ("y", C.c_void_p),
                ("x_cols", C.POINTER(C.c_void_p)), ("weights", C.c_void_p), ("valid", C.c_void_p),
                ("add_intercept", C.c_int32), ("offsets_generation", C.c_uint64), ("null_free", C.c_int32)]


class Out(C.Structure):
    _fields_ = [("coef", C.c_void_p), ("pred", C.c_void_p), ("resid", C.c_void_p), ("status", C.c_void_p)]


class CommInfo(C.Structure):          # pols_comm_info
    _fields_ = [("nranks_seen", C.c_int32), ("rank_seen", C.c_int32), ("device", C.c_int32), ("rccl_version", C.c_int32),
                ("pci_bus_id", C.c_char * 32)]


class ArrowColumn(C.Structure):       # pols_arrow_column
    _fields_ = [("schema", C.c_void_p), ("chunks", C.POINTER(C.c_void_p)), ("n_chunks", C.c_int32)]


class StatsOut(C.Structure):
    _fields_ = [("r2", C.c_void_p), ("mae", C.c_void_p), ("mse", C.c_void_p),
                ("std_err", C.c_void_p), ("t_values", C.c_void_p), ("p_values", C.c_void_p)]


EXPORTS = [
    "pols_device_count", "pols_version", "pols_last_error", "pols_create", "pols_destroy", "pols_set_stream",
    "pols_use_private_stream",
    "pols_synchronize", "pols_set_option",
    "pols_ols_params_default", "pols_rls_params_default", "pols_rolling_params_default",
    "pols_least_squares", "pols_recursive_least_squares", "pols_rolling_least_squares", "pols_predict", "pols_predict_policy",
    "pols_least_squares_statistics", "pols_multi_target_least_squares",
    "pols_layout_create", "pols_layout_destroy", "pols_layout_n_rows", "pols_layout_n_groups", "pols_layout_is_identity",
    "pols_layout_group_offsets", "pols_layout_group_keys", "pols_layout_take", "pols_layout_untake", "pols_layout_row_groups",
    "pols_partition_groups", "pols_comm_unique_id", "pols_comm_create", "pols_comm_create_all", "pols_comm_destroy",
    "pols_comm_world_size", "pols_comm_rank", "pols_comm_query", "pols_comm_group_begin", "pols_comm_group_end", "pols_comm_allgather_rows",
    "pols_comm_gather_rows", "pols_least_squares_arrow", "pols_least_squares_statistics_arrow",
    "pols_multi_target_least_squares_arrow", "pols_recursive_least_squares_arrow", "pols_rolling_least_squares_arrow",
    "pols_predict_arrow", "pols_least_squares_sharded",
]
# measurement aids (include/pols_mi355x_debug.h): not part of the reference interface
DEBUG_EXPORTS = ["pols_timing_enable", "pols_timing_collect", "pols_last_kernel_name", "pols_stream_probe", "pols_stream_probe_ex"]
POLS_COMM_ID_BYTES = 128


def build(force: bool = False, jobs: int = 8) -> Path:
    """Compile every HIP source for gfx950 into the in-tree shared library (hipcc cross-compiles without a GPU)."""
    env = dict(os.environ)
    args = ["make", "-C", str(CSRC), f"-j{jobs}"]
    if force:
        args.append("-B")
    subprocess.run(args, check=True, env=env, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C polars_ols_amd/csrc`). polars_ols_amd has no CPU fallback.")
        L = C.CDLL(str(LIB_PATH))
        L.pols_version.restype = C.c_char_p
        L.pols_last_error.restype = C.c_char_p
        L.pols_last_kernel_name.restype = C.c_char_p
        L.pols_last_kernel_name.argtypes = [C.c_void_p]
        L.pols_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.pols_destroy.argtypes = [C.c_void_p]
        L.pols_destroy.restype = None
        L.pols_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.pols_use_private_stream.argtypes = [C.c_void_p]
        L.pols_synchronize.argtypes = [C.c_void_p]
        L.pols_timing_enable.argtypes = [C.c_void_p, C.c_int]
        L.pols_timing_collect.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
        L.pols_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.pols_ols_params_default.argtypes = [C.POINTER(OlsParams)]
        L.pols_rls_params_default.argtypes = [C.POINTER(RlsParams)]
        L.pols_rolling_params_default.argtypes = [C.POINTER(RollingParams)]
        L.pols_least_squares.argtypes = [C.c_void_p, C.POINTER(Batch), C.POINTER(OlsParams), C.POINTER(Out)]
        L.pols_recursive_least_squares.argtypes = [C.c_void_p, C.POINTER(Batch), C.POINTER(RlsParams), C.POINTER(Out)]
        L.pols_rolling_least_squares.argtypes = [C.c_void_p, C.POINTER(Batch), C.POINTER(RollingParams), C.POINTER(Out)]
        L.pols_predict.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_void_p, C.c_int64, C.c_void_p]
        L.pols_stream_probe.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_void_p]
        L.pols_stream_probe_ex.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_void_p, C.c_int]
        L.pols_predict_policy.argtypes = [C.c_void_p, C.POINTER(Batch), C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
        L.pols_multi_target_least_squares.argtypes = [C.c_void_p, C.POINTER(Batch), C.POINTER(C.c_void_p), C.c_int32,
                                                      C.POINTER(OlsParams), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]
        L.pols_least_squares_statistics.argtypes = [C.c_void_p, C.POINTER(Batch), C.POINTER(OlsParams), C.POINTER(Out),
                                                    C.POINTER(StatsOut)]
        L.pols_layout_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]
        L.pols_layout_destroy.argtypes = [C.c_void_p]
        L.pols_layout_destroy.restype = None
        for fn in (L.pols_layout_n_rows, L.pols_layout_n_groups):
            fn.argtypes, fn.restype = [C.c_void_p], C.c_int64
        L.pols_layout_is_identity.argtypes = [C.c_void_p]
        for fn in (L.pols_layout_group_offsets, L.pols_layout_group_keys):
            fn.argtypes, fn.restype = [C.c_void_p], C.POINTER(C.c_int64)
        for fn in (L.pols_layout_take, L.pols_layout_untake):
            fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, C.c_int]
        L.pols_layout_row_groups.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.pols_least_squares_arrow.argtypes = [C.c_void_p, C.POINTER(ArrowColumn), C.POINTER(ArrowColumn), C.c_int32, C.POINTER(ArrowColumn),
                                               C.POINTER(C.c_int64), C.c_int64, C.c_int32, C.POINTER(OlsParams), C.c_int32, C.c_void_p, C.c_void_p]
        _ac = C.POINTER(ArrowColumn)
        _common = [C.c_void_p, _ac, _ac, C.c_int32, _ac, C.POINTER(C.c_int64), C.c_int64, C.c_int32]
        L.pols_least_squares_statistics_arrow.argtypes = _common + [C.POINTER(OlsParams), C.c_void_p, C.c_void_p]
        L.pols_multi_target_least_squares_arrow.argtypes = _common + [C.POINTER(OlsParams), C.c_void_p, C.c_void_p]
        L.pols_recursive_least_squares_arrow.argtypes = _common + [C.POINTER(RlsParams), C.c_int32, C.c_void_p, C.c_void_p]
        L.pols_rolling_least_squares_arrow.argtypes = _common + [C.POINTER(RollingParams), C.c_int32, C.c_void_p, C.c_void_p]
        L.pols_predict_arrow.argtypes = [C.c_void_p, _ac, _ac, C.c_int32, C.c_int32, C.c_int32, C.c_char_p, C.c_void_p, C.c_void_p]
        L.pols_least_squares_sharded.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.POINTER(Batch), C.POINTER(OlsParams),
                                                 C.POINTER(Out), C.c_int32]
        L.pols_partition_groups.argtypes = [C.POINTER(C.c_int64), C.c_int64, C.c_int, C.POINTER(C.c_int64)]
        L.pols_comm_unique_id.argtypes = [C.c_void_p]
        L.pols_comm_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.pols_comm_create_all.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)]
        L.pols_comm_destroy.argtypes, L.pols_comm_destroy.restype = [C.c_void_p], None
        L.pols_comm_world_size.argtypes = [C.c_void_p]
        L.pols_comm_rank.argtypes = [C.c_void_p]
        L.pols_comm_query.argtypes = [C.c_void_p, C.POINTER(CommInfo)]
        L.pols_comm_allgather_rows.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int64, C.c_void_p]
        L.pols_comm_gather_rows.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int64, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc < 0:
        msg = lib().pols_last_error().decode(errors="replace")
        raise (PolsPanic if rc == -4 else PolsError)(rc, msg)
