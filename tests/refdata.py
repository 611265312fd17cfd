"""Seeded synthetic frames shaped like the reference's own test fixture.

``make_data`` re-states ``_make_data`` of the reference test-suite
(reference tests/test_ols.py:22-51): ``np.random.default_rng(0)``, x ~ N(0,1)
(n x k), y = sum of the first floor(k(1-sparsity)) features + N(0, scale),
optional non-contiguous integer ``group`` key, optional 10 % nulls (NaN here:
the frames in this repo are dict-of-numpy, and NaN is what the plugin turns a
null into for the "ignore" policy, reference src/expressions.rs:53).
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np


def make_data(n_samples: int = 5_000, n_features: int = 2, n_groups: Optional[int] = None,
              scale: float = 0.1, sparsity: float = 0.0) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(0)
    x = rng.normal(size=(n_samples, n_features))
    eps = rng.normal(size=n_samples, scale=scale)
    out = {"x": x, "y": x[:, : int(n_features * (1.0 - sparsity))].sum(1) + eps}
    for i in range(n_features):
        out[f"x{i + 1}"] = np.ascontiguousarray(x[:, i])
    if n_groups is not None:
        out["group"] = rng.integers(n_groups, size=n_samples)
    return out


def notebook_make_data(n_samples: int = 2_000, n_features: int = 5, n_groups: int = 5, noise: float = 0.1,
                       sparsity: float = 0.0) -> Dict[str, np.ndarray]:
    """The seeded frame of the reference's demo notebook (notebooks/polars_ols_demo.ipynb cell 1, `_make_data`): same RNG call
    order (x, eps, group, sample_weights) -- note the MINUS sign of the target, unlike tests/test_ols.py:22-51."""
    rng = np.random.default_rng(0)
    x = rng.normal(size=(n_samples, n_features))
    eps = rng.normal(size=n_samples, scale=noise)
    out = {"x": x, "y": -1 * x[:, : int(n_features * (1.0 - sparsity))].sum(1) + eps}
    for i in range(n_features):
        out[f"x{i + 1}"] = np.ascontiguousarray(x[:, i])
    out["group"] = rng.integers(0, n_groups, size=n_samples)
    out["sample_weights"] = rng.uniform(0, 1, size=n_samples)
    return out


def insert_nulls(d: Dict[str, np.ndarray], columns: Sequence[str], frac: float = 0.1, seed: int = 7):
    """10 % missing values per listed column (reference tests/test_ols.py:42-50)."""
    rng = np.random.default_rng(seed)
    d = {k: np.array(v, copy=True) for k, v in d.items()}
    n = len(d["y"])
    for c in columns:
        m = rng.random(n) < frac
        if c == "y":
            d["y"][m] = np.nan
        else:
            j = int(c[1:]) - 1
            d["x"][m, j] = np.nan
            d[c][m] = np.nan
    return d


def sort_by_group(group: np.ndarray):
    """Stable sort-by-key: (order, offsets, keys) -- what Polars' .over() gather does on the host."""
    order = np.argsort(group, kind="stable")
    keys, counts = np.unique(group, return_counts=True)
    offsets = np.zeros(len(keys) + 1, dtype=np.int64)
    np.cumsum(counts, out=offsets[1:])
    return order, offsets, keys


def synthetic_groups(n_groups: int, n_rows: int, n_features: int, seed: int = 0, dtype=np.float64,
                     noise: float = 0.1, with_weights: bool = False):
    """Bench-shaped frame: contiguous equal-size groups, x ~ N(0,1), beta = 1, y = x.beta + N(0, noise)."""
    rng = np.random.default_rng(seed)
    N = n_groups * n_rows
    cols = [rng.standard_normal(N).astype(dtype) for _ in range(n_features)]
    y = (np.sum(cols, axis=0) + noise * rng.standard_normal(N)).astype(dtype)
    offsets = np.arange(n_groups + 1, dtype=np.int64) * n_rows
    out = {"y": y, "cols": cols, "offsets": offsets}
    if with_weights:
        w = rng.uniform(0.0, 1.0, N)
        out["w"] = (w / w.mean()).astype(dtype)
    return out
