"""Counter-based synthetic frames: Philox-4x32-10 keyed by (seed; row, column, stream), the SAME integer arithmetic in
numpy on the host and in torch on the device, so a host can regenerate any slice of a device-resident frame bit for bit -- a parity test of a
full-size frame needs no device-to-host copy of its inputs.  Test / bench infrastructure: nothing in polars_ols_amd/ imports this.

The variates use no transcendental function (those differ by an ulp between libraries): a "normal" value is the centred, scaled sum of the
eight 16-bit halves of one Philox block (Irwin-Hall, n = 8: mean 0, variance 1, support +-4.9 -- indistinguishable from N(0, 1) for a
regression frame), a uniform one is (word + 0.5) * 2^-32.  Every step is an exact integer operation or ONE correctly rounded IEEE
multiplication (not a division: torch turns a tensor / scalar into a multiplication by the reciprocal on the GPU, numpy divides).

Frame recipe (mirrors the reference's tests/test_ols.py:22-51): x_j ~ N(0, 1) for every column j, beta = 1, y = sum_j x_j + 0.1 N(0, 1),
weights w ~ U(0, 1) / mean; column j uses counter word 2 = j (the target's noise: j = 0xFFFF0000, the weights: 0xFFFF0001).
"""
from __future__ import annotations

import numpy as np

_M0, _M1, _W0, _W1, _MASK = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85, 0xFFFFFFFF
COL_NOISE, COL_WEIGHT = 0xFFFF0000, 0xFFFF0001


def _mulhilo(c: int, b):
    """(hi, lo) 32-bit halves of c * b for a constant c < 2^32 and an int64 array b of values < 2^32 -- through 16-bit limbs, so that no
    intermediate exceeds 2^34 (torch has no uint64; the same code runs on numpy arrays)."""
    ch, cl = c >> 16, c & 0xFFFF
    bh, bl = b >> 16, b & 0xFFFF
    mid = bh * cl + bl * ch
    lo64 = bl * cl + ((mid & 0xFFFF) << 16)
    return bh * ch + (mid >> 16) + (lo64 >> 32), lo64 & _MASK


def philox4x32(c0, c1, c2, c3, k0: int, k1: int):
    """Philox-4x32-10 (Salmon et al., SC'11; Random123): counter words c0..c3 (int64 arrays of values < 2^32, numpy or torch), key (k0, k1)."""
    for _ in range(10):
        h0, l0 = _mulhilo(_M0, c0)
        h1, l1 = _mulhilo(_M1, c2)
        c0, c1, c2, c3 = h1 ^ c1 ^ k0, l1, h0 ^ c3 ^ k1, l0
        k0, k1 = (k0 + _W0) & _MASK, (k1 + _W1) & _MASK
    return c0, c1, c2, c3


def _block(seed: int, col: int, row_lo: int, row_hi: int, device):
    if device is None:
        rows = np.arange(row_lo, row_hi, dtype=np.int64)
        zeros = np.zeros_like(rows)
    else:
        import torch

        rows = torch.arange(row_lo, row_hi, dtype=torch.int64, device=device)
        zeros = torch.zeros_like(rows)
    return philox4x32(rows & _MASK, rows >> 32, zeros + (col & _MASK), zeros, seed & _MASK, (seed >> 32) & _MASK)


_IH_MEAN, _IH_INV_STD = 8 * 32767.5, 1.0 / float(np.sqrt(8 * (65536.0 ** 2 - 1.0) / 12.0))
_TWO_M32 = 2.0 ** -32


def normal(seed: int, col: int, row_lo: int, row_hi: int, dtype=np.float64, device=None):
    """Rows [row_lo, row_hi) of column `col` of frame `seed`: ~N(0, 1), bit-identical on the host (device=None: numpy) and on a torch device."""
    w = _block(seed, col, row_lo, row_hi, device)
    s = (w[0] & 0xFFFF) + (w[0] >> 16) + (w[1] & 0xFFFF) + (w[1] >> 16) + (w[2] & 0xFFFF) + (w[2] >> 16) + (w[3] & 0xFFFF) + (w[3] >> 16)
    if device is None:
        return ((s.astype(np.float64) - _IH_MEAN) * _IH_INV_STD).astype(dtype)
    import torch

    return ((s.to(torch.float64) - _IH_MEAN) * _IH_INV_STD).to(dtype)


def uniform(seed: int, col: int, row_lo: int, row_hi: int, dtype=np.float64, device=None):
    w0 = _block(seed, col, row_lo, row_hi, device)[0]
    if device is None:
        return ((w0.astype(np.float64) + 0.5) * _TWO_M32).astype(dtype)
    import torch

    return ((w0.to(torch.float64) + 0.5) * _TWO_M32).to(dtype)


def frame_columns(seed: int, n_features: int, row_lo: int, row_hi: int, dtype=np.float64, device=None, weights: bool = False):
    """(y, [x_0 .. x_{k-1}], w or None) for rows [row_lo, row_hi) of frame `seed`.  The target is summed in f64 in column order and rounded
    once, on both sides; the weights are NOT normalised by their mean here (a whole-frame quantity: callers divide by it themselves)."""
    cols64 = [normal(seed, j, row_lo, row_hi, np.float64 if device is None else _t64(), device) for j in range(n_features)]
    y = normal(seed, COL_NOISE, row_lo, row_hi, np.float64 if device is None else _t64(), device) * 0.1
    for c in cols64:
        y = y + c
    cast = (lambda a: a.astype(dtype)) if device is None else (lambda a: a.to(dtype))
    w = cast(uniform(seed, COL_WEIGHT, row_lo, row_hi, np.float64 if device is None else _t64(), device)) if weights else None
    return cast(y), [cast(c) for c in cols64], w


def _t64():
    import torch

    return torch.float64
