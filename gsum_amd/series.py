"""Order-by-order series arithmetic on the host: y_n <-> c_n and geometric sums.

O(n * orders) elementwise numpy that feeds the hot path; it stays on the host
exactly as in the reference (gsum/helpers.py:71-182).
"""
from __future__ import annotations

import numpy as np

__all__ = ["coefficients", "partials", "geometric_sum"]


def coefficients(y, ratio, ref=1, orders=None):
    """Extract dimensionless coefficients ``c_n`` from partial sums ``y_n``.

    ``c_0 = y_0/(ref ratio^o_0)``, ``c_n = (y_n - y_{n-1})/(ref ratio^o_n)``; same contract and
    errors as gsum/helpers.py:71-101.
    """
    y = np.asarray(y)
    if y.ndim != 2:
        raise ValueError("y must be 2d")
    if orders is None:
        orders = np.arange(y.shape[-1])
    if len(orders) != y.shape[-1]:
        raise ValueError("partials and orders must have the same length")
    ref, ratio, orders = np.atleast_1d(ref, ratio, orders)
    scale = ref[:, None] * ratio[:, None] ** orders
    c = np.empty(y.shape, dtype=np.result_type(y, float))
    c[..., 0] = y[..., 0]
    c[..., 1:] = np.diff(y, axis=-1)
    return c / scale


def partials(coeffs, ratio, ref=1, orders=None):
    """Partial sums ``y_k = ref * sum_{n<=k} c_n ratio^o_n`` (gsum/helpers.py:104-146)."""
    coeffs = np.asarray(coeffs)
    if orders is None:
        orders = np.arange(coeffs.shape[-1])
    ratio = np.atleast_1d(ratio)
    if ratio.ndim == 1:
        ratio = ratio[:, None]
    ref = np.atleast_1d(ref)
    if ref.ndim == 1:
        ref = ref[:, None]
    return np.cumsum(ref * coeffs * ratio ** orders, axis=-1)


def geometric_sum(x, start, end, excluded=None):
    """``sum_{i=start}^{end} x^i`` with the orders in ``excluded`` left out (gsum/helpers.py:149-182)."""
    if end < start:
        raise ValueError("end must be greater than or equal to start")
    s = (x ** start - x ** (end + 1)) / (1 - x)
    if excluded is not None:
        for n in np.atleast_1d(excluded):
            if start <= n <= end:
                s -= x ** n
    return s
