"""Host-side tables of the device scale-space build (`mad_space_build`, include/mad_amd.h).

The filter weights and the spline operator are tiny (tens to a few thousand numbers) and must be the very
numbers scipy uses, so they are formed here with numpy exactly as scipy forms them and handed to the C-ABI:

* `gaussian_kernel1d` restates `scipy.ndimage._filters._gaussian_kernel1d` (the weights behind
  `gaussian_filter` / `gaussian_laplace`, MapSpace.py:144,171,182) -- `tests/test_host.py` holds it to scipy's;
* `spline_tables` describes `scipy.interpolate.interp1d(kind="cubic")` on the sites 0..n-1 evaluated at
  0, 0.5, ..., n-1 (MapSpace.py:137-146,206-214): not-a-knot cubic B-spline interpolation.  It returns the
  banded LU factors of the collocation matrix (bandwidth 2, no pivoting needed) and, per output site, the four
  non-zero basis values and the index of the first coefficient they multiply.
"""
import functools

import numpy as np


def kernel_radius(sigma, truncate=4.0):
    return int(truncate * float(sigma) + 0.5)


def gaussian_kernel1d(sigma, order, radius):
    sigma2 = sigma * sigma
    x = np.arange(-radius, radius + 1)
    phi_x = np.exp(-0.5 / sigma2 * x ** 2)
    phi_x = phi_x / phi_x.sum()
    if order == 0:
        return phi_x
    exponent_range = np.arange(order + 1)
    q = np.zeros(order + 1)
    q[0] = 1
    D = np.diag(exponent_range[1:], 1)
    P = np.diag(np.ones(order) / -sigma2, -1)
    Q_deriv = D + P
    for _ in range(order):
        q = Q_deriv.dot(q)
    q = (x[:, None] ** exponent_range).dot(q)
    return q * phi_x


def _bspline_basis_all(t, k, x):
    """Values of the k+1 B-splines that are non-zero at x (Cox - de Boor), and the index of the first."""
    n = len(t) - k - 1
    # knot span: t[mu] <= x < t[mu+1], clamped so that the last site belongs to the last span
    mu = int(np.searchsorted(t, x, side="right") - 1)
    mu = min(max(mu, k), n - 1)
    b = np.zeros(k + 1)
    b[0] = 1.0
    for j in range(1, k + 1):
        saved = 0.0
        for r in range(j):
            left, right = t[mu + 1 + r - j], t[mu + 1 + r]
            term = b[r] / (right - left)
            b[r] = saved + (right - x) * term
            saved = (x - left) * term
        b[j] = saved
    return b, mu - k


@functools.lru_cache(maxsize=16)
def spline_tables(n):
    """-> (lu float64[5, n], ev_w float64[2n-1, 4], ev_i int32[2n-1]) for a line of n samples (n >= 4)."""
    if n < 4:
        raise ValueError("a cubic not-a-knot spline needs at least 4 samples")
    k = 3
    x = np.arange(n, dtype=np.float64)
    t = np.r_[(x[0],) * (k + 1), x[2:-2], (x[-1],) * (k + 1)]      # not-a-knot: sites 1 and n-2 are not knots
    A = np.zeros((n, n))
    for i in range(n):
        b, first = _bspline_basis_all(t, k, x[i])
        A[i, first:first + k + 1] = b
    # banded LU, bandwidth 2 on both sides, no pivoting (the matrix is well conditioned, cond ~ 3)
    U = A.copy()
    L = np.zeros((n, 2))        # L[i, 0] multiplies row i-2, L[i, 1] multiplies row i-1
    for c in range(n - 1):
        for r in range(c + 1, min(c + 3, n)):
            if U[r, c] != 0.0:
                m = U[r, c] / U[c, c]
                L[r, 1 - (r - c - 1)] = m
                U[r, c:c + 4] -= m * U[c, c:c + 4]
                U[r, c] = 0.0
    if np.abs(np.triu(U, 3)).max() > 0 or np.abs(np.tril(U, -1)).max() > 0:
        raise AssertionError("collocation matrix is not banded as expected")
    lu = np.zeros((5, n))
    lu[0], lu[1] = L[:, 0], L[:, 1]
    lu[2] = np.diag(U)
    lu[3, :-1] = np.diag(U, 1)
    lu[4, :-2] = np.diag(U, 2)
    xi = np.arange(0, n - 0.5, 0.5)
    ev_w = np.zeros((len(xi), 4))
    ev_i = np.zeros(len(xi), np.int32)
    for m, xv in enumerate(xi):
        b, first = _bspline_basis_all(t, k, xv)
        ev_w[m], ev_i[m] = b, first
    return np.ascontiguousarray(lu), np.ascontiguousarray(ev_w), ev_i


def spline_apply(y, axis=0):
    """numpy evaluation of the same operator (used by the CPU tests to hold the tables to scipy)."""
    y = np.moveaxis(np.asarray(y, dtype=np.float64), axis, 0)
    n = y.shape[0]
    lu, ev_w, ev_i = spline_tables(n)
    z = np.zeros_like(y)
    for i in range(n):
        z[i] = y[i]
        if i >= 1:
            z[i] = z[i] - lu[1, i] * z[i - 1]
        if i >= 2:
            z[i] = z[i] - lu[0, i] * z[i - 2]
    c = np.zeros_like(y)
    for i in range(n - 1, -1, -1):
        s = z[i]
        if i + 1 < n:
            s = s - lu[3, i] * c[i + 1]
        if i + 2 < n:
            s = s - lu[4, i] * c[i + 2]
        c[i] = s / lu[2, i]
    out = np.zeros((2 * n - 1,) + y.shape[1:])
    for m in range(2 * n - 1):
        i0 = ev_i[m]
        out[m] = ((ev_w[m, 0] * c[i0] + ev_w[m, 1] * c[i0 + 1]) + ev_w[m, 2] * c[i0 + 2]) + ev_w[m, 3] * c[i0 + 3]
    return np.moveaxis(out, 0, axis)
