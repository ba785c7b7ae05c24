"""Synthetic log-densities (value and gradient) used by the parity tests and
bench (TEST INFRASTRUCTURE, see package docstring).

All are batched: ``fn(q: (N, D) float32) -> (logp: (N,) float32, grad: (N, D) float32)``.
Reductions are accumulated in fp64 and rounded once to fp32 (oracle/fp.py) so
the CPU and GPU evaluations agree bit-for-bit; the element-wise gradient uses
single correctly-rounded fp32 operations.

* ``diag_gaussian``   SURVEY.md section 8(d) C1/C2/C4: ``logp = -1/2 sum q_i^2 / sigma_i^2``
  (reference fixture: tests/fixtures.py:60-78 ``std_normal_logdensity``)
* ``neal_funnel``     tests/fixtures.py:81-98 ``neal_funnel_logdensity``
* ``ar1_gaussian``    SURVEY.md 8(d) C5: Sigma_ij = rho^|i-j| (tridiagonal precision)
"""
from __future__ import annotations

import numpy as np

from .fp import exp_cr, f32, f64


def diag_gaussian(inv_var):
    """inv_var: (D,) float32 = 1/sigma^2.  grad = -(q*inv_var), logp = 1/2 sum q*grad."""
    iv = np.asarray(inv_var, dtype=f32)

    def fn(q):
        q = np.asarray(q, dtype=f32)
        g = -(q * iv)  # one fp32 rounding
        logp = (0.5 * np.sum(q.astype(f64) * g.astype(f64), axis=-1)).astype(f32)
        return logp, g.astype(f32)

    return fn


def neal_funnel():
    """y = q[:,0] ~ N(0, 9); q[:,1:] ~ N(0, e^y).

    logp = -1/2 (y/3)^2 - 1/2 e^{-y} S - 1/2 (D-1) y,   S = sum v^2 (fp64 accumulate)
    d/dy = -y/9 + 1/2 e^{-y} S - 1/2 (D-1) ;  d/dv = -e^{-y} v
    Scalars are evaluated in fp64 from fp32 inputs and rounded once.
    """

    def fn(q):
        q = np.asarray(q, dtype=f32)
        D = q.shape[-1]
        y = q[:, 0].astype(f64)
        v = q[:, 1:]
        S = np.sum(v.astype(f64) ** 2, axis=-1)
        ey32 = exp_cr(-q[:, 0])  # fp32 e^{-y}, used element-wise
        ey = ey32.astype(f64)
        logp = (-0.5 * (y / 3.0) ** 2 - 0.5 * ey * S - 0.5 * (D - 1) * y).astype(f32)
        g = np.empty_like(q)
        g[:, 0] = (-y / 9.0 + 0.5 * ey * S - 0.5 * (D - 1)).astype(f32)
        g[:, 1:] = -(ey32[:, None] * v)
        return logp, g

    return fn


def ar1_gaussian(rho: float, D: int):
    """Zero-mean Gaussian with Sigma_ij = rho^|i-j|.  Precision is tridiagonal:
    P = 1/(1-rho^2) * tridiag(-rho, [1, 1+rho^2, ..., 1+rho^2, 1], -rho).
    grad = -(P q) evaluated as fma chain; logp = 1/2 sum q*grad (fp64 accumulate)."""
    from .fp import fma32

    c = f32(1.0 / (1.0 - rho * rho))
    diag = np.full(D, 1.0 + rho * rho, dtype=f32)
    diag[0] = diag[-1] = 1.0
    diag = (diag * c).astype(f32)
    off = f32(-rho) * c

    def fn(q):
        q = np.asarray(q, dtype=f32)
        left = np.zeros_like(q)
        right = np.zeros_like(q)
        left[:, 1:] = q[:, :-1]
        right[:, :-1] = q[:, 1:]
        t = diag * q
        t = fma32(off, left, t)
        t = fma32(off, right, t)
        g = -t
        logp = (0.5 * np.sum(q.astype(f64) * g.astype(f64), axis=-1)).astype(f32)
        return logp, g.astype(f32)

    return fn


def ar1_covariance(rho: float, D: int) -> np.ndarray:
    i = np.arange(D)
    return (rho ** np.abs(i[:, None] - i[None, :])).astype(f32)
