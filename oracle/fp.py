"""Floating-point helpers for the oracle (TEST INFRASTRUCTURE, see package docstring).

fp32 semantics with three conventions shared with the HIP kernels:

* ``fma32(a, b, c)``   -- correctly rounded fp32 ``a*b + c`` (one rounding), the
  instruction XLA:CPU emits for ``x + step_size*coef*grad``
  (blackjax/mcmc/integrators.py:200,236) on an FMA machine and that
  ``v_fma_f32`` executes on CDNA4.
* ``dot64(a, b)``      -- sum of fp32 products accumulated in fp64, rounded
  once to fp32.  The reference's ``jnp.dot`` (blackjax/mcmc/metrics.py:269)
  has an implementation-defined fp32 summation order; the fp64 accumulation is
  order independent to ~1e-16 so CPU and GPU agree bit-for-bit after the final
  rounding (up to a ~1e-9 probability per reduction).
* ``*_cr`` scalar functions -- evaluate in fp64, round once to fp32
  ("correctly rounded" fp32 functions for all practical purposes).
"""
from __future__ import annotations

import numpy as np

f32 = np.float32
f64 = np.float64

_TIE_LOW = np.uint64(1 << 28)
_LOW_MASK = np.uint64((1 << 29) - 1)


def fma32(a, b, c):
    """Correctly rounded fp32 fused multiply-add, vectorised.

    a*b is exact in fp64 (24+24 <= 53 bits).  s = a*b + c is rounded to 53
    bits; rounding that again to 24 bits can double-round only when the fp64
    result sits exactly on an fp32 rounding midpoint while the true sum does
    not.  TwoSum recovers the fp64 rounding error and nudges such midpoints in
    its direction before the final fp32 rounding.
    """
    a, b, c = np.broadcast_arrays(
        np.asarray(a, dtype=f32), np.asarray(b, dtype=f32), np.asarray(c, dtype=f32)
    )
    shape = a.shape
    a = a.astype(f64).ravel()
    b = b.astype(f64).ravel()
    c = c.astype(f64).ravel()
    with np.errstate(invalid="ignore", over="ignore"):
        prod = a * b
        t = prod + c
        bb = t - prod
        err = (prod - (t - bb)) + (c - bb)  # TwoSum error term (exact)
        tie = ((t.view(np.uint64) & _LOW_MASK) == _TIE_LOW) & np.isfinite(t)
        if tie.any():
            up = tie & (err > 0)
            dn = tie & (err < 0)
            t[up] = np.nextafter(t[up], np.inf)
            t[dn] = np.nextafter(t[dn], -np.inf)
        return t.astype(f32).reshape(shape)


def dot64(a, b, axis=-1):
    """fp32 x fp32 products (exact in fp64) summed in fp64, rounded once to fp32."""
    a = np.asarray(a, dtype=f32).astype(f64)
    b = np.asarray(b, dtype=f32).astype(f64)
    return np.sum(a * b, axis=axis).astype(f32)


def exp_cr(x):
    with np.errstate(over="ignore", under="ignore"):
        return np.exp(np.asarray(x, dtype=f32).astype(f64)).astype(f32)


def log_cr(x):
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.log(np.asarray(x, dtype=f32).astype(f64)).astype(f32)


def log1p_cr(x):
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.log1p(np.asarray(x, dtype=f32).astype(f64)).astype(f32)


def sqrt32(x):
    with np.errstate(invalid="ignore"):
        return np.sqrt(np.asarray(x, dtype=f32))  # IEEE correctly rounded


def logaddexp_cr(a, b):
    """jnp.logaddexp in fp32: evaluated in fp64, rounded once (-inf safe)."""
    a = np.asarray(a, dtype=f32).astype(f64)
    b = np.asarray(b, dtype=f32).astype(f64)
    return np.logaddexp(a, b).astype(f32)


def expit_cr(x):
    """jax.scipy.special.expit = 1/(1+exp(-x)) in fp64, rounded once."""
    x = np.asarray(x, dtype=f32).astype(f64)
    with np.errstate(over="ignore"):
        return (1.0 / (1.0 + np.exp(-x))).astype(f32)


def mfma_k_order(K: int, tile: int = 16):
    """Summation order of the engine's fp32 MFMA GEMMs (blackjax_amd/csrc/bjx_dense.hip): K is
    walked in tiles of 16; inside a tile the MFMA step u = 0..7 accumulates k = u and then
    k = 8 + u (v_mfma_f32_32x32x2_f32: lanes 0-31 carry the first k of the pair, lanes 32-63 the
    second).  Indices >= K (zero padding of the last tile) are dropped: fma(0, b, acc) == acc."""
    order = []
    for t0 in range(0, K, tile):
        for u in range(tile // 2):
            for k in (t0 + u, t0 + tile // 2 + u):
                if k < K:
                    order.append(k)
    return np.asarray(order, dtype=np.int32)


def gemm_f32chain(a, b_kn, k_order=None):
    """NumPy statement of an fp32 dot evaluated as ONE fmaf chain per output element in a given k
    order: C[m][n] = fma(a[m][k_last], b[k_last][n], ... fma(a[m][k_0], b[k_0][n], +0)).  The
    reference's dense products are ``jnp.dot(..., precision="highest")`` in fp32
    (blackjax/util.py:23-61) whose summation order is implementation-defined; this is that
    arithmetic for the order the engine's MFMA kernels use.  Slow (one vectorised fma32 per k):
    the C port (oracle/cport.py::gemm_f32chain) is the fast twin, checked against this one."""
    a = np.asarray(a, dtype=f32)
    b_kn = np.asarray(b_kn, dtype=f32)
    K = a.shape[1]
    if k_order is None:
        k_order = mfma_k_order(K)
    acc = np.zeros((a.shape[0], b_kn.shape[1]), f32)
    for k in k_order:
        acc = fma32(a[:, k:k + 1], b_kn[k:k + 1, :], acc)
    return acc
