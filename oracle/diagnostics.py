"""Oracle for ``effective_sample_size`` (TEST INFRASTRUCTURE, see package docstring).

Follows blackjax/diagnostics.py:157-304 statement by statement (NumPy, explicit loops for the two
``lax.scan`` passes).  JAX gather semantics are restated explicitly: an out-of-range read index is
clamped, an out-of-range ``.at[].set`` is dropped.
"""
from __future__ import annotations

import numpy as np
from scipy.fft import next_fast_len


def effective_sample_size(input_array, chain_axis: int = 0, sample_axis: int = 1):
    x = np.asarray(input_array)
    x = np.moveaxis(x, (chain_axis, sample_axis), (0, 1))  # (M, T, ...)
    dtype = x.dtype if x.dtype in (np.float32, np.float64) else np.float64
    x = x.astype(dtype)
    M, T = x.shape[:2]
    assert T > 1, f"The input array must have at least 2 samples, got only {T}."
    event = x.shape[2:]
    x = x.reshape(M, T, -1)

    has_var = np.any(x != x[:, :1], axis=(0, 1))  # diagnostics.py:204-209
    mean_chain = x.mean(axis=1, keepdims=True)
    centered = x - mean_chain
    m = next_fast_len(2 * T)
    f = np.fft.rfft(centered, n=m, axis=1)
    f = f * np.conjugate(f)
    autocov = np.fft.irfft(f, n=m, axis=1)[:, :T].astype(dtype) / dtype.type(T)
    mean_autocov = autocov.mean(0)  # (T, E)
    mean_var0 = mean_autocov[0] * T / (T - 1.0)
    degenerate = np.isfinite(mean_var0) & (~has_var | (mean_var0 <= 0.0))
    weighted_var = mean_var0 * (T - 1.0) / T
    if M > 1:
        weighted_var = weighted_var + mean_chain[:, 0].var(axis=0, ddof=1)
    weighted_var = np.where(degenerate, 1.0, weighted_var)

    T_even = T - T % 2
    rho = np.concatenate([np.ones_like(mean_var0)[None],
                          1.0 - (mean_var0 - mean_autocov[1:T_even]) / weighted_var], axis=0)
    rho_even, rho_odd = rho[0::2].copy(), rho[1::2].copy()
    K, E = rho_even.shape

    mask0 = (rho_even + rho_odd) > 0.0
    carry = np.ones(E, bool)
    max_t = np.zeros(E, int)
    mask = np.zeros_like(mask0)
    for t in range(K):  # positive_sequence_body_fn
        carry = carry & mask0[t]
        max_t = np.where(carry, t, max_t)
        mask[t] = carry
    idx = max_t + 1
    idx_read = np.minimum(idx, K - 1)  # gather clamps
    cols = np.arange(E)
    rho_odd = np.where(mask, rho_odd, 0.0)
    mask_even = mask.copy()
    ok = idx < K  # scatter drops out-of-range updates
    mask_even[idx[ok], cols[ok]] = rho_even[idx_read[ok], cols[ok]] > 0
    rho_even = np.where(mask_even, rho_even, 0.0)

    rho_sum = rho_even + rho_odd
    upd_mask = np.zeros_like(mask0)
    upd_val = np.zeros_like(rho_sum)
    prev = rho_sum[0]
    for t in range(K):  # monotone_sequence_body_fn
        um = rho_sum[t] > prev
        prev = np.where(um, prev, rho_sum[t])
        upd_mask[t], upd_val[t] = um, prev
    rho_even_f = np.where(upd_mask, upd_val / 2.0, rho_even)
    rho_odd_f = np.where(upd_mask, upd_val / 2.0, rho_odd)

    ess_raw = M * T
    tau = -1.0 + 2.0 * np.sum(rho_even_f + rho_odd_f, axis=0) - rho_even_f[idx_read, cols]
    tau = np.maximum(tau, 1 / np.log10(ess_raw))
    ess = ess_raw / tau
    ess = np.where(degenerate, 0.0, ess)
    return ess.reshape(event).astype(dtype)


# ----------------------------------------------------------------------------- rank-normalised family
def potential_scale_reduction(input_array, chain_axis: int = 0, sample_axis: int = 1):
    """blackjax/diagnostics.py:39-89."""
    x = np.moveaxis(np.asarray(input_array), (chain_axis, sample_axis), (0, 1))
    assert x.shape[0] > 1
    n = x.shape[1]
    between = n * x.mean(1).var(0, ddof=1)
    within = x.var(1, ddof=1).mean(0)
    return np.sqrt((between / within + n - 1) / n)


def split_chains(x):
    """diagnostics.py:341-360: (M, T, ...) -> (2M, T//2, ...), an odd last draw is dropped."""
    half = x.shape[1] // 2
    return np.concatenate([x[:, :half], x[:, half:2 * half]], axis=0)


def rank_normalize(x):
    """diagnostics.py:363-401: Blom plotting position of the pooled ranks (stable double argsort)."""
    from scipy.special import ndtri

    M, T = x.shape[:2]
    n = M * T
    flat = x.reshape(n, *x.shape[2:])
    ranks = np.argsort(np.argsort(flat, axis=0, kind="stable"), axis=0, kind="stable").astype(x.dtype) + 1
    z = ndtri((ranks - x.dtype.type(3.0 / 8)) / x.dtype.type(n + 1.0 / 4)).astype(x.dtype)
    return z.reshape(x.shape)


def _std_axes(input_array, chain_axis, sample_axis):
    x = np.asarray(input_array)
    x = x.astype(x.dtype if x.dtype in (np.float32, np.float64) else np.float64)
    return np.moveaxis(x, (chain_axis % x.ndim, sample_axis % x.ndim), (0, 1))


def rhat(input_array, chain_axis: int = 0, sample_axis: int = 1):
    """diagnostics.py:92-155: max of the split-R-hat of the rank-normalised draws and of the
    rank-normalised draws folded about the pooled median."""
    xs = split_chains(_std_axes(input_array, chain_axis, sample_axis))
    bulk = potential_scale_reduction(rank_normalize(xs))
    flat = xs.reshape(xs.shape[0] * xs.shape[1], *xs.shape[2:])
    folded = np.abs(xs - np.median(flat, axis=0)).astype(xs.dtype)
    tail = potential_scale_reduction(rank_normalize(folded))
    return np.maximum(bulk, tail)


def ess_bulk(input_array, chain_axis: int = 0, sample_axis: int = 1):
    """diagnostics.py:404-443."""
    xs = split_chains(_std_axes(input_array, chain_axis, sample_axis))
    return effective_sample_size(rank_normalize(xs))


def ess_tail(input_array, chain_axis: int = 0, sample_axis: int = 1, prob: float = 0.90):
    """diagnostics.py:446-522."""
    xs = split_chains(_std_axes(input_array, chain_axis, sample_axis))
    flat = xs.reshape(xs.shape[0] * xs.shape[1], *xs.shape[2:])
    q_lo = np.quantile(flat, (1.0 - prob) / 2.0, axis=0).astype(xs.dtype)
    q_hi = np.quantile(flat, (1.0 + prob) / 2.0, axis=0).astype(xs.dtype)
    lower = effective_sample_size((xs <= q_lo[None, None]).astype(xs.dtype))
    upper = effective_sample_size((xs >= q_hi[None, None]).astype(xs.dtype))
    return np.minimum(lower, upper)
