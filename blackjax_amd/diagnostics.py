"""``effective_sample_size`` on device (blackjax/diagnostics.py:157-304).

FFT autocovariance via ``torch.fft`` (rocFFT on MI355X); the two sequential ``lax.scan`` passes of
the reference (Geyer's initial positive / initial monotone sequences) are closed-form prefix
operations here (``cumprod`` / ``cummin``), so the whole diagnostic is a handful of device ops with
no host synchronisation.  This is an offline diagnostic, not part of the sampling hot path.
"""
from __future__ import annotations

import math

import torch

from ._diag_rank import ess_bulk, ess_tail, rhat  # noqa: E402,F401
from ._diag_rhat import potential_scale_reduction  # noqa: E402,F401

# NOTE: as in the reference, ``rhat`` is the rank-normalised split-R-hat (diagnostics.py:92-155), not
# the classic ``potential_scale_reduction`` (39-89).
__all__ = ["effective_sample_size", "potential_scale_reduction", "rhat", "ess_bulk", "ess_tail"]


def _next_fast_len(n: int) -> int:
    """scipy.fft.next_fast_len(n) (complex transform: smallest 11-smooth integer >= n)."""
    try:
        from scipy.fft import next_fast_len

        return int(next_fast_len(n))
    except Exception:  # pragma: no cover - scipy is present in the supported images
        while True:
            m = n
            for p in (2, 3, 5, 7, 11):
                while m % p == 0:
                    m //= p
            if m == 1:
                return n
            n += 1


def effective_sample_size(input_array: torch.Tensor, chain_axis: int = 0,
                          sample_axis: int = 1) -> torch.Tensor:
    """ESS with chain and sample axes squeezed; zero where the within-chain variance is
    numerically zero (diagnostics.py:157-304)."""
    x = torch.movedim(input_array, (chain_axis, sample_axis), (0, 1))
    if x.dtype not in (torch.float32, torch.float64):
        x = x.double()
    M, T = x.shape[:2]
    assert T > 1, f"The input array must have at least 2 samples, got only {T}."
    event = x.shape[2:]
    x = x.reshape(M, T, -1)

    has_var = (x != x[:, :1]).any(dim=0).any(dim=0)
    mean_chain = x.mean(dim=1, keepdim=True)
    centered = x - mean_chain
    m = _next_fast_len(2 * T)
    f = torch.fft.rfft(centered, n=m, dim=1)
    f = f * torch.conj(f)
    autocov = torch.fft.irfft(f, n=m, dim=1)[:, :T] / T
    mean_autocov = autocov.mean(dim=0)  # (T, E)
    mean_var0 = mean_autocov[0] * T / (T - 1.0)
    degenerate = torch.isfinite(mean_var0) & (~has_var | (mean_var0 <= 0.0))
    weighted_var = mean_var0 * (T - 1.0) / T
    if M > 1:
        weighted_var = weighted_var + mean_chain[:, 0].var(dim=0, unbiased=True)
    weighted_var = torch.where(degenerate, torch.ones_like(weighted_var), weighted_var)

    T_even = T - T % 2
    rho = torch.cat([torch.ones_like(mean_var0)[None],
                     1.0 - (mean_var0 - mean_autocov[1:T_even]) / weighted_var], dim=0)
    rho_even, rho_odd = rho[0::2], rho[1::2]
    K, E = rho_even.shape

    # Geyer's initial positive sequence: mask = running AND of (rho_even + rho_odd > 0)
    mask0 = (rho_even + rho_odd) > 0.0
    mask = torch.cumprod(mask0.to(torch.int32), dim=0).bool()
    n_true = mask.sum(dim=0)
    max_t = torch.clamp(n_true - 1, min=0)
    idx = max_t + 1
    idx_read = torch.clamp(idx, max=K - 1)  # JAX gather clamps an out-of-range index
    cols = torch.arange(E, device=x.device)
    rho_odd = torch.where(mask, rho_odd, torch.zeros_like(rho_odd))
    lag = torch.arange(K, device=x.device)[:, None]
    at_idx = lag == idx[None, :]  # empty column when idx == K: JAX scatter drops that update
    mask_even = torch.where(at_idx, (rho_even[idx_read, cols] > 0)[None, :], mask)
    rho_even = torch.where(mask_even, rho_even, torch.zeros_like(rho_even))

    # Geyer's initial monotone sequence: running minimum of the pair sums
    rho_sum = rho_even + rho_odd
    run_min = torch.cummin(rho_sum, dim=0).values
    prev_min = torch.cat([rho_sum[:1], run_min[:-1]], dim=0)
    upd_mask = rho_sum > prev_min
    rho_even_f = torch.where(upd_mask, run_min / 2.0, rho_even)
    rho_odd_f = torch.where(upd_mask, run_min / 2.0, rho_odd)

    ess_raw = M * T
    tau = -1.0 + 2.0 * (rho_even_f + rho_odd_f).sum(dim=0) - rho_even_f[idx_read, cols]
    tau = torch.clamp(tau, min=1.0 / math.log10(ess_raw))
    ess = ess_raw / tau
    ess = torch.where(degenerate, torch.zeros_like(ess), ess)
    return ess.reshape(event)
