"""Rank-normalised split-chain diagnostics on device (blackjax/diagnostics.py:92-155, 328-522;
Vehtari et al. 2021): ``rhat``, ``ess_bulk``, ``ess_tail``.

Offline diagnostics over retained draws ``(chains, draws, ...)``: device sorts (rocPRIM through
``torch.sort``), ``torch.special.ndtri`` and the FFT-based ``effective_sample_size`` -- no host
synchronisation.  Not part of the sampling hot path.
"""
from __future__ import annotations

import torch


def _to_standard_axes(x: torch.Tensor, chain_axis: int, sample_axis: int) -> torch.Tensor:
    """diagnostics.py:328-338."""
    if x.dtype not in (torch.float32, torch.float64):
        x = x.double()
    nd = x.ndim
    return torch.movedim(x, (chain_axis % nd, sample_axis % nd), (0, 1))


def _split_chains(x: torch.Tensor) -> torch.Tensor:
    """diagnostics.py:341-360: ``(M, T, ...) -> (2M, T // 2, ...)``; an odd last draw is dropped."""
    half = x.shape[1] // 2
    return torch.cat([x[:, :half], x[:, half:2 * half]], dim=0)


def _rank_normalize(x: torch.Tensor) -> torch.Tensor:
    """diagnostics.py:363-401: ``z = ndtri((rank - 3/8) / (n + 1/4))`` over the pooled draws.  The
    reference's double ``argsort`` is one stable sort plus the inverse permutation (a scatter)."""
    M, T = x.shape[:2]
    n = M * T
    flat = x.reshape(n, -1)
    order = torch.argsort(flat, dim=0, stable=True)
    ranks = torch.empty_like(order)
    src = torch.arange(n, device=x.device).unsqueeze(1).expand_as(order)
    ranks.scatter_(0, order, src)
    r = (ranks + 1).to(x.dtype)
    z = torch.special.ndtri((r - 3.0 / 8) / (n + 1.0 / 4))
    return z.reshape(x.shape)


def _quantile0(flat: torch.Tensor, q: float) -> torch.Tensor:
    """``jnp.quantile(flat, q, axis=0)`` (linear interpolation) without torch.quantile's size limit."""
    n = flat.shape[0]
    s = torch.sort(flat, dim=0).values
    pos = q * (n - 1)
    lo = int(pos // 1)
    hi = min(lo + 1, n - 1)
    w_hi = pos - lo
    return s[lo] * (1.0 - w_hi) + s[hi] * w_hi


def rhat(input_array: torch.Tensor, chain_axis: int = 0, sample_axis: int = 1) -> torch.Tensor:
    """Rank-normalised split-R-hat (diagnostics.py:92-155): the maximum of the split-R-hat of the
    rank-normalised draws (bulk) and of the rank-normalised draws folded about the pooled median."""
    from ._diag_rhat import potential_scale_reduction

    xs = _split_chains(_to_standard_axes(input_array, chain_axis, sample_axis))
    bulk = potential_scale_reduction(_rank_normalize(xs))
    flat = xs.reshape(xs.shape[0] * xs.shape[1], *xs.shape[2:])
    folded = (xs - _quantile0(flat, 0.5)).abs()
    tail = potential_scale_reduction(_rank_normalize(folded))
    return torch.maximum(bulk, tail)


def ess_bulk(input_array: torch.Tensor, chain_axis: int = 0, sample_axis: int = 1) -> torch.Tensor:
    """Bulk effective sample size (diagnostics.py:404-443)."""
    from .diagnostics import effective_sample_size

    xs = _split_chains(_to_standard_axes(input_array, chain_axis, sample_axis))
    return effective_sample_size(_rank_normalize(xs))


def ess_tail(input_array: torch.Tensor, chain_axis: int = 0, sample_axis: int = 1,
             prob: float = 0.90) -> torch.Tensor:
    """Tail effective sample size (diagnostics.py:446-522): min of the ESS of the lower- and
    upper-tail indicators at the pooled ``(1 -+ prob) / 2`` quantiles of the split chains."""
    from .diagnostics import effective_sample_size

    xs = _split_chains(_to_standard_axes(input_array, chain_axis, sample_axis))
    flat = xs.reshape(xs.shape[0] * xs.shape[1], *xs.shape[2:])
    q_lo = _quantile0(flat, (1.0 - prob) / 2.0)
    q_hi = _quantile0(flat, (1.0 + prob) / 2.0)
    lower = effective_sample_size((xs <= q_lo[None, None]).to(xs.dtype))
    upper = effective_sample_size((xs >= q_hi[None, None]).to(xs.dtype))
    return torch.minimum(lower, upper)
