"""Host-side metric preparation (blackjax/mcmc/metrics.py:180-218, 701-729).

The Gaussian-Euclidean metric itself (momentum draw, kinetic energy, U-turn dot
products) is evaluated inside the HIP kernels; this module only classifies the
``inverse_mass_matrix`` argument and, for a dense matrix, factorises it once.
"""
from __future__ import annotations

from typing import NamedTuple, Optional

import torch


class Metric(NamedTuple):
    kind: str  # "diag" | "dense"
    imm: torch.Tensor  # (D,), (N, D) [diag]  or (D, D) [dense]
    imm_stride: int  # diag: 0 shared, D per chain
    mass_sqrt_t: Optional[torch.Tensor]  # dense: (L^{-T})^T = L^{-1}, row-major (D, D)
    imm_t: Optional[torch.Tensor] = None  # shared dense: the transpose of imm, row-major (bjx_dense_apply_imm_t)


_DENSE_CACHE: dict = {}


def default_metric(inverse_mass_matrix, n_chains: int, dim: int, device) -> Metric:
    """Classify ``inverse_mass_matrix`` like metrics.py:180-218 / _format_covariance 701-729.

    * 1-d ``(D,)``: diagonal, shared by all chains.
    * 2-d ``(N, D)`` with ``N == n_chains`` and ``N != D``: per-chain diagonal (what a
      vmapped ``window_adaptation(...).run`` produces in the reference).  For the
      ambiguous square case ``N == D`` a 2-d array is a dense matrix, exactly as in the
      reference; pass ``PerChainDiag(imm)`` to force the per-chain reading.
    * 2-d ``(D, D)``: dense, shared by all chains.
    """
    per_chain = False
    if isinstance(inverse_mass_matrix, PerChainDiag):
        inverse_mass_matrix = inverse_mass_matrix.imm
        per_chain = True
    if isinstance(inverse_mass_matrix, PerChainDiagTensor):  # what window_adaptation returns
        inverse_mass_matrix = inverse_mass_matrix.as_subclass(torch.Tensor)
        per_chain = True
    imm = torch.as_tensor(inverse_mass_matrix, dtype=torch.float32, device=device)
    if imm.ndim == 1:
        if imm.shape[0] != dim:
            raise ValueError(f"inverse_mass_matrix has {imm.shape[0]} entries, position has {dim}")
        return Metric("diag", imm.contiguous(), 0, None)
    if imm.ndim == 2 and (per_chain or (imm.shape[0] == n_chains and imm.shape[0] != imm.shape[1])):
        if imm.shape != (n_chains, dim):
            raise ValueError(
                f"per-chain inverse_mass_matrix must be ({n_chains}, {dim}), got {tuple(imm.shape)}")
        return Metric("diag", imm.contiguous(), dim, None)
    if imm.ndim == 2 and imm.shape[0] == imm.shape[1]:
        if imm.shape[0] != dim:
            raise ValueError(f"inverse_mass_matrix is {tuple(imm.shape)}, position has {dim} dims")
        return _dense_metric(imm.contiguous())
    if imm.ndim == 3 and imm.shape[1] == imm.shape[2]:
        # one dense matrix per chain (a vmapped dense warmup's output)
        if imm.shape[0] != n_chains or imm.shape[1] != dim:
            raise ValueError(
                f"per-chain dense inverse_mass_matrix must be ({n_chains}, {dim}, {dim}), "
                f"got {tuple(imm.shape)}")
        return _dense_metric(imm.contiguous())
    raise ValueError(
        "The mass matrix has the wrong number of dimensions:"
        f" expected 1 or 2, got {imm.ndim}."
    )


class PerChainDiag:
    """Marker wrapper: treat a 2-d array as per-chain diagonals even when N == D."""

    def __init__(self, imm):
        self.imm = imm


class PerChainDiagTensor(torch.Tensor):
    """An ``(N, D)`` tensor TAGGED as "one inverse-mass diagonal per chain".  ``window_adaptation``
    returns ``parameters["inverse_mass_matrix"]`` as this type, so the reference idiom
    ``nuts(logdensity_fn, **parameters)`` round-trips unambiguously even when ``N == D`` (where a
    plain square 2-d array means a dense matrix, metrics.py:180-218).  It is an ordinary tensor in
    every other respect (indexing, ``.cpu()``, arithmetic keep working) -- but the TAG DOES NOT
    PROPAGATE: every torch operation on it (slicing, ``torch.diag``, ``mean`` ...) returns a plain
    ``torch.Tensor``, so a dense ``(D, D)`` matrix a user derives from the warm-up output is read as the
    dense matrix it is.  Only the object ``window_adaptation`` returned carries the tag."""

    __torch_function__ = torch._C._disabled_torch_function_impl

    @staticmethod
    def tag(imm: torch.Tensor) -> "PerChainDiagTensor":
        if imm.ndim != 2:
            raise ValueError(f"per-chain diagonals must be (N, D), got {tuple(imm.shape)}")
        return imm.as_subclass(PerChainDiagTensor)


def _dense_metric(imm: torch.Tensor) -> Metric:
    """metrics.py:711-715: ``L = cholesky(imm, lower)``; ``mass_matrix_sqrt = L^{-T}``.
    Factorised once per distinct matrix (fp64, rounded once to fp32) instead of inside
    every kernel call as the reference does (hmc.py:289)."""
    key = (imm.data_ptr(), imm._version, tuple(imm.shape))
    hit = _DENSE_CACHE.get(key)
    if hit is not None:
        return hit
    L = torch.linalg.cholesky(imm.double())
    eye = torch.eye(imm.shape[-1], dtype=torch.float64, device=imm.device)
    if imm.ndim == 3:
        eye = eye.expand(imm.shape).contiguous()
    Linv = torch.linalg.solve_triangular(L, eye, upper=False)  # L^{-1} = (L^{-T})^T
    # "dense": one matrix shared by all chains (MFMA GEMMs); "dense_pc": one matrix per chain
    m = Metric("dense" if imm.ndim == 2 else "dense_pc", imm, 0, Linv.float().contiguous(),
               imm.t().contiguous() if imm.ndim == 2 else None)
    if len(_DENSE_CACHE) > 8:
        _DENSE_CACHE.clear()
    _DENSE_CACHE[key] = m
    return m
