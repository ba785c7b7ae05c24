"""Synthetic log-densities with fused HIP value-and-gradient kernels.

Each object is a PyTorch callable ``f(q: (N, D)) -> (logp: (N,), grad: (N, D))`` and
plays the role of the user's ``logdensity_fn`` in the bench and parity tests.  The
engine accepts ANY torch callable (see ``_util.value_and_grad``); these exist so the
benchmark's gradient evaluation moves the minimum 2 words per element.

* ``DiagGaussian``  logp = -1/2 sum q_i^2 inv_var_i      (reference fixture tests/fixtures.py:60-78)
* ``NealFunnel``    reference fixture tests/fixtures.py:81-98
* ``AR1Gaussian``   Sigma_ij = rho^|i-j|
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from ._util import check_batch


class _Target:
    _bjx_value_and_grad = True
    _bjx_returns_pair = True
    _bjx_capturable = True  # one kernel launch over static buffers: safe to record in a HIP graph

    def _alloc(self, q):
        q = check_batch(q, "q")
        return q, torch.empty(q.shape[0], dtype=torch.float32, device=q.device), torch.empty_like(q)


class DiagGaussian(_Target):
    def __init__(self, inv_var: torch.Tensor):
        self.inv_var = check_batch(inv_var, "inv_var")

    def _bjx_fused_target(self, dim: int):
        """(kind, parameter vector) for ``bjx_nuts_async_t.target_kind`` or None when the tick kernels
        cannot evaluate this target themselves (rows of at most 128 floats take another reduction order
        in the stand-alone kernel)."""
        if dim > 128 and tuple(self.inv_var.shape) == (dim,):
            return 2, self.inv_var
        return None

    def __call__(self, q):
        q, logp, g = self._alloc(q)
        N, D = q.shape
        if self.inv_var.shape != (D,):
            raise ValueError(f"inv_var has shape {tuple(self.inv_var.shape)}, expected ({D},)")
        _lib.call("bjx_target_diag_gaussian", _lib.current_stream(), N, D, self.inv_var.data_ptr(),
                  q.data_ptr(), logp.data_ptr(), g.data_ptr())
        return logp, g


class NealFunnel(_Target):
    def _bjx_fused_target(self, dim: int):
        return 1, None

    def __call__(self, q):
        q, logp, g = self._alloc(q)
        N, D = q.shape
        _lib.call("bjx_target_neal_funnel", _lib.current_stream(), N, D, q.data_ptr(),
                  logp.data_ptr(), g.data_ptr())
        return logp, g


class AR1Gaussian(_Target):
    def __init__(self, rho: float, dim: int):
        self.rho, self.dim = float(rho), int(dim)
        c = np.float32(1.0 / (1.0 - self.rho * self.rho))
        self.d_edge = float(np.float32(1.0) * c)
        self.d_mid = float(np.float32(1.0 + self.rho * self.rho) * c)
        self.off = float(np.float32(-self.rho) * c)

    def covariance(self, device) -> torch.Tensor:
        i = torch.arange(self.dim, device=device)
        return (self.rho ** (i[:, None] - i[None, :]).abs().double()).float()

    def __call__(self, q):
        q, logp, g = self._alloc(q)
        N, D = q.shape
        if D != self.dim:
            raise ValueError(f"expected dim {self.dim}, got {D}")
        _lib.call("bjx_target_ar1_gaussian", _lib.current_stream(), N, D, self.d_edge, self.d_mid,
                  self.off, q.data_ptr(), logp.data_ptr(), g.data_ptr())
        return logp, g
