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


class DeviceTarget(_Target):
    """A log-density written by the USER as HIP device code, compiled at run time (``blackjax_amd.rtc``: hiprtc)
    into (a) a stand-alone ``(logp, grad)`` kernel -- so the object is an ordinary recordable callable under the
    external-callable contract -- and (b) the engine's whole-transition kernel (``hmc(..., fuse_target=True)``),
    where it is evaluated in registers between two leapfrogs.  Both call the same ``eval``, so the two paths give
    the same bits.

    ``source`` defines a struct (default name ``Target``) with the interface of ``csrc/bjx_traj_dev.h``::

        struct Target {
          template <int NI> struct Ctx { ... };          // per-chain constants in registers; may be empty
          template <int NI> static __device__ void init(Ctx<NI>& c, int64_t D, const float* params);
          template <int NI> static __device__ void eval(const Ctx<NI>& c, int64_t D, const float* params,
                                                        const F4 (&x)[NI], bool need_logp, F4 (&g)[NI], float& lp);
        };

    ``x`` / ``g`` hold the chain's row in NI 16-byte pieces per lane (piece k of lane l = columns
    4 (l + 64 k) .. + 3; guard with ``j < D``); the 64 lanes of a wave call ``eval`` together and ``lp`` must be
    the same in every lane (``bjx::wave_sum``).  ``params``: a float32 device tensor handed to both functions
    (data, hyper-parameters), or ``None``.  Rows of at most 1 024 floats, ``D % 4 == 0``.

    The numerics of user code are the user's: compiled with ``-ffp-contract=off`` like the library, nothing is
    checked against an oracle -- ``tests/test_device_target_gpu.py`` checks the machinery on a target whose
    gradient autograd reproduces."""

    def __init__(self, source: str, params: torch.Tensor | None = None, struct: str = "Target"):
        self.source, self.struct = str(source), str(struct)
        if params is not None:
            if not (isinstance(params, torch.Tensor) and params.dtype == torch.float32 and params.is_contiguous()):
                raise ValueError("params must be a contiguous float32 tensor")
        self.params = params
        self._code = None
        self._module = None

    def code_object(self) -> bytes:
        """The compiled gfx950 code object (hiprtc cross-compiles: no GPU needed)."""
        if self._code is None:
            from . import rtc

            self._code = rtc.compile(rtc.TARGET_TU % {"source": self.source, "struct": self.struct},
                                     f"bjx_device_target_{self.struct}.hip")
        return self._code

    def module(self):
        if self._module is None:
            from . import rtc

            self._module = rtc.Module(self.code_object())
        return self._module

    def nuts_module(self):
        """The free-running NUTS multi-tick kernel compiled around this target (``run(..., fuse_target=True)``;
        a separate, larger code object: ~10 s on first use)."""
        if getattr(self, "_nuts_module", None) is None:
            from . import rtc

            self._nuts_module = rtc.Module(rtc.compile(rtc.NUTS_TU % {"source": self.source, "struct": self.struct},
                                                       f"bjx_device_target_nuts_{self.struct}.hip"))
        return self._nuts_module

    def _params_ptr(self, device):
        if self.params is None:
            return 0
        if self.params.device != device:
            raise ValueError(f"params live on {self.params.device}, the chains on {device}")
        return self.params.data_ptr()

    def _bjx_fused_target(self, dim: int):
        return ("rtc", self) if dim % 4 == 0 and dim <= 1024 else None

    def __call__(self, q):
        import ctypes

        from . import rtc

        q, logp, g = self._alloc(q)
        N, D = q.shape
        if D % 4 != 0 or D > 1024:
            raise ValueError("DeviceTarget: rows of at most 1 024 floats, a multiple of 4")
        if N == 0:
            return logp, g
        grid = min((N + 3) // 4, 1 << 20)
        self.module().launch(f"bjx_rtc_eval_{rtc.ni_for(D)}", grid, 256, _lib.current_stream(),
                             ctypes.c_longlong(N), ctypes.c_longlong(D), ctypes.c_void_p(self._params_ptr(q.device)),
                             ctypes.c_void_p(q.data_ptr()), ctypes.c_void_p(logp.data_ptr()),
                             ctypes.c_void_p(g.data_ptr()))
        return logp, g


class ElementwiseRowsTarget(_Target):
    """``from_elementwise`` for rows the register-resident ``DeviceTarget`` cannot hold (D > 1 024 or D % 4 != 0): the
    generated element-wise value-and-gradient arithmetic in a row-loop kernel (``elementwise.trace(...).rows_source``,
    hiprtc).  An ordinary recordable external callable ``f(q) -> (logp, grad)`` moving 8 bytes per element; NOT
    available to ``fuse_target=True`` (the engine's kernels keep a row in registers)."""

    def __init__(self, src, dim: int):
        self.elementwise, self.dim = src, int(dim)
        self.params = src.params
        self._module = None

    def code_object(self) -> bytes:
        from . import elementwise, rtc

        return rtc.compile(elementwise.ROWS_TU % {"source": self.elementwise.rows_source}, "bjx_elementwise_rows.hip")

    def module(self):
        if self._module is None:
            from . import rtc

            self._module = rtc.Module(self.code_object())
        return self._module

    def _bjx_fused_target(self, dim: int):
        return None

    def __call__(self, q):
        import ctypes

        q, logp, g = self._alloc(q)
        N, D = q.shape
        if D != self.dim:
            raise ValueError(f"this target was generated for D = {self.dim}, got {D}")
        if N == 0:
            return logp, g
        if self.params is not None and self.params.device != q.device:
            raise ValueError(f"params live on {self.params.device}, the chains on {q.device}")
        grid = min((N + 3) // 4, 1 << 20)
        self.module().launch("bjx_rtc_ew_rows", grid, 256, _lib.current_stream(), ctypes.c_longlong(N),
                             ctypes.c_longlong(D), ctypes.c_void_p(0 if self.params is None else self.params.data_ptr()),
                             ctypes.c_void_p(q.data_ptr()), ctypes.c_void_p(logp.data_ptr()), ctypes.c_void_p(g.data_ptr()))
        return logp, g


def from_elementwise(fn, dim: int, device="cuda"):
    """A plain PyTorch log-density of the element-wise + row-sum shape -> ONE generated HIP value-and-gradient kernel
    (``blackjax_amd.elementwise``: torch.fx trace, forward-mode derivative, hiprtc): an ordinary external callable
    ``f(q) -> (logp, grad)`` that moves 8 bytes per element instead of autograd's 50-80 -- a ``DeviceTarget`` for rows
    of at most 1 024 floats with D % 4 == 0, an ``ElementwiseRowsTarget`` (row-loop kernel) for any other D.
    The reference gets this fusion from ``jax.value_and_grad`` under XLA (mcmc/integrators.py:189,204).
    ``hmc / nuts(logdensity_fn)`` try this themselves on a plain PyTorch function (``_util.value_and_grad``)."""
    from .elementwise import from_elementwise as _f

    return _f(fn, dim, device)
