"""``jax.scipy`` stand-in (see ``tests/refshim/jax/__init__.py``: NOT JAX)."""
from __future__ import annotations

import sys
import types

import torch

from .. import _Missing, _wrap, asarray


def _t(x):
    return asarray(x).as_subclass(torch.Tensor)


linalg = types.ModuleType("jax.scipy.linalg")


def _cholesky(a, lower=False):
    L = torch.linalg.cholesky(_t(a))
    return _wrap(L if lower else L.mT.contiguous())


def _solve_triangular(a, b, trans=0, lower=False, unit_diagonal=False):
    at, bt = _t(a), _t(b)
    if trans in (1, "T", 2, "C"):
        at, lower = at.mT, not lower
    vec = bt.ndim == 1
    out = torch.linalg.solve_triangular(at, bt.unsqueeze(-1) if vec else bt, upper=not lower, unitriangular=unit_diagonal)
    return _wrap(out.squeeze(-1) if vec else out)


linalg.cholesky, linalg.solve_triangular = _cholesky, _solve_triangular
linalg.inv = lambda a: _wrap(torch.linalg.inv(_t(a)))
linalg.solve = lambda a, b, **k: _wrap(torch.linalg.solve(_t(a), _t(b)))
linalg.cho_solve = lambda c_and_lower, b: _wrap(torch.cholesky_solve(
    _t(b).unsqueeze(-1) if _t(b).ndim == 1 else _t(b), _t(c_and_lower[0]), upper=not c_and_lower[1]).squeeze(-1)
    if _t(b).ndim == 1 else torch.cholesky_solve(_t(b), _t(c_and_lower[0]), upper=not c_and_lower[1]))
linalg.__getattr__ = lambda item: _Missing(f"jax.scipy.linalg.{item}")

special = types.ModuleType("jax.scipy.special")
special.expit = lambda x: _wrap(torch.sigmoid(_t(x).to(torch.float32) if not _t(x).is_floating_point() else _t(x)))
special.logsumexp = lambda a, axis=None, b=None, keepdims=False: _wrap(
    torch.logsumexp(_t(a), dim=tuple(range(_t(a).ndim)) if axis is None else axis, keepdim=keepdims))
special.logit = lambda x: _wrap(torch.logit(_t(x)))
special.erf = lambda x: _wrap(torch.erf(_t(x)))
special.erfinv = lambda x: _wrap(torch.erfinv(_t(x)))
special.gammaln = lambda x: _wrap(torch.lgamma(_t(x)))
special.ndtri = lambda x: _wrap(torch.special.ndtri(_t(x)))
special.__getattr__ = lambda item: _Missing(f"jax.scipy.special.{item}")

stats = types.ModuleType("jax.scipy.stats")
stats.__path__ = []
stats.__getattr__ = lambda item: _Missing(f"jax.scipy.stats.{item}")
_norm = types.ModuleType("jax.scipy.stats.norm")
_norm.logpdf = lambda x, loc=0.0, scale=1.0: _wrap(
    -0.5 * ((_t(x) - _t(loc)) / _t(scale)) ** 2 - torch.log(_t(scale)) - 0.9189385332046727)
_norm.__getattr__ = lambda item: _Missing(f"jax.scipy.stats.norm.{item}")
stats.norm = _norm
_expon = types.ModuleType("jax.scipy.stats.expon")
_expon.logpdf = lambda x, loc=0.0, scale=1.0: _wrap(torch.where(
    (_t(x) - _t(loc)) / _t(scale) >= 0, -((_t(x) - _t(loc)) / _t(scale)) - torch.log(_t(scale).to(torch.float32)),
    torch.tensor(-float("inf"))))
_expon.__getattr__ = lambda item: _Missing(f"jax.scipy.stats.expon.{item}")
stats.expon = _expon
sys.modules["jax.scipy.stats.expon"] = _expon
_mvn = types.ModuleType("jax.scipy.stats.multivariate_normal")


def _mvn_logpdf(x, mean, cov):
    xt, mt, ct = _t(x), _t(mean), _t(cov)
    d = xt - mt
    L = torch.linalg.cholesky(ct)
    z = torch.linalg.solve_triangular(L, d.unsqueeze(-1), upper=False).squeeze(-1)
    k = d.shape[-1]
    return _wrap(-0.5 * (k * 1.8378770664093453 + 2.0 * torch.log(torch.diagonal(L)).sum() + (z * z).sum(-1)))


_mvn.logpdf = _mvn_logpdf
_mvn.__getattr__ = lambda item: _Missing(f"jax.scipy.stats.multivariate_normal.{item}")
stats.multivariate_normal = _mvn
sys.modules["jax.scipy.stats.multivariate_normal"] = _mvn

sys.modules["jax.scipy.linalg"], sys.modules["jax.scipy.special"] = linalg, special
sys.modules["jax.scipy.stats"], sys.modules["jax.scipy.stats.norm"] = stats, _norm


def __getattr__(item):
    if item.startswith("__") and item.endswith("__"):
        raise AttributeError(item)
    if item in ("linalg", "special", "stats"):
        return sys.modules["jax.scipy." + item]
    return _Missing(f"jax.scipy.{item}")
