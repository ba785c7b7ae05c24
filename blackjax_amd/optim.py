"""Scalar gradient transformations for the ChEES trajectory-length optimiser.

The reference takes any ``optax.GradientTransformation`` (chees_adaptation.py:741, 474-481) and only
ever applies it to ONE scalar, ``log(trajectory_length)``.  optax is a third-party dependency that
is not part of the reference tree (pinned 0.2.8, uv.lock:2197-2198); ``adam`` and ``sgd`` restate its
published update rules for a scalar parameter, in fp32 operation by operation (pow in fp64, rounded
once).  Any object with ``init(params) -> state`` and ``update(grad, state, params) -> (update, state)``
can be passed instead.
"""
from __future__ import annotations

from typing import NamedTuple

import numpy as np

f32 = np.float32
f64 = np.float64


class ScaleByAdamState(NamedTuple):
    count: int
    mu: np.float32
    nu: np.float32


class _Adam:
    def __init__(self, learning_rate, b1, b2, eps, eps_root):
        self.learning_rate, self.b1, self.b2 = float(learning_rate), float(b1), float(b2)
        self.eps, self.eps_root = float(eps), float(eps_root)

    def init(self, params):
        del params
        return ScaleByAdamState(0, f32(0.0), f32(0.0))

    def update(self, grad, state, params=None):
        """scale_by_adam: ``mu = (1-b1) g + b1 mu``, ``nu = (1-b2) g^2 + b2 nu``, bias correction
        ``m / (1 - b^count)``, ``u = mu_hat / (sqrt(nu_hat + eps_root) + eps)``; then ``-lr * u``."""
        del params
        g = f32(grad)
        b1, b2 = f32(self.b1), f32(self.b2)
        mu = f32(f32(f32(1.0 - self.b1) * g) + f32(b1 * state.mu))
        nu = f32(f32(f32(1.0 - self.b2) * f32(g * g)) + f32(b2 * state.nu))
        count = state.count + 1
        c1 = f32(f32(1.0) - f32(np.power(f64(b1), f64(count))))
        c2 = f32(f32(1.0) - f32(np.power(f64(b2), f64(count))))
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            mu_hat = f32(mu / c1)
            nu_hat = f32(nu / c2)
            u = f32(mu_hat / f32(f32(np.sqrt(f32(nu_hat + f32(self.eps_root)))) + f32(self.eps)))
            return f32(f32(-self.learning_rate) * u), ScaleByAdamState(count, mu, nu)


def adam(learning_rate, b1=0.9, b2=0.999, eps=1e-8, eps_root=0.0):
    """``optax.adam`` for one scalar parameter."""
    return _Adam(learning_rate, b1, b2, eps, eps_root)


class _SGD:
    def __init__(self, learning_rate):
        self.learning_rate = float(learning_rate)

    def init(self, params):
        del params
        return ()

    def update(self, grad, state, params=None):
        del params
        return f32(f32(-self.learning_rate) * f32(grad)), state


def sgd(learning_rate):
    """``optax.sgd`` (no momentum) for one scalar parameter."""
    return _SGD(learning_rate)
