"""NOT optax.  ``chees_adaptation.py`` imports optax for ONE thing, ``optax.adam(learning_rate)`` (its default optimiser
for the log trajectory length) plus ``apply_updates``.  Optax is a third-party dependency that is not under
``/root/reference`` (pinned ``optax`` in ``uv.lock``); this module restates the published Adam update in the form optax
uses (Kingma & Ba 2015; ``optax.scale_by_adam`` followed by ``scale(-learning_rate)``: bias-corrected first and second
moments, ``eps = 1e-8`` OUTSIDE the square root, ``eps_root = 0``) so that the REST of the reference's ChEES code -- which
is under ``/root/reference`` -- can run on ``tests/refshim``.  Nothing computed through this module pins optax itself."""
from __future__ import annotations

from typing import Any, Callable, NamedTuple

import jax
import jax.numpy as jnp


class GradientTransformation(NamedTuple):
    init: Callable
    update: Callable


OptState = Any
Params = Any
Updates = Any


class ScaleByAdamState(NamedTuple):
    count: Any
    mu: Any
    nu: Any


def adam(learning_rate, b1=0.9, b2=0.999, eps=1e-8, eps_root=0.0):
    def init(params):
        zeros = jax.tree.map(jnp.zeros_like, params)
        return ScaleByAdamState(jnp.zeros((), jnp.int32), zeros, jax.tree.map(jnp.zeros_like, params))

    def update(grads, state, params=None):
        count = state.count + 1
        mu = jax.tree.map(lambda g, m: (1 - b1) * g + b1 * m, grads, state.mu)
        nu = jax.tree.map(lambda g, v: (1 - b2) * (g * g) + b2 * v, grads, state.nu)
        c = count.astype(jnp.float32)
        bc1 = 1 - b1 ** c
        bc2 = 1 - b2 ** c
        updates = jax.tree.map(lambda m, v: -learning_rate * ((m / bc1) / (jnp.sqrt(v / bc2 + eps_root) + eps)), mu, nu)
        return updates, ScaleByAdamState(count, mu, nu)

    return GradientTransformation(init, update)


def apply_updates(params, updates):
    return jax.tree.map(lambda p, u: p + u, params, updates)


def sgd(learning_rate):
    """``optax.sgd`` without momentum: updates = -learning_rate * grads."""

    def init(params):
        return ()

    def update(grads, state, params=None):
        return jax.tree.map(lambda g: -learning_rate * g, grads), state

    return GradientTransformation(init, update)
