"""``jax.lax`` stand-in: structured control flow as Python control flow (see ``tests/refshim/jax/__init__.py``)."""
from __future__ import annotations

import numpy as np
import torch

from . import _Missing, _stack, _wrap, asarray, tree_flatten, tree_unflatten


def __getattr__(item):
    if item.startswith("__") and item.endswith("__"):
        raise AttributeError(item)
    return _Missing(f"jax.lax.{item}")


def _truth(p):
    return bool(p.item()) if isinstance(p, torch.Tensor) else bool(p)


def _arrays(tree):
    """Loop carries and branch outputs are arrays in JAX: Python scalars become 0-d arrays (``~False`` must be a
    logical not, not the integer -1)."""
    from . import tree_map

    return tree_map(lambda v: asarray(v) if isinstance(v, (bool, int, float)) else v, tree)


def cond(pred, true_fun, false_fun, *operands, operand=None):
    if not callable(true_fun):
        # the legacy five-argument form cond(pred, true_operand, true_fun, false_operand, false_fun)
        # (blackjax/mcmc/termination.py:66-72)
        true_operand, t_fun, false_operand, f_fun = true_fun, false_fun, operands[0], operands[1]
        return _arrays(t_fun(true_operand) if _truth(pred) else f_fun(false_operand))
    if operand is not None:
        operands = (operand,)
    return _arrays(true_fun(*operands) if _truth(pred) else false_fun(*operands))


def switch(index, branches, *operands):
    i = int(index.item()) if isinstance(index, torch.Tensor) else int(index)
    return branches[min(max(i, 0), len(branches) - 1)](*operands)


def while_loop(cond_fun, body_fun, init_val):
    val = _arrays(init_val)
    while _truth(cond_fun(val)):
        val = _arrays(body_fun(val))
    return val


def fori_loop(lower, upper, body_fun, init_val):
    lo = int(lower.item()) if isinstance(lower, torch.Tensor) else int(lower)
    hi = int(upper.item()) if isinstance(upper, torch.Tensor) else int(upper)
    val = _arrays(init_val)
    for i in range(lo, hi):
        val = _arrays(body_fun(asarray(i), val))
    return val


def scan(f, init, xs=None, length=None, reverse=False, unroll=1):
    if xs is None:
        n, leaves, treedef = int(length), [], None
    else:
        leaves, treedef = tree_flatten(xs)
        n = leaves[0].shape[0] if leaves else int(length)
    carry, ys = _arrays(init), []
    order = range(n - 1, -1, -1) if reverse else range(n)
    for i in order:
        x = None if treedef is None else tree_unflatten(treedef, [v[i] for v in leaves])
        carry, y = f(carry, x)
        carry = _arrays(carry)
        ys.append(y)
    if reverse:
        ys.reverse()
    if not ys:
        return carry, None
    y_leaves = [tree_flatten(y)[0] for y in ys]
    y_def = tree_flatten(ys[0])[1]
    stacked = [_stack([yl[j] if not isinstance(yl[j], (bool, int, float)) else asarray(yl[j]) for yl in y_leaves])
               for j in range(len(y_leaves[0]))]
    return carry, tree_unflatten(y_def, stacked)


def stop_gradient(x):
    return x.detach() if isinstance(x, torch.Tensor) else x


def mul(a, b):
    return _wrap(asarray(a) * asarray(b))


def dot(a, b, precision=None, preferred_element_type=None):
    from .numpy import dot as _dot

    return _dot(a, b)


def select(pred, on_true, on_false):
    from .numpy import where

    return where(pred, on_true, on_false)


def dynamic_slice(x, start, sizes):
    idx = tuple(slice(int(s), int(s) + int(n)) for s, n in zip(start, sizes))
    return x[idx]


def dynamic_update_slice(x, update, start):
    y = x.clone()
    idx = tuple(slice(int(s), int(s) + int(n)) for s, n in zip(start, update.shape))
    y[idx] = update
    return y


class Precision:
    HIGHEST = "highest"
    HIGH = "high"
    DEFAULT = "default"
