"""Driver helpers mirrored from blackjax/util.py."""
from __future__ import annotations

from typing import Callable

import torch

from . import random as bjx_random

__all__ = ["run_inference_algorithm", "stack_history"]


def stack_history(history):
    """Stack a list of per-step pytrees (NamedTuples / tuples / tensors / None) along a new leading
    axis -- the ``lax.scan`` output convention of the reference."""
    if not history:
        return None

    def stack(items):
        first = items[0]
        if first is None:
            return None
        if isinstance(first, torch.Tensor):
            if any(it.shape != first.shape for it in items):
                return items
            return torch.stack(items)
        if isinstance(first, tuple) and hasattr(first, "_fields"):
            return type(first)(*[stack([it[i] for it in items]) for i in range(len(first))])
        if isinstance(first, tuple):
            return tuple(stack([it[i] for it in items]) for i in range(len(first)))
        if isinstance(first, (int, float, bool)):
            return torch.tensor(items)
        return items

    return stack(history)


def run_inference_algorithm(rng_key, inference_algorithm, num_steps: int, initial_state=None,
                            initial_position=None,
                            transform: Callable = lambda state, info: (state, info),
                            *, key_layout: str = "step_major", free_running: bool = False):
    """blackjax/util.py:150-213.  ``keys = split(rng_key, num_steps)``; step ``t`` calls
    ``inference_algorithm.step(keys[t], state)`` which derives chain ``i``'s key as
    ``split(keys[t], N)[i]`` ("step_major", the vmap-inside-scan layout of
    docs/examples/howto_sample_multiple_chains.md:116-130).  ``key_layout="chain_major"``
    reproduces ``vmap`` over whole per-chain loops instead (chain key ``split(rng_key, N)[i]``,
    step key ``split(chain_key, num_steps)[t]``; tests/mcmc/test_sampling.py:1454-1465).

    Returns ``(final_state, history)`` with ``history`` stacked along a leading step axis.

    ``free_running=True`` (algorithms with a ``run`` method, i.e. NUTS): the same transitions, same
    keys, same draws, driven without lockstep across transitions (``nuts.run_free``); ``transform``
    is not applied, ``history`` is ``(positions (num_steps, N, D), NUTSRunInfo)``.
    """
    if initial_state is None and initial_position is None:
        raise ValueError("Either `initial_state` or `initial_position` must be provided.")
    if initial_state is not None and initial_position is not None:
        raise ValueError("Only one of `initial_state` or `initial_position` must be provided.")
    if key_layout not in ("step_major", "chain_major"):
        raise ValueError("key_layout must be 'step_major' or 'chain_major'")
    if initial_state is None:
        rng_key, init_key = bjx_random.split(rng_key, 2)
        initial_state = inference_algorithm.init(initial_position, init_key)
    state = initial_state
    if free_running:
        run = getattr(inference_algorithm, "run", None)
        if run is None:
            raise NotImplementedError("free_running=True needs an algorithm with a run method (blackjax_amd.nuts)")
        state, positions, info = run(rng_key, state, num_steps, key_layout=key_layout)
        return state, (positions, info)
    history = []
    if key_layout == "step_major":
        keys = bjx_random.split(rng_key, num_steps)
        for t in range(num_steps):
            state, info = inference_algorithm.step(keys[t], state)
            history.append(transform(state, info))
    else:
        run_key = bjx_random.key_words(rng_key)
        for t in range(num_steps):
            state, info = inference_algorithm.step(bjx_random.ChainMajorKey(run_key, t), state)
            history.append(transform(state, info))
    return state, stack_history(history)


# ------------------------------------------------------------------------------- pytree positions
def _tree_leaves(tree, path=()):
    """Leaves of a dict / list / tuple tree of tensors in jax.tree_util order (dict keys sorted)."""
    if isinstance(tree, dict):
        for k in sorted(tree):
            yield from _tree_leaves(tree[k], path + (k,))
    elif isinstance(tree, (list, tuple)):
        for i, v in enumerate(tree):
            yield from _tree_leaves(v, path + (i,))
    else:
        yield path, tree


def _tree_build(tree, leaves_iter):
    if isinstance(tree, dict):
        built = {k: _tree_build(tree[k], leaves_iter) for k in sorted(tree)}
        return type(tree)((k, built[k]) for k in tree) if type(tree) is not dict else {k: built[k] for k in tree}
    if isinstance(tree, tuple) and hasattr(tree, "_fields"):  # namedtuple
        return type(tree)(*[_tree_build(v, leaves_iter) for v in tree])
    if isinstance(tree, (list, tuple)):
        return type(tree)(_tree_build(v, leaves_iter) for v in tree)
    return next(leaves_iter)


def ravel_chain_pytree(tree):
    """The engine's positions are one ``(N, D)`` tensor; the reference accepts any pytree of arrays
    (a dict of parameters, say) and flattens it with ``jax.flatten_util.ravel_pytree`` where it
    needs a vector (adaptation/mass_matrix.py:288-291, util.py:66-91).  This is the batched
    counterpart: ``tree`` is a dict / list / tuple tree whose leaves are ``(N, ...)`` tensors (chain
    axis first); returns ``(flat, unravel)`` with ``flat`` the ``(N, D)`` float32 tensor of the leaves
    concatenated in ``jax.tree_util`` order (dict keys sorted) and ``unravel(x)`` mapping any
    ``(M, D)`` tensor back to a tree of ``(M, ...)`` views of it (differentiable)."""
    import torch

    leaves = [leaf for _, leaf in _tree_leaves(tree)]
    if not leaves:
        raise ValueError("ravel_chain_pytree: the tree has no leaves")
    n = leaves[0].shape[0]
    for path, leaf in _tree_leaves(tree):
        if not isinstance(leaf, torch.Tensor) or leaf.ndim < 1 or leaf.shape[0] != n:
            raise ValueError(f"leaf {path} must be a tensor with the chain axis ({n}) first")
    shapes = [tuple(leaf.shape[1:]) for leaf in leaves]
    sizes = [int(torch.Size(s).numel()) if s else 1 for s in shapes]
    flat = torch.cat([leaf.reshape(n, -1).to(torch.float32) for leaf in leaves], dim=1).contiguous()

    def unravel(x):
        if x.ndim != 2 or x.shape[1] != sum(sizes):
            raise ValueError(f"expected a (M, {sum(sizes)}) tensor, got {tuple(x.shape)}")
        parts = torch.split(x, sizes, dim=1)
        return _tree_build(tree, iter(p.reshape((x.shape[0],) + s) for p, s in zip(parts, shapes)))

    return flat, unravel


def flat_logdensity(logdensity_fn: Callable, unravel: Callable) -> Callable:
    """``logdensity_fn`` over a pytree of ``(N, ...)`` tensors -> the ``(N, D) -> (N,)`` callable the
    samplers take (gradient by autograd through ``unravel``)."""

    def fn(q):
        return logdensity_fn(unravel(q))

    return fn
