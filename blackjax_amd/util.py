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
