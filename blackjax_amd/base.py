"""Protocol types mirrored from blackjax/base.py:88-113,148-151."""
from __future__ import annotations

from typing import Callable, NamedTuple


class SamplingAlgorithm(NamedTuple):
    """``init(position, rng_key=None) -> State`` ; ``step(rng_key, state) -> (State, Info)``
    (blackjax/base.py:88-113)."""

    init: Callable
    step: Callable


class AdaptationAlgorithm(NamedTuple):
    """``run(rng_key, position, num_steps=1000) -> (AdaptationResults, info)``
    (blackjax/base.py:148-151)."""

    run: Callable
