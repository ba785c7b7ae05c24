"""Protocol types mirrored from blackjax/base.py:88-113,148-151."""
from __future__ import annotations

from typing import Callable, NamedTuple, Optional


class SamplingAlgorithm(tuple):
    """``init(position, rng_key=None) -> State`` ; ``step(rng_key, state) -> (State, Info)``
    (blackjax/base.py:88-113).  Behaves like the reference's two-field NamedTuple (``init, step =
    algorithm`` works); samplers that can advance every chain through many transitions without
    lockstep additionally expose ``run`` (``None`` otherwise), see ``blackjax_amd.nuts``."""

    _fields = ("init", "step")

    def __new__(cls, init: Callable, step: Callable, run: Optional[Callable] = None):
        self = super().__new__(cls, (init, step))
        self.run = run
        return self

    @property
    def init(self) -> Callable:
        return self[0]

    @property
    def step(self) -> Callable:
        return self[1]

    def __repr__(self):
        return f"SamplingAlgorithm(init={self[0]!r}, step={self[1]!r})"


class AdaptationAlgorithm(NamedTuple):
    """``run(rng_key, position, num_steps=1000) -> (AdaptationResults, info)``
    (blackjax/base.py:148-151)."""

    run: Callable
