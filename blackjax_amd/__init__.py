"""blackjax_amd -- MI355X-native HMC/NUTS engine behind the blackjax.hmc / blackjax.nuts /
blackjax.window_adaptation API surface (blackjax/__init__.py:70-80,104-112).

Positions are batched ``(n_chains, dim)`` float32 ROCm tensors; the log-density is a
PyTorch callable over the batch; all sampler arithmetic runs in hand-written HIP
kernels (``libbjxhip.so``, C ABI in ``include/bjx_hip.h``).  There is no CPU fallback.
"""
from __future__ import annotations

import functools as _functools

from . import dynamic_hmc as _dynamic_hmc
from . import ghmc as _ghmc
from . import hmc as _hmc
from . import nuts as _nuts
from . import adaptation, chees, diagnostics, distributed, integrators, meads, metrics, optim, random, rtc, targets, util
from .adaptation import staged_adaptation, window_adaptation
from .chees import chees_adaptation
from .meads import meads_adaptation
from .base import AdaptationAlgorithm, SamplingAlgorithm
from ._util import capturable, no_trace, returns_pair

__version__ = "0.1.0"


class GenerateSamplingAPI:
    """blackjax/__init__.py:70-80: callable that also exposes ``init`` / ``build_kernel``."""

    def __init__(self, differentiable, init, build_kernel):
        self.differentiable = differentiable
        self.init = init
        self.build_kernel = build_kernel

    def __call__(self, *args, **kwargs) -> SamplingAlgorithm:
        return self.differentiable(*args, **kwargs)


hmc = GenerateSamplingAPI(_hmc.as_top_level_api, _hmc.init, _hmc.build_kernel)
nuts = GenerateSamplingAPI(_nuts.as_top_level_api, _nuts.init, _nuts.build_kernel)
# blackjax/__init__.py:145-151: multinomial HMC shares HMCState / init with hmc
mhmc = GenerateSamplingAPI(
    _functools.partial(_hmc.as_top_level_api, build_proposal=_hmc.multinomial_hmc_proposal),
    _hmc.init,
    _functools.partial(_hmc.build_kernel, build_proposal=_hmc.multinomial_hmc_proposal),
)
multinomial_hmc = mhmc
dynamic_hmc = GenerateSamplingAPI(_dynamic_hmc.as_top_level_api, _dynamic_hmc.init,
                                  _dynamic_hmc.build_kernel)
# batched counterparts of the reference's default callables (dynamic_hmc.py:69-70) + key seeding
dynamic_hmc.next_key_fn = _dynamic_hmc.next_key_fn
dynamic_hmc.randint_steps_fn = _dynamic_hmc.randint_steps_fn
dynamic_hmc.chain_keys = _dynamic_hmc.chain_keys
dynamic_hmc.halton_sequence = _dynamic_hmc.halton_sequence
dynamic_hmc.halton_steps_fn = _dynamic_hmc.halton_steps_fn
dhmc = dynamic_hmc  # blackjax/__init__.py alias used by the ChEES examples
# blackjax/__init__.py:155-163: dynamic trajectory lengths with the multinomial (whole-trajectory) proposal
dmhmc = GenerateSamplingAPI(
    _functools.partial(_dynamic_hmc.as_top_level_api, build_proposal=_hmc.multinomial_hmc_proposal),
    _dynamic_hmc.init,
    _functools.partial(_dynamic_hmc.build_kernel, build_proposal=_hmc.multinomial_hmc_proposal),
)
hmc_family = [hmc, nuts, mhmc]  # blackjax/__init__.py:188
# Generalized HMC (blackjax/mcmc/ghmc.py), the sampler the MEADS warm-up tunes
ghmc = GenerateSamplingAPI(_ghmc.as_top_level_api, _ghmc.init, _ghmc.build_kernel)

__all__ = ["hmc", "nuts", "mhmc", "hmc_family", "multinomial_hmc", "dynamic_hmc", "dhmc", "dmhmc", "ghmc", "window_adaptation", "staged_adaptation", "chees_adaptation", "meads_adaptation", "chees", "meads", "optim", "adaptation", "diagnostics", "distributed", "util", "metrics", "integrators", "random", "rtc", "targets", "SamplingAlgorithm", "AdaptationAlgorithm", "capturable", "returns_pair", "no_trace"]
