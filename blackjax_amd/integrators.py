"""Palindromic symplectic integrators (blackjax/mcmc/integrators.py:270-369).

An integrator is its coefficient list ``[b1, a1, b2, a2, ..., b1]`` (momentum / position updates
alternating, ``generalized_two_stage_integrator`` 62-152).  ``velocity_verlet`` is implemented for
every sampler and metric; ``mclachlan``, ``yoshida`` and ``omelyan`` (any palindromic list) run through
the general-coefficient kernels: ``hmc`` and ``dynamic_hmc`` with diagonal and dense metrics
(``bjx_leapfrog_*_coef`` / ``bjx_hmc_finish_*_coef``), ``mhmc`` / ``dmhmc`` with diagonal
(``bjx_mhmc_step_diag_coef``) and -- round 4 -- dense metrics (``bjx_mhmc_step_dense_coef``), and ``nuts`` -- lockstep ``step`` with diagonal and dense metrics
(``bjx_nuts_t.int_kick / int_drift`` + ``bjx_nuts_mid``) and, round 4, ``run`` on the FREE-RUNNING tick kernels
for a diagonal metric with 16-byte rows of at most 512 floats (a leaf lasts K ticks,
``bjx_nuts_async_t.int_stages``; other shapes run the same transitions as lockstep steps);
``window_adaptation(..., free_running=True)`` takes the same integrators under the same conditions.  The non-Euclidean
integrators of the reference (isokinetic, maruyama, implicit midpoint) are out of scope.
"""
from __future__ import annotations


class Integrator:
    def __init__(self, name, coefficients):
        self.name = name
        self.coefficients = tuple(float(c) for c in coefficients)
        assert len(self.coefficients) % 2 == 1 and self.coefficients == self.coefficients[::-1]

    @property
    def num_gradients_per_step(self) -> int:
        return (len(self.coefficients) - 1) // 2

    def __repr__(self):
        return self.name


velocity_verlet = Integrator("velocity_verlet", [0.5, 1.0, 0.5])  # integrators.py:321-322

_b1 = 0.1931833275037836  # integrators.py:335-340
mclachlan = Integrator("mclachlan", [_b1, 0.5, 1 - 2 * _b1, 0.5, _b1])

_b1, _a1 = 0.11888010966548, 0.29619504261126  # integrators.py:350-356
yoshida = Integrator("yoshida", [_b1, _a1, 0.5 - _b1, 1 - 2 * _a1, 0.5 - _b1, _a1, _b1])

_b1, _a1, _b2, _a2 = (0.08398315262876693, 0.2539785108410595, 0.6822365335719091,
                      -0.03230286765269967)  # integrators.py:362-369
_b3, _a3 = 0.5 - _b1 - _b2, 1 - 2 * (_a1 + _a2)
omelyan = Integrator("omelyan", [_b1, _a1, _b2, _a2, _b3, _a3, _b3, _a2, _b2, _a1, _b1])


def check_supported(integrator, allow_general: bool = False):
    """``allow_general``: the caller implements arbitrary palindromic coefficients."""
    if integrator is velocity_verlet:
        return
    if allow_general and isinstance(integrator, Integrator):
        return
    raise NotImplementedError(
        "this sampler/metric implements the velocity_verlet integrator only; got %r "
        "(mclachlan / yoshida / omelyan: see blackjax_amd.integrators for where they are available)"
        % (integrator,))


def __getattr__(name):
    # blackjax/mcmc/integrators.py:43-53 defines IntegratorState; here it lives with the samplers that fill it
    # (blackjax_amd.hmc) -- resolved lazily because that module imports this one
    if name == "IntegratorState":
        from .hmc import IntegratorState

        return IntegratorState
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
