"""Oracle for the palindromic integrators (TEST INFRASTRUCTURE, see package docstring).

generalized_two_stage_integrator (blackjax/mcmc/integrators.py:104-150) with the coefficient lists
of velocity_verlet / mclachlan / yoshida / omelyan (321-322, 335-369); momentum update
``p + (step_size*coef)*grad`` (236) and position update ``q + (step_size*coef)*kinetic_grad`` (200)
are single fmas; the kinetic gradient is ``imm * p`` (224, 242).
"""
from __future__ import annotations

import numpy as np

from . import hmc as ohmc
from . import prng
from .fp import exp_cr, f32, fma32

velocity_verlet = [0.5, 1.0, 0.5]
_b1 = 0.1931833275037836
mclachlan = [_b1, 0.5, 1 - 2 * _b1, 0.5, _b1]
_b1, _a1 = 0.11888010966548, 0.29619504261126
yoshida = [_b1, _a1, 0.5 - _b1, 1 - 2 * _a1, 0.5 - _b1, _a1, _b1]
_b1, _a1, _b2, _a2 = 0.08398315262876693, 0.2539785108410595, 0.6822365335719091, -0.03230286765269967
_b3, _a3 = 0.5 - _b1 - _b2, 1 - 2 * (_a1 + _a2)
omelyan = [_b1, _a1, _b2, _a2, _b3, _a3, _b3, _a2, _b2, _a1, _b1]


def one_step(state: ohmc.IntegratorState, step_size, logdensity_fn, metric, coefficients):
    """integrators.py:104-150 for a batch of chains; ``step_size`` scalar or (N,)."""
    q, p, logp, g = state
    eps = ohmc._col(step_size) if np.ndim(step_size) else f32(step_size)
    for i, coef in enumerate(coefficients[:-1]):
        c = (eps * f32(coef)).astype(f32) if np.ndim(eps) else f32(eps * f32(coef))
        if i % 2 == 0:
            p = fma32(c, g, p)
            v = ohmc.linear_map(metric, metric.inverse_mass_matrix, p)
        else:
            q = fma32(c, v, q)
            logp, g = logdensity_fn(q)
    c = (eps * f32(coefficients[-1])).astype(f32) if np.ndim(eps) else f32(eps * f32(coefficients[-1]))
    p = fma32(c, g, p)
    return ohmc.IntegratorState(q, p, np.asarray(logp, f32), np.asarray(g, f32))


def hmc_kernel(rng_key, state: ohmc.HMCState, logdensity_fn, step_size, inverse_mass_matrix,
               num_integration_steps: int, coefficients, divergence_threshold: float = 1000.0,
               chain_offset: int = 0, metric=None):
    """blackjax.hmc(..., integrator=<palindromic integrator>): hmc.py:279-312 / 153-176."""
    N, D = state.position.shape
    if metric is None:
        metric = ohmc.default_metric(inverse_mass_matrix, n_chains=N)
    kk = prng.split(ohmc.chain_keys(rng_key, N, chain_offset), 2)
    p0 = ohmc.sample_momentum(metric, kk[:, 0], D)
    z0 = ohmc.IntegratorState(state.position, p0, state.logdensity, state.logdensity_grad)
    z = z0
    for _ in range(num_integration_steps):
        z = one_step(z, step_size, logdensity_fn, metric, coefficients)
    e0, e1 = ohmc.hmc_energy(metric, z0), ohmc.hmc_energy(metric, z)
    delta = ohmc.safe_energy_diff(e0, e1)
    p_acc = np.minimum(exp_cr(delta), f32(1.0))
    acc = prng.uniform(kk[:, 1], ()) < p_acc
    new = ohmc.HMCState(np.where(acc[:, None], z.position, state.position).astype(f32),
                        np.where(acc, z.logdensity, state.logdensity).astype(f32),
                        np.where(acc[:, None], z.logdensity_grad, state.logdensity_grad).astype(f32))
    return new, (p_acc, acc, (-delta) > f32(divergence_threshold), e1, z)
