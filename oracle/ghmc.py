"""Oracle for the Generalized HMC transition (TEST INFRASTRUCTURE, see package docstring).

Batched restatement with a leading chain axis; chain ``i`` of ``kernel(rng_key, state, ...)`` equals
the reference's single-chain ``blackjax.ghmc.build_kernel()(jax.random.split(rng_key, N)[i], ...)``
(the layout ``meads_adaptation`` uses: ``keys = split(rng_key, num_chains + 1)``,
meads_adaptation.py:521-522).

Reference lines followed
* GHMCState / init                     blackjax/mcmc/ghmc.py:32-64
* momentum metric from an inverse scale blackjax/mcmc/ghmc.py:67-86 (legacy diagonal form: scale ** 2)
* kernel                               blackjax/mcmc/ghmc.py:116-198
* update_momentum                      blackjax/mcmc/ghmc.py:203-223
* hmc_proposal.generate (L = 1)        blackjax/mcmc/hmc.py:153-176
* nonreversible_slice_sampling         blackjax/mcmc/proposal.py:243-264
* flip_momentum                        blackjax/mcmc/hmc.py:95-112

The per-dimension inverse-scale form of ``momentum_inverse_scale`` (what MEADS passes) and -- ``metric=`` --
any Gaussian-Euclidean metric of oracle/hmc.py, e.g. a dense inverse mass matrix (the reference's rich-metric
branch, ghmc.py:67-86: a 2-d array passes straight to ``default_metric``).  Low-rank metrics are not restated.
"""
from __future__ import annotations

from typing import Callable, NamedTuple

import numpy as np

from . import hmc as ohmc
from . import prng
from .fp import exp_cr, f32, log_cr, sqrt32


class GHMCState(NamedTuple):  # ghmc.py:32-50
    position: np.ndarray  # (N, D)
    momentum: np.ndarray  # (N, D)
    logdensity: np.ndarray  # (N,)
    logdensity_grad: np.ndarray  # (N, D)
    slice: np.ndarray  # (N,)


def _chain_keys(rng_key, n, chain_offset, override):
    return prng.split(rng_key, n, offset=chain_offset) if override is None else override


def init(position, logdensity_fn: Callable, rng_key, chain_offset: int = 0,
         chain_keys_override=None) -> GHMCState:
    """ghmc.py:53-64, chain i with key split(rng_key, N)[i] (meads_adaptation.py:726-727)."""
    position = np.asarray(position, dtype=f32)
    N, D = position.shape
    logp, grad = logdensity_fn(position)
    keys = _chain_keys(rng_key, N, chain_offset, chain_keys_override)
    kk = prng.split(keys, 2)
    momentum = prng.normal(kk[:, 0], (D,))  # generate_gaussian_noise(mu=0, sigma=1), util.py:66-91
    sl = prng.uniform(kk[:, 1], (), -1.0, 1.0)
    return GHMCState(position, momentum, np.asarray(logp, f32), np.asarray(grad, f32), sl.astype(f32))


def _per_chain(x, n):
    x = np.asarray(x, dtype=f32)
    return np.broadcast_to(x, (n,)).astype(f32) if x.ndim == 0 else x


def kernel(rng_key, state: GHMCState, logdensity_fn, step_size, momentum_inverse_scale, alpha, delta,
           divergence_threshold: float = 1000.0, chain_offset: int = 0, chain_keys_override=None, metric=None):
    """ghmc.py:116-198.  ``step_size``, ``alpha``, ``delta``: scalars or (N,);
    ``momentum_inverse_scale``: scalar, (D,) or (N, D) inverse scale (squared into the inverse
    mass matrix, ghmc.py:86)."""
    q, p_prev, logp, g, sl = state
    N, D = q.shape
    if metric is None:
        scale = np.asarray(momentum_inverse_scale, dtype=f32)
        imm = (scale * scale).astype(f32)
        if imm.ndim == 0:
            imm = np.full(D, imm, f32)
        metric = ohmc.default_metric(imm, n_chains=N, per_chain_diag=imm.ndim == 2)
    keys = _chain_keys(rng_key, N, chain_offset, chain_keys_override)
    kk = prng.split(keys, 2)  # key_momentum, key_noise (noise_fn = 0, ghmc.py:89-91)
    a = _per_chain(alpha, N)[:, None]
    d = _per_chain(delta, N)
    # update_momentum, ghmc.py:216-221: prev * sqrt(1 - alpha) + sqrt(alpha) * fresh (two products, one sum)
    fresh = ohmc.sample_momentum(metric, kk[:, 0], D)
    s1 = sqrt32((f32(1.0) - a).astype(f32))
    s2 = sqrt32(a)
    p = ((p_prev * s1).astype(f32) + (s2 * fresh).astype(f32)).astype(f32)
    # slice = ((slice + 1 + delta + 0) % 2) - 1, ghmc.py:176
    t = (((sl + f32(1.0)).astype(f32) + d).astype(f32) + f32(0.0)).astype(f32)
    sl_now = (np.mod(t, f32(2.0)).astype(f32) - f32(1.0)).astype(f32)

    z0 = ohmc.IntegratorState(q, p, logp, g)
    z1 = ohmc.velocity_verlet(z0, step_size, logdensity_fn, metric)  # one step, hmc.py:118
    end = ohmc.IntegratorState(z1.position, (f32(-1.0) * z1.momentum).astype(f32), z1.logdensity,
                               z1.logdensity_grad)
    e0 = ohmc.hmc_energy(metric, z0)
    e1 = ohmc.hmc_energy(metric, end)
    dE = ohmc.safe_energy_diff(e0, e1)
    is_div = (-dE) > f32(divergence_threshold)
    # nonreversible_slice_sampling, proposal.py:253-256
    p_acc = np.minimum(exp_cr(dE), f32(1.0))
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        acc = log_cr(np.abs(sl_now)) <= dE
        accf = acc.astype(f32)
        factor = ((exp_cr((-dE).astype(f32)) * accf).astype(f32) + (f32(1.0) - accf).astype(f32)).astype(f32)
        sl_next = (sl_now * factor).astype(f32)
    am = acc[:, None]
    # sampled state, then hmc.flip_momentum once more (ghmc.py:188): accepted -> +p1, rejected -> -p
    mom = np.where(am, (f32(-1.0) * end.momentum).astype(f32), (f32(-1.0) * p).astype(f32)).astype(f32)
    new_state = GHMCState(
        np.where(am, end.position, q).astype(f32), mom,
        np.where(acc, end.logdensity, logp).astype(f32),
        np.where(am, end.logdensity_grad, g).astype(f32), sl_next)
    info = ohmc.HMCInfo(p, p_acc, acc, is_div, e1, end, 1)
    return new_state, info
