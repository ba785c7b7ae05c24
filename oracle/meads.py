"""Oracle for MEADS, the cross-chain warm-up of Generalized HMC (TEST INFRASTRUCTURE).

Reference lines followed (diagonal momentum metric, ``low_rank_rank=None``)
* MEADSAdaptationState, base()          blackjax/adaptation/meads_adaptation.py:31-212
* meads_adaptation: one_step            blackjax/adaptation/meads_adaptation.py:511-722 (low-rank branches excluded)
* meads_adaptation: run                 blackjax/adaptation/meads_adaptation.py:724-787
* maximum_eigenvalue                    blackjax/adaptation/meads_adaptation.py:790-817

Numerics contract (as everywhere in this repo): reductions over chains are accumulated in fp64 and
rounded once to fp32 -- per-fold standard deviations, Gram traces -- the scalar heuristics that
follow are fp32 operation by operation (transcendentals in fp64, rounded once).  XLA's own fp32
reduction order is unspecified, so this part is compared with a tolerance (parity unpinned at the
bit level, like the ChEES ensemble means).
"""
from __future__ import annotations

from typing import NamedTuple

import numpy as np

from . import ghmc as oghmc
from . import prng
from .fp import exp_cr, f32, f64, sqrt32


class MEADSAdaptationState(NamedTuple):  # meads_adaptation.py:31-52
    current_iteration: int
    step_size: np.ndarray  # (K,)
    position_sigma: np.ndarray  # (K, D)
    alpha: np.ndarray  # (K,)
    delta: np.ndarray  # (K,)


def maximum_eigenvalue(x) -> np.float32:
    """meads_adaptation.py:790-817 for an (n, d) batch: with S = X X^T,
    (sum S^2 - sum diag(S)^2) / (n (n - 1)) / (sum diag(S) / n).  ||X X^T||_F = ||X^T X||_F, so the
    smaller Gram matrix is formed; fp64 accumulation, one rounding."""
    x = np.asarray(x, dtype=f32).astype(f64)
    n, d = x.shape
    gram = x @ x.T if n <= d else x.T @ x
    row_sq = np.sum(x * x, axis=1)
    lam = np.sum(row_sq) / n
    with np.errstate(divide="ignore", invalid="ignore"):
        lam_sq = (np.sum(gram * gram) - np.sum(row_sq * row_sq)) / (n * (n - 1))
        return f32(lam_sq / lam)


def fold_std(x) -> np.ndarray:
    """Population standard deviation over the chain axis (jnp.std, ddof = 0), fp64, one rounding."""
    return np.std(np.asarray(x, dtype=f32).astype(f64), axis=-2).astype(f32)


def step_size_of(precond_grads, multiplier) -> np.float32:
    """Algorithm 3 line 8 (meads_adaptation.py:583-588): min(multiplier / sqrt(lambda_max), 1)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.minimum((f32(multiplier) / sqrt32(maximum_eigenvalue(precond_grads))).astype(f32), f32(1.0))


def damping_of(precond_pos, eps, t, damping_slowdown):
    """Algorithm 3 lines 9-10 (meads_adaptation.py:604-614): positions centred within the fold."""
    x = np.asarray(precond_pos, dtype=f32)
    centred = (x - x.astype(f64).mean(axis=0).astype(f32)).astype(f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        g1 = (f32(1.0) / sqrt32(maximum_eigenvalue(centred))).astype(f32)
        g2 = (f32(damping_slowdown) / (f32(t + 1) * f32(eps)).astype(f32)).astype(f32)
        gamma = np.maximum(g1, g2)
        alpha = (f32(1.0) - exp_cr(((f32(-2.0) * f32(eps)).astype(f32) * gamma).astype(f32))).astype(f32)
    return alpha, (alpha / f32(2.0)).astype(f32)


def base_init(positions, grads, num_folds=4, step_size_multiplier=0.5, damping_slowdown=1.0):
    """base().init (meads_adaptation.py:148-166): parameters from ALL chains, replicated per fold."""
    positions = np.asarray(positions, f32)
    sd = fold_std(positions)
    mean = positions.astype(f64).mean(axis=0).astype(f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        normalized = ((positions - mean).astype(f32) / sd).astype(f32)
    eps = step_size_of((np.asarray(grads, f32) * sd).astype(f32), step_size_multiplier)
    with np.errstate(divide="ignore", invalid="ignore"):
        g1 = (f32(1.0) / sqrt32(maximum_eigenvalue(normalized))).astype(f32)
        g2 = (f32(damping_slowdown) / (f32(1) * eps).astype(f32)).astype(f32)
        gamma = np.maximum(g1, g2)
        alpha = (f32(1.0) - exp_cr(((f32(-2.0) * eps).astype(f32) * gamma).astype(f32))).astype(f32)
    delta = (alpha / f32(2.0)).astype(f32)
    K = num_folds
    return MEADSAdaptationState(0, np.full(K, eps, f32), np.repeat(sd[None], K, axis=0), np.full(K, alpha, f32),
                                np.full(K, delta, f32))


def one_step(rng_key, states: oghmc.GHMCState, ad: MEADSAdaptationState, logdensity_fn, num_folds,
             step_size_multiplier=0.5, damping_slowdown=1.0):
    """meads_adaptation.py:511-722 (diagonal branch).  Returns (states, adaptation state, HMCInfo)."""
    K = num_folds
    N, D = states.position.shape
    n = N // K
    t = ad.current_iteration
    keys = prng.split(rng_key, N + 1)
    chain_keys, shuffle_key = keys[:N], keys[N]
    pos = states.position.reshape(K, n, D)
    grads = states.logdensity_grad.reshape(K, n, D)
    scales = np.stack([fold_std(pos[k]) for k in range(K)])  # (K, D)
    eps_own = np.array([step_size_of((grads[k] * scales[k]).astype(f32), step_size_multiplier) for k in range(K)], f32)
    eps_rolled = np.roll(eps_own, 1)
    scales_rolled = np.roll(scales, 1, axis=0)
    alphas, deltas = np.empty(K, f32), np.empty(K, f32)
    for k in range(K):
        with np.errstate(divide="ignore", invalid="ignore"):
            pp = (pos[k] / scales[k]).astype(f32)
        alphas[k], deltas[k] = damping_of(pp, eps_rolled[k], t, damping_slowdown)
    new_states, info = oghmc.kernel(None, states, logdensity_fn, np.repeat(eps_rolled, n),
                                    np.repeat(scales_rolled, n, axis=0), np.repeat(alphas, n),
                                    np.repeat(deltas, n), chain_keys_override=chain_keys)
    if K > 1:  # the fold t mod K does not advance (Algorithm 3 line 4)
        skipped = np.repeat(np.arange(K) == t % K, n)
        new_states = oghmc.GHMCState(*[
            np.where(skipped.reshape((N,) + (1,) * (a.ndim - 1)), b, a) for a, b in zip(new_states, states)])
    new_ad = MEADSAdaptationState(t + 1, eps_rolled, scales_rolled, alphas, deltas)
    if K > 1 and (t + 1) % K == 0:
        perm = prng.permutation(shuffle_key, N)
        new_states = oghmc.GHMCState(*[a[perm] for a in new_states])
    return new_states, new_ad, info


def run(rng_key, positions, logdensity_fn, num_steps, num_folds=4, step_size_multiplier=0.5,
        damping_slowdown=1.0):
    """meads_adaptation.py:724-787: returns (last states, parameters, history of (states, adaptation))."""
    positions = np.asarray(positions, f32)
    N = positions.shape[0]
    kk = prng.split(rng_key, 2)
    key_init, key_adapt = kk[0], kk[1]
    states = oghmc.init(positions, logdensity_fn, key_init)
    ad = base_init(positions, states.logdensity_grad, num_folds, step_size_multiplier, damping_slowdown)
    hist = []
    for key in prng.split(key_adapt, num_steps):
        states, ad, info = one_step(key, states, ad, logdensity_fn, num_folds, step_size_multiplier,
                                    damping_slowdown)
        hist.append((states, ad, info))
    params = {
        "step_size": f32(ad.step_size.astype(f64).mean()),
        "momentum_inverse_scale": ad.position_sigma.astype(f64).mean(axis=0).astype(f32),
        "alpha": f32(ad.alpha.astype(f64).mean()),
        "delta": f32(ad.delta.astype(f64).mean()),
    }
    return states, params, hist
