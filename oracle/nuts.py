"""Oracle for the NUTS transition (TEST INFRASTRUCTURE, see package docstring).

A deliberately plain per-chain restatement (Python loops over chains, doublings and leapfrog
steps; NumPy over the D axis) of

* nuts.build_kernel.kernel / iterative_nuts_proposal     blackjax/mcmc/nuts.py:113-145,223-321
* dynamic_multiplicative_expansion                      blackjax/mcmc/trajectory.py:580-727
* dynamic_progressive_integration                       trajectory.py:242-395
* append / reorder / merge trajectories                 trajectory.py:62-125
* proposal_generator, progressive_uniform/biased        blackjax/mcmc/proposal.py:51-105,118-176
* iterative_uturn_numpyro                               blackjax/mcmc/termination.py:31-106
* gaussian_euclidean.is_turning                         blackjax/mcmc/metrics.py:272-304

Chain ``i`` of the batched kernel equals the reference's single-chain kernel called with
``jax.random.split(rng_key, N)[i]`` (or the override keys).  Only for small cases.
"""
from __future__ import annotations

from typing import NamedTuple

import numpy as np

from . import hmc as ohmc
from . import prng
from .fp import dot64, exp_cr, expit_cr, f32, logaddexp_cr

NEG_INF = f32(-np.inf)


class NUTSInfo(NamedTuple):  # nuts.py:36-74
    momentum: np.ndarray
    is_divergent: np.ndarray
    is_turning: np.ndarray
    energy: np.ndarray
    trajectory_leftmost_state: ohmc.IntegratorState
    trajectory_rightmost_state: ohmc.IntegratorState
    num_trajectory_expansions: np.ndarray
    num_integration_steps: np.ndarray
    acceptance_rate: np.ndarray


def is_turning(metric, p_left, p_right, p_sum):
    """metrics.py:297-304 (single chain rows of shape (1, D))."""
    v_left = ohmc.linear_map(metric, metric.inverse_mass_matrix, p_left)
    v_right = ohmc.linear_map(metric, metric.inverse_mass_matrix, p_right)
    rho = (p_sum - ((p_right + p_left).astype(f32) / f32(2.0)).astype(f32)).astype(f32)
    return bool((dot64(v_left, rho) <= 0)[0] or (dot64(v_right, rho) <= 0)[0])


def leaf_idx_to_ckpt_idxs(n: int):
    """termination.py:75-84."""
    idx_max = bin(n >> 1).count("1")
    num_subtrees = bin(((~n) & (n + 1)) - 1).count("1")
    return idx_max - num_subtrees + 1, idx_max


def is_iterative_turning(metric, r_ckpts, r_sum_ckpts, idx_min, idx_max, r_sum, r):
    """termination.py:86-104 (explicit idx_min / idx_max as in tests/mcmc/test_uturn.py)."""
    i, turning = idx_max, False
    while i >= idx_min and not turning:
        subtree_r_sum = ((r_sum - r_sum_ckpts[i]).astype(f32) + r_ckpts[i]).astype(f32)
        turning = is_turning(metric, r_ckpts[i], r, subtree_r_sum)
        i -= 1
    return turning


def _chain_metric(metric: ohmc.Metric, i: int) -> ohmc.Metric:
    if metric.is_dense and metric.inverse_mass_matrix.ndim == 3:
        return ohmc.Metric(metric.inverse_mass_matrix[i], metric.mass_matrix_sqrt[i], True)
    if metric.is_dense or metric.inverse_mass_matrix.ndim == 1:
        return metric
    return ohmc.Metric(metric.inverse_mass_matrix[i:i + 1], metric.mass_matrix_sqrt[i:i + 1], False)


def _one_chain(key_integrator, z0, logdensity_fn, eps, metric, max_depth, thr, coefficients=None):
    """iterative_nuts_proposal.propose for ONE chain (arrays of shape (1, D) / (1,))."""
    D = z0.position.shape[1]
    ckpt_r = np.zeros((max_depth, 1, D), f32)  # termination.py:46-54
    ckpt_rs = np.zeros((max_depth, 1, D), f32)
    H0 = ohmc.hmc_energy(metric, z0)  # nuts.py:282
    prop_state, prop_energy, prop_w, prop_slpa = z0, H0, f32(0.0), NEG_INF  # nuts.py:283-285
    left = right = z0
    msum = z0.momentum.copy()
    n_states = 0
    depth, div, turn = 0, False, False
    while depth < max_depth and not div and not turn:  # trajectory.py:622-629
        subkey = prng.fold_in(key_integrator, depth)  # trajectory.py:645
        kd, kt, kp = prng.split(subkey, 3)
        direction = 1 if bool(prng.uniform(kd, ()) < f32(0.5)) else -1  # trajectory.py:650
        zr = right if direction > 0 else left
        deps = f32(direction) * f32(eps)
        # ---- dynamic_progressive_integration (trajectory.py:273-393)
        s, sdiv, sturn = 0, False, False
        sub_first = None
        while s < 2 ** depth and not sturn and not sdiv:
            znew = ohmc.integrator_step(zr, deps, logdensity_fn, metric, coefficients)  # trajectory.py:323
            e_new = ohmc.hmc_energy(metric, znew)
            w = ohmc.safe_energy_diff(H0, e_new)[0]  # proposal.py:91-95
            new_slpa = np.minimum(w, f32(0.0))
            sdiv = bool(-w > f32(thr))
            if s == 0:  # trajectory.py:329-334
                sub_first, sub_msum, sub_n = znew, znew.momentum.copy(), 1
                sp_state, sp_energy, sp_w, sp_slpa = znew, e_new, w, new_slpa
            else:
                sub_msum = (sub_msum + znew.momentum).astype(f32)  # append_to_trajectory
                sub_n += 1
                with np.errstate(invalid="ignore"):
                    pa = expit_cr(f32(w - sp_w))  # progressive_uniform_sampling, proposal.py:118-143
                acc = bool(prng.uniform(prng.fold_in(kt, s), ()) < pa)
                W = logaddexp_cr(sp_w, w)
                S = logaddexp_cr(sp_slpa, new_slpa)
                if acc:
                    sp_state, sp_energy = znew, e_new
                sp_w, sp_slpa = W, S
            idx_min, idx_max = leaf_idx_to_ckpt_idxs(s)  # termination.py:56-73
            if s % 2 == 0:
                ckpt_r[idx_max] = znew.momentum
                ckpt_rs[idx_max] = sub_msum
            sturn = is_iterative_turning(metric, ckpt_r, ckpt_rs, idx_min, idx_max, sub_msum,
                                         znew.momentum)
            zr = znew
            s += 1
        sub_left, sub_right = (sub_first, zr) if direction > 0 else (zr, sub_first)  # 376-385
        # ---- merge (trajectory.py:678-715)
        if sdiv or sturn:
            prop_slpa = logaddexp_cr(prop_slpa, sp_slpa)
        else:
            with np.errstate(invalid="ignore", over="ignore"):
                pa = np.minimum(exp_cr(f32(sp_w - prop_w)), f32(1.0))  # progressive_biased_sampling
            acc = bool(prng.uniform(kp, ()) < pa)
            W = logaddexp_cr(prop_w, sp_w)
            S = logaddexp_cr(prop_slpa, sp_slpa)
            if acc:
                prop_state, prop_energy = sp_state, sp_energy
            prop_w, prop_slpa = W, S
        if direction > 0:
            right = sub_right
            msum = (msum + sub_msum).astype(f32)
        else:
            left = sub_left
            msum = (sub_msum + msum).astype(f32)
        n_states += sub_n
        turn = sturn or is_turning(metric, left.momentum, right.momentum, msum)
        div = sdiv
        depth += 1
    acc_rate = (exp_cr(prop_slpa) / f32(n_states)).astype(f32)  # nuts.py:303-305
    return prop_state, prop_energy, left, right, depth, n_states, acc_rate, div, turn


def kernel(rng_key, state: ohmc.HMCState, logdensity_fn, step_size, inverse_mass_matrix,
           max_num_doublings: int = 10, divergence_threshold: float = 1000.0,
           chain_offset: int = 0, chain_keys_override=None, per_chain_diag=False, coefficients=None,
           metric=None):
    """nuts.py:113-145, batched by looping over chains.  ``coefficients``: palindromic integrator
    (nuts.py:150-158 ``integrator=``; None = velocity Verlet)."""
    N, D = state.position.shape
    if metric is None:  # ``metric``: a prepared Metric, e.g. default_metric(..., dense_accum="f32chain")
        metric = ohmc.default_metric(inverse_mass_matrix, n_chains=N, per_chain_diag=per_chain_diag)
    keys = (ohmc.chain_keys(rng_key, N, chain_offset) if chain_keys_override is None
            else chain_keys_override)
    kk = prng.split(keys, 2)  # nuts.py:133
    p0 = ohmc.sample_momentum(metric, kk[:, 0], D)  # nuts.py:136
    eps = np.broadcast_to(np.asarray(step_size, f32), (N,))
    outs = []
    for i in range(N):
        z0 = ohmc.IntegratorState(state.position[i:i + 1], p0[i:i + 1], state.logdensity[i:i + 1],
                                  state.logdensity_grad[i:i + 1])
        outs.append(_one_chain(kk[i, 1], z0, logdensity_fn, eps[i], _chain_metric(metric, i),
                               max_num_doublings, divergence_threshold, coefficients))

    def cat_state(idx):
        return ohmc.IntegratorState(*[np.concatenate([getattr(o[idx], f) for o in outs], 0)
                                      for f in ohmc.IntegratorState._fields])

    prop = cat_state(0)
    new_state = ohmc.HMCState(prop.position, prop.logdensity, prop.logdensity_grad)
    info = NUTSInfo(
        p0,
        np.array([o[7] for o in outs]),
        np.array([o[8] for o in outs]),
        np.concatenate([o[1] for o in outs]).astype(f32),
        cat_state(2), cat_state(3),
        np.array([o[4] for o in outs], np.int32),
        np.array([o[5] for o in outs], np.int32),
        np.concatenate([np.atleast_1d(o[6]) for o in outs]).astype(f32),
    )
    return new_state, info
