"""Oracle for Stan-style window adaptation (TEST INFRASTRUCTURE, see package docstring).

Reference lines followed
* dual_averaging init/update/final      blackjax/optimizers/dual_averaging.py:87-127
* dual_averaging_adaptation(target)     blackjax/adaptation/step_size.py:119-148
* welford_algorithm                     blackjax/adaptation/mass_matrix.py:390-442
* mass_matrix_adaptation init/update/final   mass_matrix.py:218-256,288-291,335-357
* _make_engine fast/slow/slow_final/final    blackjax/adaptation/staged_adaptation.py:173-307
* build_schedule                        staged_adaptation.py:315-405
* run / one_step                        staged_adaptation.py:731-754,860-876,968-981
* window_adaptation (validation + delegate)  blackjax/adaptation/window_adaptation.py:296-444

Batched semantics: adaptation is PER CHAIN (one step size, one inverse mass matrix per
chain), i.e. chain ``i`` of ``run(rng_key, positions)`` equals the reference's
``window_adaptation(...).run(jax.random.split(rng_key, N)[i], position_i)`` -- the vmapped
warmup of docs/examples/howto_progress_bar.md:147-189 ("chain-major" key layout: chain key
``c_i = split(rng_key, N)[i]``, step key ``split(c_i, T)[t]``).

Scalar conventions (shared with the HIP kernels): the dual-averaging recursion is evaluated
operation by operation in fp32 WITHOUT fusing (no fma), pow/exp/log are fp64 rounded once;
the Welford m2 update ``m2 + delta*updated_delta`` is one fma; the window-end blend is
``fma(beta_prev, imm_prev, beta_data*cov) + beta_ident*1e-3``.
"""
from __future__ import annotations

from typing import NamedTuple

import numpy as np

from . import hmc as ohmc
from . import prng
from .fp import exp_cr, f32, f64, fma32, log_cr


# ----------------------------------------------------------------------------- dual averaging
class DualAveragingState(NamedTuple):  # dual_averaging.py:24-50 / step_size.py
    log_step_size: np.ndarray  # (N,)
    log_step_size_avg: np.ndarray
    step: int
    avg_error: np.ndarray
    mu: np.ndarray


def da_init(x_init) -> DualAveragingState:  # dual_averaging.py:87-99
    x = np.asarray(x_init, dtype=f32)
    mu = log_cr((f32(10.0) * x).astype(f32))
    return DualAveragingState(log_cr(x), np.zeros_like(x), 1, np.zeros_like(x), mu)


def da_update(state: DualAveragingState, gradient, t0=10, gamma=0.05, kappa=0.75):
    """dual_averaging.py:101-123 (``gradient = target - acceptance_rate``, step_size.py:144)."""
    log_x, log_x_avg, step, avg_error, mu = state
    g = np.asarray(gradient, dtype=f32)
    reg = f32(step + t0)
    eta = f32(np.power(f64(step), f64(-kappa)))  # step ** (-kappa), fp64 rounded once
    inv_reg = f32(1.0) / reg
    avg_error = ((f32(1.0) - inv_reg) * avg_error).astype(f32) + (g / reg).astype(f32)
    avg_error = avg_error.astype(f32)
    coef = f32(np.sqrt(f32(step))) / f32(gamma)
    new_log_x = (mu - (coef * avg_error).astype(f32)).astype(f32)
    new_log_x_avg = ((eta * log_x).astype(f32) + ((f32(1.0) - eta) * log_x_avg).astype(f32)).astype(f32)
    return DualAveragingState(new_log_x, new_log_x_avg, step + 1, avg_error, mu)


def da_final(state: DualAveragingState):  # dual_averaging.py:125-127
    return exp_cr(state.log_step_size_avg)


# ----------------------------------------------------------------------------- Welford
class WelfordState(NamedTuple):  # mass_matrix.py:364-388
    mean: np.ndarray  # (N, D)
    m2: np.ndarray  # (N, D) diag | (N, D, D) dense
    sample_size: int


def welford_init(N, D, is_diag=True) -> WelfordState:  # mass_matrix.py:390-408
    return WelfordState(np.zeros((N, D), f32), np.zeros((N, D) if is_diag else (N, D, D), f32), 0)


def welford_update(state: WelfordState, value, is_diag=True) -> WelfordState:  # mass_matrix.py:410-435
    mean, m2, n = state
    n = n + 1
    value = np.asarray(value, f32)
    delta = (value - mean).astype(f32)
    mean = (mean + (delta / f32(n)).astype(f32)).astype(f32)
    upd = (value - mean).astype(f32)
    if is_diag:
        m2 = fma32(delta, upd, m2)
    else:
        m2 = fma32(upd[:, :, None], delta[:, None, :], m2)  # outer(updated_delta, delta)
    return WelfordState(mean, m2, n)


def welford_final(state: WelfordState):  # mass_matrix.py:437-442
    mean, m2, n = state
    return (m2 / f32(n - 1)).astype(f32), n, mean


class MassMatrixState(NamedTuple):  # mass_matrix.py:33-56
    inverse_mass_matrix: np.ndarray
    wc_state: WelfordState


def mm_final(state: MassMatrixState, is_diag=True, shrinkage=0.0) -> MassMatrixState:
    """mass_matrix.py:335-357."""
    prev, wc = state
    cov, count, mean = welford_final(wc)
    N, D = mean.shape
    denom = f32(f32(count + 5) + f32(shrinkage))
    beta_data = f32(count) / denom
    beta_prev = f32(shrinkage) / denom
    beta_ident = f32(5.0) / denom
    reg = f32(beta_ident * f32(1e-3))
    blend = fma32(beta_prev, prev, (beta_data * cov).astype(f32))
    if is_diag:
        imm = (blend + reg).astype(f32)
    else:
        imm = (blend + (reg * np.eye(D, dtype=f32))).astype(f32)
    return MassMatrixState(imm, welford_init(N, D, is_diag))


# ----------------------------------------------------------------------------- schedule
def build_schedule(num_steps, initial_buffer_size=75, final_buffer_size=50, first_window_size=25):
    """staged_adaptation.py:315-405 -> list of (stage, is_middle_window_end)."""
    schedule = []
    if num_steps < 20:
        schedule += [(0, False)] * num_steps
    else:
        if initial_buffer_size + first_window_size + final_buffer_size > num_steps:
            initial_buffer_size = int(0.15 * num_steps)
            final_buffer_size = int(0.1 * num_steps)
            first_window_size = num_steps - initial_buffer_size - final_buffer_size
        schedule += [(0, False)] * initial_buffer_size
        final_buffer_start = num_steps - final_buffer_size
        next_window_size = first_window_size
        next_window_start = initial_buffer_size
        while next_window_start < final_buffer_start:
            current_start, current_size = next_window_start, next_window_size
            if 3 * current_size <= final_buffer_start - current_start:
                next_window_size = 2 * current_size
            else:
                current_size = final_buffer_start - current_start
            next_window_start = current_start + current_size
            schedule += [(1, False)] * (next_window_start - 1 - current_start)
            schedule.append((1, True))
        schedule += [(0, False)] * (num_steps - final_buffer_start)
    return schedule


# ----------------------------------------------------------------------------- engine
class StagedAdaptationState(NamedTuple):  # staged_adaptation.py:69-103
    ss_state: DualAveragingState
    imm_state: MassMatrixState
    step_size: np.ndarray  # (N,)
    inverse_mass_matrix: np.ndarray  # (N, D) | (N, D, D)


def adapt_init(N, D, initial_step_size, is_diag=True, initial_imm=None) -> StagedAdaptationState:
    """staged_adaptation.py:173-184 + mass_matrix.py:218-256."""
    if initial_imm is None:
        imm = np.ones((N, D), f32) if is_diag else np.tile(np.eye(D, dtype=f32), (N, 1, 1))
    else:
        imm = np.broadcast_to(np.asarray(initial_imm, f32), (N,) + np.shape(initial_imm)).copy()
    eps0 = np.full(N, initial_step_size, f32)
    return StagedAdaptationState(da_init(eps0), MassMatrixState(imm, welford_init(N, D, is_diag)),
                                 eps0, imm)


def adapt_update(ws: StagedAdaptationState, stage, is_window_end, position, acceptance_rate,
                 target=0.8, is_diag=True, shrinkage=0.0) -> StagedAdaptationState:
    """staged_adaptation.py:186-297 (fast_update / slow_update / slow_final dispatch)."""
    grad = (f32(target) - np.asarray(acceptance_rate, f32)).astype(f32)  # step_size.py:144
    imm_state = ws.imm_state
    if stage == 1:
        imm_state = MassMatrixState(imm_state.inverse_mass_matrix,
                                    welford_update(imm_state.wc_state, position, is_diag))
    ss = da_update(ws.ss_state, grad)
    ws = StagedAdaptationState(ss, imm_state, exp_cr(ss.log_step_size), imm_state.inverse_mass_matrix)
    if is_window_end:
        imm_state = mm_final(ws.imm_state, is_diag, shrinkage)
        ss = da_init(da_final(ws.ss_state))
        ws = StagedAdaptationState(ss, imm_state, exp_cr(ss.log_step_size), imm_state.inverse_mass_matrix)
    return ws


def window_adaptation_run(rng_key, position, logdensity_fn, num_steps, num_integration_steps,
                          is_mass_matrix_diagonal=True, initial_step_size=1.0,
                          target_acceptance_rate=0.8, initial_inverse_mass_matrix=None,
                          imm_shrinkage_to_previous=0.0, chain_offset=0, kernel_fn=None,
                          chain_keys_override=None, schedule=None):
    """window_adaptation(hmc, ...).run (window_adaptation.py:296-444 ->
    staged_adaptation.py:860-876,968-981), batched per chain, chain-major keys.
    ``chain_keys_override``: the chain keys ``c_i`` of an arbitrary subset of global chain indices
    (chains are independent, so any subset of a big run can be checked on its own)."""
    N, D = position.shape
    state = ohmc.init(position, logdensity_fn)
    ws = adapt_init(N, D, initial_step_size, is_mass_matrix_diagonal, initial_inverse_mass_matrix)
    chain_keys = (prng.split(rng_key, N, offset=chain_offset) if chain_keys_override is None
                  else np.asarray(chain_keys_override, np.uint32))  # c_i
    if schedule is None:  # staged_adaptation's schedule_fn hook: any list of (stage, is_window_end)
        schedule = build_schedule(num_steps)
    history = []
    for t, (stage, is_end) in enumerate(schedule):
        keys_t = prng.fold_in(chain_keys, np.uint32(t))  # split(c_i, T)[t]
        if kernel_fn is None:
            state, info = ohmc.kernel(None, state, logdensity_fn, ws.step_size,
                                      ws.inverse_mass_matrix, num_integration_steps,
                                      chain_keys_override=keys_t,
                                      per_chain_diag=is_mass_matrix_diagonal)
        else:
            state, info = kernel_fn(keys_t, state, ws.step_size, ws.inverse_mass_matrix)
        ws = adapt_update(ws, stage, is_end, state.position, info.acceptance_rate,
                          target_acceptance_rate, is_mass_matrix_diagonal, imm_shrinkage_to_previous)
        history.append((info.acceptance_rate.copy(), ws.step_size.copy()))
    step_size = da_final(ws.ss_state)  # staged_adaptation.py:301-305
    return state, {"step_size": step_size, "inverse_mass_matrix": ws.imm_state.inverse_mass_matrix}, history
