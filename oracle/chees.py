"""Oracle for ChEES-HMC pooled (cross-chain) adaptation (TEST INFRASTRUCTURE, see package docstring).

Reference lines followed
* ChEESAdaptationState / base.init / base.update(compute_parameters)
                                         blackjax/adaptation/chees_adaptation.py:28-58, 250-571
* weighted_empirical_mean                chees_adaptation.py:239-247
* chees_adaptation.run / one_step        chees_adaptation.py:737-1025
* diagonal mass-matrix gate              chees_adaptation.py:61-90
* length floor (eig state, power iteration, floor arithmetic)   chees_adaptation.py:112-236
* cgl_update_batch                       blackjax/adaptation/metric_buffers.py:396-451
* halton_sequence                        blackjax/mcmc/dynamic_hmc.py:205-215
* dual averaging                         blackjax/optimizers/dual_averaging.py:87-127 (oracle.adaptation)

Third-party arithmetic absent from /root/reference: ``optax`` (pinned 0.2.8, uv.lock:2197-2198).
``adam``/``sgd`` below restate optax's published update rules (scale_by_adam: first/second moment
EMA with bias correction ``m/(1-b^t)``, update ``-lr * m_hat / (sqrt(v_hat + eps_root) + eps)``;
sgd: ``-lr * g``).  **Parity unpinned** for optax bit patterns (no optax install reachable); the
statistical pins of tests/adaptation/test_adaptation.py:77-152 anchor the behaviour.

Numerics contract (shared with the HIP kernels): every cross-chain or cross-dimension sum is
accumulated in fp64 over exact products of fp32 operands and rounded once; elementwise arithmetic
is fp32 op by op (no fma); scalar pow/exp/log are fp64 rounded once.  The reference folds the
ensemble into its diagonal Welford accumulator row by row (chees_adaptation.py:816-824); the
engine and this oracle use the batch (CGL) merge of the same statistics, equal up to rounding
(``welford_fold_rows`` is the literal row-by-row fold, kept to pin that equivalence).
"""
from __future__ import annotations

from typing import Callable, NamedTuple

import numpy as np

from . import adaptation as oad
from . import hmc as ohmc
from . import prng
from .fp import dot64, exp_cr, f32, f64, log_cr

OPTIMAL_TARGET_ACCEPTANCE_RATE = 0.651  # chees_adaptation.py:21
LOG_UPDATE_CLIP = 0.35  # :23
EPS_FLOAT = 1e-20  # :25
CHEES_LENGTH_FLOOR_FACTOR = np.pi / 2  # :112
LENGTH_FLOOR_RECOMPUTE_INTERVAL = 32  # :120
LENGTH_FLOOR_POWER_ITERATIONS = 5  # :121
LENGTH_FLOOR_FINAL_POWER_ITERATIONS = 20  # :122
LENGTH_FLOOR_LAMBDA_EPS = 1e-6  # :127


# ----------------------------------------------------------------------------- optimizers (optax)
class AdamState(NamedTuple):
    count: int
    mu: np.float32
    nu: np.float32


class Adam:
    """optax.adam(learning_rate, b1, b2, eps, eps_root) on one scalar parameter."""

    def __init__(self, learning_rate, b1=0.9, b2=0.999, eps=1e-8, eps_root=0.0):
        self.lr, self.b1, self.b2, self.eps, self.eps_root = (float(learning_rate), float(b1), float(b2),
                                                              float(eps), float(eps_root))

    def init(self, params):
        return AdamState(0, f32(0.0), f32(0.0))

    def update(self, grad, state, params=None):
        g = f32(grad)
        b1, b2 = f32(self.b1), f32(self.b2)
        mu = f32(f32(f32(1.0 - self.b1) * g) + f32(b1 * state.mu))
        nu = f32(f32(f32(1.0 - self.b2) * f32(g * g)) + f32(b2 * state.nu))
        count = state.count + 1
        c1 = f32(f32(1.0) - f32(np.power(f64(b1), f64(count))))
        c2 = f32(f32(1.0) - f32(np.power(f64(b2), f64(count))))
        with np.errstate(divide="ignore", invalid="ignore"):
            mu_hat = f32(mu / c1)
            nu_hat = f32(nu / c2)
            u = f32(mu_hat / f32(f32(np.sqrt(f32(nu_hat + f32(self.eps_root)))) + f32(self.eps)))
        return f32(f32(-self.lr) * u), AdamState(count, mu, nu)


class SGD:
    """optax.sgd(learning_rate)."""

    def __init__(self, learning_rate):
        self.lr = float(learning_rate)

    def init(self, params):
        return ()

    def update(self, grad, state, params=None):
        return f32(f32(-self.lr) * f32(grad)), state


# ----------------------------------------------------------------------------- Halton
def halton_sequence(i: int, max_bits: int = 10) -> np.float32:
    """dynamic_hmc.py:205-215: radical inverse (base 2) of ``i + 1`` over ``max_bits`` bits.
    Exact in fp32 for max_bits <= 24."""
    max_bits = int(max_bits)
    if max_bits >= 32:
        raise ValueError(f"max_bits ({max_bits}) must be less than bit width of dtype int32 (32)")
    v = f32(0.0)
    for k in range(max_bits):
        bit = ((int(i) + 1) >> k) & 1
        v = f32(v + f32(bit) * f32(0.5 / (1 << k)))
    return v


# ----------------------------------------------------------------------------- ChEES state/update
class ChEESAdaptationState(NamedTuple):  # chees_adaptation.py:28-58
    step_size: np.float32
    log_step_size_moving_average: np.float32
    trajectory_length: np.float32
    log_trajectory_length_moving_average: np.float32
    da_state: oad.DualAveragingState
    optim_state: tuple
    random_generator_arg: int
    step: int


def weighted_empirical_mean(x, w):
    """chees_adaptation.py:239-247."""
    finite = np.isfinite(x)
    x_safe = np.where(finite, x, f32(0.0)).astype(f32)
    w = np.where(finite.all(axis=-1), w, f32(0.0)).astype(f32)
    num = (w.astype(f64)[:, None] * x_safe.astype(f64)).sum(axis=0).astype(f32)
    den = f32(f32(w.astype(f64).sum()) + f32(EPS_FLOAT))
    with np.errstate(divide="ignore", invalid="ignore"):
        return (num / den).astype(f32)


def nanmean0(x):
    """jnp.nanmean(x, axis=0)."""
    ok = ~np.isnan(x)
    s = np.where(ok, x, 0.0).astype(f64).sum(axis=0).astype(f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        return (s / ok.sum(axis=0).astype(f32)).astype(f32)


def chain_criterion(proposed_positions, proposed_momentums, initial_positions, w, imm, whiten=True):
    """chees_adaptation.py:376-466 up to the per-chain factor
    ``(|dx'|^2 - |dx|^2) * <dx', v'>`` (whitened by the diagonal metric when ``whiten``)."""
    pm = weighted_empirical_mean(proposed_positions, w)
    im = nanmean0(initial_positions)
    return criterion_given_means(proposed_positions, proposed_momentums, initial_positions, pm, im, imm,
                                 whiten)


def criterion_given_means(proposed_positions, proposed_momentums, initial_positions, pm, im, imm,
                          whiten=True):
    """chees_adaptation.py:387-466 given the two ensemble means (which, under chain sharding, come
    from all-reduced sums)."""
    with np.errstate(invalid="ignore", over="ignore"):
        pc = (proposed_positions - pm).astype(f32)
        ic = (initial_positions - im).astype(f32)
        if whiten:
            inv_sqrt = (f32(1.0) / np.sqrt(np.asarray(imm, f32))).astype(f32)
            pc_w = (pc * inv_sqrt).astype(f32)
            ic_w = (ic * inv_sqrt).astype(f32)
            vel_w = ((proposed_momentums * np.asarray(imm, f32)).astype(f32) * inv_sqrt).astype(f32)
        else:
            pc_w, ic_w, vel_w = pc, ic, np.asarray(proposed_momentums, f32)
        diff = (dot64(pc_w, pc_w) - dot64(ic_w, ic_w)).astype(f32)
        return (diff * dot64(pc_w, vel_w)).astype(f32)


def base(jitter_generator: Callable, next_random_arg_fn: Callable, optim, target_acceptance_rate: float,
         decay_rate: float, max_leapfrog_steps: int, whiten_criterion: bool = True):
    """chees_adaptation.py:250-571."""

    def init(random_generator_arg, step_size):  # :513-523
        s = f32(step_size)
        return ChEESAdaptationState(s, f32(0.0), s, f32(0.0), oad.da_init(s), optim.init(s),
                                    random_generator_arg, 1)

    def update(state: ChEESAdaptationState, proposed_positions, proposed_momentums, initial_positions,
               acceptance_probabilities, is_divergent, inverse_mass_matrix):  # :307-511
        (step_size, log_ss_ma, traj_len, log_tl_ma, da_state, optim_state, rga, step) = state
        acc = np.asarray(acceptance_probabilities, f32)
        div = np.asarray(is_divergent, bool)
        nd = ~div
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            inv_acc = (f32(1.0) / acc).astype(f32)
            mean_inv = f32(f32(inv_acc[nd].astype(f64).sum()) / f32(nd.sum()))
            hm = f32(f32(1.0) / mean_inv)  # :358-360
        hm = hm if np.isfinite(hm) else f32(0.0)  # :362
        da_new = oad.da_update(da_state, f32(f32(target_acceptance_rate) - hm))  # :363
        ss_new = f32(exp_cr(da_new.log_step_size))  # :364
        if np.isfinite(ss_new):  # :365-370
            new_step_size, new_da, new_log_ss = ss_new, da_new, f32(da_new.log_step_size)
        else:
            new_step_size, new_da, new_log_ss = step_size, da_state, f32(da_state.log_step_size)
        uw = f32(np.power(f64(step), f64(-decay_rate)))  # :371
        new_log_ss_ma = f32(f32(f32(f32(1.0) - uw) * log_ss_ma) + f32(uw * new_log_ss))  # :372-374

        w = np.where(nd, acc, f32(0.0)).astype(f32)  # :376
        per_chain = chain_criterion(proposed_positions, proposed_momentums, initial_positions, w,
                                    inverse_mass_matrix, whiten_criterion)
        scale = f32(f32(jitter_generator(rga)) * traj_len)  # :461-462
        with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
            tg = (scale * per_chain).astype(f32)
            num = f32((acc[nd].astype(f64) * tg[nd].astype(f64)).sum())
            den = f32((acc[nd] + f32(EPS_FLOAT)).astype(f32).astype(f64).sum())
            grad = f32(num / den)  # :468-471
        log_tl = f32(log_cr(traj_len))  # :473
        upd, optim_new = optim.update(grad, optim_state, log_tl)  # :474-476
        upd = f32(np.clip(upd, f32(-LOG_UPDATE_CLIP), f32(LOG_UPDATE_CLIP))) if not np.isnan(upd) else upd
        log_tl_new = f32(log_tl + upd)  # :481
        if not np.isfinite(log_tl_new):  # :482-489
            log_tl_new, optim_new = log_tl, optim_state
        new_log_tl_ma = f32(f32(f32(f32(1.0) - uw) * log_tl_ma) + f32(uw * log_tl_new))  # :490-492
        new_tl = f32(exp_cr(new_log_tl_ma))
        new_tl = f32(np.clip(new_tl, new_step_size, f32(f32(max_leapfrog_steps) * new_step_size)))  # :497-501
        return ChEESAdaptationState(new_step_size, new_log_ss_ma, new_tl, new_log_tl_ma, new_da, optim_new,
                                    next_random_arg_fn(rga), step + 1)

    return init, update


# ----------------------------------------------------------------------------- pooled moments
class MomentBlock(NamedTuple):  # metric_buffers.py:171-215
    count: np.float32
    mean: np.ndarray  # (D,)
    m2: np.ndarray  # (D,) diagonal | (D, D) dense


def cgl_update_batch(block: MomentBlock, batch) -> MomentBlock:
    """metric_buffers.py:396-451."""
    batch = np.asarray(batch, f32)
    n_a, mean_a, m2_a = block
    n_b = f32(batch.shape[0])
    mean_b = (batch.astype(f64).sum(axis=0) / f64(batch.shape[0])).astype(f32)
    centered = (batch - mean_b).astype(f32)
    if m2_a.ndim == 1:
        m2_b = (centered.astype(f64) ** 2).sum(axis=0).astype(f32)
    else:
        m2_b = (centered.astype(f64).T @ centered.astype(f64)).astype(f32)
    n_ab = f32(n_a + n_b)
    delta = (mean_b - mean_a).astype(f32)
    mean_ab = (mean_a + (delta * f32(n_b / n_ab)).astype(f32)).astype(f32)
    coef = f32(f32(n_a * n_b) / n_ab)
    if m2_a.ndim == 1:
        cross = ((delta * delta).astype(f32) * coef).astype(f32)
    else:
        cross = (np.outer(delta, delta).astype(f32) * coef).astype(f32)
    m2_ab = ((m2_a + m2_b).astype(f32) + cross).astype(f32)
    return MomentBlock(n_ab, mean_ab, m2_ab)


def welford_fold_rows(mean, m2, n, batch):
    """The reference's literal pooling (chees_adaptation.py:816-824): mass_matrix.py:410-435
    applied to one row at a time."""
    mean, m2 = np.array(mean, f32), np.array(m2, f32)
    for x in np.asarray(batch, f32):
        n += 1
        delta = (x - mean).astype(f32)
        mean = (mean + (delta / f32(n)).astype(f32)).astype(f32)
        m2 = (m2 + (delta * (x - mean).astype(f32)).astype(f32)).astype(f32)
    return mean, m2, n


def mass_matrix_engagement_threshold(num_dim: int) -> int:  # chees_adaptation.py:61-72
    return max(64, int(2 * np.sqrt(num_dim)))


def diagonal_mass_matrix_or_fallback(block: MomentBlock, threshold: int, num_dim: int):  # :75-90
    if block.count >= threshold:
        return np.maximum((block.m2 / f32(block.count - f32(1.0))).astype(f32), f32(EPS_FLOAT))
    return np.ones(num_dim, f32)


def power_iteration_lambda_max(matrix, v0, num_iterations):  # :147-166
    v = np.asarray(v0, f32)
    m64 = matrix.astype(f64)
    for _ in range(num_iterations):
        v_next = (m64 @ v.astype(f64)).astype(f32)
        norm = f32(np.sqrt(f32((v_next.astype(f64) ** 2).sum())))
        v = (v_next / (norm if norm > 0 else f32(1.0))).astype(f32)
    mv = (m64 @ v.astype(f64)).astype(f32)
    return f32((v.astype(f64) * mv.astype(f64)).sum()), v


def recompute_eig_state(cov_block: MomentBlock, imm, eigenvector,
                        num_iterations=LENGTH_FLOOR_POWER_ITERATIONS):  # :169-189
    cov = (cov_block.m2 / f32(max(cov_block.count - f32(1.0), f32(1.0)))).astype(f32)
    inv_sqrt_d = (f32(1.0) / np.sqrt(np.asarray(imm, f32))).astype(f32)
    whitened = ((cov * inv_sqrt_d[:, None]).astype(f32) * inv_sqrt_d[None, :]).astype(f32)
    lam, vec = power_iteration_lambda_max(whitened, eigenvector, num_iterations)
    return vec, f32(max(lam, f32(LENGTH_FLOOR_LAMBDA_EPS)))


def apply_length_floor(trajectory_length, lambda_max, engaged, enable, max_leapfrog_steps=1000,
                       step_size=0.1):  # :192-236
    if not enable:
        return f32(trajectory_length), False
    floor_value = f32(f32(CHEES_LENGTH_FLOOR_FACTOR) * f32(np.sqrt(f32(lambda_max)))) if engaged else f32(0.0)
    cap = f32(f32(max_leapfrog_steps) * f32(step_size))
    consumed = f32(min(max(f32(trajectory_length), floor_value), cap))
    return consumed, bool(engaged and floor_value > cap)


# ----------------------------------------------------------------------------- run
def integration_steps(jitter: np.float32, num_leapfrog_steps: np.float32) -> int:  # :767-771
    return int(np.ceil(f32(f32(jitter) * f32(num_leapfrog_steps))))


def run(logdensity_fn, rng_key, positions, step_size, optim, num_steps=1000, *, num_chains=None,
        jitter_generator=None, jitter_amount=1.0, target_acceptance_rate=OPTIMAL_TARGET_ACCEPTANCE_RATE,
        decay_rate=0.5, max_leapfrog_steps=1000, max_sampling_steps=1000, mass_matrix_estimation=None,
        mass_matrix_window_fraction=0.5, whiten_criterion=True, length_floor=True, chain_offset=0,
        record=None):
    """chees_adaptation(...).run (chees_adaptation.py:737-1025).  Returns
    ``(last_state, random_generator_arg, parameters, history)``; ``record(t, state, info, adapt)``
    is called after every step when given."""
    if mass_matrix_estimation not in (None, "diagonal"):
        raise ValueError(f"mass_matrix_estimation must be None or 'diagonal', got {mass_matrix_estimation!r}.")
    if not 0.0 <= mass_matrix_window_fraction <= 1.0:
        raise ValueError(f"mass_matrix_window_fraction must be in [0.0, 1.0], got {mass_matrix_window_fraction}.")
    positions = np.asarray(positions, f32)
    N, D = positions.shape
    estimate_mm = mass_matrix_estimation == "diagonal"
    enable_floor = estimate_mm and length_floor
    rng_key = np.asarray(rng_key, np.uint32)
    ja, jb = f32(jitter_amount), f32(1.0 - jitter_amount)
    if jitter_generator is not None:  # :756-760
        rng_key, carry_key = prng.split(rng_key, 2)
        jitter_gn = lambda i: f32(f32(f32(jitter_generator(prng.fold_in(carry_key, np.uint32(i)))) * ja) + jb)
    else:  # :761-765
        max_bits = int(np.ceil(np.log2(num_steps + max_sampling_steps)))
        jitter_gn = lambda i: f32(f32(halton_sequence(i, max_bits) * ja) + jb)
    init, update = base(jitter_gn, lambda i: i + 1, optim, target_acceptance_rate, decay_rate,
                        max_leapfrog_steps, whiten_criterion)
    window_start = int(mass_matrix_window_fraction * num_steps) if estimate_mm else num_steps
    threshold = mass_matrix_engagement_threshold(D)

    state = ohmc.init(positions, logdensity_fn)
    adapt = init(0, step_size)
    mm = MomentBlock(f32(0.0), np.zeros(D, f32), np.zeros(D, f32))
    cov = MomentBlock(f32(0.0), np.zeros(D, f32), np.zeros((D, D), f32))
    eigvec, lambda_max = (np.ones(D, f32) / f32(np.sqrt(f32(D)))).astype(f32), f32(1.0)
    keys_step = prng.split(rng_key, num_steps)
    history = []
    for t in range(num_steps):
        current_imm = diagonal_mass_matrix_or_fallback(mm, threshold, D) if estimate_mm else np.ones(D, f32)
        consumed = adapt.trajectory_length
        if enable_floor:
            consumed, _ = apply_length_floor(adapt.trajectory_length, lambda_max, mm.count >= threshold,
                                             True, max_leapfrog_steps, adapt.step_size)
        L = integration_steps(jitter_gn(adapt.random_generator_arg), f32(consumed / adapt.step_size))
        new_state, info = ohmc.kernel(keys_step[t], state, logdensity_fn, adapt.step_size, current_imm, L,
                                      chain_offset=chain_offset)
        adapt = update(adapt, info.proposal.position, info.proposal.momentum, state.position,
                       info.acceptance_rate, info.is_divergent, current_imm)
        in_window = t >= window_start
        if estimate_mm and in_window:
            mm = cgl_update_batch(mm, new_state.position)
        if enable_floor and in_window:
            cov = cgl_update_batch(cov, new_state.position)
            if t % LENGTH_FLOOR_RECOMPUTE_INTERVAL == 0:
                eigvec, lambda_max = recompute_eig_state(cov, current_imm, eigvec)
        state = new_state
        if record is not None:
            history.append(record(t, state, info, adapt))

    final_imm = diagonal_mass_matrix_or_fallback(mm, threshold, D) if estimate_mm else np.ones(D, f32)
    step_size_ma = f32(exp_cr(adapt.log_step_size_moving_average))
    floor_clipped = False
    if enable_floor:  # :971-991
        eigvec, lambda_max = recompute_eig_state(cov, final_imm, eigvec, LENGTH_FLOOR_FINAL_POWER_ITERATIONS)
        tl_ma = f32(exp_cr(adapt.log_trajectory_length_moving_average))
        consumed_ma, floor_clipped = apply_length_floor(tl_ma, lambda_max, mm.count >= threshold, True,
                                                        max_leapfrog_steps, step_size_ma)
        num_leapfrog = f32(consumed_ma / step_size_ma)
    else:  # :993-996
        num_leapfrog = f32(exp_cr(f32(adapt.log_trajectory_length_moving_average
                                      - adapt.log_step_size_moving_average)))
    parameters = {"step_size": step_size_ma, "inverse_mass_matrix": final_imm,
                  "integration_steps_params": (num_leapfrog,), "jitter_gn": jitter_gn,
                  "floor_clipped_by_cap": floor_clipped}
    rga = np.full(N, adapt.random_generator_arg, np.int32)
    return state, rga, parameters, (adapt, history)
