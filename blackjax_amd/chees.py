"""ChEES-HMC pooled (cross-chain) adaptation behind the ``blackjax.chees_adaptation`` API surface
(blackjax/adaptation/chees_adaptation.py).

One step size and one trajectory length for the whole ensemble, tuned from statistics of ALL chains:
the harmonic-mean acceptance rate drives dual averaging (358-374) and the ChEES criterion drives an
optimiser step on ``log(trajectory_length)`` (376-501).  Optionally a pooled diagonal
``inverse_mass_matrix`` (``mass_matrix_estimation="diagonal"``, 788-887) and the slow-direction
trajectory-length floor (112-236).

Division of labour: everything that touches an ``(N, D)`` array is a HIP kernel of
``include/bjx_pool.h`` (column / row reductions in fp64) or the HMC transition itself (the shared
trajectory length makes every warm-up step a plain ``blackjax_amd.hmc`` transition -- the headline
hot path); the scalar recursions run on the host in fp32, one device->host read of four doubles per
step.  Under chain sharding the fp64 sums are all-reduced over ``process_group`` (RCCL): this is the
one part of the engine with a data-path collective (two small all-reduces per step).

Differences from the reference, by design: the ensemble is folded into the diagonal accumulator with
the batch (CGL) merge of ``cgl_update_batch`` instead of a sequential row-by-row Welford scan
(816-824) -- equal up to rounding; the dense ``centered^T centered`` of the floor's covariance block
is a plain library GEMM (rocBLAS through ``torch.matmul``).
"""
from __future__ import annotations

from typing import Callable, NamedTuple, Optional

import numpy as np
import torch

from . import _lib, hmc
from . import random as bjx_random
from ._util import check_batch
from .adaptation import (AdaptationAlgorithm, AdaptationResults, _stack_history,
                         return_all_adapt_info)
from .distributed import all_reduce_sum_
from .dynamic_hmc import DynamicHMCState, halton_sequence, halton_steps_fn

__all__ = ["ChEESAdaptationState", "DualAveragingState", "base", "chees_adaptation",
           "weighted_empirical_mean", "OPTIMAL_TARGET_ACCEPTANCE_RATE"]

f32 = np.float32
f64 = np.float64

OPTIMAL_TARGET_ACCEPTANCE_RATE = 0.651  # chees_adaptation.py:21
LOG_UPDATE_CLIP = 0.35  # :23
EPS_FLOAT = 1e-20  # :25
CHEES_LENGTH_FLOOR_FACTOR = np.pi / 2  # :112
_LENGTH_FLOOR_RECOMPUTE_INTERVAL = 32  # :120
_LENGTH_FLOOR_POWER_ITERATIONS = 5  # :121
_LENGTH_FLOOR_FINAL_POWER_ITERATIONS = 20  # :122
_LENGTH_FLOOR_LAMBDA_EPS = 1e-6  # :127


# ----------------------------------------------------------------------------- host scalar recursions
class DualAveragingState(NamedTuple):  # optimizers/dual_averaging.py:24-50
    log_x: np.float32
    log_x_avg: np.float32
    step: int
    avg_error: np.float32
    mu: np.float32


def _da_init(x_init) -> DualAveragingState:  # dual_averaging.py:87-99
    x = f32(x_init)
    return DualAveragingState(f32(np.log(f64(x))), f32(0.0), 1, f32(0.0), f32(np.log(f64(f32(10.0) * x))))


def _da_update(state: DualAveragingState, gradient, t0=10, gamma=0.05, kappa=0.75):  # :101-123
    log_x, log_x_avg, step, avg_error, mu = state
    g = f32(gradient)
    reg = f32(step + t0)
    eta = f32(np.power(f64(step), f64(-kappa)))
    avg_error = f32(f32(f32(f32(1.0) - f32(f32(1.0) / reg)) * avg_error) + f32(g / reg))
    coef = f32(f32(np.sqrt(f32(step))) / f32(gamma))
    new_log_x = f32(mu - f32(coef * avg_error))
    new_log_x_avg = f32(f32(eta * log_x) + f32(f32(f32(1.0) - eta) * log_x_avg))
    return DualAveragingState(new_log_x, new_log_x_avg, step + 1, avg_error, mu)


class ChEESAdaptationState(NamedTuple):  # chees_adaptation.py:28-58
    step_size: np.float32
    log_step_size_moving_average: np.float32
    trajectory_length: np.float32
    log_trajectory_length_moving_average: np.float32
    da_state: DualAveragingState
    optim_state: tuple
    random_generator_arg: int
    step: int


def _exp32(x) -> np.float32:
    with np.errstate(over="ignore"):
        return f32(np.exp(f64(x)))


# ----------------------------------------------------------------------------- device statistics
class _Workspace:
    """Scratch for the pooled reductions of one (N, D) ensemble (no allocation inside the C ABI)."""

    def __init__(self, N: int, D: int, device):
        self.N, self.D, self.device = N, D, device
        nbytes = int(_lib.load().bjx_pool_workspace_bytes(N, D))
        self.scratch = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=device)
        self.stats = torch.empty(4 * D, dtype=torch.float64, device=device)
        self.scalars = torch.empty(4, dtype=torch.float64, device=device)
        self.colsum = torch.empty(D + 1, dtype=torch.float64, device=device)
        self.w = torch.empty(N, dtype=torch.float32, device=device)
        self.crit = torch.empty(N, dtype=torch.float32, device=device)
        self.pm = torch.empty(D, dtype=torch.float32, device=device)
        self.im = torch.empty(D, dtype=torch.float32, device=device)
        self.isq = torch.empty(D, dtype=torch.float32, device=device)


_workspaces: dict = {}


def _workspace(N: int, D: int, device) -> _Workspace:
    key = (N, D, device.type, device.index)
    ws = _workspaces.get(key)
    if ws is None:
        if len(_workspaces) > 4:
            _workspaces.clear()
        ws = _workspaces[key] = _Workspace(N, D, device)
    return ws


def _ensemble_means(ws: _Workspace, q_prop, acc, is_div, q_init, imm, group):
    """weights -> column statistics -> (all-reduce) -> means (chees_adaptation.py:376-386)."""
    N, D = q_prop.shape
    stream = _lib.current_stream()
    _lib.call("bjx_chees_weights_colstats", stream, N, D, q_prop.data_ptr(), acc.data_ptr(), is_div.data_ptr(),
              q_init.data_ptr(), ws.w.data_ptr(), ws.scratch.data_ptr(), ws.stats.data_ptr())
    all_reduce_sum_(ws.stats, group)
    _lib.call("bjx_chees_means", stream, D, ws.stats.data_ptr(), _lib.ptr(imm), ws.pm.data_ptr(),
              ws.im.data_ptr(), ws.isq.data_ptr() if imm is not None else None)


def weighted_empirical_mean(x: torch.Tensor, w: torch.Tensor, group=None) -> torch.Tensor:
    """chees_adaptation.py:239-247 for an ``(N, D)`` batch (rows with a non-finite entry get weight 0)."""
    x = check_batch(x, "x")
    N, D = x.shape
    ws = _workspace(N, D, x.device)
    no_div = torch.zeros(N, dtype=torch.bool, device=x.device)
    _ensemble_means(ws, x, w.to(torch.float32).contiguous(), no_div, x, None, group)
    return ws.pm.clone()


def _ensemble_scalars(q_prop, p_prop, q_init, acc, is_div, imm, scale, group) -> np.ndarray:
    """-> float64[4]: sum 1/acc, #non-divergent, sum acc*tg, sum (acc + 1e-20) over non-divergent
    chains of ALL ranks (chees_adaptation.py:358-360, 376-471)."""
    q_prop = check_batch(q_prop, "proposed_positions")
    p_prop = check_batch(p_prop, "proposed_momentums")
    q_init = check_batch(q_init, "initial_positions")
    acc = check_batch(acc, "acceptance_probabilities")
    N, D = q_prop.shape
    if is_div.dtype != torch.bool or not is_div.is_cuda:
        raise ValueError("is_divergent must be a device bool tensor")
    is_div = is_div.contiguous()
    ws = _workspace(N, D, q_prop.device)
    stream = _lib.current_stream()
    _ensemble_means(ws, q_prop, acc, is_div, q_init, imm, group)
    _lib.call("bjx_chees_criterion", stream, N, D, q_prop.data_ptr(), p_prop.data_ptr(), q_init.data_ptr(),
              ws.pm.data_ptr(), ws.im.data_ptr(), _lib.ptr(imm),
              ws.isq.data_ptr() if imm is not None else None, ws.crit.data_ptr())
    _lib.call("bjx_chees_scalars", stream, N, acc.data_ptr(), is_div.data_ptr(), ws.crit.data_ptr(),
              float(scale), ws.scalars.data_ptr())
    all_reduce_sum_(ws.scalars, group)
    return ws.scalars.cpu().numpy()  # the one device->host read of the step


def base(jitter_generator: Callable, next_random_arg_fn: Callable, optim, target_acceptance_rate: float,
         decay_rate: float, max_leapfrog_steps: int, _whiten_criterion: bool = True, *,
         process_group=None):
    """chees_adaptation.py:250-571: ``init(random_generator_arg, step_size)`` and
    ``update(state, proposed_positions, proposed_momentums, initial_positions,
    acceptance_probabilities, is_divergent, inverse_mass_matrix)`` on batched device tensors.
    ``inverse_mass_matrix=None`` means the identity metric (the ``jnp.ones`` of 835)."""

    def init(random_generator_arg, step_size: float):  # :513-523
        s = f32(step_size)
        return ChEESAdaptationState(s, f32(0.0), s, f32(0.0), _da_init(s), optim.init(s),
                                    random_generator_arg, 1)

    def update(adaptation_state: ChEESAdaptationState, proposed_positions, proposed_momentums,
               initial_positions, acceptance_probabilities, is_divergent, inverse_mass_matrix=None):
        imm = None
        if inverse_mass_matrix is not None and _whiten_criterion:
            imm = check_batch(inverse_mass_matrix, "inverse_mass_matrix")
            if imm.ndim != 1:
                raise ValueError("ChEES supports a shared diagonal inverse_mass_matrix of shape (D,)")
        scale = f32(f32(jitter_generator(adaptation_state.random_generator_arg))
                    * adaptation_state.trajectory_length)  # :461-462
        sums = _ensemble_scalars(proposed_positions, proposed_momentums, initial_positions,
                                 acceptance_probabilities, is_divergent, imm, scale, process_group)
        return scalar_update(adaptation_state, sums)

    def scalar_update(adaptation_state: ChEESAdaptationState, sums) -> ChEESAdaptationState:
        """The host half of ``compute_parameters`` (chees_adaptation.py:358-374, 468-511) from the four
        pooled sums of ``bjx_chees_scalars``."""
        (step_size, log_ss_ma, traj_len, log_tl_ma, da_state, optim_state, rga, step) = adaptation_state
        s_inv, n_ok, s_num, s_den = (f64(v) for v in sums)
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            hm = f32(f32(1.0) / f32(f32(s_inv) / f32(n_ok)))  # :358-360
            hm = hm if np.isfinite(hm) else f32(0.0)  # :362
            da_new = _da_update(da_state, f32(f32(target_acceptance_rate) - hm))  # :363
            ss_new = _exp32(da_new.log_x)  # :364
            if np.isfinite(ss_new):  # :365-370
                new_step_size, new_da, new_log_ss = ss_new, da_new, da_new.log_x
            else:
                new_step_size, new_da, new_log_ss = step_size, da_state, da_state.log_x
            uw = f32(np.power(f64(step), f64(-decay_rate)))  # :371
            new_log_ss_ma = f32(f32(f32(f32(1.0) - uw) * log_ss_ma) + f32(uw * new_log_ss))  # :372-374

            grad = f32(f32(s_num) / f32(s_den))  # :468-471
            log_tl = f32(np.log(f64(traj_len)))  # :473
            upd, optim_new = optim.update(grad, optim_state, log_tl)  # :474-476
            upd = f32(upd)
            if not np.isnan(upd):
                upd = f32(min(max(upd, f32(-LOG_UPDATE_CLIP)), f32(LOG_UPDATE_CLIP)))  # :478-480
            log_tl_new = f32(log_tl + upd)  # :481
            if not np.isfinite(log_tl_new):  # :482-489
                log_tl_new, optim_new = log_tl, optim_state
            new_log_tl_ma = f32(f32(f32(f32(1.0) - uw) * log_tl_ma) + f32(uw * log_tl_new))  # :490-492
            new_tl = _exp32(new_log_tl_ma)
            hi = f32(f32(max_leapfrog_steps) * new_step_size)
            new_tl = f32(min(max(new_tl, new_step_size), hi))  # :497-501
        return ChEESAdaptationState(new_step_size, new_log_ss_ma, new_tl, new_log_tl_ma, new_da, optim_new,
                                    next_random_arg_fn(rga), step + 1)

    update.scalar_update = scalar_update
    return init, update


# ----------------------------------------------------------------------------- pooled moment blocks
class MomentBlock(NamedTuple):  # adaptation/metric_buffers.py:171-215
    count: np.float32
    mean: torch.Tensor  # (D,)
    m2: torch.Tensor  # (D,) diagonal | (D, D) dense


def _batch_mean(ws: _Workspace, batch, group):
    """-> (global batch size as fp32, mean_b) : metric_buffers.py:428-429 over the chains of all ranks."""
    N, D = batch.shape
    stream = _lib.current_stream()
    _lib.call("bjx_pool_colsum", stream, N, D, batch.data_ptr(), None, ws.scratch.data_ptr(),
              ws.colsum.data_ptr())
    n_total = float(N)
    if group is not None:
        ws.colsum[D] = float(N)
        all_reduce_sum_(ws.colsum, group)
        n_total = float(ws.colsum[D].item())
    mean_b = torch.empty(D, dtype=torch.float32, device=batch.device)
    _lib.call("bjx_pool_mean", stream, D, ws.colsum.data_ptr(), n_total, mean_b.data_ptr())
    return f32(n_total), mean_b


def _cgl_update_diag(block: MomentBlock, batch, group) -> MomentBlock:
    """cgl_update_batch (metric_buffers.py:396-451) for a diagonal block, over all ranks."""
    N, D = batch.shape
    ws = _workspace(N, D, batch.device)
    stream = _lib.current_stream()
    n_b, mean_b = _batch_mean(ws, batch, group)
    _lib.call("bjx_pool_colsum", stream, N, D, batch.data_ptr(), mean_b.data_ptr(), ws.scratch.data_ptr(),
              ws.colsum.data_ptr())
    all_reduce_sum_(ws.colsum[:D], group)
    mean, m2 = block.mean.clone(), block.m2.clone()
    _lib.call("bjx_pool_merge_diag", stream, D, float(block.count), float(n_b), mean_b.data_ptr(),
              ws.colsum.data_ptr(), mean.data_ptr(), m2.data_ptr())
    return MomentBlock(f32(block.count + n_b), mean, m2)


def _cgl_update_dense(block: MomentBlock, batch, group) -> MomentBlock:
    """cgl_update_batch for the dense (D, D) block of the length floor."""
    N, D = batch.shape
    ws = _workspace(N, D, batch.device)
    n_b, mean_b = _batch_mean(ws, batch, group)
    centered = torch.empty_like(batch)
    _lib.call("bjx_pool_center", _lib.current_stream(), N, D, batch.data_ptr(), mean_b.data_ptr(),
              centered.data_ptr())
    m2_b = centered.t() @ centered  # plain library GEMM (rocBLAS), fp32
    all_reduce_sum_(m2_b, group)
    n_a = block.count
    n_ab = f32(n_a + n_b)
    delta = mean_b - block.mean
    mean_ab = block.mean + delta * float(f32(n_b / n_ab))
    coef = float(f32(f32(n_a * n_b) / n_ab))
    m2_ab = (block.m2 + m2_b) + torch.outer(delta, delta) * coef
    return MomentBlock(n_ab, mean_ab, m2_ab)


def _mass_matrix_engagement_threshold(num_dim: int) -> int:  # chees_adaptation.py:61-72
    return max(64, int(2 * np.sqrt(num_dim)))


def _diagonal_mass_matrix_or_fallback(block: MomentBlock, threshold: int, num_dim: int):  # :75-90
    if block.count >= threshold:
        imm = torch.empty_like(block.m2)
        _lib.call("bjx_pool_final_diag", _lib.current_stream(), num_dim, float(block.count),
                  block.m2.data_ptr(), imm.data_ptr())
        return imm
    return None  # identity metric


def _power_iteration_lambda_max(matrix, v0, num_iterations: int):  # :147-166
    v = v0
    for _ in range(num_iterations):
        v_next = matrix @ v
        norm = torch.linalg.vector_norm(v_next)
        v = v_next / torch.where(norm > 0.0, norm, torch.ones_like(norm))
    return torch.dot(v, matrix @ v), v


def _recompute_eig_state(cov_block: MomentBlock, imm, eigenvector,
                         num_iterations: int = _LENGTH_FLOOR_POWER_ITERATIONS):  # :169-189
    cov = cov_block.m2 / float(max(f32(cov_block.count - f32(1.0)), f32(1.0)))
    if imm is not None:
        inv_sqrt_d = 1.0 / torch.sqrt(imm)
        cov = cov * inv_sqrt_d[:, None] * inv_sqrt_d[None, :]
    lam, vec = _power_iteration_lambda_max(cov, eigenvector, num_iterations)
    return vec, f32(max(f32(lam.item()), f32(_LENGTH_FLOOR_LAMBDA_EPS)))


def _apply_length_floor(trajectory_length, lambda_max, engaged: bool, enable: bool,
                        max_leapfrog_steps: int = 1000, step_size: float = 0.1):  # :192-236
    if not enable:
        return f32(trajectory_length), False
    floor_value = (f32(f32(CHEES_LENGTH_FLOOR_FACTOR) * f32(np.sqrt(f32(lambda_max)))) if engaged
                   else f32(0.0))
    cap = f32(f32(max_leapfrog_steps) * f32(step_size))
    consumed = f32(min(max(f32(trajectory_length), floor_value), cap))
    return consumed, bool(engaged and floor_value > cap)


# ----------------------------------------------------------------------------- public API
class _AdaptationInfo:
    """The stacked per-step info plus the ``floor_clipped_by_cap`` flag (chees_adaptation.py:1007-1021)."""

    def __init__(self, info_obj, floor_flag):
        self._wrapped_info = info_obj
        self.floor_clipped_by_cap = floor_flag

    def __getattr__(self, name):
        return getattr(self._wrapped_info, name)


def chees_adaptation(logdensity_fn: Callable, num_chains: int, *, jitter_generator: Optional[Callable] = None,
                     jitter_amount: float = 1.0,
                     target_acceptance_rate: float = OPTIMAL_TARGET_ACCEPTANCE_RATE,
                     decay_rate: float = 0.5, max_leapfrog_steps: int = 1000,
                     adaptation_info_fn: Optional[Callable] = return_all_adapt_info,
                     mass_matrix_estimation: Optional[str] = None,
                     mass_matrix_window_fraction: float = 0.5, _whiten_criterion: bool = True,
                     _length_floor: bool = True, chain_offset: int = 0,
                     process_group=None, fuse_target: bool = False) -> AdaptationAlgorithm:
    """blackjax/adaptation/chees_adaptation.py:574-1025.

    ``num_chains`` is the number of chains THIS process holds (``positions.shape[0]``);
    ``chain_offset`` its first global chain index; ``process_group`` the ``torch.distributed`` group
    whose ranks pool their statistics (``None``: this process only).  ``jitter_generator`` takes a
    host key (``uint32[2]``) and returns a float in [0, 1] (e.g. ``blackjax_amd.random.uniform``).

    ``run(rng_key, positions, step_size, optim, num_steps=1000, *, max_sampling_steps=1000)`` returns
    ``(AdaptationResults(last_states, parameters), info)`` with ``parameters`` ready for
    ``blackjax_amd.dynamic_hmc(logdensity_fn, **parameters)``.

    ``fuse_target=True`` (engine-resident or ``targets.DeviceTarget`` log-densities; OUTSIDE the external-callable
    contract): every warm-up transition is one launch (``hmc.build_fused_target_kernel``), the pooled statistics
    are unchanged; results equal the default warm-up's bit for bit."""
    if mass_matrix_estimation not in (None, "diagonal"):
        raise ValueError("mass_matrix_estimation must be None or 'diagonal', got "
                         f"{mass_matrix_estimation!r}.")
    if not 0.0 <= mass_matrix_window_fraction <= 1.0:
        raise ValueError("mass_matrix_window_fraction must be in [0.0, 1.0], got "
                         f"{mass_matrix_window_fraction}.")
    estimate_mass_matrix = mass_matrix_estimation == "diagonal"
    enable_length_floor = estimate_mass_matrix and _length_floor

    def run(rng_key, positions, step_size: float, optim, num_steps: int = 1000, *,
            max_sampling_steps: int = 1000):
        positions = check_batch(positions, "positions")
        if positions.shape[0] != num_chains:
            raise ValueError("initial `positions` leading dimension must be equal to the `num_chains`")
        N, D = positions.shape
        dev = positions.device
        num_steps = int(num_steps)
        ja, jb = f32(jitter_amount), f32(1.0 - jitter_amount)

        next_random_arg_fn = lambda i: i + 1  # :753
        if jitter_generator is not None:  # :756-760
            ks = bjx_random.split(rng_key, 2)
            rng_key, carry_key = ks[0], ks[1]
            jitter_gn = lambda i: f32(f32(f32(jitter_generator(bjx_random.fold_in(carry_key, int(i)))) * ja) + jb)
            integration_steps_fn = _generator_steps_fn(jitter_gn)
        else:  # :761-765
            max_bits = int(np.ceil(np.log2(num_steps + max_sampling_steps)))
            jitter_gn = lambda i: f32(f32(halton_sequence(int(i), max_bits) * ja) + jb)
            integration_steps_fn = halton_steps_fn(max_bits, float(jitter_amount))

        kernel = hmc.build_fused_target_kernel() if fuse_target else hmc.build_kernel()
        init, update = base(jitter_gn, next_random_arg_fn, optim, target_acceptance_rate, decay_rate,
                            max_leapfrog_steps, _whiten_criterion, process_group=process_group)
        window_start = int(mass_matrix_window_fraction * num_steps) if estimate_mass_matrix else num_steps
        threshold = _mass_matrix_engagement_threshold(D)

        state = hmc.init(positions, logdensity_fn)
        adapt = init(0, step_size)
        zeros = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        mm = MomentBlock(f32(0.0), zeros(D), zeros(D)) if estimate_mass_matrix else None
        cov = MomentBlock(f32(0.0), zeros(D), zeros(D, D)) if enable_length_floor else None
        eigvec = torch.full((D,), 1.0 / float(np.sqrt(f32(D))), dtype=torch.float32, device=dev)
        lambda_max = f32(1.0)
        ones = torch.ones(D, dtype=torch.float32, device=dev)
        keys_step = bjx_random.split(rng_key, num_steps)
        history = []
        for t in range(num_steps):  # one_step, :826-914
            est = _diagonal_mass_matrix_or_fallback(mm, threshold, D) if estimate_mass_matrix else None
            current_imm = ones if est is None else est
            consumed = adapt.trajectory_length
            if enable_length_floor:
                consumed, _ = _apply_length_floor(adapt.trajectory_length, lambda_max,
                                                  bool(mm.count >= threshold), True, max_leapfrog_steps,
                                                  adapt.step_size)
            L = int(np.ceil(f32(jitter_gn(adapt.random_generator_arg) * f32(consumed / adapt.step_size))))
            new_state, info = kernel(keys_step[t], state, logdensity_fn, float(adapt.step_size),
                                     current_imm, L, chain_offset=chain_offset)
            adapt = update(adapt, info.proposal.position, info.proposal.momentum, state.position,
                           info.acceptance_rate, info.is_divergent, est)
            in_window = t >= window_start
            if estimate_mass_matrix and in_window:
                mm = _cgl_update_diag(mm, new_state.position, process_group)
            if enable_length_floor and in_window:
                cov = _cgl_update_dense(cov, new_state.position, process_group)
                if t % _LENGTH_FLOOR_RECOMPUTE_INTERVAL == 0:
                    eigvec, lambda_max = _recompute_eig_state(cov, est, eigvec)
            state = new_state
            if adaptation_info_fn is not None:
                rga = torch.full((N,), int(adapt.random_generator_arg), dtype=torch.int32, device=dev)
                history.append(adaptation_info_fn(
                    DynamicHMCState(state.position, state.logdensity, state.logdensity_grad, rga), info,
                    adapt))

        est = _diagonal_mass_matrix_or_fallback(mm, threshold, D) if estimate_mass_matrix else None
        final_imm = ones if est is None else est
        step_size_ma = _exp32(adapt.log_step_size_moving_average)
        floor_clipped_by_cap = False
        if enable_length_floor:  # :971-991
            eigvec, lambda_max = _recompute_eig_state(cov, est, eigvec, _LENGTH_FLOOR_FINAL_POWER_ITERATIONS)
            tl_ma = _exp32(adapt.log_trajectory_length_moving_average)
            consumed_ma, floor_clipped_by_cap = _apply_length_floor(
                tl_ma, lambda_max, bool(mm.count >= threshold), True, max_leapfrog_steps, step_size_ma)
            num_leapfrog_steps = f32(consumed_ma / step_size_ma)
        else:  # :993-996
            num_leapfrog_steps = _exp32(f32(adapt.log_trajectory_length_moving_average
                                            - adapt.log_step_size_moving_average))
        parameters = {
            "step_size": float(step_size_ma),
            "inverse_mass_matrix": final_imm,
            "next_random_arg_fn": next_random_arg_fn,
            "integration_steps_fn": integration_steps_fn,
            "integration_steps_params": (float(num_leapfrog_steps),),
        }
        rga = torch.full((N,), int(adapt.random_generator_arg), dtype=torch.int32, device=dev)
        last_states = DynamicHMCState(state.position, state.logdensity, state.logdensity_grad, rga)
        info_out = _AdaptationInfo(_stack_history(history), floor_clipped_by_cap)
        return AdaptationResults(last_states, parameters), info_out

    return AdaptationAlgorithm(run)


def _generator_steps_fn(jitter_gn: Callable):
    """``integration_steps_fn`` for a user jitter generator (chees_adaptation.py:758-771): the
    counters of an ensemble are few distinct values, evaluated on the host."""

    def steps_fn(random_generator_arg: torch.Tensor, num_leapfrog_steps: float):
        vals = random_generator_arg.to(torch.int64)
        out = torch.empty_like(vals, dtype=torch.int32)
        for v in torch.unique(vals).tolist():
            out[vals == v] = int(np.ceil(f32(jitter_gn(v) * f32(num_leapfrog_steps))))
        return out

    return steps_fn
