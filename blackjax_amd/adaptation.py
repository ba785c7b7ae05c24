"""Stan-style window adaptation on MI355X behind ``blackjax.window_adaptation``.

Mirrors blackjax/adaptation/window_adaptation.py:296-444 (argument validation + delegate) and the
single-chain engine of blackjax/adaptation/staged_adaptation.py (``_make_engine`` 111-307,
``build_schedule`` 315-405, ``run`` 860-876 / 968-981) with dual averaging
(optimizers/dual_averaging.py:53-129, adaptation/step_size.py:65-150) and the Welford
mass-matrix estimator (adaptation/mass_matrix.py:111-444).

Semantics: adaptation is PER CHAIN, exactly as in the reference where many chains means
``jax.vmap(warmup.run)(jax.random.split(rng_key, N), positions)``: every chain owns a step size
and an inverse mass matrix (diagonal ``(N, D)``, or dense ``(N, D, D)`` with
``is_mass_matrix_diagonal=False``); chain ``i`` uses key ``split(split(rng_key, N)[i], T)[t]``
at step ``t`` (chain-major layout).  No collective is needed when chains
are sharded over GPUs.  All per-chain arithmetic runs in HIP kernels (``bjx_adapt.hip``).
"""
from __future__ import annotations

from typing import Callable, NamedTuple, Optional

import numpy as np
import torch

from . import _lib, integrators, metrics
from ._util import check_batch
from .base import AdaptationAlgorithm
from .random import ChainMajorKey, key_words

__all__ = ["window_adaptation", "staged_adaptation", "build_schedule", "free_running_table", "AdaptationResults", "AdaptationInfo",
           "return_all_adapt_info", "get_filter_adapt_info_fn", "DualAveragingAdaptationState",
           "WelfordAlgorithmState", "MassMatrixAdaptationState", "StagedAdaptationState"]


# ------------------------------------------------------------------------------- result types
class AdaptationResults(NamedTuple):  # adaptation/base.py:21-23
    state: NamedTuple
    parameters: dict


class AdaptationInfo(NamedTuple):  # adaptation/base.py:26-29
    state: NamedTuple
    info: NamedTuple
    adaptation_state: NamedTuple


def return_all_adapt_info(state, info, adaptation_state):
    """adaptation/base.py:32-36.  NOTE: with thousands of chains this retains every step's
    (N, D) tensors; pass ``get_filter_adapt_info_fn(...)`` to keep only what is needed.  Passed EXPLICITLY it is
    always honoured; ``window_adaptation``'s default (``_default_adapt_info``) behaves like it but is replaced by
    ``_scalars_only_adapt_info`` (with a warning) when one step's record would exceed ``ALL_INFO_MAX_BYTES``."""
    return AdaptationInfo(state, info, adaptation_state)


def _default_adapt_info(state, info, adaptation_state):
    """The default ``adaptation_info_fn`` of ``window_adaptation`` / ``staged_adaptation``: a private sentinel that
    records what ``return_all_adapt_info`` records (adaptation/base.py:32-36), distinguishable from an explicit
    ``adaptation_info_fn=return_all_adapt_info`` (the explicit argument must never be downgraded)."""
    return AdaptationInfo(state, info, adaptation_state)


# One step of return_all_adapt_info holds ~11 (N, D) tensors (state, proposal, momentum, trajectory, Welford, metric):
# above this many bytes PER STEP the default keeps the per-chain scalars only (1 GiB at 32 768 x 4 096: a 1 000-step
# warm-up would otherwise retain 5+ TiB -- the reference's default is fatal at this scale)
ALL_INFO_MAX_BYTES = 256 << 20
_WARNED_INFO = set()  # (n, d) shapes the downgrade was announced for


def _scalars_only_adapt_info(state, info, adaptation_state):
    """What the default ``adaptation_info_fn`` keeps for large ensembles: every per-chain SCALAR of the step --
    ``state.logdensity``, the (N,) fields of ``info`` (acceptance rate, accept / divergence flags, energy, step
    counts), the dual-averaging state and the step size -- and none of the (N, D) tensors."""

    n = state.position.shape[0]

    def small(t):  # per-chain scalars: (N,) tensors (and Python scalars)
        if isinstance(t, torch.Tensor):
            return t if (t.ndim == 0 or (t.ndim == 1 and t.shape[0] == n)) else None
        return t

    def filt(tup):
        if tup is None:
            return None
        if isinstance(tup, tuple) and hasattr(tup, "_fields"):
            # the metric is dropped BY NAME: a shared (D,) diagonal has the shape of a per-chain scalar when N == D
            return type(tup)(*[None if k == "inverse_mass_matrix" else filt(v) for k, v in zip(tup._fields, tup)])
        return small(tup)

    return AdaptationInfo(filt(state), filt(info), filt(adaptation_state))


def _select_info_fn(adaptation_info_fn, n: int, d: int, is_mass_matrix_diagonal: bool):
    """What ``window_adaptation.run`` records per step, and whether the Welford buffers may be updated in place.
    Only the DEFAULT (``_default_adapt_info``) is downgraded to per-chain scalars for large ensembles, with a warning
    per shape; an explicit ``return_all_adapt_info`` -- or any user function -- is honoured as given."""
    info_fn = adaptation_info_fn
    if info_fn is _default_adapt_info and 11 * 4 * n * d > ALL_INFO_MAX_BYTES:
        info_fn = _scalars_only_adapt_info
        if (n, d) not in _WARNED_INFO:  # once per shape: every downgraded shape is announced
            _WARNED_INFO.add((n, d))
            import warnings

            warnings.warn(
                f"blackjax_amd.window_adaptation: the default adaptation_info_fn would retain "
                f"~{11 * 4 * n * d / 2**30:.1f} GiB per warm-up step at {n} x {d}; keeping the per-chain scalars of every "
                "step instead (state.logdensity, info.acceptance_rate / flags / energy, the dual-averaging state, "
                "step sizes).  Pass adaptation_info_fn=blackjax_amd.adaptation.return_all_adapt_info explicitly "
                "(it is then honoured) or get_filter_adapt_info_fn(...) to choose.", RuntimeWarning, stacklevel=3)
    # the Welford buffers may be updated in place when no per-step record can hold on to them
    in_place = bool(is_mass_matrix_diagonal) and (
        info_fn is None or info_fn is _scalars_only_adapt_info or getattr(info_fn, "_bjx_keeps_no_welford", False))
    return info_fn, in_place


def get_filter_adapt_info_fn(state_keys=frozenset(), info_keys=frozenset(),
                             adapt_state_keys=frozenset()):
    """adaptation/base.py:39-60: keep only the named fields (others become ``None``)."""

    def filter_tuple(tup, key_set):
        return type(tup)(*[v if k in key_set else None for k, v in zip(tup._fields, tup)])

    def filter_fn(state, info, adaptation_state):
        return AdaptationInfo(filter_tuple(state, state_keys), filter_tuple(info, info_keys),
                              filter_tuple(adaptation_state, adapt_state_keys))

    # no field of the adaptation state is kept: the engine may then update the Welford buffers in place
    filter_fn._bjx_keeps_no_welford = not (set(adapt_state_keys) & {"imm_state"})
    return filter_fn


class DualAveragingAdaptationState(NamedTuple):  # adaptation/step_size.py:26-62
    log_step_size: torch.Tensor  # (N,)
    log_step_size_avg: torch.Tensor
    step: int  # identical for every chain (same schedule)
    avg_error: torch.Tensor
    mu: torch.Tensor


class WelfordAlgorithmState(NamedTuple):  # adaptation/mass_matrix.py:364-388
    mean: torch.Tensor  # (N, D)
    m2: torch.Tensor  # (N, D)
    sample_size: int


class MassMatrixAdaptationState(NamedTuple):  # adaptation/mass_matrix.py:33-56
    inverse_mass_matrix: torch.Tensor  # (N, D)
    wc_state: WelfordAlgorithmState


class StagedAdaptationState(NamedTuple):  # adaptation/staged_adaptation.py:69-103
    ss_state: DualAveragingAdaptationState
    imm_state: MassMatrixAdaptationState
    step_size: torch.Tensor  # (N,)
    inverse_mass_matrix: torch.Tensor  # (N, D)


# ------------------------------------------------------------------------------- schedule (host)
def build_schedule(num_steps: int, initial_buffer_size: int = 75, final_buffer_size: int = 50,
                   first_window_size: int = 25) -> list:
    """Stan warmup schedule, list of ``(stage, is_middle_window_end)``
    (adaptation/staged_adaptation.py:315-405)."""
    schedule = []
    if num_steps < 20:
        return [(0, False)] * num_steps
    if initial_buffer_size + first_window_size + final_buffer_size > num_steps:
        initial_buffer_size = int(0.15 * num_steps)
        final_buffer_size = int(0.1 * num_steps)
        first_window_size = num_steps - initial_buffer_size - final_buffer_size
    schedule += [(0, False)] * initial_buffer_size
    final_buffer_start = num_steps - final_buffer_size
    size, start = first_window_size, initial_buffer_size
    while start < final_buffer_start:
        cur_start, cur_size = start, size
        if 3 * cur_size <= final_buffer_start - cur_start:
            size = 2 * cur_size
        else:
            cur_size = final_buffer_start - cur_start
        start = cur_start + cur_size
        schedule += [(1, False)] * (start - 1 - cur_start)
        schedule.append((1, True))
    schedule += [(0, False)] * (num_steps - final_buffer_start)
    return schedule


# ------------------------------------------------------------------------------- engine
_DA_T0, _DA_GAMMA, _DA_KAPPA = 10.0, 0.05, 0.75  # dual_averaging.py:53-55 defaults


def _da_init(x_in: torch.Tensor, from_log_avg: bool) -> tuple:
    n = x_in.shape[0]
    outs = [torch.empty_like(x_in) for _ in range(5)]
    _lib.call("bjx_da_init", _lib.current_stream(), n, int(from_log_avg), x_in.data_ptr(),
              *[o.data_ptr() for o in outs])
    log_x, log_x_avg, avg_err, mu, step_size = outs
    return DualAveragingAdaptationState(log_x, log_x_avg, 1, avg_err, mu), step_size


def _da_update(ss: DualAveragingAdaptationState, acceptance_rate: torch.Tensor, target: float):
    n = acceptance_rate.shape[0]
    log_x, log_x_avg, avg_err, step_size = (torch.empty_like(acceptance_rate) for _ in range(4))
    _lib.call("bjx_da_update", _lib.current_stream(), n, ss.step, float(target), _DA_T0, _DA_GAMMA,
              _DA_KAPPA, acceptance_rate.data_ptr(), ss.log_step_size.data_ptr(),
              ss.log_step_size_avg.data_ptr(), ss.avg_error.data_ptr(), ss.mu.data_ptr(),
              log_x.data_ptr(), log_x_avg.data_ptr(), avg_err.data_ptr(), step_size.data_ptr())
    return DualAveragingAdaptationState(log_x, log_x_avg, ss.step + 1, avg_err, ss.mu), step_size


def _welford_update(wc: WelfordAlgorithmState, position: torch.Tensor, in_place: bool = False) -> WelfordAlgorithmState:
    """mass_matrix.py:410-435; ``wc.m2`` is (N, D) [diagonal] or (N, D, D) [dense].  ``in_place`` (diagonal
    estimator only: the kernel is element-wise, every element read and written by one thread): update ``wc``'s
    own buffers instead of allocating two fresh (N, D) tensors per slow step -- only when nothing retains the
    previous state (round 5: at 32 768 x 4 096 that is 1 GiB of allocator traffic per warm-up step)."""
    n, d = position.shape
    if in_place and wc.m2.ndim == 2:
        mean, m2 = wc.mean, wc.m2
    else:
        mean, m2 = torch.empty_like(position), torch.empty_like(wc.m2)
    name = "bjx_welford_update_diag" if wc.m2.ndim == 2 else "bjx_welford_update_dense"
    _lib.call(name, _lib.current_stream(), n, d, wc.sample_size + 1, position.data_ptr(),
              wc.mean.data_ptr(), wc.m2.data_ptr(), mean.data_ptr(), m2.data_ptr())
    return WelfordAlgorithmState(mean, m2, wc.sample_size + 1)


def _mm_final(mm: MassMatrixAdaptationState, shrinkage: float) -> MassMatrixAdaptationState:
    """mass_matrix.py:335-357 (diagonal and dense)."""
    m2 = mm.wc_state.m2
    n, d = m2.shape[0], m2.shape[1]
    imm = torch.empty_like(m2)
    prev = mm.inverse_mass_matrix
    if m2.ndim == 2:
        _lib.call("bjx_welford_final_diag", _lib.current_stream(), n, d, mm.wc_state.sample_size,
                  float(shrinkage), m2.data_ptr(), prev.data_ptr(), 0 if prev.ndim == 1 else d,
                  imm.data_ptr())
    else:
        _lib.call("bjx_welford_final_dense", _lib.current_stream(), n, d, mm.wc_state.sample_size,
                  float(shrinkage), m2.data_ptr(), prev.data_ptr(), 1 if prev.ndim == 3 else 0,
                  imm.data_ptr())
    mean0 = torch.zeros_like(mm.wc_state.mean)
    return MassMatrixAdaptationState(imm, WelfordAlgorithmState(mean0, torch.zeros_like(m2), 0))


def free_running_table(num_steps: int, imm_shrinkage_to_previous: float = 0.0) -> "np.ndarray":
    """Host table for a free-running warm-up (``bjx_nuts_async_t.adapt_tab``, include/bjx_nuts.h):
    one row per warm-up step with the schedule flags and every fp32 scalar that depends on the step
    index only, evaluated exactly as the lockstep entry points evaluate them on the host
    (``bjx_da_update``: ``step + t0``, its reciprocal, ``step^-kappa``, ``sqrt(step)/gamma``;
    ``bjx_welford_final_diag``: ``n - 1`` and the blend coefficients of mass_matrix.py:339-343)."""
    f32 = np.float32
    tab = np.zeros((num_steps, _lib.NUTS_ADAPT_COLS), f32)
    AT = _lib.NUTS_AT
    da_step, wel = 1, 0
    shrink = f32(imm_shrinkage_to_previous)
    for t, (stage, is_window_end) in enumerate(build_schedule(num_steps)):
        tab[t, AT["FLAGS"]] = (1 if stage == 1 else 0) | (2 if is_window_end else 0)
        if stage == 1:
            wel += 1
            tab[t, AT["WEL_N"]] = wel
        reg = f32(da_step) + f32(_DA_T0)
        tab[t, AT["DA_REG"]] = reg
        tab[t, AT["DA_INV_REG"]] = f32(1.0) / reg
        tab[t, AT["DA_ETA"]] = f32(float(da_step) ** (-float(f32(_DA_KAPPA))))
        tab[t, AT["DA_COEF"]] = np.sqrt(f32(da_step)) / f32(_DA_GAMMA)
        da_step += 1
        if is_window_end:
            denom = f32(wel + 5) + shrink
            tab[t, AT["FIN_NM1"]] = f32(wel - 1)
            tab[t, AT["FIN_BETA_DATA"]] = f32(wel) / denom
            tab[t, AT["FIN_BETA_PREV"]] = shrink / denom
            tab[t, AT["FIN_REG"]] = (f32(5.0) / denom) * f32(1e-3)
            wel, da_step = 0, 1
    return tab


def _stack_history(history):
    """Stack per-step records along a new leading axis, like the ``lax.scan`` output of the
    reference (staged_adaptation.py:870-874)."""
    if not history:
        return None

    def stack(items):
        first = items[0]
        if first is None:
            return None
        if isinstance(first, torch.Tensor):
            if any(not isinstance(it, torch.Tensor) or it.shape != first.shape for it in items):
                return items  # e.g. the shared (D,) initial imm vs the per-chain (N, D) adapted ones
            return torch.stack(items)
        if isinstance(first, tuple) and hasattr(first, "_fields"):
            return type(first)(*[stack([it[i] for it in items]) for i in range(len(first))])
        if isinstance(first, (int, float, bool)) or (hasattr(first, "dtype") and hasattr(first, "item")
                                                     and getattr(first, "ndim", 1) == 0):
            return torch.tensor([it.item() if hasattr(it, "item") else it for it in items])
        return items

    return stack(history)


def window_adaptation(algorithm, logdensity_fn: Callable, is_mass_matrix_diagonal: bool = True,
                      initial_inverse_mass_matrix=None, imm_shrinkage_to_previous: float = 0.0,
                      initial_step_size: float = 1.0, target_acceptance_rate: float = 0.80,
                      adaptation_info_fn: Optional[Callable] = _default_adapt_info,
                      integrator=integrators.velocity_verlet, _schedule_fn: Optional[Callable] = None,
                      fuse_target: bool = False, **extra_parameters) -> AdaptationAlgorithm:
    """blackjax/adaptation/window_adaptation.py:296-444.  ``algorithm`` is ``blackjax_amd.hmc`` or
    ``blackjax_amd.nuts``; ``extra_parameters`` are forwarded to its kernel (e.g.
    ``num_integration_steps=...``).  ``adaptation_info_fn=None`` records nothing.

    ``run(rng_key, position, num_steps)`` returns ``(AdaptationResults, AdaptationInfo)`` as in the
    reference.  ``run(..., free_running=True)`` (NUTS, diagonal metric, default integrator) returns
    ``(AdaptationResults, NUTSRunInfo)`` instead: the chains adapt inside the free-running kernels,
    so there is no per-step hook for ``adaptation_info_fn`` -- combining the flag with a custom
    ``adaptation_info_fn`` or integrator raises rather than silently ignoring them.
    ``parameters["inverse_mass_matrix"]`` is a ``metrics.PerChainDiagTensor`` (an ``(N, D)`` tensor
    tagged as per-chain diagonals) so that ``algorithm(logdensity_fn, **parameters)`` is
    unambiguous even when ``N == D``.

    ``fuse_target=True`` with ``blackjax_amd.hmc`` (engine-resident or ``targets.DeviceTarget`` log-densities,
    diagonal mass matrix, velocity Verlet; OUTSIDE the external-callable contract): every warm-up transition is
    one launch (``hmc.build_fused_target_kernel``); results equal the default warm-up's bit for bit.  For NUTS
    the switch belongs to ``run(..., free_running=True, fuse_target=True)``."""
    if initial_inverse_mass_matrix is not None:
        imm0 = torch.as_tensor(initial_inverse_mass_matrix)
        if is_mass_matrix_diagonal:
            if imm0.ndim != 1:
                raise ValueError(
                    "is_mass_matrix_diagonal=True requires "
                    f"initial_inverse_mass_matrix.ndim == 1, got ndim={imm0.ndim}")
        elif imm0.ndim != 2 or imm0.shape[0] != imm0.shape[1]:
            raise ValueError(
                "is_mass_matrix_diagonal=False requires initial_inverse_mass_matrix to be a 2-D "
                f"square array, got shape={tuple(imm0.shape)}")
    if imm_shrinkage_to_previous < 0.0:
        raise ValueError(
            f"imm_shrinkage_to_previous must be >= 0.0, got {imm_shrinkage_to_previous}")
    mcmc_kernel = algorithm.build_kernel(integrator)  # the sampler validates the integrator
    step_extra = extra_parameters  # what every step of the kernel receives
    if fuse_target:
        from .hmc import build_fused_target_kernel as _fused, build_kernel as _hmc_build_kernel

        if getattr(algorithm, "build_kernel", None) is not _hmc_build_kernel:
            raise NotImplementedError("window_adaptation(fuse_target=True) is the hmc switch; NUTS takes "
                                      "run(..., free_running=True, fuse_target=True)")
        if integrator is not integrators.velocity_verlet or not is_mass_matrix_diagonal:
            raise NotImplementedError("fuse_target=True: velocity Verlet and a diagonal mass matrix")
        fused_kwargs = {}
        if "divergence_threshold" in extra_parameters:
            # a build-time argument of the fused kernel, not a per-step one: it stays in the
            # returned parameters, but is not forwarded to every step
            fused_kwargs["divergence_threshold"] = extra_parameters["divergence_threshold"]
            step_extra = {k: v for k, v in extra_parameters.items() if k != "divergence_threshold"}
        mcmc_kernel = _fused(**fused_kwargs)

    def _run_free_running(rng_key, state, imm, ss, eps0, num_steps, chain_offset, fuse_target=False):
        """The same warm-up with free-running chains (nuts.run_free, include/bjx_nuts.h adapt_*): every
        chain carries its own dual-averaging / Welford state through its own sequence of trees and
        adapts when IT finishes a transition, so a warm-up no longer lasts as long as the deepest tree
        of every step.  Chains are independent in the reference's vmapped warm-up, hence the results
        are those of ``run`` bit for bit.  NUTS with a diagonal metric; the per-step record is the
        ``NUTSRunInfo`` of the run (with ``step_size``), not ``adaptation_info_fn``'s.
        ``fuse_target``: ``nuts.run_free``'s switch (engine-resident log-densities only)."""
        from .nuts import build_kernel as _nuts_build_kernel, run_free as _nuts_run_free

        if getattr(algorithm, "build_kernel", None) is not _nuts_build_kernel:
            raise NotImplementedError("free_running=True is implemented for blackjax_amd.nuts")
        if not is_mass_matrix_diagonal:
            raise NotImplementedError("free_running=True needs a diagonal mass matrix")
        if adaptation_info_fn not in (None, return_all_adapt_info, _default_adapt_info):
            raise ValueError("free_running=True records a NUTSRunInfo, not adaptation_info_fn's output: "
                             "pass adaptation_info_fn=None (or the default) with it")
        if integrator is not integrators.velocity_verlet:
            from .nuts import free_running_supports

            if fuse_target or not free_running_supports(integrator, "diag", state.position.shape[1]):
                raise NotImplementedError(
                    "free_running=True with a multi-stage integrator: no fuse_target, at most "
                    "NUTS_MAX_MID middle stages (nuts.free_running_supports)")
        if _schedule_fn is not None:
            raise NotImplementedError("free_running=True uses the Stan schedule (build_schedule)")
        n, d = state.position.shape
        dev = state.position.device
        extra = dict(extra_parameters)
        max_depth = int(extra.pop("max_num_doublings", 10))
        div_thr = float(extra.pop("divergence_threshold", 1000))
        if extra:
            raise TypeError(f"unexpected parameters for a free-running NUTS warm-up: {sorted(extra)}")
        tab = free_running_table(num_steps, imm_shrinkage_to_previous)
        imm_pc = (imm if imm.ndim == 2 else imm.expand(n, d)).contiguous().clone()
        ad = {"tab": tab, "target": float(target_acceptance_rate), "log_x": ss.log_step_size.clone(),
              "log_x_avg": ss.log_step_size_avg.clone(), "avg_err": ss.avg_error.clone(), "mu": ss.mu.clone(),
              "step_size": eps0.clone(), "mean": torch.zeros_like(state.position),
              "m2": torch.zeros_like(state.position), "imm": imm_pc}
        state, _, run_info = _nuts_run_free(
            rng_key, state, logdensity_fn, ad["step_size"], imm_pc, num_steps, max_depth,
            divergence_threshold=div_thr, chain_offset=chain_offset, key_layout="chain_major",
            store_positions=False, adaptation=ad, fuse_target=fuse_target, integrator=integrator)
        step_size = torch.empty_like(ad["log_x_avg"])
        _lib.call("bjx_exp", _lib.current_stream(), n, ad["log_x_avg"].data_ptr(), step_size.data_ptr())
        parameters = {"step_size": step_size,
                      "inverse_mass_matrix": metrics.PerChainDiagTensor.tag(imm_pc), **extra_parameters}
        return AdaptationResults(state, parameters), run_info

    def run(rng_key, position, num_steps: Optional[int] = None, *, chain_offset: int = 0,
            free_running: bool = False, fuse_target: bool = False):
        """staged_adaptation.py:756-786,860-876,968-981 (single-chain path, batched over chains).  ``num_steps=None``
        is the reference's sentinel for "not given": 1 000 steps (its other meaning belongs to ``metric="auto"``).
        ``free_running=True`` (NUTS, diagonal metric): see ``_run_free_running``; with it,
        ``fuse_target=True`` evaluates a ``blackjax_amd.targets`` log-density inside the tick kernels
        (``nuts.run_free``: same results, outside the external-callable contract)."""
        if fuse_target and not free_running:
            raise ValueError("fuse_target=True needs free_running=True")
        num_steps = 1000 if num_steps is None else int(num_steps)
        position = check_batch(position, "position")
        n, d = position.shape
        run_key = key_words(rng_key)
        state = algorithm.init(position, logdensity_fn)
        # adapt_init: staged_adaptation.py:173-184
        if initial_inverse_mass_matrix is None:
            if is_mass_matrix_diagonal:
                imm = torch.ones(d, dtype=torch.float32, device=position.device)
            else:  # mass_matrix.py:243-244: identity, shared by all chains until the first window end
                imm = torch.eye(d, dtype=torch.float32, device=position.device)
        else:
            imm = torch.as_tensor(initial_inverse_mass_matrix, dtype=torch.float32,
                                  device=position.device).contiguous()
        eps0 = torch.full((n,), float(initial_step_size), dtype=torch.float32, device=position.device)
        ss, _ = _da_init(eps0, from_log_avg=False)
        if free_running:
            return _run_free_running(rng_key, state, imm, ss, eps0, int(num_steps), chain_offset, fuse_target)
        zeros = torch.zeros_like(position)
        # Welford second moments: (N, D) diagonal, or (N, D, D) dense -- one matrix PER CHAIN, the
        # semantics of a vmapped dense warmup (N * D^2 words: meant for moderate N * D^2)
        m2_0 = (torch.zeros_like(position) if is_mass_matrix_diagonal
                else torch.zeros((n, d, d), dtype=torch.float32, device=position.device))
        ws = StagedAdaptationState(
            ss, MassMatrixAdaptationState(imm, WelfordAlgorithmState(zeros, m2_0, 0)), eps0, imm)
        history = []
        info = None
        info_fn, welford_in_place = _select_info_fn(adaptation_info_fn, n, d, is_mass_matrix_diagonal)
        schedule = build_schedule(int(num_steps)) if _schedule_fn is None else _as_schedule(
            _schedule_fn(int(num_steps)), int(num_steps))
        for t, (stage, is_window_end) in enumerate(schedule):
            # one_step: staged_adaptation.py:731-754
            imm_arg = ws.inverse_mass_matrix
            if is_mass_matrix_diagonal and imm_arg.ndim == 2:  # per-chain diagonals, not a dense matrix
                imm_arg = metrics.PerChainDiag(imm_arg)
            state, info = mcmc_kernel(ChainMajorKey(run_key, t), state, logdensity_fn, ws.step_size,
                                      imm_arg, chain_offset=chain_offset, **step_extra)
            imm_state = ws.imm_state
            if stage == 1:  # slow_update: staged_adaptation.py:200-231
                imm_state = MassMatrixAdaptationState(
                    imm_state.inverse_mass_matrix,
                    _welford_update(imm_state.wc_state, state.position, in_place=welford_in_place))
            ss, step_size = _da_update(ws.ss_state, info.acceptance_rate, target_acceptance_rate)
            ws = StagedAdaptationState(ss, imm_state, step_size, imm_state.inverse_mass_matrix)
            if is_window_end:  # slow_final: staged_adaptation.py:233-249
                imm_state = _mm_final(ws.imm_state, imm_shrinkage_to_previous)
                ss, step_size = _da_init(ws.ss_state.log_step_size_avg, from_log_avg=True)
                ws = StagedAdaptationState(ss, imm_state, step_size, imm_state.inverse_mass_matrix)
            if info_fn is not None:
                history.append(info_fn(state, info, ws))
        # final: staged_adaptation.py:299-305
        step_size = torch.empty_like(ws.ss_state.log_step_size_avg)
        _lib.call("bjx_exp", _lib.current_stream(), n, ws.ss_state.log_step_size_avg.data_ptr(),
                  step_size.data_ptr())
        imm_final = ws.imm_state.inverse_mass_matrix
        if imm_final.ndim == 1:  # fewer than 20 steps: no window ever ended
            imm_final = imm_final.expand(n, d).contiguous()
        elif not is_mass_matrix_diagonal and imm_final.ndim == 2:
            imm_final = imm_final.expand(n, d, d).contiguous()
        if is_mass_matrix_diagonal:  # (N, D) per-chain diagonals, tagged so N == D is not read as dense
            imm_final = metrics.PerChainDiagTensor.tag(imm_final)
        parameters = {"step_size": step_size, "inverse_mass_matrix": imm_final, **extra_parameters}
        return AdaptationResults(state, parameters), _stack_history(history)

    return AdaptationAlgorithm(run)


def _as_schedule(sched, num_steps: int) -> list:
    """A ``schedule_fn`` result -- ``(num_steps, 2)`` array-like of ``(stage, is_window_end)`` as in
    staged_adaptation.py:315-405 -- as the list of pairs the engine loops over."""
    rows = [(int(a), bool(b)) for a, b in np.asarray(sched).reshape(-1, 2).tolist()]
    if len(rows) != num_steps or any(st not in (0, 1) for st, _ in rows):
        raise ValueError("schedule_fn(num_steps) must return num_steps rows of (stage in {0, 1}, is_window_end)")
    return rows


_RECIPES = {"welford_diag": True, "welford_dense": False}


def staged_adaptation(algorithm, logdensity_fn: Callable, metric: str = "welford_diag", *,
                      max_grad_budget=None, n_chains: int = 1, imm_shrinkage_to_previous: float = 0.0,
                      initial_inverse_mass_matrix=None, initial_step_size: float = 1.0,
                      target_acceptance_rate: float = 0.80,
                      adaptation_info_fn: Optional[Callable] = _default_adapt_info,
                      integrator=integrators.velocity_verlet, schedule_fn: Optional[Callable] = None,
                      initial_metric_state=None, **extra_parameters) -> AdaptationAlgorithm:
    """The engine entry point of blackjax/adaptation/staged_adaptation.py:519-983, of which
    ``window_adaptation`` is the compatibility shim (window_adaptation.py:427-444).

    ``metric``: the registry names of the Welford recipes, ``"welford_diag"`` (default) and
    ``"welford_dense"`` (adaptation/metric_recipes.py:961-989); ``schedule_fn(num_steps)`` replaces
    the Stan schedule exactly as in the reference (an explicit callable is always honoured).
    As everywhere in this engine the chain axis is native: ``run(rng_key, position)`` takes ``(N, D)``
    positions and adapts every chain on its own (the reference's ``n_chains = 1`` path under ``vmap``),
    so ``n_chains`` must stay 1.

    Out of scope: ``metric="auto"`` -- the experimental
    meta-adaptation controller with low-rank escalation and its multi-chain pooled gate
    (adaptation/meta/, ~3 200 lines) --, ``"fisher_diag"``, ``MetricCore`` / ``MetricRecipe`` objects,
    ``initial_metric_state``.  For thousands of chains the pooled warm-up this engine offers is
    ``chees_adaptation`` (what the reference recommends for that regime,
    docs/examples/howto_sample_multiple_chains.md:246)."""
    if n_chains < 1:
        raise ValueError(f"staged_adaptation: n_chains must be >= 1, got {n_chains}.")
    if n_chains > 1 and metric != "auto":
        raise ValueError(
            "staged_adaptation: n_chains > 1 is only supported with metric='auto' "
            "(the multi-chain pooled gate is implemented in the meta-adaptation "
            "controller). For other metric strings pass n_chains=1 (default) and "
            "vmap the warmup call externally.")
    if not isinstance(metric, str):  # staged_adaptation.py:509-515 (this engine has no MetricRecipe / MetricCore objects)
        raise TypeError(
            f"staged_adaptation: metric must be a str, MetricRecipe, or MetricCore "
            f"(got {type(metric).__name__}). "
            f"Pass a registry name (e.g. 'welford_diag') or construct a "
            f"MetricRecipe or MetricCore directly.")
    if metric not in _RECIPES:
        raise NotImplementedError(
            f"staged_adaptation(metric={metric!r}): this engine builds the Welford recipes "
            f"{sorted(_RECIPES)}; 'auto' (experimental meta-adaptation), 'fisher_diag' and MetricCore / "
            "MetricRecipe objects are outside the hot path it replaces (SURVEY.md section 2)")
    if initial_metric_state is not None or max_grad_budget is not None:
        raise NotImplementedError("initial_metric_state / max_grad_budget belong to the meta-adaptation path")
    return window_adaptation(algorithm, logdensity_fn, is_mass_matrix_diagonal=_RECIPES[metric],
                             initial_inverse_mass_matrix=initial_inverse_mass_matrix,
                             imm_shrinkage_to_previous=imm_shrinkage_to_previous,
                             initial_step_size=initial_step_size,
                             target_acceptance_rate=target_acceptance_rate,
                             adaptation_info_fn=adaptation_info_fn, integrator=integrator,
                             _schedule_fn=schedule_fn, **extra_parameters)
