"""Batched dynamic HMC (random trajectory length per chain and transition) behind the
``blackjax.dynamic_hmc`` API surface (blackjax/mcmc/dynamic_hmc.py).

Mirrors ``DynamicHMCState`` (39-52), ``init`` (55-61), ``build_kernel`` (65-126) and
``as_top_level_api`` (129-223).  Every chain carries its own ``random_generator_arg`` (one threefry
key per chain, ``(N, 2)`` uint32 stored as int32 bit patterns on the device), draws its own number
of integration steps from it and advances it, exactly as the vmapped reference does.

The transition itself is ``blackjax_amd.hmc``'s with a per-chain trajectory length: the leapfrog
kernel takes the ``(N,)`` step counts and leaves finished chains untouched
(``bjx_leapfrog_diag_masked``); the host loops to the longest trajectory of the batch.
"""
from __future__ import annotations

from typing import Callable, NamedTuple

import numpy as np
import torch

from . import _lib, integrators, metrics
from ._util import check_batch, eval_logdensity, step_size_args, value_and_grad
from .base import SamplingAlgorithm
from .hmc import HMCInfo, IntegratorState
from .random import key_spec

__all__ = ["DynamicHMCState", "init", "build_kernel", "as_top_level_api", "chain_keys",
           "next_key_fn", "randint_steps_fn", "halton_sequence", "halton_steps_fn", "rescale", "halton_trajectory_length"]


class DynamicHMCState(NamedTuple):
    """blackjax/mcmc/dynamic_hmc.py:39-52, batched."""

    position: torch.Tensor
    logdensity: torch.Tensor
    logdensity_grad: torch.Tensor
    random_generator_arg: torch.Tensor  # (N, 2) int32: per-chain threefry key words (bit patterns)


def chain_keys(rng_key, n_chains: int, device, chain_offset: int = 0) -> torch.Tensor:
    """``jax.random.split(rng_key, N)`` as a device tensor: the usual way to seed
    ``random_generator_arg`` for N chains."""
    from . import random as bjx_random

    k = bjx_random.split(rng_key, n_chains, offset=chain_offset).view(np.int32)
    return torch.as_tensor(k, device=device).contiguous()


def next_key_fn(keys: torch.Tensor) -> torch.Tensor:
    """Default ``next_random_arg_fn``: ``lambda key: jax.random.split(key)[1]`` per chain
    (dynamic_hmc.py:69)."""
    out = torch.empty_like(keys)
    _lib.call("bjx_keys_child", _lib.current_stream(), keys.shape[0], keys.data_ptr(), 1,
              out.data_ptr())
    return out


def randint_steps_fn(keys: torch.Tensor, minval: int = 1, maxval: int = 10) -> torch.Tensor:
    """Default ``integration_steps_fn``: ``lambda key: jax.random.randint(key, (), 1, 10)`` per chain
    (dynamic_hmc.py:70); extra ``integration_steps_params`` replace the bounds."""
    out = torch.empty(keys.shape[0], dtype=torch.int32, device=keys.device)
    _lib.call("bjx_keys_randint", _lib.current_stream(), keys.shape[0], keys.data_ptr(), int(minval),
              int(maxval), out.data_ptr())
    return out


def halton_sequence(i: int, max_bits: int = 10) -> np.float32:
    """blackjax/mcmc/dynamic_hmc.py:205-215 for a host integer: the ``(i+1)``-th element of the
    base-2 Halton sequence over ``max_bits`` bits (exact in fp32 for ``max_bits <= 24``)."""
    max_bits = int(max_bits)
    if max_bits >= 32:
        raise ValueError(f"max_bits ({max_bits}) must be less than bit width of dtype int32 (32)")
    v = np.float32(0.0)
    for k in range(max_bits):
        if ((int(i) + 1) >> k) & 1:
            v = np.float32(v + np.float32(0.5 / (1 << k)))
    return v


def rescale(mu) -> np.float32:
    """blackjax/mcmc/adjusted_mclmc.py:281-288 (imported by dynamic_hmc.py:23): ``s`` such that
    ``round(U(0, 1) * s + 0.5)`` has expected value ``mu`` (fp32, as the reference computes it)."""
    mu = np.float32(mu)
    k = np.floor(np.float32(2.0) * mu - np.float32(1.0))
    x = k * (mu - np.float32(0.5) * (k + np.float32(1.0))) / (k + np.float32(1.0) - mu)
    return np.float32(k + x)


def halton_trajectory_length(i: int, trajectory_length_adjustment: float, max_bits: int = 10) -> int:
    """blackjax/mcmc/dynamic_hmc.py:218-223 for a host integer: a quasi-random number of integration steps with mean
    ``trajectory_length_adjustment`` -- ``rint(0.5 + halton_sequence(i) * rescale(adjustment))`` (round half to even,
    as ``jnp.rint``)."""
    s = rescale(trajectory_length_adjustment)
    return int(np.rint(np.float32(np.float32(0.5) + np.float32(halton_sequence(i, max_bits) * s))))


def halton_steps_fn(max_bits: int, jitter_amount: float = 1.0) -> Callable:
    """``integration_steps_fn`` of ChEES-HMC (chees_adaptation.py:762-771): per chain
    ``ceil((halton(arg) * jitter_amount + (1 - jitter_amount)) * num_leapfrog_steps)`` from an integer
    counter ``random_generator_arg`` of shape ``(N,)`` (int32, device)."""
    max_bits = int(max_bits)
    if max_bits >= 32:
        raise ValueError(f"max_bits ({max_bits}) must be less than bit width of dtype int32 (32)")
    ja, jb = float(np.float32(jitter_amount)), float(np.float32(1.0 - jitter_amount))

    def steps_fn(random_generator_arg: torch.Tensor, num_leapfrog_steps: float) -> torch.Tensor:
        arg = random_generator_arg
        if arg.ndim != 1 or arg.dtype != torch.int32 or not arg.is_cuda:
            raise ValueError("random_generator_arg must be a device (n_chains,) int32 counter tensor")
        out = torch.empty_like(arg)
        _lib.call("bjx_halton_steps", _lib.current_stream(), arg.shape[0], arg.contiguous().data_ptr(),
                  max_bits, ja, jb, float(num_leapfrog_steps), out.data_ptr())
        return out

    return steps_fn


def init(position: torch.Tensor, logdensity_fn: Callable, random_generator_arg: torch.Tensor):
    """blackjax/mcmc/dynamic_hmc.py:55-61.  ``random_generator_arg`` is per chain: ``(N, 2)`` int32
    key words (the default key-driven callables) or an ``(N,)`` int32 counter (Halton jitter)."""
    position = check_batch(position, "position")
    logp, grad = eval_logdensity(value_and_grad(logdensity_fn), position)
    rga = random_generator_arg
    n = position.shape[0]
    if (not isinstance(rga, torch.Tensor) or rga.dtype != torch.int32 or not rga.is_cuda
            or rga.shape not in ((n, 2), (n,))):
        raise ValueError("random_generator_arg must be a device int32 tensor: (n_chains, 2) key words "
                         "(see dynamic_hmc.chain_keys) or an (n_chains,) counter")
    return DynamicHMCState(position, logp, grad, rga.contiguous())


def build_kernel(integrator=integrators.velocity_verlet, divergence_threshold: float = 1000,
                 next_random_arg_fn: Callable = next_key_fn,
                 integration_steps_fn: Callable = randint_steps_fn, build_proposal=None):
    """blackjax/mcmc/dynamic_hmc.py:65-126.  ``integration_steps_fn(random_generator_arg, *params)``
    returns an ``(N,)`` int32 device tensor of trajectory lengths (>= 1)."""
    # any palindromic coefficient list (integrators.py:62-152; dynamic_hmc.py:65-71 takes `integrator=`)
    integrators.check_supported(integrator, allow_general=True)
    general = integrator is not integrators.velocity_verlet
    kick_c = integrator.coefficients[0::2]   # b1 .. b1
    drift_c = integrator.coefficients[1::2]  # a1 ..
    from .hmc import hmc_proposal, multinomial_hmc_proposal

    if build_proposal not in (None, hmc_proposal, multinomial_hmc_proposal):
        raise NotImplementedError("dynamic_hmc: build_proposal must be hmc_proposal or multinomial_hmc_proposal")
    thr = float(divergence_threshold)
    if build_proposal is multinomial_hmc_proposal:
        return _build_multinomial_kernel(thr, next_random_arg_fn, integration_steps_fn, kick_c, drift_c)

    def kernel(rng_key, state: DynamicHMCState, logdensity_fn: Callable, step_size,
               inverse_mass_matrix, integration_steps_params: tuple = (), *, chain_offset: int = 0):
        q0 = check_batch(state.position, "state.position")
        logp0 = check_batch(state.logdensity, "state.logdensity")
        g0 = check_batch(state.logdensity_grad, "state.logdensity_grad")
        N, D = q0.shape
        dev = q0.device
        k0, k1, fold = key_spec(rng_key)
        vg = value_and_grad(logdensity_fn)
        metric = metrics.default_metric(inverse_mass_matrix, N, D, dev)
        eps, eps_pc = step_size_args(step_size, N, dev)
        stream = _lib.current_stream()
        off = int(chain_offset)
        is_diag = metric.kind == "diag"
        if is_diag:
            imm_p, imm_s = metric.imm.data_ptr(), metric.imm_stride
        else:
            from . import dense

        n_steps = integration_steps_fn(state.random_generator_arg, *integration_steps_params)
        n_steps = n_steps.to(device=dev, dtype=torch.int32).contiguous()
        lo, hi = int(n_steps.min()), int(n_steps.max())  # one host sync per transition
        if lo < 1:
            raise ValueError("integration_steps_fn must return at least 1 step for every chain")

        p0 = torch.empty_like(q0)
        ke0 = torch.empty_like(logp0)
        if is_diag:
            _lib.call("bjx_hmc_momentum_diag", stream, k0, k1, off, fold, N, D, imm_p, imm_s,
                      p0.data_ptr(), ke0.data_ptr())
        else:
            dense.momentum(stream, metric, k0, k1, off, fold, N, D, p0, ke0)
        q, p = torch.empty_like(q0), torch.empty_like(q0)
        # every chain integrates the same number of steps (e.g. a shared Halton counter after ChEES
        # warmup): the plain, unmasked kernels; otherwise chains with n_steps <= l are skipped (their q
        # is unchanged, so the callable keeps returning the same (logp, g) for them)
        ns = None if lo == hi else n_steps

        def stage(n_k, ka, kb, a_c, l, q_in, p_in, g_in, p_out):
            """One position update: kicks (eps*ka) g [, (eps*kb) g], drift (eps*a_c) M^{-1} p, for the
            chains whose trajectory has more than ``l`` steps."""
            if not is_diag:  # dense metric: kick + GEMM / mat-vec + drift, masked by the chain's length
                return dense.leapfrog_coef(stream, metric, N, D, n_k, ka, kb, a_c, eps, eps_pc, q_in,
                                           p_in, g_in, q, p_out, ns, l)
            if general:
                _lib.call("bjx_leapfrog_diag_coef", stream, N, D, n_k, ka, kb, a_c, eps, _lib.ptr(eps_pc),
                          imm_p, imm_s, q_in.data_ptr(), p_in.data_ptr(), g_in.data_ptr(), q.data_ptr(),
                          p_out.data_ptr(), _lib.ptr(ns), l)
            elif ns is None:
                _lib.call("bjx_leapfrog_diag", stream, N, D, n_k, eps, _lib.ptr(eps_pc), imm_p, imm_s,
                          q_in.data_ptr(), p_in.data_ptr(), g_in.data_ptr(), q.data_ptr(), p_out.data_ptr())
            else:
                _lib.call("bjx_leapfrog_diag_masked", stream, N, D, n_k, eps, _lib.ptr(eps_pc), imm_p,
                          imm_s, q_in.data_ptr(), p_in.data_ptr(), g_in.data_ptr(), q.data_ptr(),
                          p_out.data_ptr(), ns.data_ptr(), l)
            return p_out

        # generalized_two_stage_integrator (integrators.py:104-150): one launch per position update; the
        # closing kick b_K of a step merges with the opening kick b_1 of the next (two separately
        # rounded fmas); a chain that has finished keeps its state -- its last closing kick is the
        # finish kernel's, as for velocity Verlet
        first = True
        for l in range(hi):
            for si, a_c in enumerate(drift_c):
                if first:
                    p = stage(1, kick_c[0], 0.0, a_c, 0, q0, p0, g0, p)
                    first = False
                elif si == 0:
                    p = stage(2, kick_c[-1], kick_c[0], a_c, l, q, p, g, p)
                else:
                    p = stage(1, kick_c[si], 0.0, a_c, l, q, p, g, p)
                logp, g = eval_logdensity(vg, q)

        p_end, q_new, g_new = torch.empty_like(q0), torch.empty_like(q0), torch.empty_like(q0)
        logp_new, acc_rate, energy = (torch.empty_like(logp0) for _ in range(3))
        is_acc = torch.empty(N, dtype=torch.bool, device=dev)
        is_div = torch.empty(N, dtype=torch.bool, device=dev)
        if is_diag and general:
            _lib.call("bjx_hmc_finish_diag_coef", stream, k0, k1, off, fold, N, D, kick_c[-1], eps,
                      _lib.ptr(eps_pc), imm_p, imm_s, thr, q0.data_ptr(), logp0.data_ptr(), g0.data_ptr(),
                      ke0.data_ptr(), q.data_ptr(), logp.data_ptr(), g.data_ptr(), p.data_ptr(),
                      p_end.data_ptr(), q_new.data_ptr(), logp_new.data_ptr(), g_new.data_ptr(),
                      acc_rate.data_ptr(), is_acc.data_ptr(), is_div.data_ptr(), energy.data_ptr())
        elif is_diag:
            _lib.call("bjx_hmc_finish_diag", stream, k0, k1, off, fold, N, D, eps, _lib.ptr(eps_pc), imm_p,
                      imm_s, thr, q0.data_ptr(), logp0.data_ptr(), g0.data_ptr(), ke0.data_ptr(),
                      q.data_ptr(), logp.data_ptr(), g.data_ptr(), p.data_ptr(), p_end.data_ptr(),
                      q_new.data_ptr(), logp_new.data_ptr(), g_new.data_ptr(), acc_rate.data_ptr(),
                      is_acc.data_ptr(), is_div.data_ptr(), energy.data_ptr())
        elif general:
            dense.finish_coef(stream, metric, k0, k1, off, fold, N, D, kick_c[-1], eps, eps_pc, thr, q0, logp0,
                              g0, ke0, q, logp, g, p, p_end, q_new, logp_new, g_new, acc_rate, is_acc, is_div,
                              energy)
        else:
            dense.finish(stream, metric, k0, k1, off, fold, N, D, eps, eps_pc, thr, q0, logp0, g0, ke0, q,
                         logp, g, p, p_end, q_new, logp_new, g_new, acc_rate, is_acc, is_div, energy)
        info = HMCInfo(p0, acc_rate, is_acc, is_div, energy, IntegratorState(q, p_end, logp, g),
                       n_steps)
        new_arg = next_random_arg_fn(state.random_generator_arg)
        return DynamicHMCState(q_new, logp_new, g_new, new_arg), info

    return kernel


def _build_multinomial_kernel(thr: float, next_random_arg_fn: Callable, integration_steps_fn: Callable,
                              kick_c=(0.5, 0.5), drift_c=(1.0,)):
    """blackjax.dmhmc (blackjax/__init__.py:155-163): every chain draws its own trajectory length and one
    state of ITS trajectory proportionally to exp(-H) (hmc.py:181-248 over dynamic_hmc.py:85-118).  The
    per-chain lengths mask the fused step kernel of ``blackjax_amd.mhmc`` (diagonal metric) or, with a dense
    metric (one shared matrix: MFMA GEMMs; one per chain: fp64 matrix-vector kernels), the
    masked dense leapfrog and ``bjx_mhmc_step_dense_masked`` / ``_coef`` (any palindromic integrator since round 4)."""

    def kernel(rng_key, state: DynamicHMCState, logdensity_fn: Callable, step_size,
               inverse_mass_matrix, integration_steps_params: tuple = (), *, chain_offset: int = 0):
        q0 = check_batch(state.position, "state.position")
        logp0 = check_batch(state.logdensity, "state.logdensity")
        g0 = check_batch(state.logdensity_grad, "state.logdensity_grad")
        N, D = q0.shape
        dev = q0.device
        k0, k1, fold = key_spec(rng_key)
        vg = value_and_grad(logdensity_fn)
        metric = metrics.default_metric(inverse_mass_matrix, N, D, dev)
        is_diag = metric.kind == "diag"
        eps, eps_pc = step_size_args(step_size, N, dev)
        stream = _lib.current_stream()
        off = int(chain_offset)
        imm_p, imm_s = metric.imm.data_ptr(), metric.imm_stride
        n_steps = integration_steps_fn(state.random_generator_arg, *integration_steps_params)
        n_steps = n_steps.to(device=dev, dtype=torch.int32).contiguous()
        lo, hi = int(n_steps.min()), int(n_steps.max())  # one host sync per transition
        if lo < 1:
            raise ValueError("integration_steps_fn must return at least 1 step for every chain")
        p0 = torch.empty_like(q0)
        ke0 = torch.empty_like(logp0)
        if is_diag:
            _lib.call("bjx_hmc_momentum_diag", stream, k0, k1, off, fold, N, D, imm_p, imm_s, p0.data_ptr(),
                      ke0.data_ptr())
        else:
            from . import dense

            dense.momentum(stream, metric, k0, k1, off, fold, N, D, p0, ke0)
        weight = torch.zeros_like(logp0)
        slpa = torch.full_like(logp0, float("-inf"))
        any_div = torch.zeros(N, dtype=torch.bool, device=dev)
        ever = torch.zeros(N, dtype=torch.bool, device=dev)
        pq, pp, pg = torch.empty_like(q0), torch.empty_like(q0), torch.empty_like(q0)
        plogp, penergy, acc_rate = torch.empty_like(logp0), torch.empty_like(logp0), torch.empty_like(logp0)
        q, p = torch.empty_like(q0), torch.empty_like(q0)
        b1, a1 = float(kick_c[0]), float(drift_c[0])
        if not is_diag:
            # opening kick + drift, callable, closing kick + reservoir step; the next leapfrog starts from
            # the fully kicked momentum with its own opening kick (blackjax_amd.hmc, dense branch) -- every
            # launch masked by the chain's own length
            # (any palindromic integrator since round 4: opening (b1, a1), stages in between, closing kick b1)
            general = tuple(kick_c) != (0.5, 0.5) or tuple(drift_c) != (1.0,)
            p_half = dense.leapfrog_coef(stream, metric, N, D, 1, b1, 0.0, a1, eps, eps_pc, q0, p0, g0, q, p)
            for i in range(hi):
                logp, g = eval_logdensity(vg, q)  # finished chains keep their q: same (logp, g) again, unused
                for si in range(1, len(drift_c)):  # stages 2 .. K, masked by the chain's own length
                    p_half = dense.leapfrog_coef(stream, metric, N, D, 1, float(kick_c[si]), 0.0, float(drift_c[si]),
                                                 eps, eps_pc, q, p_half, g, q, torch.empty_like(q0), n_steps, i)
                    logp, g = eval_logdensity(vg, q)
                p1 = dense.mhmc_step(stream, metric, k0, k1, off, fold, N, D, i, eps, eps_pc, thr, logp0, ke0,
                                     q, p_half, g, logp, weight, slpa, any_div, ever, pq, pp, pg, plogp,
                                     penergy, n_steps=n_steps, kick_coef=b1 if general else None)
                if i + 1 < hi:
                    p_half = dense.leapfrog_coef(stream, metric, N, D, 1, b1, 0.0, a1, eps, eps_pc, q, p1, g,
                                                 q, torch.empty_like(q0) if general else p_half, n_steps, i + 1)
        else:
            _lib.call("bjx_leapfrog_diag_coef", stream, N, D, 1, b1, 0.0, a1, eps, _lib.ptr(eps_pc), imm_p, imm_s,
                      q0.data_ptr(), p0.data_ptr(), g0.data_ptr(), q.data_ptr(), p.data_ptr(), None, 0)
        for i in range(hi if is_diag else 0):
            logp, g = eval_logdensity(vg, q)  # finished chains keep their q: same (logp, g) again, unused
            for si in range(1, len(drift_c)):  # stages 2 .. K of a multi-stage integrator, masked by length
                _lib.call("bjx_leapfrog_diag_coef", stream, N, D, 1, float(kick_c[si]), 0.0, float(drift_c[si]),
                          eps, _lib.ptr(eps_pc), imm_p, imm_s, q.data_ptr(), p.data_ptr(), g.data_ptr(),
                          q.data_ptr(), p.data_ptr(), n_steps.data_ptr(), i)
                logp, g = eval_logdensity(vg, q)
            _lib.call("bjx_mhmc_step_diag_coef", stream, k0, k1, off, fold, N, D, i, 1 if i + 1 < hi else 0,
                      eps, _lib.ptr(eps_pc), imm_p, imm_s, thr, logp0.data_ptr(), ke0.data_ptr(), q.data_ptr(),
                      p.data_ptr(), g.data_ptr(), logp.data_ptr(), weight.data_ptr(), slpa.data_ptr(),
                      any_div.data_ptr(), ever.data_ptr(), pq.data_ptr(), pp.data_ptr(), pg.data_ptr(),
                      plogp.data_ptr(), penergy.data_ptr(), n_steps.data_ptr(), b1, a1)
        _lib.call("bjx_mhmc_finish_masked", stream, N, D, n_steps.data_ptr(), q0.data_ptr(), p0.data_ptr(),
                  g0.data_ptr(), logp0.data_ptr(), ke0.data_ptr(), ever.data_ptr(), slpa.data_ptr(),
                  pq.data_ptr(), pp.data_ptr(), pg.data_ptr(), plogp.data_ptr(), penergy.data_ptr(),
                  acc_rate.data_ptr())
        info = HMCInfo(p0, acc_rate, torch.ones(N, dtype=torch.bool, device=dev), any_div, penergy,
                       IntegratorState(pq, pp, plogp, pg), n_steps)
        new_arg = next_random_arg_fn(state.random_generator_arg)
        return DynamicHMCState(pq, plogp, pg, new_arg), info

    return kernel


def as_top_level_api(logdensity_fn: Callable, step_size, inverse_mass_matrix, *,
                     divergence_threshold: int = 1000, integrator=integrators.velocity_verlet,
                     next_random_arg_fn: Callable = next_key_fn,
                     integration_steps_fn: Callable = randint_steps_fn,
                     integration_steps_params: tuple = (), build_proposal=None,
                     chain_offset: int = 0) -> SamplingAlgorithm:
    """blackjax/mcmc/dynamic_hmc.py:129-223."""
    kernel = build_kernel(integrator, divergence_threshold, next_random_arg_fn, integration_steps_fn,
                          build_proposal)

    def init_fn(position, rng_key):
        # build_sampling_algorithm forwards `rng_key` as the random_generator_arg seed
        # (dynamic_hmc.py:215-222): one key per chain = split(rng_key, N)
        return init(position, logdensity_fn, chain_keys(rng_key, position.shape[0], position.device,
                                                        chain_offset))

    def step_fn(rng_key, state):
        return kernel(rng_key, state, logdensity_fn, step_size, inverse_mass_matrix,
                      integration_steps_params, chain_offset=chain_offset)

    return SamplingAlgorithm(init_fn, step_fn)
