"""Batched NUTS on MI355X behind the ``blackjax.nuts`` API surface.

Mirrors blackjax/mcmc/nuts.py: ``NUTSInfo`` (36-74), ``build_kernel`` (77-147),
``as_top_level_api`` (150-220), ``iterative_nuts_proposal`` (223-321).  Chain ``i`` of
``step(rng_key, state)`` reproduces the reference's single-chain
``step(jax.random.split(rng_key, N)[i], state_i)``.

Scheduling (the reference's ``vmap`` makes every chain wait for the slowest one,
docs/examples/howto_sample_multiple_chains.md:152): chains advance in lockstep -- all running
chains are always at the same (doubling, leaf) position -- with ACTIVE-CHAIN COMPACTION: the
user's log-density callable and the kernels only see the chains that are still building a tree
(re-compacted at every doubling and every ``recompact_every`` leapfrogs inside a doubling).
All tree arithmetic (progressive sampling, momentum sums, U-turn checkpoints, merge) runs in
``bjx_nuts.hip``.
"""
from __future__ import annotations

import ctypes
from typing import Callable, NamedTuple

import torch

from . import _lib, integrators, metrics
from ._util import check_batch, eval_logdensity, step_size_args, value_and_grad
from .base import SamplingAlgorithm
from .hmc import HMCState, IntegratorState, init
from .random import key_spec

__all__ = ["NUTSInfo", "init", "build_kernel", "as_top_level_api"]


class NUTSInfo(NamedTuple):
    """blackjax/mcmc/nuts.py:36-74, batched."""

    momentum: torch.Tensor
    is_divergent: torch.Tensor
    is_turning: torch.Tensor
    energy: torch.Tensor
    trajectory_leftmost_state: IntegratorState
    trajectory_rightmost_state: IntegratorState
    num_trajectory_expansions: torch.Tensor
    num_integration_steps: torch.Tensor
    acceptance_rate: torch.Tensor


def build_kernel(integrator=integrators.velocity_verlet, divergence_threshold: int = 1000, *,
                 recompact_every: int = 16):
    """blackjax/mcmc/nuts.py:77-147."""
    integrators.check_supported(integrator)
    thr = float(divergence_threshold)
    F, I = _lib.NUTS_F, _lib.NUTS_I

    def kernel(rng_key, state: HMCState, logdensity_fn: Callable, step_size, inverse_mass_matrix,
               max_num_doublings: int = 10, *, chain_offset: int = 0):
        """One NUTS transition for all chains (nuts.py:113-145 + iterative_nuts_proposal 278-319)."""
        q0 = check_batch(state.position, "state.position")
        logp0 = check_batch(state.logdensity, "state.logdensity")
        g0 = check_batch(state.logdensity_grad, "state.logdensity_grad")
        N, D = q0.shape
        dev = q0.device
        max_depth = int(max_num_doublings)
        k0, k1, fold = key_spec(rng_key)
        vg = value_and_grad(logdensity_fn)
        metric = metrics.default_metric(inverse_mass_matrix, N, D, dev)
        if metric.kind != "diag":
            raise NotImplementedError("NUTS with a dense mass matrix is not implemented yet")
        eps, eps_pc = step_size_args(step_size, N, dev)
        stream = _lib.current_stream()
        off = int(chain_offset)

        p0 = torch.empty_like(q0)
        ke0 = torch.empty_like(logp0)
        _lib.call("bjx_hmc_momentum_diag", stream, k0, k1, off, fold, N, D, metric.imm.data_ptr(),
                  metric.imm_stride, p0.data_ptr(), ke0.data_ptr())

        names = ["Lq", "Lp", "Lg", "Rq", "Rp", "Rg", "msum", "Smsum", "Pq", "Pg", "Sq", "Sg"]
        bufs = {n: torch.empty_like(q0) for n in names}
        ck_r = torch.empty((N, max(max_depth, 1), D), dtype=torch.float32, device=dev)
        ck_rs = torch.empty_like(ck_r)
        fs = torch.empty((_lib.NUTS_NF, N), dtype=torch.float32, device=dev)
        is_ = torch.empty((_lib.NUTS_NI, N), dtype=torch.int32, device=dev)
        desc = _lib.NutsDesc(
            N=N, D=D, max_depth=max_depth, reserved=0, imm=metric.imm.data_ptr(),
            imm_stride=metric.imm_stride, eps_per_chain=_lib.ptr(eps_pc), eps=eps,
            divergence_threshold=thr, key0=k0, key1=k1, chain_offset=off, step_fold=fold,
            q0=q0.data_ptr(), g0=g0.data_ptr(), p0=p0.data_ptr(),
            ckpt_r=ck_r.data_ptr(), ckpt_rs=ck_rs.data_ptr(), fs=fs.data_ptr(), is_=is_.data_ptr(),
            **{n: b.data_ptr() for n, b in bufs.items()})
        dref = ctypes.byref(desc)
        _lib.call("bjx_nuts_init", stream, dref, logp0.data_ptr(), ke0.data_ptr())

        idx_doubling = None  # None = all chains, compact row b == chain b
        n_doubling = N
        for depth in range(max_depth):
            if depth > 0:
                idx_doubling = torch.nonzero(is_[I["ACTIVE"]], as_tuple=False).flatten().to(torch.int32)
                n_doubling = int(idx_doubling.shape[0])  # host sync, once per doubling
                if n_doubling == 0:
                    break
            idx_step, n_step = idx_doubling, n_doubling
            qf = torch.empty((n_step, D), dtype=torch.float32, device=dev)
            n_leaves = 1 << depth
            for s in range(n_leaves):
                if s > 0 and recompact_every and s % recompact_every == 0:
                    # drop the chains whose subtree has stopped (diverged / turned)
                    sub = is_[I["SUB_ACTIVE"]]
                    alive = sub.bool() if idx_step is None else sub[idx_step.long()].bool()
                    if idx_step is None:
                        new_idx = torch.nonzero(alive, as_tuple=False).flatten().to(torch.int32)
                    else:
                        new_idx = idx_step[alive]
                    n_new = int(new_idx.shape[0])  # host sync
                    if n_new == 0:
                        break
                    if n_new < n_step:
                        idx_step, n_step = new_idx.contiguous(), n_new
                        qf = qf[:n_step]
                _lib.call("bjx_nuts_pre", stream, dref, depth, s, n_step, _lib.ptr(idx_step),
                          qf.data_ptr())
                logp_f, gf = eval_logdensity(vg, qf)
                _lib.call("bjx_nuts_post", stream, dref, depth, s, n_step, _lib.ptr(idx_step),
                          qf.data_ptr(), logp_f.data_ptr(), gf.data_ptr())
            _lib.call("bjx_nuts_merge", stream, dref, depth, n_doubling, _lib.ptr(idx_doubling))

        info = NUTSInfo(
            p0,
            is_[I["DIV"]].bool(),
            is_[I["TURN"]].bool(),
            fs[F["PENERGY"]],
            IntegratorState(bufs["Lq"], bufs["Lp"], fs[F["LLOGP"]], bufs["Lg"]),
            IntegratorState(bufs["Rq"], bufs["Rp"], fs[F["RLOGP"]], bufs["Rg"]),
            is_[I["DEPTH"]],
            is_[I["NSTATES"]],
            fs[F["ACC"]],
        )
        return HMCState(bufs["Pq"], fs[F["PLOGP"]], bufs["Pg"]), info

    return kernel


def as_top_level_api(logdensity_fn: Callable, step_size, inverse_mass_matrix, *,
                     max_num_doublings: int = 10, divergence_threshold: int = 1000,
                     integrator=integrators.velocity_verlet, chain_offset: int = 0,
                     recompact_every: int = 16) -> SamplingAlgorithm:
    """blackjax/mcmc/nuts.py:150-220."""
    kernel = build_kernel(integrator, divergence_threshold, recompact_every=recompact_every)

    def init_fn(position, rng_key=None):
        del rng_key
        return init(position, logdensity_fn)

    def step_fn(rng_key, state):
        return kernel(rng_key, state, logdensity_fn, step_size, inverse_mass_matrix,
                      max_num_doublings, chain_offset=chain_offset)

    return SamplingAlgorithm(init_fn, step_fn)
