"""Batched HMC on MI355X behind the ``blackjax.hmc`` API surface.

Mirrors blackjax/mcmc/hmc.py: ``HMCState`` (38-49), ``HMCInfo`` (52-87), ``init``
(90-92), ``build_kernel`` (251-314), ``as_top_level_api`` (317-414).  The chain
axis is native: every field carries a leading ``N``; chain ``i`` of
``step(rng_key, state)`` reproduces the reference's single-chain
``step(jax.random.split(rng_key, N)[i], state_i)`` (the vmap layout of
docs/examples/howto_sample_multiple_chains.md:120-127).

The arithmetic runs in libbjxhip's HIP kernels (include/bjx_hip.h); this module
only sequences launches around the user's PyTorch log-density callable.
"""
from __future__ import annotations

from typing import Callable, NamedTuple

import torch

from . import _lib, integrators, metrics
from ._util import (new_graph, record_graph, check_batch, eval_logdensity, is_capturable, step_size_args, value_and_grad,
                    warn_eager_driver)
from .base import SamplingAlgorithm
from .random import key_spec

__all__ = ["HMCState", "HMCInfo", "IntegratorState", "init", "flip_momentum", "build_kernel", "as_top_level_api"]


class HMCState(NamedTuple):
    """blackjax/mcmc/hmc.py:38-49, batched: (N, D), (N,), (N, D)."""

    position: torch.Tensor
    logdensity: torch.Tensor
    logdensity_grad: torch.Tensor


class IntegratorState(NamedTuple):
    """blackjax/mcmc/integrators.py:43-53, batched."""

    position: torch.Tensor
    momentum: torch.Tensor
    logdensity: torch.Tensor
    logdensity_grad: torch.Tensor


class HMCInfo(NamedTuple):
    """blackjax/mcmc/hmc.py:52-87, batched."""

    momentum: torch.Tensor
    acceptance_rate: torch.Tensor
    is_accepted: torch.Tensor
    is_divergent: torch.Tensor
    energy: torch.Tensor
    proposal: IntegratorState
    num_integration_steps: int


def flip_momentum(state: IntegratorState) -> IntegratorState:
    """blackjax/mcmc/hmc.py:95-112: the end state of a trajectory with its momentum negated (time reversibility).  The
    transition kernels do this inside their finishing launch (``k_hmc_finish_*``: the flipped momentum is what ``HMCInfo.
    proposal`` carries); this is the stand-alone function of the reference for code that builds on it."""
    return IntegratorState(state.position, -1.0 * state.momentum, state.logdensity, state.logdensity_grad)


def init(position: torch.Tensor, logdensity_fn: Callable) -> HMCState:
    """blackjax/mcmc/hmc.py:90-92."""
    position = check_batch(position, "position")
    if position.ndim != 2:
        raise ValueError(f"position must be (n_chains, dim), got {tuple(position.shape)}")
    logp, grad = eval_logdensity(value_and_grad(logdensity_fn), position)
    return HMCState(position, logp, grad)


_FUSE_FIRST = __import__("os").environ.get("BJX_HMC_FUSE_FIRST", "1") != "0"  # A/B switch (NOTEBOOK.md section 5)


def _launch_leapfrog(stream, metric, N, D, n_kicks, eps, eps_pc, q_in, p_in, g, q_out, p_out):
    if metric.kind == "diag":
        _lib.call("bjx_leapfrog_diag", stream, N, D, n_kicks, eps, _lib.ptr(eps_pc),
                  metric.imm.data_ptr(), metric.imm_stride, q_in.data_ptr(), p_in.data_ptr(),
                  g.data_ptr(), q_out.data_ptr(), p_out.data_ptr())
        return p_out
    from . import dense

    return dense.leapfrog(stream, metric, N, D, n_kicks, eps, eps_pc, q_in, p_in, g, q_out, p_out)


class _ProposalKind:
    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return self.name


#: selectors for ``build_kernel(build_proposal=...)`` (blackjax/mcmc/hmc.py:115-178, 181-248)
hmc_proposal = _ProposalKind("hmc_proposal")
multinomial_hmc_proposal = _ProposalKind("multinomial_hmc_proposal")


def _build_mhmc_kernel(thr: float, kick_c=(0.5, 0.5), drift_c=(1.0,)):
    """blackjax.mhmc: ``build_kernel(build_proposal=multinomial_hmc_proposal)`` (hmc.py:181-248 with
    trajectory.static_progressive_integration 170-232).  Instead of the trajectory end point, one
    state of the whole trajectory is drawn proportionally to exp(-H) by progressive (reservoir)
    sampling; there is no Metropolis rejection (``is_accepted`` is always True)."""

    def kernel(rng_key, state: HMCState, logdensity_fn: Callable, step_size,
               inverse_mass_matrix, num_integration_steps: int, *, chain_offset: int = 0):
        q0 = check_batch(state.position, "state.position")
        logp0 = check_batch(state.logdensity, "state.logdensity")
        g0 = check_batch(state.logdensity_grad, "state.logdensity_grad")
        N, D = q0.shape
        L = int(num_integration_steps)
        if L < 0:
            raise ValueError("num_integration_steps must be >= 0")
        k0, k1, fold = key_spec(rng_key)
        vg = value_and_grad(logdensity_fn)
        metric = metrics.default_metric(inverse_mass_matrix, N, D, q0.device)
        eps, eps_pc = step_size_args(step_size, N, q0.device)
        stream = _lib.current_stream()
        off = int(chain_offset)
        dev = q0.device
        is_diag = metric.kind == "diag"
        if is_diag:
            imm_p, imm_s = metric.imm.data_ptr(), metric.imm_stride

        p0 = torch.empty_like(q0)
        ke0 = torch.empty_like(logp0)
        if is_diag:
            _lib.call("bjx_hmc_momentum_diag", stream, k0, k1, off, fold, N, D, imm_p, imm_s,
                      p0.data_ptr(), ke0.data_ptr())
        else:
            from . import dense

            dense.momentum(stream, metric, k0, k1, off, fold, N, D, p0, ke0)
        weight = torch.zeros_like(logp0)
        slpa = torch.full_like(logp0, float("-inf"))
        any_div = torch.zeros(N, dtype=torch.bool, device=dev)
        ever = torch.zeros(N, dtype=torch.bool, device=dev)
        pq, pp, pg = torch.empty_like(q0), torch.empty_like(q0), torch.empty_like(q0)
        plogp, penergy = torch.empty_like(logp0), torch.empty_like(logp0)
        acc_rate = torch.empty_like(logp0)
        general = tuple(kick_c) != (0.5, 0.5) or tuple(drift_c) != (1.0,)
        if L > 0 and is_diag and general:
            # any palindromic integrator [b1, a1, ..., b1] (integrators.py:104-150): the fused step kernel
            # closes a step with (eps b1) g and opens the next with (eps b1) g, (eps a1) M^{-1} p; the
            # stages in between are plain kick + drift launches, each followed by the callable
            b1, a1 = float(kick_c[0]), float(drift_c[0])
            q, p = torch.empty_like(q0), torch.empty_like(q0)
            _lib.call("bjx_leapfrog_diag_coef", stream, N, D, 1, b1, 0.0, a1, eps, _lib.ptr(eps_pc), imm_p,
                      imm_s, q0.data_ptr(), p0.data_ptr(), g0.data_ptr(), q.data_ptr(), p.data_ptr(), None, 0)
            for i in range(L):
                logp, g = eval_logdensity(vg, q)
                for si in range(1, len(drift_c)):
                    _lib.call("bjx_leapfrog_diag_coef", stream, N, D, 1, float(kick_c[si]), 0.0,
                              float(drift_c[si]), eps, _lib.ptr(eps_pc), imm_p, imm_s, q.data_ptr(),
                              p.data_ptr(), g.data_ptr(), q.data_ptr(), p.data_ptr(), None, 0)
                    logp, g = eval_logdensity(vg, q)
                _lib.call("bjx_mhmc_step_diag_coef", stream, k0, k1, off, fold, N, D, i,
                          1 if i + 1 < L else 0, eps, _lib.ptr(eps_pc), imm_p, imm_s, thr,
                          logp0.data_ptr(), ke0.data_ptr(), q.data_ptr(), p.data_ptr(), g.data_ptr(),
                          logp.data_ptr(), weight.data_ptr(), slpa.data_ptr(), any_div.data_ptr(),
                          ever.data_ptr(), pq.data_ptr(), pp.data_ptr(), pg.data_ptr(),
                          plogp.data_ptr(), penergy.data_ptr(), None, b1, a1)
        elif L > 0 and general:
            # dense metric x any palindromic integrator (round 4): opening (b1, a1), the stages in between and the
            # re-opening of the next step are bjx_leapfrog_dense_coef launches (kick prologue + GEMM / mat-vec +
            # drift epilogue), the closing kick b1 + reservoir step is bjx_mhmc_step_dense_coef
            from . import dense

            b1, a1 = float(kick_c[0]), float(drift_c[0])
            q, p_half = torch.empty_like(q0), torch.empty_like(q0)
            p_half = dense.leapfrog_coef(stream, metric, N, D, 1, b1, 0.0, a1, eps, eps_pc, q0, p0, g0, q, p_half)
            for i in range(L):
                logp, g = eval_logdensity(vg, q)
                for si in range(1, len(drift_c)):
                    p_half = dense.leapfrog_coef(stream, metric, N, D, 1, float(kick_c[si]), 0.0, float(drift_c[si]),
                                                 eps, eps_pc, q, p_half, g, q, torch.empty_like(q0))
                    logp, g = eval_logdensity(vg, q)
                p1 = dense.mhmc_step(stream, metric, k0, k1, off, fold, N, D, i, eps, eps_pc, thr, logp0,
                                     ke0, q, p_half, g, logp, weight, slpa, any_div, ever, pq, pp, pg,
                                     plogp, penergy, kick_coef=b1)
                if i + 1 < L:
                    p_half = dense.leapfrog_coef(stream, metric, N, D, 1, b1, 0.0, a1, eps, eps_pc, q, p1, g, q,
                                                 torch.empty_like(q0))
        elif L > 0 and is_diag:
            q, p = torch.empty_like(q0), torch.empty_like(q0)
            _lib.call("bjx_leapfrog_diag", stream, N, D, 1, eps, _lib.ptr(eps_pc), imm_p, imm_s,
                      q0.data_ptr(), p0.data_ptr(), g0.data_ptr(), q.data_ptr(), p.data_ptr())
            for i in range(L):
                logp, g = eval_logdensity(vg, q)
                _lib.call("bjx_mhmc_step_diag", stream, k0, k1, off, fold, N, D, i,
                          1 if i + 1 < L else 0, eps, _lib.ptr(eps_pc), imm_p, imm_s, thr,
                          logp0.data_ptr(), ke0.data_ptr(), q.data_ptr(), p.data_ptr(), g.data_ptr(),
                          logp.data_ptr(), weight.data_ptr(), slpa.data_ptr(), any_div.data_ptr(),
                          ever.data_ptr(), pq.data_ptr(), pp.data_ptr(), pg.data_ptr(),
                          plogp.data_ptr(), penergy.data_ptr())
        elif L > 0:
            # dense metric (shared matrix: MFMA GEMMs; per-chain matrices: fp64 matrix-vector kernels):
            # opening kick + drift, callable, then closing kick + reservoir step; the next leapfrog
            # starts from the fully kicked momentum with its own (separately rounded) opening kick
            q, p_half = torch.empty_like(q0), torch.empty_like(q0)
            p_half = dense.leapfrog(stream, metric, N, D, 1, eps, eps_pc, q0, p0, g0, q, p_half)
            for i in range(L):
                logp, g = eval_logdensity(vg, q)
                p1 = dense.mhmc_step(stream, metric, k0, k1, off, fold, N, D, i, eps, eps_pc, thr, logp0,
                                     ke0, q, p_half, g, logp, weight, slpa, any_div, ever, pq, pp, pg,
                                     plogp, penergy)
                if i + 1 < L:
                    p_half = dense.leapfrog(stream, metric, N, D, 1, eps, eps_pc, q, p1, g, q, p_half)
        _lib.call("bjx_mhmc_finish", stream, N, D, L, q0.data_ptr(), p0.data_ptr(), g0.data_ptr(),
                  logp0.data_ptr(), ke0.data_ptr(), ever.data_ptr(), slpa.data_ptr(), pq.data_ptr(),
                  pp.data_ptr(), pg.data_ptr(), plogp.data_ptr(), penergy.data_ptr(),
                  acc_rate.data_ptr())
        info = HMCInfo(p0, acc_rate, torch.ones(N, dtype=torch.bool, device=dev), any_div, penergy,
                       IntegratorState(pq, pp, plogp, pg), L)
        return HMCState(pq, plogp, pg), info

    return kernel


def _default_chain_block():
    import os

    v = os.environ.get("BJX_CHAIN_BLOCK", "")
    if v in ("", "auto"):
        return "auto"
    return int(v)  # 0 = all chains in one launch


# MI355X: 256 MiB Infinity Cache in front of HBM.  A chain block whose q, p and g (3 arrays) take
# 192 MiB leaves room for the shared vectors and the callable's scratch.
_IC_WORKING_SET_BYTES = 192 << 20


def auto_chain_block(n_chains: int, dim: int, arrays: int = 3) -> int:
    """Chains per block for ``chain_block="auto"``: the largest multiple of 1024 chains whose
    working set (``arrays`` fp32 arrays per element: q, p, g, plus a per-chain inverse mass matrix
    when there is one) fits the Infinity-Cache budget; all chains at once when the whole batch fits
    or the blocks would be too small to amortise a launch."""
    blk = (_IC_WORKING_SET_BYTES // (4 * int(arrays) * max(int(dim), 1))) // 1024 * 1024
    if blk < 1024 or blk >= n_chains:
        return int(n_chains)
    return int(blk)


class _GraphedTrajectory:
    """HIP-graph capture of the inner loop of one chain block:

        callable(Wq) ; [leapfrog(2 kicks, in place on Wq/Wp) ; callable(Wq)] x (L-1)

    over STATIC workspace buffers (positions ``Wq``, momenta ``Wp``, per-chain step sizes,
    inverse mass matrix), so one captured graph serves every block of every transition: the
    caller fills the workspace (the first kick+drift of a trajectory writes straight into it),
    replays, and reads the end state back.  Removes the per-launch host cost that would
    otherwise dominate once a block is small enough to live in the Infinity Cache.
    """

    def __init__(self, n, D, L, vg, imm_per_chain, device, owner=None):
        self.n, self.D, self.L = n, D, L
        self.owner = owner  # the user's callable: held so that id(owner) in the graph key stays unique
        self.Wq = torch.empty((n, D), dtype=torch.float32, device=device)
        self.Wp = torch.empty((n, D), dtype=torch.float32, device=device)
        self.eps = torch.ones(n, dtype=torch.float32, device=device)
        self.imm = torch.ones((n, D) if imm_per_chain else (D,), dtype=torch.float32, device=device)
        self.imm_stride = D if imm_per_chain else 0
        self.Wq.zero_()
        self.Wp.zero_()
        self._vg = vg
        # warm-up on a side stream (allocator / lazy-init work must not happen inside capture)
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            self._body()
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        self.graph = new_graph()
        with record_graph(self.graph):
            self.logp, self.g = self._body()

    def _body(self):
        stream = _lib.current_stream()
        logp, g = eval_logdensity(self._vg, self.Wq)
        for _ in range(self.L - 1):
            _lib.call("bjx_leapfrog_diag", stream, self.n, self.D, 2, 0.0, self.eps.data_ptr(),
                      self.imm.data_ptr(), self.imm_stride, self.Wq.data_ptr(), self.Wp.data_ptr(),
                      g.data_ptr(), self.Wq.data_ptr(), self.Wp.data_ptr())
            # release the consumed gradient first so the (stream-ordered) allocator hands the same
            # block to the callable again: the block's working set stays q, p, g (3 arrays)
            del logp, g
            logp, g = eval_logdensity(self._vg, self.Wq)
        return logp, g


_SIDE_STREAMS: dict = {}


def _side_streams(dev, n):
    pool = _SIDE_STREAMS.setdefault(dev.index, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=dev))
    return pool[:n]


def build_kernel(integrator=integrators.velocity_verlet, divergence_threshold: float = 1000,
                 build_proposal=None, *, chain_block=None, use_graph="auto", streams="auto"):
    """blackjax/mcmc/hmc.py:251-314.  ``build_proposal`` other than the default endpoint
    proposal (hmc_proposal, 115-178) is out of scope.

    ``chain_block``: chains are independent, so a transition may be run block by block
    (``chain_block`` chains at a time through all L leapfrogs) instead of launch by launch
    over all N chains.  With a block whose q/p/g working set fits the 256 MiB Infinity Cache
    the L-step loop re-reads its state from on-die cache instead of HBM.  Results are
    identical for any blocking (per-chain keys depend only on the global chain index).
    ``chain_block="auto"`` sizes the block for the 256 MiB Infinity Cache (``auto_chain_block``);
    measured at 65 536 x 1 024, L = 50: +12 % whole-transition throughput over one block.

    ``streams``: chain blocks advanced concurrently, each on its own HIP stream (blocks are
    independent; results are identical for any value), so that one block's callable and launch
    ramps overlap another block's kernels.  Measured: 65 536 x 1 024 diagonal, two blocks of 8 192
    in flight 234.5 vs 228.0 M/s with one stream (+2.9 %; the host then issues launches for two
    queues); 16 384 x 512 dense, two half batches: the GEMM launches shorten from 91 to 83 us each
    but the transition does not (2.46 vs 2.42 ms).  ``"auto"`` = 1.

    ``use_graph``: capture the per-block inner loop (the user's callable included) in a HIP
    graph (diagonal metric; the callable must be capturable: static shapes, no host sync).
    ``"auto"`` (default): do so for callables DECLARED recordable (``blackjax_amd.targets``, or any
    callable passed through ``blackjax_amd.capturable``) when a block is small enough for its
    launches to be bound by the host's launch rate (at most 2^21 elements: a leapfrog launch is
    then under ~10 us of GPU work; at 65 536 x 1 024 graphs measured no faster than plain
    launches), with a fall-back to plain launches if the recording fails; every other callable
    is driven with plain launches.  ``True`` records whatever the callable is.  As under
    ``jax.jit`` in the reference, a recorded callable is replayed as recorded: Python-side state
    it reads (a minibatch index, say) is frozen at recording time.
    """
    if use_graph not in (True, False, "auto"):
        raise ValueError("use_graph must be True, False or 'auto'")
    thr = float(divergence_threshold)
    if build_proposal is multinomial_hmc_proposal:
        integrators.check_supported(integrator, allow_general=True)
        return _build_mhmc_kernel(thr, integrator.coefficients[0::2], integrator.coefficients[1::2])
    if build_proposal not in (None, hmc_proposal):
        raise NotImplementedError(
            "build_proposal must be hmc_proposal (default) or multinomial_hmc_proposal")
    # any palindromic coefficient list [b1, a1, ..., b1] (integrators.py:62-152); the higher-order
    # ones (mclachlan / yoshida / omelyan) run through the general-coefficient kernels (diag metric)
    integrators.check_supported(integrator, allow_general=True)
    general = integrator is not integrators.velocity_verlet
    kick_c = integrator.coefficients[0::2]   # b1 .. b1
    drift_c = integrator.coefficients[1::2]  # a1 ..
    if chain_block is None:
        chain_block = _default_chain_block()
    graphs: dict = {}
    not_capturable: dict = {}    # id -> callable whose capture failed once ("auto" mode; the
    #                              reference keeps the id from being reused by another object)
    seen: dict = {}              # "auto" mode: calls per (shape, L, callable)

    def kernel(rng_key, state: HMCState, logdensity_fn: Callable, step_size,
               inverse_mass_matrix, num_integration_steps: int, *, chain_offset: int = 0):
        """One HMC transition for all chains (hmc.py:279-312 + hmc_proposal.generate 153-176)."""
        q0 = check_batch(state.position, "state.position")
        logp0 = check_batch(state.logdensity, "state.logdensity")
        g0 = check_batch(state.logdensity_grad, "state.logdensity_grad")
        N, D = q0.shape
        L = int(num_integration_steps)
        if L < 0:
            raise ValueError("num_integration_steps must be >= 0")
        k0, k1, fold = key_spec(rng_key)
        vg = value_and_grad(logdensity_fn)
        metric = metrics.default_metric(inverse_mass_matrix, N, D, q0.device)
        eps, eps_pc = step_size_args(step_size, N, q0.device)
        stream = _lib.current_stream()
        off = int(chain_offset)
        dev = q0.device
        graphed = use_graph is True and L >= 1 and metric.kind == "diag" and not general

        p0 = torch.empty_like(q0)
        ke0 = torch.empty_like(logp0)
        p_end = torch.empty_like(q0)
        q_new = torch.empty_like(q0)
        g_new = torch.empty_like(q0)
        logp_new = torch.empty_like(logp0)
        acc_rate = torch.empty_like(logp0)
        energy = torch.empty_like(logp0)
        is_acc = torch.empty(N, dtype=torch.bool, device=dev)
        is_div = torch.empty(N, dtype=torch.bool, device=dev)

        cb = chain_block
        n_streams = streams
        if n_streams == "auto":
            n_streams = 1
        if cb == "auto":  # Infinity-Cache tiling pays for the streaming (diagonal) kernels only
            if metric.kind == "diag":
                cb = auto_chain_block(N, D, 3 + (1 if metric.imm_stride else 0))
            elif metric.kind == "dense" and int(n_streams) > 1:
                cb = -(-N // int(n_streams) // 128) * 128  # equal blocks of whole 128-row GEMM tiles
            else:
                cb = N
        blk = N if not cb or cb >= N else int(cb)
        n_blocks = (N + blk - 1) // blk if N else 0
        fn_id = id(logdensity_fn)  # graphs are keyed on the USER's callable (and hold it)
        if (use_graph == "auto" and L >= 2 and N > 0 and metric.kind == "diag" and not general
                and blk * D <= (1 << 21) and is_capturable(logdensity_fn)
                and fn_id not in not_capturable):
            gkey = (min(blk, N), D, L, fn_id, metric.imm_stride != 0, dev.index)
            # record on the SECOND call with a given shape and trajectory length (a one-off call --
            # or a caller that varies L from step to step -- should not pay for a recording), and
            # keep at most 8 recordings per kernel
            seen[gkey] = seen.get(gkey, 0) + 1
            try:
                if gkey not in graphs and seen[gkey] >= 2 and len(graphs) < 8:
                    graphs[gkey] = _GraphedTrajectory(min(blk, N), D, L, vg, metric.imm_stride != 0, dev,
                                                      owner=logdensity_fn)
                graphed = gkey in graphs
            except RuntimeError:  # the callable cannot be recorded (or is broken: the plain path re-raises)
                not_capturable[fn_id] = logdensity_fn
                torch.cuda.synchronize(dev)
        if (use_graph == "auto" and L >= 2 and N > 0 and metric.kind == "diag" and not general
                and blk * D <= (1 << 21) and not is_capturable(logdensity_fn)):
            warn_eager_driver(logdensity_fn, "hmc")  # small launches from Python: say so once
        single = n_blocks <= 1 and not graphed
        # end-of-trajectory state (HMCInfo.proposal): per-block work buffers are copied out
        # unless the whole batch is one un-graphed block, in which case they ARE the result
        q_end = torch.empty_like(q0) if L > 0 else q0
        p_work = torch.empty_like(q0) if (L > 0 and not graphed) else None
        g_end = torch.empty_like(q0) if (L > 0 and not single) else None
        logp_end = torch.empty_like(logp0) if (L > 0 and not single) else None
        if L == 0:
            g_end, logp_end = g0, logp0

        def block_args(b):
            s, e = b * blk, min(N, (b + 1) * blk)
            sl = slice(s, e)
            if metric.kind == "dense_pc":  # per-chain matrices travel with their chains
                m = metric._replace(imm=metric.imm[sl], mass_sqrt_t=metric.mass_sqrt_t[sl])
            else:
                m = metric if metric.imm_stride == 0 else metric._replace(imm=metric.imm[sl])
            return e - s, sl, m, (None if eps_pc is None else eps_pc[sl]), off + s

        # plain velocity-Verlet trajectory on a diagonal metric with rows long enough for the row-per-wave
        # momentum kernel: the first kick + drift ride along with the (RNG-bound) momentum draw
        fused_first = (metric.kind == "diag" and L > 0 and not graphed and not general and D > 128 and _FUSE_FIRST)

        def launch_first(b, stream_):
            n, sl, m, eb, boff = block_args(b)
            _lib.call("bjx_hmc_momentum_kick_diag", stream_, k0, k1, boff, fold, n, D, m.imm.data_ptr(),
                      m.imm_stride, eps, _lib.ptr(eb), q0[sl].data_ptr(), g0[sl].data_ptr(),
                      p0[sl].data_ptr(), ke0[sl].data_ptr(), q_end[sl].data_ptr(), p_work[sl].data_ptr())

        def run_block(b):
            """One chain block's transition as a generator: yields after every log-density
            evaluation so that several blocks can be advanced in turn on their own streams."""
            nonlocal g_end, logp_end
            stream = _lib.current_stream()
            n, sl, m, eb, boff = block_args(b)
            if fused_first:
                q, p = q_end[sl], p_work[sl]
                launch_first(b, stream)
            elif m.kind == "diag":
                _lib.call("bjx_hmc_momentum_diag", stream, k0, k1, boff, fold, n, D, m.imm.data_ptr(),
                          m.imm_stride, p0[sl].data_ptr(), ke0[sl].data_ptr())
            else:
                from . import dense

                dense.momentum(stream, m, k0, k1, boff, fold, n, D, p0[sl], ke0[sl])

            if L == 0:
                q, p, logp, g = q0[sl], p0[sl], logp0[sl], g0[sl]
                eps_fin, eps_pc_fin = 0.0, None
            elif graphed:
                gkey = (n, D, L, fn_id, m.imm_stride != 0, dev.index)
                ctx = graphs.get(gkey)
                if ctx is None:
                    ctx = graphs[gkey] = _GraphedTrajectory(n, D, L, vg, m.imm_stride != 0, dev,
                                                            owner=logdensity_fn)
                if eb is None:
                    ctx.eps.fill_(eps)
                else:
                    ctx.eps.copy_(eb)
                ctx.imm.copy_(m.imm)
                # first kick + drift writes straight into the static workspace
                _launch_leapfrog(stream, m, n, D, 1, eps, eb, q0[sl], p0[sl], g0[sl], ctx.Wq, ctx.Wp)
                ctx.graph.replay()
                q, p, logp, g = ctx.Wq, ctx.Wp, ctx.logp, ctx.g
                eps_fin, eps_pc_fin = eps, eb
            elif general:
                # generalized_two_stage_integrator (integrators.py:104-150): one launch per position
                # update; the closing kick b_K of a step merges with the opening kick b_1 of the next
                q, p = q_end[sl], p_work[sl]

                def stage(n_k, ka, kb, a_c, q_in, p_in, g_in, p_out):
                    if m.kind == "diag":
                        _lib.call("bjx_leapfrog_diag_coef", stream, n, D, n_k, ka, kb, a_c, eps, _lib.ptr(eb),
                                  m.imm.data_ptr(), m.imm_stride, q_in.data_ptr(), p_in.data_ptr(),
                                  g_in.data_ptr(), q.data_ptr(), p_out.data_ptr(), None, 0)
                        return p_out
                    from . import dense

                    return dense.leapfrog_coef(stream, m, n, D, n_k, ka, kb, a_c, eps, eb, q_in, p_in,
                                               g_in, q, p_out)

                first = True
                for _ in range(L):
                    for si, a_c in enumerate(drift_c):
                        if first:
                            p = stage(1, kick_c[0], 0.0, a_c, q0[sl], p0[sl], g0[sl], p)
                            first = False
                        elif si == 0:
                            p = stage(2, kick_c[-1], kick_c[0], a_c, q, p, g, p)
                        else:
                            p = stage(1, kick_c[si], 0.0, a_c, q, p, g, p)
                        logp, g = eval_logdensity(vg, q)
                        yield
                        stream = _lib.current_stream()
                eps_fin, eps_pc_fin = eps, eb
            else:
                if not fused_first:
                    q, p = q_end[sl], p_work[sl]
                    p = _launch_leapfrog(stream, m, n, D, 1, eps, eb, q0[sl], p0[sl], g0[sl], q, p)
                logp, g = eval_logdensity(vg, q)
                for _ in range(L - 1):
                    yield
                    stream = _lib.current_stream()
                    p = _launch_leapfrog(stream, m, n, D, 2, eps, eb, q, p, g, q, p)
                    logp, g = eval_logdensity(vg, q)
                yield
                stream = _lib.current_stream()
                eps_fin, eps_pc_fin = eps, eb

            if m.kind == "diag" and general:
                _lib.call("bjx_hmc_finish_diag_coef", stream, k0, k1, boff, fold, n, D, kick_c[-1],
                          eps_fin, _lib.ptr(eps_pc_fin), m.imm.data_ptr(), m.imm_stride, thr,
                          q0[sl].data_ptr(), logp0[sl].data_ptr(), g0[sl].data_ptr(),
                          ke0[sl].data_ptr(), q.data_ptr(), logp.data_ptr(), g.data_ptr(),
                          p.data_ptr(), p_end[sl].data_ptr(), q_new[sl].data_ptr(),
                          logp_new[sl].data_ptr(), g_new[sl].data_ptr(), acc_rate[sl].data_ptr(),
                          is_acc[sl].data_ptr(), is_div[sl].data_ptr(), energy[sl].data_ptr())
            elif m.kind == "diag":
                _lib.call("bjx_hmc_finish_diag", stream, k0, k1, boff, fold, n, D, eps_fin,
                          _lib.ptr(eps_pc_fin), m.imm.data_ptr(), m.imm_stride, thr,
                          q0[sl].data_ptr(), logp0[sl].data_ptr(), g0[sl].data_ptr(),
                          ke0[sl].data_ptr(), q.data_ptr(), logp.data_ptr(), g.data_ptr(),
                          p.data_ptr(), p_end[sl].data_ptr(), q_new[sl].data_ptr(),
                          logp_new[sl].data_ptr(), g_new[sl].data_ptr(), acc_rate[sl].data_ptr(),
                          is_acc[sl].data_ptr(), is_div[sl].data_ptr(), energy[sl].data_ptr())
            elif general:
                from . import dense

                dense.finish_coef(stream, m, k0, k1, boff, fold, n, D, kick_c[-1], eps_fin, eps_pc_fin, thr,
                                  q0[sl], logp0[sl], g0[sl], ke0[sl], q, logp, g, p, p_end[sl], q_new[sl],
                                  logp_new[sl], g_new[sl], acc_rate[sl], is_acc[sl], is_div[sl], energy[sl])
            else:
                from . import dense

                dense.finish(stream, m, k0, k1, boff, fold, n, D, eps_fin, eps_pc_fin, thr, q0[sl],
                             logp0[sl], g0[sl], ke0[sl], q, logp, g, p, p_end[sl], q_new[sl],
                             logp_new[sl], g_new[sl], acc_rate[sl], is_acc[sl], is_div[sl],
                             energy[sl])
            if L > 0:
                if single:
                    g_end, logp_end = g, logp
                else:
                    g_end[sl].copy_(g)
                    logp_end[sl].copy_(logp)
                    if graphed:
                        q_end[sl].copy_(q)


        # Blocks are independent, so up to `n_streams` of them are advanced in turn, each on its own
        # HIP stream: one block's callable (bandwidth-bound), the start-up of its launches and the
        # output drain of a dense-metric GEMM then overlap another block's kernels.
        ns = 1 if (graphed or n_blocks <= 1) else min(int(n_streams), n_blocks)
        if ns <= 1:
            for b in range(n_blocks):
                for _ in run_block(b):
                    pass
        else:
            main = torch.cuda.current_stream(dev)
            pool = _side_streams(dev, ns)
            for st_ in pool:
                st_.wait_stream(main)
            for w0 in range(0, n_blocks, ns):
                active = [(run_block(b), pool[b - w0]) for b in range(w0, min(w0 + ns, n_blocks))]
                while active:
                    for item in list(active):
                        with torch.cuda.stream(item[1]):
                            try:
                                next(item[0])
                            except StopIteration:
                                active.remove(item)
            for st_ in pool:
                main.wait_stream(st_)

        if n_blocks == 0 and L > 0:
            g_end, logp_end = g0, logp0
        info = HMCInfo(p0, acc_rate, is_acc, is_div, energy,
                       IntegratorState(q_end, p_end, logp_end, g_end), L)
        return HMCState(q_new, logp_new, g_new), info

    return kernel


def build_fused_target_kernel(divergence_threshold: float = 1000, *, with_info_arrays: bool = True):
    """A whole transition per launch for log-densities the ENGINE evaluates itself (``bjx_hmc_trajectory_diag``):
    ``blackjax_amd.targets.NealFunnel`` / ``DiagGaussian``, diagonal metric, velocity Verlet, 128 < D <= 1 024,
    D % 4 == 0.  OUTSIDE the external-callable contract (the reference calls ``logdensity_fn`` between two
    leapfrogs) -- opt-in through ``hmc(..., fuse_target=True)``; state and info are bit for bit those of
    ``build_kernel()``'s kernel.  ``with_info_arrays=False``: ``HMCInfo.momentum`` and ``.proposal`` are ``None``
    and their five arrays are not written."""
    thr = float(divergence_threshold)

    def kernel(rng_key, state: HMCState, logdensity_fn: Callable, step_size,
               inverse_mass_matrix, num_integration_steps: int, *, chain_offset: int = 0):
        q0 = check_batch(state.position, "state.position")
        logp0 = check_batch(state.logdensity, "state.logdensity")
        g0 = check_batch(state.logdensity_grad, "state.logdensity_grad")
        N, D = q0.shape
        dev = q0.device
        L = int(num_integration_steps)
        metric = metrics.default_metric(inverse_mass_matrix, N, D, dev)
        spec = getattr(logdensity_fn, "_bjx_fused_target", None)
        spec = spec(D) if callable(spec) else None
        if spec is None or metric.kind != "diag" or D % 4 != 0 or not 128 < D <= 1024 or L < 1:
            raise NotImplementedError(
                "fuse_target=True needs a blackjax_amd.targets log-density the engine can evaluate in place "
                "(NealFunnel; DiagGaussian), a diagonal metric, 128 < D <= 1024 with D % 4 == 0 and at least "
                "one integration step")
        k0, k1, fold = key_spec(rng_key)
        eps, eps_pc = step_size_args(step_size, N, dev)
        q_new, g_new, logp_new = torch.empty_like(q0), torch.empty_like(g0), torch.empty_like(logp0)
        acc_rate, energy = torch.empty_like(logp0), torch.empty_like(logp0)
        is_acc = torch.empty(N, dtype=torch.bool, device=dev)
        is_div = torch.empty(N, dtype=torch.bool, device=dev)
        p0 = q1 = p_end = logp1 = g1 = None
        if with_info_arrays:
            p0, q1, p_end, g1 = (torch.empty_like(q0) for _ in range(4))
            logp1 = torch.empty_like(logp0)
        if spec[0] == "rtc":
            # a user-written device target (targets.DeviceTarget): the same trajectory code, compiled around it
            # by hiprtc (csrc/bjx_traj_dev.h, blackjax_amd/rtc.py)
            from . import rtc

            tgt = spec[1]
            ptr = lambda t: 0 if t is None else t.data_ptr()  # noqa: E731
            args = rtc.TrajArgs(k0, k1, int(chain_offset), fold, N, D, L, eps, ptr(eps_pc), metric.imm.data_ptr(),
                                metric.imm_stride, thr, tgt._params_ptr(dev), q0.data_ptr(), logp0.data_ptr(),
                                g0.data_ptr(), ptr(p0), ptr(q1), ptr(p_end), ptr(logp1), ptr(g1), q_new.data_ptr(),
                                logp_new.data_ptr(), g_new.data_ptr(), acc_rate.data_ptr(), energy.data_ptr(),
                                is_acc.data_ptr(), is_div.data_ptr())
            tgt.module().launch(f"bjx_rtc_traj_{rtc.ni_for(D)}", min((N + 3) // 4, 65536), 256,
                                _lib.current_stream(), args)
        else:
            _lib.call("bjx_hmc_trajectory_diag", _lib.current_stream(), k0, k1, int(chain_offset), fold, N, D, L,
                      eps, _lib.ptr(eps_pc), metric.imm.data_ptr(), metric.imm_stride, thr, int(spec[0]),
                      _lib.ptr(spec[1]), q0.data_ptr(), logp0.data_ptr(), g0.data_ptr(), _lib.ptr(p0),
                      _lib.ptr(q1), _lib.ptr(p_end), _lib.ptr(logp1), _lib.ptr(g1), q_new.data_ptr(),
                      logp_new.data_ptr(), g_new.data_ptr(), acc_rate.data_ptr(), is_acc.data_ptr(),
                      is_div.data_ptr(), energy.data_ptr())
        proposal = IntegratorState(q1, p_end, logp1, g1) if with_info_arrays else None
        return HMCState(q_new, logp_new, g_new), HMCInfo(p0, acc_rate, is_acc, is_div, energy, proposal, L)

    return kernel


def as_top_level_api(logdensity_fn: Callable, step_size, inverse_mass_matrix,
                     num_integration_steps: int, *, divergence_threshold: float = 1000,
                     integrator=integrators.velocity_verlet, build_proposal=None,
                     chain_offset: int = 0, chain_block=None,
                     use_graph="auto", streams="auto", fuse_target=False) -> SamplingAlgorithm:
    """blackjax/mcmc/hmc.py:317-414.  ``chain_offset`` is this process' first global chain
    index when the chains of one run are sharded over several GPUs.  ``fuse_target=True`` (or ``"lean"``: without
    the momentum / proposal arrays of ``HMCInfo``): ``build_fused_target_kernel`` -- one launch per transition for
    the library's own log-densities, outside the external-callable contract, identical results."""
    if fuse_target:
        if integrator is not integrators.velocity_verlet or build_proposal not in (None, hmc_proposal):
            raise NotImplementedError("fuse_target=True: velocity Verlet with the endpoint proposal")
        kernel = build_fused_target_kernel(divergence_threshold, with_info_arrays=fuse_target != "lean")
    else:
        kernel = build_kernel(integrator, divergence_threshold, build_proposal,
                              chain_block=chain_block, use_graph=use_graph, streams=streams)

    def init_fn(position, rng_key=None):
        del rng_key
        return init(position, logdensity_fn)

    def step_fn(rng_key, state):
        return kernel(rng_key, state, logdensity_fn, step_size, inverse_mass_matrix,
                      num_integration_steps, chain_offset=chain_offset)

    return SamplingAlgorithm(init_fn, step_fn)
