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
``bjx_nuts_lockstep.hip`` / ``bjx_nuts_tick.hip`` / ``bjx_nuts_spec.hip`` (device functions: ``bjx_nuts_chain.h``, ``bjx_nuts_tick_dev.h``).

Two drivers with identical results:
* eager (``use_graph=False``): three launches per leapfrog (pre, callable, post) issued from Python;
* graph (``use_graph=True``): chunks of up to 16 leapfrogs are captured once in HIP graphs over a
  static workspace and replayed; everything that changes between replays (doubling, leaf index,
  active row count, keys) lives in a small device control block.  This removes the per-launch host
  cost, which otherwise dominates because a NUTS transition is hundreds of short launches.
The default ``use_graph="auto"`` takes the graph driver and falls back to the eager one for a
callable that cannot be recorded.
"""
from __future__ import annotations

import ctypes
from typing import Callable, NamedTuple, Optional

import torch

from . import _lib, integrators, metrics
from ._util import (new_graph, record_graph, check_batch, eval_logdensity, is_capturable, step_size_args, value_and_grad,
                    warn_eager_driver)
from .base import SamplingAlgorithm
from .hmc import HMCState, IntegratorState, init
from .random import key_spec

__all__ = ["NUTSInfo", "NUTSRunInfo", "init", "build_kernel", "run_free", "as_top_level_api"]

_BUFS = ["Lq", "Lp", "Lg", "Rq", "Rp", "Rg", "msum", "Smsum", "Pq", "Pg", "Sq", "Sg"]


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


def _make_info(p0, bufs, fs, is_, clone: bool):
    F, I = _lib.NUTS_F, _lib.NUTS_I
    c = (lambda t: t.clone()) if clone else (lambda t: t)
    info = NUTSInfo(
        p0,
        is_[I["DIV"]].bool(),
        is_[I["TURN"]].bool(),
        c(fs[F["PENERGY"]]),
        IntegratorState(c(bufs["Lq"]), c(bufs["Lp"]), c(fs[F["LLOGP"]]), c(bufs["Lg"])),
        IntegratorState(c(bufs["Rq"]), c(bufs["Rp"]), c(fs[F["RLOGP"]]), c(bufs["Rg"])),
        c(is_[I["DEPTH"]]),
        c(is_[I["NSTATES"]]),
        c(fs[F["ACC"]]),
    )
    return HMCState(c(bufs["Pq"]), c(fs[F["PLOGP"]]), c(bufs["Pg"])), info


def _dense_fields(kind, imm, N, D, max_depth, device):
    """Extra descriptor fields + buffers for a dense metric (``bjx_nuts_t.Mdense`` ...): the
    velocities M^{-1} p of the two trajectory ends and of the checkpointed momenta."""
    if kind == "diag":
        return {"fields": dict(Mdense=0, Mdense_stride=0, v0=0, Lv=0, Rv=0, ckpt_v=0), "keep": ()}
    f32 = dict(dtype=torch.float32, device=device)
    lv, rv = torch.empty((N, D), **f32), torch.empty((N, D), **f32)
    ck_v = torch.empty((N, max(max_depth, 1), D), **f32)
    return {"fields": dict(Mdense=imm.data_ptr(), Mdense_stride=D * D if kind == "dense_pc" else 0,
                           v0=0, Lv=lv.data_ptr(), Rv=rv.data_ptr(), ckpt_v=ck_v.data_ptr()),
            "keep": (lv, rv, ck_v)}


class _GraphWorkspace:
    """Static device buffers + captured chunk graphs for the ``use_graph`` driver."""

    MAX_CHUNK = 16
    MIN_BUCKET = int(__import__("os").environ.get("BJX_NUTS_MIN_BUCKET", "256"))  # smallest recorded batch

    def __init__(self, N, D, max_depth, vg, imm_shape, kind, thr, device, owner=None,
                 kick_c=(0.5, 0.5), drift_c=(1.0,)):
        self.N, self.D, self.max_depth, self.vg = N, D, max_depth, vg
        self.kick_c, self.drift_c = tuple(kick_c), tuple(drift_c)  # palindromic integrator [b1, a1, ..., b1]
        self.owner = owner  # the user's callable: held so that id(owner) in the workspace key stays unique
        f32 = dict(dtype=torch.float32, device=device)
        self.bufs = {n: torch.empty((N, D), **f32) for n in _BUFS}
        self.ck_r = torch.empty((N, max(max_depth, 1), D), **f32)
        self.ck_rs = torch.empty_like(self.ck_r)
        self.fs = torch.empty((_lib.NUTS_NF, N), **f32)
        self.is_ = torch.empty((_lib.NUTS_NI, N), dtype=torch.int32, device=device)
        self.eps = torch.ones(N, **f32)
        self.imm = torch.ones(imm_shape, **f32)  # static copy of the (diag or dense) metric
        self.idx = torch.arange(N, dtype=torch.int32, device=device)
        self.qf = torch.zeros((N, D), **f32)
        self.ctl = torch.zeros(8, dtype=torch.int64, device=device)
        self.gemm = kind == "dense_gemm"
        self.dense = _dense_fields("dense" if self.gemm else kind, self.imm, N, D, max_depth, device)
        if self.gemm:
            self.pc, self.vc = torch.empty((N, D), **f32), torch.empty((N, D), **f32)
            self.imm_t = torch.ones(imm_shape, **f32)  # static copy of the transposed matrix (bjx_dense_apply_imm_t)
            self.dense["fields"]["v_pre"] = self.vc.data_ptr()
        self.desc = _lib.NutsDesc(
            N=N, D=D, max_depth=max_depth, reserved=0, imm=self.imm.data_ptr(),
            imm_stride=D if (kind == "diag" and len(imm_shape) == 2) else 0,
            eps_per_chain=self.eps.data_ptr(), eps=0.0,
            divergence_threshold=thr, key0=0, key1=0, chain_offset=0, step_fold=-1,
            q0=0, g0=0, p0=0, ckpt_r=self.ck_r.data_ptr(), ckpt_rs=self.ck_rs.data_ptr(),
            fs=self.fs.data_ptr(), is_=self.is_.data_ptr(),
            **{n: b.data_ptr() for n, b in self.bufs.items()}, **self.dense["fields"],
            int_kick=self.kick_c[0], int_drift=self.drift_c[0])
        self.graphs: dict = {}
        self.count_pinned = torch.zeros(1, dtype=torch.int64).pin_memory()  # lagged reads of ctl[2] (kernel_graph)
        self.count_event = torch.cuda.Event()

    def bucket(self, n_rows: int) -> int:
        cap = self.MIN_BUCKET
        while cap < n_rows:
            cap *= 2
        return min(cap, self.N)

    def _chunk_body(self, k, n_cap):
        stream = _lib.current_stream()
        dref = ctypes.byref(self.desc)
        qf = self.qf[:n_cap]
        idx_p, ctl_p = self.idx.data_ptr(), self.ctl.data_ptr()

        def velocities(s_off, gf_t, kick):
            """GEMM mode (shared dense metric): vc[b] = M^{-1} (p_end + (dir eps kick) g), one GEMM over
            the chunk's row capacity"""
            _lib.call("bjx_nuts_dense_kick", stream, dref, 0, s_off, n_cap, idx_p, ctl_p, _lib.ptr(gf_t), kick,
                      self.pc.data_ptr())
            _lib.call("bjx_dense_apply_imm_t", stream, n_cap, self.D, self.pc.data_ptr(), self.imm.data_ptr(),
                      self.imm_t.data_ptr(), self.vc.data_ptr())

        if self.gemm:
            velocities(0, None, self.kick_c[0])
        _lib.call("bjx_nuts_pre_ctl", stream, dref, 0, n_cap, idx_p, ctl_p, qf.data_ptr())
        for i in range(k):
            logp_f, gf = eval_logdensity(self.vg, qf)
            for si in range(1, len(self.drift_c)):  # stages 2 .. K of a multi-stage integrator
                if self.gemm:
                    velocities(i, gf, self.kick_c[si])
                _lib.call("bjx_nuts_mid", stream, dref, n_cap, idx_p, ctl_p,
                          qf.data_ptr(), gf.data_ptr(), self.kick_c[si], self.drift_c[si])
                logp_f, gf = eval_logdensity(self.vg, qf)
            if self.gemm:
                velocities(i, gf, self.kick_c[-1])
            # post(i) fused with pre(i+1) inside the chunk (row order is fixed within a chunk); in GEMM
            # mode the next leaf's opening velocity is its own product, so pre is a launch of its own
            fuse = 0 if self.gemm else (1 if i < k - 1 else 0)
            _lib.call("bjx_nuts_post_ctl", stream, dref, i, n_cap, idx_p, ctl_p, qf.data_ptr(),
                      logp_f.data_ptr(), gf.data_ptr(), fuse)
            if self.gemm and i < k - 1:
                velocities(i + 1, None, self.kick_c[0])
                _lib.call("bjx_nuts_pre_ctl", stream, dref, i + 1, n_cap, idx_p, ctl_p, qf.data_ptr())
        return logp_f, gf

    def chunk_graph(self, k, n_cap):
        g = self.graphs.get((k, n_cap))
        if g is None:
            dev = self.qf.device
            # capture with n_rows = 0 in the control block: the kernels touch no chain state
            saved = self.ctl.clone()
            self.ctl.zero_()
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                self._chunk_body(1, n_cap)  # warm-up outside capture
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            graph = new_graph()
            with record_graph(graph):
                keep = self._chunk_body(k, n_cap)
            self.ctl.copy_(saved)
            g = self.graphs[(k, n_cap)] = (graph, keep)
        return g[0]

    def set_ctl(self, depth, s_base, n_rows, k0, k1, fold, off):
        _lib.call("bjx_nuts_set_ctl", _lib.current_stream(), self.ctl.data_ptr(), depth, s_base,
                  n_rows, k0, k1, fold, off)


def build_kernel(integrator=integrators.velocity_verlet, divergence_threshold: int = 1000, *,
                 recompact_every: int = 16, use_graph="auto", graph_sync_every: int = 4,
                 dense_gemm="auto"):
    """blackjax/mcmc/nuts.py:77-147.  ``use_graph``: ``True`` = the HIP-graph driver, ``False`` =
    plain launches (three per leaf from Python: host-bound once a leaf is a few microseconds of GPU
    work), ``"auto"`` (default) = the graph driver for callables DECLARED recordable
    (``blackjax_amd.targets``, or anything passed through ``blackjax_amd.capturable``) -- with a
    fall-back to plain launches should the recording fail -- and plain launches for every other
    callable.  As under ``jax.jit`` in the reference, a recorded callable is
    replayed as recorded: Python-side state it reads is frozen at recording time (``use_graph=False``
    for such a callable).  ``graph_sync_every``: in graph mode the host reads the active-row
    count back only every that many 16-leapfrog chunks (to stop early / shrink the callable's
    batch); compaction itself happens on the device every chunk."""
    if use_graph not in (True, False, "auto"):
        raise ValueError("use_graph must be True, False or 'auto'")
    if dense_gemm not in (True, False, "auto"):
        raise ValueError("dense_gemm must be True, False or 'auto'")
    # any palindromic coefficient list [b1, a1, ..., b1] (integrators.py:62-152, nuts.py:150-158): a leaf
    # is then pre (kick b1, drift a1) -> callable -> [mid (kick b_i, drift a_i) -> callable] ... -> post
    integrators.check_supported(integrator, allow_general=True)
    kick_c = integrator.coefficients[0::2]
    drift_c = integrator.coefficients[1::2]
    thr = float(divergence_threshold)
    I = _lib.NUTS_I
    workspaces: dict = {}
    sync_every = int(graph_sync_every)

    def _common(rng_key, state, logdensity_fn, step_size, inverse_mass_matrix, chain_offset):
        q0 = check_batch(state.position, "state.position")
        logp0 = check_batch(state.logdensity, "state.logdensity")
        g0 = check_batch(state.logdensity_grad, "state.logdensity_grad")
        N, D = q0.shape
        k0, k1, fold = key_spec(rng_key)
        vg = value_and_grad(logdensity_fn)
        metric = metrics.default_metric(inverse_mass_matrix, N, D, q0.device)
        eps, eps_pc = step_size_args(step_size, N, q0.device)
        stream = _lib.current_stream()
        p0 = torch.empty_like(q0)
        ke0 = torch.empty_like(logp0)
        v0 = None
        if metric.kind == "diag":
            _lib.call("bjx_hmc_momentum_diag", stream, k0, k1, int(chain_offset), fold, N, D,
                      metric.imm.data_ptr(), metric.imm_stride, p0.data_ptr(), ke0.data_ptr())
        else:
            # dense metric.  One matrix per chain, and small shared-matrix problems: fp64-accumulated
            # matrix-vector kernels throughout (bit-compatible with the oracle's "f64" mode).  A SHARED
            # matrix with D >= 128 (dense_gemm="auto"; True forces it at any D): every product v = M^{-1} p of a leaf is ONE fp32 MFMA GEMM
            # over the live rows (bjx_nuts_dense_kick -> bjx_dense_apply_imm -> kernels reading
            # bjx_nuts_t.v_pre) instead of D^2 words per chain -- the oracle's "f32chain" mode
            # (metrics.py:263-304 with util.py:58-61).
            from . import dense

            # "auto" looks at the metric alone (shared matrix, D >= 128), never at the local batch size: a
            # chain's arithmetic must not depend on how many chains share its process
            gemm = metric.kind == "dense" and (dense_gemm is True or (dense_gemm == "auto" and D >= 128))
            v0 = dense.momentum(stream, metric, k0, k1, int(chain_offset), fold, N, D, p0, ke0,
                                force_pc=not gemm)
            metric = metric._replace(kind="dense_gemm") if gemm else metric
        return q0, logp0, g0, N, D, k0, k1, fold, vg, metric, eps, eps_pc, stream, p0, ke0, v0

    def kernel_eager(rng_key, state: HMCState, logdensity_fn: Callable, step_size,
                     inverse_mass_matrix, max_num_doublings: int = 10, *, chain_offset: int = 0):
        (q0, logp0, g0, N, D, k0, k1, fold, vg, metric, eps, eps_pc, stream, p0, ke0,
         v0) = _common(rng_key, state, logdensity_fn, step_size, inverse_mass_matrix, chain_offset)
        dev = q0.device
        max_depth = int(max_num_doublings)
        bufs = {n: torch.empty_like(q0) for n in _BUFS}
        ck_r = torch.empty((N, max(max_depth, 1), D), dtype=torch.float32, device=dev)
        ck_rs = torch.empty_like(ck_r)
        fs = torch.empty((_lib.NUTS_NF, N), dtype=torch.float32, device=dev)
        is_ = torch.empty((_lib.NUTS_NI, N), dtype=torch.int32, device=dev)
        gemm = metric.kind == "dense_gemm"
        dense_f = _dense_fields("dense" if gemm else metric.kind, metric.imm, N, D, max_depth, dev)
        if v0 is not None:
            dense_f["fields"]["v0"] = v0.data_ptr()
        pc = vc = None
        if gemm:  # compact kicked momenta and their velocities (one GEMM per product)
            pc, vc = torch.empty_like(q0), torch.empty_like(q0)
            dense_f["fields"]["v_pre"] = vc.data_ptr()
        desc = _lib.NutsDesc(
            N=N, D=D, max_depth=max_depth, reserved=0, imm=metric.imm.data_ptr(),
            imm_stride=metric.imm_stride, eps_per_chain=_lib.ptr(eps_pc), eps=eps,
            divergence_threshold=thr, key0=k0, key1=k1, chain_offset=int(chain_offset),
            step_fold=fold, q0=q0.data_ptr(), g0=g0.data_ptr(), p0=p0.data_ptr(),
            ckpt_r=ck_r.data_ptr(), ckpt_rs=ck_rs.data_ptr(), fs=fs.data_ptr(), is_=is_.data_ptr(),
            **{n: b.data_ptr() for n, b in bufs.items()}, **dense_f["fields"],
            int_kick=kick_c[0], int_drift=drift_c[0])
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
            need_pre = True
            for s in range(n_leaves):
                if s > 0 and recompact_every and s % recompact_every == 0:
                    # drop the chains whose subtree has stopped (diverged / turned)
                    sub = is_[I["SUB_ACTIVE"]]
                    if idx_step is None:
                        new_idx = torch.nonzero(sub, as_tuple=False).flatten().to(torch.int32)
                    else:
                        new_idx = idx_step[sub[idx_step.long()].bool()]
                    n_new = int(new_idx.shape[0])  # host sync
                    if n_new == 0:
                        break
                    if n_new < n_step:
                        idx_step, n_step = new_idx.contiguous(), n_new
                        qf = qf[:n_step]
                def velocities(gf_t, kick):
                    """GEMM mode: vc[b] = M^{-1} (p_end + (dir eps kick) g) for the live rows"""
                    _lib.call("bjx_nuts_dense_kick", stream, dref, depth, s, n_step, _lib.ptr(idx_step), None,
                              _lib.ptr(gf_t), kick, pc.data_ptr())
                    _lib.call("bjx_dense_apply_imm_t", stream, n_step, D, pc.data_ptr(), metric.imm.data_ptr(),
                              metric.imm_t.data_ptr(), vc.data_ptr())

                if need_pre:
                    if gemm:
                        velocities(None, kick_c[0])
                    _lib.call("bjx_nuts_pre", stream, dref, depth, s, n_step, _lib.ptr(idx_step),
                              qf.data_ptr())
                logp_f, gf = eval_logdensity(vg, qf)
                for si in range(1, len(drift_c)):  # stages 2 .. K of a multi-stage integrator
                    if gemm:
                        velocities(gf, kick_c[si])
                    _lib.call("bjx_nuts_mid", stream, dref, n_step, _lib.ptr(idx_step), None, qf.data_ptr(),
                              gf.data_ptr(), kick_c[si], drift_c[si])
                    logp_f, gf = eval_logdensity(vg, qf)
                if gemm:
                    velocities(gf, kick_c[-1])
                # fuse the next leaf's opening half into post unless rows are re-compacted before it
                # (never in GEMM mode: the next leaf's opening velocity is its own product)
                nxt = s + 1
                fuse = (not gemm) and nxt < n_leaves and not (recompact_every and nxt % recompact_every == 0)
                _lib.call("bjx_nuts_post", stream, dref, depth, s, n_step, _lib.ptr(idx_step),
                          qf.data_ptr(), logp_f.data_ptr(), gf.data_ptr(), 1 if fuse else 0)
                need_pre = not fuse
            _lib.call("bjx_nuts_merge", stream, dref, depth, n_doubling, _lib.ptr(idx_doubling))
        return _make_info(p0, bufs, fs, is_, clone=False)

    def kernel_graph(rng_key, state: HMCState, logdensity_fn: Callable, step_size,
                     inverse_mass_matrix, max_num_doublings: int = 10, *, chain_offset: int = 0):
        (q0, logp0, g0, N, D, k0, k1, fold, vg, metric, eps, eps_pc, stream, p0, ke0,
         v0) = _common(rng_key, state, logdensity_fn, step_size, inverse_mass_matrix, chain_offset)
        max_depth = int(max_num_doublings)
        off = int(chain_offset)
        wkey = (N, D, max_depth, id(logdensity_fn), metric.kind, tuple(metric.imm.shape), q0.device.index)
        ws = workspaces.get(wkey)
        if ws is None:
            ws = workspaces[wkey] = _GraphWorkspace(N, D, max_depth, vg, tuple(metric.imm.shape),
                                                    metric.kind, thr, q0.device, owner=logdensity_fn,
                                                    kick_c=kick_c, drift_c=drift_c)
        if eps_pc is None:
            ws.eps.fill_(eps)
        else:
            ws.eps.copy_(eps_pc)
        ws.imm.copy_(metric.imm)
        if ws.gemm:
            ws.imm_t.copy_(metric.imm_t)
        d = ws.desc
        d.key0, d.key1, d.chain_offset, d.step_fold = k0, k1, off, fold
        d.q0, d.g0, d.p0 = q0.data_ptr(), g0.data_ptr(), p0.data_ptr()
        d.v0 = v0.data_ptr() if v0 is not None else 0
        dref = ctypes.byref(d)
        _lib.call("bjx_nuts_init", stream, dref, logp0.data_ptr(), ke0.data_ptr())
        chunk_max = ws.MAX_CHUNK
        idx_ptr, ctl_ptr = ws.idx.data_ptr(), ws.ctl.data_ptr()
        # Lagged reads of the live-row count (round 5).  The host needs that count at every doubling (to stop, to size
        # the recorded batch, to save the doubling's row list for the merge) and every few chunks inside a doubling.  A
        # blocking .item() at those points drains the queue: the GPU then idles for the host's whole wake-up + launch
        # latency, ~13 times per transition -- 20 us each on a fast host, 100 us on a slow one (lockstep `step` measured
        # 62-80 M/s from box to box).  Instead the count is copied to pinned memory behind the compaction, the NEXT
        # chunk is queued with the row capacity known so far (an upper bound: the recorded kernels take the actual
        # count from the device control block), and only then does the host wait for the copy -- the GPU is busy with
        # that chunk meanwhile.  Cost: when the count turns out to be zero the chunk queued ahead ran over no rows
        # (its kernels exit at once) -- at most once per transition.  Same kernels in the same order: results unchanged.
        # Same-call A/B at C3 (three pairs): 72.1 / 71.9 M/s blocking, 73.0 / 73.3 / 73.6 lagged (+2 % on a fast host).
        def read_count_async():
            ws.count_pinned.copy_(ws.ctl[2:3], non_blocking=True)
            ws.count_event.record()

        def read_count_wait() -> int:
            ws.count_event.synchronize()
            return int(ws.count_pinned[0])

        n_prev = N
        for depth in range(max_depth):
            # chains still doubling -> ws.idx / ctl[2] (device-side compaction)
            _lib.call("bjx_nuts_compact", stream, dref, I["ACTIVE"], N, None, idx_ptr, ctl_ptr)
            n_leaves = 1 << depth
            # deep doublings are a handful of chains and pure launch latency (two dependent kernels per
            # leaf): longer recorded chunks there -- fewer compaction / control-block launches and
            # replay boundaries per leaf
            k = min(chunk_max if depth < 7 else 4 * chunk_max, n_leaves)
            n_cap = ws.bucket(n_prev)  # capacity for the chunk queued ahead of the count
            if depth > 0:
                read_count_async()
            ws.set_ctl(depth, 0, -1, k0, k1, fold, off)
            ws.chunk_graph(k, n_cap).replay()  # chunk 0 of this doubling, queued before the host knows the count
            n_doubling = N if depth == 0 else read_count_wait()
            if n_doubling == 0:
                break
            n_prev = n_doubling
            idx_doubling = ws.idx[:n_doubling].clone()  # (chunk 0 does not touch the row list)
            n_cap = ws.bucket(n_doubling)
            stopped = False
            for j, s_base in enumerate(range(k, n_leaves, k), start=1):
                poll = bool(sync_every and j % sync_every == 0)
                if n_cap > ws.MIN_BUCKET or poll:
                    # drop the chains whose subtree has stopped -- on the device, no host sync.  Once the batch
                    # is at its smallest recorded size nothing can shrink any more (the leaf kernels skip
                    # stopped chains themselves), so the compaction launch is only made where the host reads
                    # the live count (round 4: ~16 us per chunk in the deep doublings)
                    _lib.call("bjx_nuts_compact", stream, dref, I["SUB_ACTIVE"], -1, idx_ptr, idx_ptr,
                              ctl_ptr)
                    if poll:
                        read_count_async()
                ws.set_ctl(depth, s_base, -1, k0, k1, fold, off)
                ws.chunk_graph(k, n_cap).replay()
                if poll:  # occasional read: early exit / smaller batch for the chunks after this one
                    n_now = read_count_wait()
                    if n_now == 0:
                        stopped = True
                        break
                    n_cap = ws.bucket(n_now)
            del stopped
            _lib.call("bjx_nuts_merge", stream, dref, depth, n_doubling, idx_doubling.data_ptr())
        return _make_info(p0, ws.bufs, ws.fs, ws.is_, clone=True)

    not_capturable: dict = {}  # id -> callable whose capture failed once (the reference pins the id)

    def kernel_auto(rng_key, state: HMCState, logdensity_fn: Callable, step_size,
                    inverse_mass_matrix, max_num_doublings: int = 10, *, chain_offset: int = 0):
        if (is_capturable(logdensity_fn) and id(logdensity_fn) not in not_capturable
                and state.position.shape[0] > 0):
            try:
                return kernel_graph(rng_key, state, logdensity_fn, step_size, inverse_mass_matrix,
                                    max_num_doublings, chain_offset=chain_offset)
            except RuntimeError:
                # recording the callable failed (or the callable itself is broken, in which case the
                # plain driver below raises the same error again); the transition restarts from
                # `state`, which the graph driver never modifies
                not_capturable[id(logdensity_fn)] = logdensity_fn
                torch.cuda.synchronize(state.position.device)
        elif not is_capturable(logdensity_fn):
            warn_eager_driver(logdensity_fn, "nuts")
        return kernel_eager(rng_key, state, logdensity_fn, step_size, inverse_mass_matrix,
                            max_num_doublings, chain_offset=chain_offset)

    if use_graph == "auto":
        return kernel_auto
    return kernel_graph if use_graph else kernel_eager


def _os_environ():
    import os

    return os.environ


def free_running_supports(integrator, metric_kind: str, dim: int) -> bool:
    """Can the free-running tick kernels integrate with ``integrator``?  Velocity Verlet: always.  A general
    palindromic list with at most ``NUTS_MAX_MID`` middle stages: on the low-traffic kernels (diagonal metric, 16-byte
    rows of at most 1 024 floats) and -- round 6 -- on the general tick kernel for every other row length and for
    per-chain dense metrics (a leaf lasts K ticks either way).  What is left to lockstep steps is ONE shared dense
    matrix with D >= 128, whose products run on the MFMA GEMM (``run`` is then made of such steps by design)."""
    del dim
    if integrator is integrators.velocity_verlet:
        return True
    return metric_kind in ("diag", "dense") and integrator.num_gradients_per_step - 1 <= _lib.NUTS_MAX_MID


_SIDE_STREAMS: dict = {}  # (device index, main stream handle) -> a stream that runs CONCURRENTLY with it, or False


def _concurrent_side_stream(dev):
    """A stream whose launches overlap those of the current stream, or None.  HIP streams share a few hardware
    queues (four by default) and two streams on one queue run in order, so the pair is probed once
    (``bjx_stream_probe``: a kernel on the candidate waits for a kernel launched after it on the current stream) and
    the answer cached per (device, current stream)."""
    main = torch.cuda.current_stream(dev)
    key = (dev.index, main.cuda_stream)
    hit = _SIDE_STREAMS.get(key)
    if hit is not None:
        return hit or None
    flag = torch.zeros(2, dtype=torch.int32, device=dev)
    found = False
    for _ in range(8):
        cand = torch.cuda.Stream(device=dev)
        if cand.cuda_stream == main.cuda_stream:
            continue
        flag.zero_()
        cand.wait_stream(main)
        _lib.call("bjx_stream_probe", cand.cuda_stream, main.cuda_stream, flag.data_ptr(), 2000)
        torch.cuda.synchronize(dev)
        if int(flag[1].item()) == 1:
            found = cand
            break
    _SIDE_STREAMS[key] = found
    return found or None


_SPEC_STATS: dict = {}  # counters of the last run's speculative tail (run_free fills it; empty when the tail was not taken)
_DENSE_GEMM_CAP = 0  # run_free(dense_gemm=True): default size of a tick's momentum list (0 = a quarter of the ensemble)
_WARNED_LOCKSTEP_RUN: set = set()


def _warn_lockstep_run(integrator, metric_kind: str, dim: int) -> None:
    """One-time note (per integrator / metric kind / dimension) that ``run`` is made of lockstep ``step`` calls here:
    identical draws, but a transition then lasts as long as the ensemble's deepest tree (about 3 x slower at the
    C3 shape).  VERDICT r4 item 6: never degrade silently."""
    k = (tuple(integrator.coefficients), metric_kind, int(dim))
    if k in _WARNED_LOCKSTEP_RUN:
        return
    _WARNED_LOCKSTEP_RUN.add(k)
    import warnings

    warnings.warn(
        f"blackjax_amd.nuts.run: a {integrator.num_gradients_per_step}-gradient integrator with a {metric_kind!r} metric at "
        f"D = {dim} has no free-running tick kernel (they serve diagonal metrics, D % 4 == 0, D <= 1024 for "
        "multi-stage integrators); the run is made of lockstep `step` calls instead -- the same draws, but every "
        "transition waits for the deepest tree of the ensemble.", RuntimeWarning, stacklevel=3)


def auto_row_block(n_rows: int, dim: int) -> int:
    """Rows per group of the free-running schedule.  Default: ALL rows in one group.  Ticking the
    ensemble in Infinity-Cache-sized row groups (the analogue of ``hmc.auto_chain_block``; pass
    ``row_block=n`` to ``run_free``) was measured SLOWER at the C3 shape (32 768 x
    256: one group 138.6 M/s, groups of 16 384 / 8 192 / 4 096 rows 124.6 / 114.2 / 88.9 M/s; on the
    final kernels 184-191 vs 164 / 131): a tick moves more than the cache holds between two uses of
    a row whatever the grouping, and the smaller launches cost (NOTEBOOK.md section 7).  Round 6: two half-size
    groups on two CONCURRENT streams (one group's tick under the other's callable): 276 vs 279 M/s at T = 100, no gain."""
    return int(n_rows)


class NUTSRunInfo(NamedTuple):
    """Per-(transition, chain) records of a free-running run, each ``(num_steps, N)``: the scalar
    fields of ``NUTSInfo`` plus the log-density of the accepted state."""

    logdensity: torch.Tensor
    acceptance_rate: torch.Tensor
    energy: torch.Tensor
    num_integration_steps: torch.Tensor
    num_trajectory_expansions: torch.Tensor
    is_divergent: torch.Tensor
    is_turning: torch.Tensor
    step_size: Optional[torch.Tensor] = None  # adaptation runs only: the step size after transition t's update


def run_free(rng_key, state: HMCState, logdensity_fn: Callable, step_size, inverse_mass_matrix,
             num_steps: int, max_num_doublings: int = 10, *, divergence_threshold: float = 1000,
             chain_offset: int = 0, key_layout: str = "step_major", store_positions: bool = True,
             sync_every=None, use_graph="auto", graph_max_rows: int = 8192, adaptation=None,
             row_block=None, fuse_target: bool = False, integrator=integrators.velocity_verlet,
             dense_gemm: bool = False, dense_gemm_cap=None, spec_rows=None, keep_ends: bool = False,
             _handle: Optional[dict] = None):
    """``num_steps`` NUTS transitions of every chain WITHOUT lockstep (include/bjx_nuts.h,
    "free-running chains"): per tick each chain integrates one leapfrog of its own current tree and
    a chain that completes a transition starts its next one at once, so the user callable always
    sees N useful rows.  In a lockstep ``step`` a transition lasts as long as the deepest tree of
    the whole ensemble (2^max_depth - 1 dependent leaves when a single chain goes deep); here the
    run lasts as long as the chain with the largest TOTAL number of leapfrogs.

    Chain ``c`` at transition ``t`` uses exactly the key ``step`` would give it
    (``key_layout="step_major"``: ``split(split(rng_key, num_steps)[t], N)[c]``, the layout of
    ``run_inference_algorithm``; ``"chain_major"``: ``split(split(rng_key, N)[c], num_steps)[t]``),
    so the draws are identical to ``num_steps`` calls of ``step``.

    ``key_layout="step"`` (``num_steps`` must be 1): chain ``c`` uses ``split(rng_key, N)[c]`` -- the one
    transition ``step(rng_key, state)`` makes.

    The host syncs once per ``sync_every`` ticks (default 16; 128 with ``fuse_target``, where a whole chunk
    of ticks is one launch) (to stop, and to drop finished chains from the
    callable's batch once fewer than half of its rows are still running).  In the tail of a run,
    where a few deep trees are all that is left, a tick is a few microseconds of GPU work and the
    loop is bound by the host's launch rate; there (``use_graph="auto"``: once at most
    ``graph_max_rows`` rows are live) chunks of ``4 * sync_every`` ticks + callable invocations are
    recorded once per batch size and replayed as one HIP graph.  A callable that cannot be captured
    (it synchronises, say) makes "auto" fall back to plain launches; ``use_graph=True`` captures from
    the second chunk on whatever the batch size and lets a capture error propagate, ``False`` never
    captures.

    ``adaptation``: per-chain window adaptation carried by the chains themselves (include/bjx_nuts.h,
    ``adapt_*`` fields; built by ``window_adaptation(...).run(..., free_running=True)``): a dict with the
    host table ``tab`` (``(num_steps, NUTS_ADAPT_COLS)`` float32), ``target`` and the device buffers
    ``log_x, log_x_avg, avg_err, mu, step_size`` (N,), ``mean, m2, imm`` (N, D), all updated in place;
    ``step_size`` / ``inverse_mass_matrix`` arguments are then ignored in favour of those buffers.

    ``fuse_target=True`` (off by default; only for log-densities the library itself ships,
    ``blackjax_amd.targets.NealFunnel`` / ``DiagGaussian``, diagonal metric, ``D % 4 == 0``, ``D <= 512``):
    the tick kernels evaluate the log-density of the position they just produced THEMSELVES, with the
    device function the stand-alone target kernel runs, so a tick is one launch instead of two and the
    results are bit for bit those of the default path.  This leaves the external-callable contract of
    the engine (any PyTorch callable between two ticks) -- it shows what that contract costs: the tail of
    a run is two dependent launches per leapfrog, ~10-14 us, against one (NOTEBOOK.md section 7).

    ``integrator``: any palindromic coefficient list of ``blackjax_amd.integrators`` (round 4).  With K > 1
    gradients per leapfrog a leaf lasts K ticks -- K - 1 middle stages (kick b_i, drift a_i) and the tick that
    closes the leaf and does its bookkeeping (``bjx_nuts_async_t.int_stages``) -- on the low-traffic tick
    kernels (``free_running_supports``); results equal ``num_steps`` lockstep steps with the same integrator.

    ``dense_gemm=True`` (ONE shared dense inverse mass matrix, velocity Verlet): every product ``M^-1 p`` of a
    tick is one fp32 MFMA GEMM over the live rows -- the arithmetic of the lockstep ``step`` for this metric, so
    ``run(T)`` equals ``T`` steps bit for bit -- instead of one fp64 matrix-vector product per chain
    (``bjx_nuts_async_t.gemm_*``); at most ``dense_gemm_cap`` chains (default: a quarter of the ensemble) start a
    transition per tick, the others wait one tick.

    ``spec_rows`` (default 128, ``BJX_NUTS_SPEC_ROWS`` overrides, 0 = off): once at most that many chains are live the
    run continues on the TWO-STREAM SPECULATIVE TAIL (include/bjx_nuts.h, "Speculative tail"; diagonal metric,
    ``D % 4 == 0``, ``D <= 1024``, one-gradient integrators, recordable external callable, no adaptation): per
    leapfrog the latency-critical stream runs the callable and a light integrator that follows the key's direction
    schedule, while the tree bookkeeping -- the unchanged tick arithmetic -- replays the pushed gradients on a second
    stream and restarts the integrator at every transition end.  Same results bit for bit; the callable is
    additionally evaluated at a few positions past the end of each transition (wasted work, 2-3 leapfrogs).

    ``keep_ends`` (lean tick kernels: diagonal metric, ``D % 4 == 0``, ``D <= 1024``): every chain leaves the two ends of
    its LAST transition's trajectory in the work arrays (``bjx_nuts_async_t.keep_ends``) -- what ``step`` needs for
    ``NUTSInfo.trajectory_leftmost_state`` / ``rightmost_state``.

    ``_handle`` (private; ``key_layout="step"``, a plain key, no adaptation / fuse_target / dense_gemm): a dict that makes
    the call PERSISTENT -- every buffer is owned by the workspace and the per-call inputs (state, key, step size,
    metric) are copied into static buffers, so the recorded tail sequences survive the call; the dict then holds
    ``rerun(rng_key, state, step_size, inverse_mass_matrix)``, which repeats the run on the same workspace without
    any allocation or recording, and ``work`` (the arrays ``keep_ends`` fills).  Results of a persistent call are clones.

    Returns ``(final_state, positions, info)``: ``positions`` is ``(num_steps, N, D)`` (``None`` when
    ``store_positions=False``), ``info`` a ``NUTSRunInfo``."""
    import numpy as np

    from . import random as bjx_random

    q = check_batch(state.position, "state.position").clone()
    logp = check_batch(state.logdensity, "state.logdensity").clone()
    g = check_batch(state.logdensity_grad, "state.logdensity_grad").clone()
    N, D = q.shape
    dev = q.device
    T = int(num_steps)
    max_depth = int(max_num_doublings)
    if max_depth < 1:
        raise ValueError("free-running chains need max_num_doublings >= 1")
    if key_layout not in ("step_major", "chain_major", "step"):
        raise ValueError("key_layout must be 'step_major', 'chain_major' or 'step'")
    if key_layout == "step" and T != 1:
        raise ValueError("key_layout='step' is the single transition of step(): num_steps must be 1")
    vg = value_and_grad(logdensity_fn)
    f32 = dict(dtype=torch.float32, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)
    persistent = _handle is not None
    if persistent and (key_layout not in ("step", "step_major") or adaptation is not None or fuse_target or dense_gemm
                       or bjx_random.key_spec(rng_key)[2] >= 0 or N == 0 or T == 0):
        raise ValueError("a persistent free-running workspace serves key_layout='step' / 'step_major' with a plain key, "
                         "without adaptation, fuse_target or dense_gemm")
    # A persistent ``step_major`` workspace is built for a CAPACITY of transitions (``_handle["capacity"]`` >= T): the
    # recorded launches carry ``n_steps`` by value, so a call of T <= capacity transitions starts every chain's counter at
    # t0 = capacity - T and uses rows t0 .. capacity - 1 of the key table and of the output arrays (the kernels index
    # both by the chain's absolute counter: csrc/bjx_nuts_tick_dev.h, async_ctx and the transition end)
    Tc = T
    if persistent and key_layout == "step_major":
        Tc = max(T, int(_handle.get("capacity", T)))
    cur = {"T": T, "t0": Tc - T}
    adapt_fields = {}
    out_step_size = None
    if adaptation is not None:
        ad = adaptation
        for name, shape in (("log_x", (N,)), ("log_x_avg", (N,)), ("avg_err", (N,)), ("mu", (N,)),
                            ("step_size", (N,)), ("mean", (N, D)), ("m2", (N, D)), ("imm", (N, D))):
            b = ad[name]
            if not (isinstance(b, torch.Tensor) and b.is_cuda and b.dtype == torch.float32
                    and tuple(b.shape) == shape and b.is_contiguous()):
                raise ValueError(f"adaptation['{name}'] must be a contiguous float32 device tensor of shape {shape}")
        tab_host = np.ascontiguousarray(ad["tab"], dtype=np.float32)
        if tab_host.shape != (T, _lib.NUTS_ADAPT_COLS):
            raise ValueError(f"adaptation['tab'] must have shape ({T}, {_lib.NUTS_ADAPT_COLS})")
        adapt_tab = torch.as_tensor(tab_host, device=dev)
        out_step_size = torch.empty((T, N), **f32)
        step_size, inverse_mass_matrix = ad["step_size"], metrics.PerChainDiag(ad["imm"])
        adapt_fields = dict(
            adapt_tab=adapt_tab.data_ptr(), adapt_target=float(ad["target"]), adapt_reserved=0.0,
            adapt_log_x=ad["log_x"].data_ptr(), adapt_log_x_avg=ad["log_x_avg"].data_ptr(),
            adapt_avg_err=ad["avg_err"].data_ptr(), adapt_mu=ad["mu"].data_ptr(),
            adapt_step_size=ad["step_size"].data_ptr(), adapt_mean=ad["mean"].data_ptr(),
            adapt_m2=ad["m2"].data_ptr(), adapt_imm=ad["imm"].data_ptr(),
            out_step_size=out_step_size.data_ptr())
    metric = metrics.default_metric(inverse_mass_matrix, N, D, dev)
    integrators.check_supported(integrator, allow_general=True)
    general = integrator is not integrators.velocity_verlet
    kick_c, drift_c = integrator.coefficients[0::2], integrator.coefficients[1::2]
    if general and (fuse_target or not free_running_supports(integrator, metric.kind, D)):
        raise NotImplementedError(
            f"free-running ticks with a multi-stage integrator: at most {_lib.NUTS_MAX_MID + 1} gradients per leapfrog "
            "and no fuse_target")
    if metric.kind != "diag" and adaptation is not None:
        raise NotImplementedError("free-running per-chain adaptation is implemented for the diagonal metric")
    gemm = bool(dense_gemm)
    if gemm and (metric.kind != "dense" or general or fuse_target):
        raise NotImplementedError("dense_gemm=True: one shared dense inverse mass matrix, velocity Verlet, no fuse_target")
    eps, eps_pc = step_size_args(step_size, N, dev)
    eps_buf = imm_buf = None
    if persistent:  # static copies: recorded launches carry the descriptors by value
        eps_buf = torch.empty(N, **f32)
        eps_buf.fill_(eps) if eps_pc is None else eps_buf.copy_(eps_pc)
        imm_buf = metric.imm.clone()
        metric = metric._replace(imm=imm_buf)
        eps, eps_pc = 0.0, eps_buf
    if adaptation is not None and (eps_pc is None or eps_pc.data_ptr() != adaptation["step_size"].data_ptr()
                                   or metric.imm.data_ptr() != adaptation["imm"].data_ptr()
                                   or metric.imm_stride != D):
        raise RuntimeError("adaptation buffers were copied on the way to the kernels")
    info = NUTSRunInfo(torch.empty((Tc, N), **f32), torch.empty((Tc, N), **f32), torch.empty((Tc, N), **f32),
                       torch.empty((Tc, N), **i32), torch.empty((Tc, N), **i32),
                       torch.empty((Tc, N), dtype=torch.bool, device=dev),
                       torch.empty((Tc, N), dtype=torch.bool, device=dev), out_step_size)
    positions = torch.empty((Tc, N, D), **f32) if store_positions else None
    if T == 0 or N == 0:
        return HMCState(q, logp, g), positions, info

    t_first = 0
    if key_layout == "step_major":
        keys = bjx_random.split(rng_key, T)
        step_keys = torch.zeros((Tc, 2), **i32)
        step_keys[Tc - T:].copy_(torch.as_tensor(np.ascontiguousarray(keys).view(np.int32).reshape(T, 2)))
        k0 = k1 = 0
    elif key_layout == "step":
        k0, k1, fold = bjx_random.key_spec(rng_key)
        step_keys = None
        if fold < 0:  # a plain key: the transition's own key; a ChainMajorKey: (run key, transition index)
            step_keys = torch.as_tensor(np.array([[k0, k1]], dtype=np.uint32).view(np.int32), device=dev)
            k0 = k1 = 0
        else:
            t_first = int(fold)
    else:
        step_keys = None
        k0, k1 = bjx_random.key_words(rng_key)
    bufs = {n: torch.empty_like(q) for n in _BUFS}
    ck_r = torch.empty((N, max_depth, D), **f32)
    ck_rs = torch.empty_like(ck_r)
    fs = torch.empty((_lib.NUTS_NF, N), **f32)
    is_ = torch.zeros((_lib.NUTS_NI, N), **i32)
    p = torch.empty_like(q)
    qf = torch.zeros_like(q)
    t_done = torch.full((N,), Tc - T, **i32)
    phase = torch.zeros(N, **i32)
    # work buffers of the low-traffic tick kernels (include/bjx_nuts.h: rec / front_p; diagonal metric)
    dense_f = _dense_fields(metric.kind, metric.imm, N, D, max_depth, dev)
    v0 = None
    if metric.kind != "diag":  # the tick kernel draws p = L^{-T} z and v0 = M^{-1} p per chain
        v0 = torch.empty_like(q)
        dense_f["fields"]["v0"] = v0.data_ptr()
    gemm_bufs = None
    if gemm:
        # compact kicked momenta / their velocities (one GEMM per product), and the momentum list of a tick
        cap = dense_gemm_cap if dense_gemm_cap is not None else _DENSE_GEMM_CAP
        cap = int(cap) if cap and int(cap) > 0 else max(128, -(-N // 4))
        cap = min(N, -(-cap // 128) * 128)
        gemm_bufs = (torch.zeros_like(q), torch.zeros_like(q), torch.zeros((cap, D), **f32),
                     torch.zeros((cap, D), **f32), torch.zeros((cap, D), **f32), cap,
                     metric.mass_sqrt_t.t().contiguous())  # L^{-T} row-major: read as stored by the TN GEMM kernel
        dense_f["fields"]["v_pre"] = gemm_bufs[1].data_ptr()
    rec = torch.zeros((N, _lib.NUTS_REC_WORDS), **i32) if v0 is None else None
    front_p = torch.empty_like(q) if v0 is None else None
    end_list = torch.empty((2, N), **i32)
    end_count = torch.zeros(2, **i32)
    n_done = torch.zeros(1, **i32)
    desc = _lib.NutsDesc(
        N=N, D=D, max_depth=max_depth, reserved=0, imm=metric.imm.data_ptr(),
        imm_stride=metric.imm_stride, eps_per_chain=_lib.ptr(eps_pc), eps=eps,
        divergence_threshold=float(divergence_threshold), key0=k0, key1=k1,
        chain_offset=int(chain_offset), step_fold=-1, q0=q.data_ptr(), g0=g.data_ptr(),
        p0=p.data_ptr(), ckpt_r=ck_r.data_ptr(), ckpt_rs=ck_rs.data_ptr(), fs=fs.data_ptr(),
        is_=is_.data_ptr(), **{n: b.data_ptr() for n, b in bufs.items()}, **dense_f["fields"],
        int_kick=kick_c[0] if general else 0.0, int_drift=drift_c[0] if general else 0.0)
    run = _lib.NutsAsync(
        step_keys=_lib.ptr(step_keys), t_first=t_first, n_steps=Tc, q=q.data_ptr(), g=g.data_ptr(),
        logp=logp.data_ptr(), p=p.data_ptr(), t=t_done.data_ptr(), phase=phase.data_ptr(),
        n_done=n_done.data_ptr(), rows=None, n_rows=N, out_position=_lib.ptr(positions),
        out_logdensity=info.logdensity.data_ptr(), out_acceptance_rate=info.acceptance_rate.data_ptr(),
        out_energy=info.energy.data_ptr(),
        out_num_integration_steps=info.num_integration_steps.data_ptr(),
        out_num_trajectory_expansions=info.num_trajectory_expansions.data_ptr(),
        out_is_divergent=info.is_divergent.data_ptr(), out_is_turning=info.is_turning.data_ptr(),
        rec=_lib.ptr(rec), front_p=_lib.ptr(front_p), end_list=end_list.data_ptr(),
        end_count=end_count.data_ptr(), mass_sqrt_t=_lib.ptr(metric.mass_sqrt_t if v0 is not None else None),
        v0=_lib.ptr(v0), **adapt_fields)
    if keep_ends:
        if rec is None or fuse_target or D % 4 != 0 or D > 1024:
            raise NotImplementedError("keep_ends is served by the lean tick kernels: diagonal metric, D % 4 == 0, D <= 1024, "
                                      "external callable")
        run.keep_ends = 1
    if gemm_bufs is not None:
        (run.gemm_pc, run.gemm_vc, run.gemm_z, run.gemm_pm, run.gemm_vm) = (b.data_ptr() for b in gemm_bufs[:5])
        run.gemm_cap = gemm_bufs[5]
        run.gemm_mass_sqrt = gemm_bufs[6].data_ptr()  # both matrices read as stored (bjx_dense_matmul_bt / apply_imm_t)
        run.gemm_imm_t = metric.imm_t.data_ptr()
    if general and len(drift_c) > 1:  # middle stages (b_2, a_2), ..., (b_K, a_K); the closing kick b_1 is int_kick
        run.int_stages = len(drift_c)
        for i in range(1, len(drift_c)):
            run.int_mid_kick[i - 1], run.int_mid_drift[i - 1] = kick_c[i], drift_c[i]
    fused = False
    rtc_module = None
    # with an engine-resident target a whole chunk of ticks is ONE launch (k_nuts_async_multi) at every batch size:
    # measured at C3 (NOTEBOOK.md section 7) 322 / 340 / 220 M/s at T = 20 / 100 / 400 against 190 / 212 / 112 with
    # one launch per tick -- the one-tick target kernels were removed in round 5
    if fuse_target:
        spec = getattr(logdensity_fn, "_bjx_fused_target", None)
        spec = spec(D) if callable(spec) else None
        if spec is None or metric.kind != "diag" or D % 4 != 0 or D > 512 or rec is None:
            raise NotImplementedError(
                "fuse_target=True needs a blackjax_amd.targets log-density the tick kernels can evaluate "
                "(NealFunnel; DiagGaussian with D > 128), a diagonal metric, D % 4 == 0 and D <= 512")
        if spec[0] == "rtc":
            # a user-written device target (targets.DeviceTarget): the multi-tick kernel of csrc/bjx_nuts_tick_dev.h is
            # compiled around it by hiprtc (blackjax_amd/rtc.py) and launched through the module API; every
            # chunk of ticks is one launch (the library's one-tick kernels do not know the target)
            rtc_target = spec[1]
            rtc_module = rtc_target.nuts_module()
            run.target_kind, run.target_vec = _lib.NUTS_TARGET_USER, rtc_target._params_ptr(dev) or None
        else:
            run.target_kind, run.target_vec = int(spec[0]), _lib.ptr(spec[1])
        fused_keep = spec[1]  # noqa: F841  (keeps the parameter vector / the compiled module alive for the run)
        fused = True
    dref, rref = ctypes.byref(desc), ctypes.byref(run)
    stream = _lib.current_stream()
    # K ticks per leaf; + one tick per transition: the busy-phase leaf kernel finishes a transition in the
    # tick AFTER the one that completed its tree (deferred transition ends, k_nuts_async_tick3<.., DEFER>)
    def ticks_bound(T_):
        b_ = T_ * (((1 << max_depth) - 1) * max(1, len(drift_c) if general else 1) + 1) + 4
        if gemm_bufs is not None:
            # a chain may wait for a slot of the momentum list.  Slots are handed out by atomicAdd in arbitrary order and
            # a waiting chain competes again in the next tick, so ceil(N / cap) - 1 ticks per start is the EXPECTED wait,
            # not a strict bound: twice that plus a constant, so that a starved chain cannot trip the "did not
            # finish within its tick bound" error spuriously (the bound only exists to stop a runaway loop)
            b_ += 2 * T_ * (-(-N // gemm_bufs[5])) + 64
        return b_

    max_ticks = ticks_bound(T)
    if sync_every is None:
        sync_every = 128 if fused else 16
    sync_every = int(_os_environ().get("BJX_NUTS_SYNC_EVERY", sync_every))
    sync_every = max(2, int(sync_every) + (int(sync_every) & 1))  # even (GEMM mode alternates its momentum lists per tick)
    import os as _os

    graph_max_rows = int(_os.environ.get("BJX_NUTS_TAIL_ROWS", graph_max_rows))
    if use_graph not in (True, False, "auto"):
        raise ValueError("use_graph must be True, False or 'auto'")
    can_record = use_graph is True or (use_graph == "auto" and is_capturable(logdensity_fn))
    if use_graph == "auto" and not can_record:
        warn_eager_driver(logdensity_fn, "nuts.run")

    def multi_tick(run_k, qf_t, lp_t, g_t):
        """One launch = ``run_k.ticks_per_launch`` ticks of every row of ``run_k`` (engine-resident target)."""
        if rtc_module is None:
            _lib.call("bjx_nuts_async_tick", _lib.current_stream(), dref, ctypes.byref(run_k), qf_t.data_ptr(),
                      lp_t.data_ptr(), g_t.data_ptr())
        elif run_k.n_rows > 0:
            from . import rtc

            rtc_module.launch(rtc.nuts_kernel_name(D), min(int(run_k.n_rows), 1 << 20), 64, _lib.current_stream(),
                              desc, run_k, ctypes.c_void_p(qf_t.data_ptr()), ctypes.c_void_p(lp_t.data_ptr()),
                              ctypes.c_void_p(g_t.data_ptr()))

    def make_run(rows, n_rows):
        r = _lib.NutsAsync()
        ctypes.memmove(ctypes.byref(r), ctypes.byref(run), ctypes.sizeof(run))  # copy of the template
        r.rows, r.n_rows = _lib.ptr(rows), n_rows
        return r

    class _Group:
        """A set of compact rows ticked together: (qf, logp_f, gf) batch + its row list."""

        def __init__(self, rows, n_rows, qf_g):
            self.rows, self.n_rows, self.qf = rows, n_rows, qf_g
            self.run = make_run(rows, n_rows)
            self.rref = ctypes.byref(self.run)
            self.logp_f = torch.zeros(n_rows, **f32)  # the first tick only starts transitions
            self.gf = torch.zeros_like(qf_g)
            self.graph = None  # (CUDAGraph, static logp_f, static gf, n_ticks)
            self.eager_chunks = 0

        def chunk(self, n_ticks, logp_f, gf):
            """``n_ticks`` ticks, each followed by the callable on the group's batch."""
            if fused:
                # engine-resident target + one-launch ticks: the whole chunk is ONE launch, every wave
                # advancing its chain n_ticks times (bjx_nuts_async_t.ticks_per_launch)
                self.run.tick, self.run.ticks_per_launch = 0, n_ticks
                multi_tick(self.run, self.qf, logp_f, gf)
                return logp_f, gf
            for i in range(n_ticks):
                self.run.tick = i & 1
                _lib.call("bjx_nuts_async_tick", _lib.current_stream(), dref, self.rref, self.qf.data_ptr(),
                          logp_f.data_ptr(), gf.data_ptr())
                logp_f, gf = eval_logdensity(vg, self.qf)
            return logp_f, gf

        def advance(self, n_ticks, record):
            nonlocal can_record
            if self.graph is not None and self.graph[3] == n_ticks:
                self.graph[0].replay()
                return
            self.graph = None
            self.logp_f, self.gf = self.chunk(n_ticks, self.logp_f, self.gf)
            self.eager_chunks += 1
            if record:
                # recording does not execute anything, so the chains' state is untouched by it
                lp_s, g_s = self.logp_f.clone(), self.gf.clone()
                try:
                    cg = new_graph()
                    with record_graph(cg):
                        lp_e, g_e = self.chunk(n_ticks, lp_s, g_s)
                        if lp_e is not lp_s:
                            lp_s.copy_(lp_e)
                            g_s.copy_(g_e)
                    self.graph = (cg, lp_s, g_s, n_ticks)
                    self.logp_f, self.gf = lp_s, g_s
                except Exception:
                    if use_graph is True:
                        raise
                    can_record = False  # "auto": this callable cannot be captured -- plain launches
                    torch.cuda.synchronize()

    src_work = torch.empty(N, **i32)
    n_out = torch.zeros(1, **i32)

    class _Tail:
        """The tail of a run (at most ``cap`` live rows) on FIXED-CAPACITY buffers: the kernels read
        the live-row count from device memory (``n_rows_dev``, include/bjx_nuts.h), so ONE recorded
        chunk of ticks per (buffer set, view size) serves every batch size in between -- compaction
        only rewrites the row list, the pending positions and the count.  Two buffer sets because a
        compaction cannot work in place.

        Round 3: the launch geometry and the callable's batch follow TIERS of the live-row count
        (``cap``, then 4 096, 2 048, 512, 128, 32 rows: prefix views of the same buffers, one recording each) instead
        of staying at ``cap`` to the end: with three live chains the callable used to be evaluated on
        all ``cap`` rows every tick.  Round 6: ``cap`` is up to 8 192 rows (2 048 before) -- between 2 048 and 8 192 live
        rows a tick is 16-20 us of GPU work and plain launches from Python took 33 us (NOTEBOOK section 19).  Rows of a view beyond the live count always hold a
        VALID position (the buffers start as copies of a current chain state and are only ever
        overwritten with pending positions), so a log-density that validates its support never sees
        zeros or garbage there."""

        TIERS = (32, 128, 512, 2048, 4096)

        def __init__(self, cap, n_ticks, reps):
            # a recorded sequence of n_ticks ticks, replayed `reps` times per host sync: recording
            # costs ~40 us per tick, so a short sequence pays for itself within a few hundred ticks
            self.cap, self.n_ticks, self.reps = cap, n_ticks, reps
            self.tiered = True  # tail sequences follow tiers of the live-row count (cap / 512 / 128 / 32 rows)
            self.rows = [torch.zeros(cap, **i32) for _ in range(2)]
            # valid positions everywhere.  repeat(), not expand().contiguous(): with cap == 1 the latter IS q[:1] (already
            # contiguous, no copy), and the last live chain's pending positions were then written into chain 0's state
            # row (round 5: found by the row-width sweep of tests/test_nuts_v3_gpu.py; latent since round 3)
            self.qf = [q[:1].repeat(cap, 1) for _ in range(2)]
            self.n_dev = [torch.zeros(1, **i32) for _ in range(2)]
            self.run = []
            for k in range(2):
                r = make_run(self.rows[k], cap)
                r.n_rows_dev = self.n_dev[k].data_ptr()
                self.run.append(r)
            self.graph = {}   # (buffer set, view rows) -> (CUDAGraph, ticks per replay)
            self.cur, self.n_cur, self.view = 0, 0, cap
            if fused:  # in/out (logp, grad) of the pending positions, one pair per buffer set
                self.lp = [torch.zeros(cap, **f32) for _ in range(2)]
                self.g = [torch.zeros((cap, D), **f32) for _ in range(2)]

        def tier(self, n_active):
            if not self.tiered:
                return self.cap
            for v in self.TIERS:
                if n_active <= v and v < self.cap:
                    return v
            return self.cap

        def _after_compaction(self, k, n_active):
            self.view = self.tier(n_active)
            self.run[k].n_rows = self.view
            self.cur, self.n_cur = k, n_active
            if fused:  # (logp, grad) of the gathered positions: one stand-alone evaluation, then in place
                lp, g_ = eval_logdensity(vg, self.qf[k][:self.view])
                self.lp[k][:self.view].copy_(lp)
                self.g[k][:self.view].copy_(g_)

        def chunk_ticks(self):
            """Ticks per recorded sequence: the few-row tiers are pure launch latency (two dependent
            kernels per tick), so their sequences are longer -- fewer replay boundaries per tick."""
            mult = 4
            return self.n_ticks * (mult if (self.tiered and self.view <= 128) else 1)

        def enter(self, groups_in, n_active):
            off = 0
            for g_ in groups_in:
                _lib.call("bjx_nuts_async_compact", stream, dref, g_.rref, g_.qf.data_ptr(),
                          self.rows[0][off:].data_ptr(), self.qf[0][off:].data_ptr(), src_work.data_ptr(),
                          n_out.data_ptr())
                off += int(n_out.item()) if len(groups_in) > 1 else n_active
            assert off == n_active, (off, n_active)
            self.n_dev[0].fill_(n_active)
            self._after_compaction(0, n_active)

        def wants_compaction(self, n_active):
            if self.tiered:
                return self.tier(n_active) < self.view
            return n_active <= self.n_cur // 2 and self.n_cur > 64

        def compact(self, n_active):
            x, y = self.cur, self.cur ^ 1
            _lib.call("bjx_nuts_async_compact", stream, dref, ctypes.byref(self.run[x]), self.qf[x].data_ptr(),
                      self.rows[y].data_ptr(), self.qf[y].data_ptr(), src_work.data_ptr(),
                      self.n_dev[y].data_ptr())
            self._after_compaction(y, n_active)

        def _body(self, k, n_ticks):
            """``n_ticks`` x (callable on the pending positions, tick).  CALLABLE FIRST: what carries
            over from one sequence to the next is then ``qf`` alone -- a fixed buffer the last tick
            wrote -- and a replay needs no copies of the callable's outputs into static inputs (round 2
            copied logp and the gradient once per sequence: 0.85 us per tick in the deep tail)."""
            qf_v = self.qf[k][:self.view]
            rref_k = ctypes.byref(self.run[k])
            if fused:
                lp, g_ = self.lp[k][:self.view], self.g[k][:self.view]
                self.run[k].tick, self.run[k].ticks_per_launch = 0, n_ticks  # the whole sequence as one launch
                multi_tick(self.run[k], qf_v, lp, g_)
                return
            for i in range(n_ticks):
                lp, g_ = eval_logdensity(vg, qf_v)
                self.run[k].tick = i & 1
                _lib.call("bjx_nuts_async_tick", _lib.current_stream(), dref, rref_k, qf_v.data_ptr(),
                          lp.data_ptr(), g_.data_ptr())

        def advance(self):
            """-> ticks issued"""
            nonlocal can_record
            k = self.cur
            hit = self.graph.get((k, self.view))
            if hit is not None:
                for _ in range(self.reps):
                    hit[0].replay()
                return hit[1] * self.reps
            n = self.chunk_ticks()
            self._body(k, self.n_ticks)  # one short plain sequence first (kernels, allocator warm)
            issued = self.n_ticks
            if can_record:
                try:
                    cg = new_graph()
                    with record_graph(cg):
                        self._body(k, n)
                    self.graph[(k, self.view)] = (cg, n)
                except Exception:
                    if use_graph is True:
                        raise
                    can_record = False
                    torch.cuda.synchronize()
            return issued

    spec_rows = int(_os.environ.get("BJX_NUTS_SPEC_ROWS", 128 if spec_rows is None else spec_rows))
    spec_ok = (spec_rows > 0 and can_record and rec is not None and not fused and gemm_bufs is None
               and adaptation is None and not (general and len(drift_c) > 1) and D % 4 == 0 and D <= 1024
               and max_depth <= 30)
    side_stream = _concurrent_side_stream(dev) if spec_ok else None  # probed once per (device, stream), then cached
    spec_ok = spec_ok and side_stream is not None

    class _SpecTail:
        """The two-stream speculative tail (include/bjx_nuts.h): stream A = the current stream, [callable, integrate]
        per leapfrog as one recorded sequence; stream B = ``side``, one long-lived bookkeeping launch per sequence."""

        RING = 64      # ring slots per row
        LEAD = 4       # records the integrator may run ahead of the bookkeeper's consumed count (2: stalls; 4 .. 63: flat at
                       # D = 256, where a record costs the bookkeeper 3.2 us against 4.9 us per leaf on stream A)
        SEQ = int(_os.environ.get("BJX_NUTS_SPEC_SEQ", 64))  # leapfrogs per recorded sequence of stream A (32 / 64 / 128 measured within 4 % of each other)
        SEQ0 = 8       # plain leapfrogs before the recording (kernels, allocator warm)
        REPS = int(_os.environ.get("BJX_NUTS_SPEC_REPS", 1))  # sequences queued per host poll (round 6: 1 -- the host learns of the
                       # run's end one batch late, and a batch of 128 leapfrogs past the end cost `step` 4 % at C3)
        TIMEOUT_US = 20000

        def __init__(self, cap):
            self.cap = cap
            self.rows = torch.zeros(cap, **i32)
            self.qf = q[:1].repeat(cap, 1)  # valid positions everywhere (see _Tail)
            self.n_dev = torch.zeros(1, **i32)
            mk = lambda: torch.zeros((cap, D), **f32)  # noqa: E731
            self.fp, self.qf_book = mk(), mk()
            self.ends = [mk() for _ in range(6)]
            self.iw = torch.zeros((cap, _lib.NUTS_SPEC_IW), **i32)
            self.bw = torch.zeros((cap, _lib.NUTS_SPEC_IW), **i32)
            self.ring_g = torch.zeros((cap, self.RING, D), **f32)
            self.ring_tag = torch.zeros((cap, self.RING, _lib.NUTS_SPEC_TAG), **i32)
            self.avail = torch.zeros(cap, **i32)
            self.ack = torch.zeros(cap, dtype=torch.int64, device=dev)
            self.a_seq = torch.zeros(1, **i32)
            self.dbg = torch.zeros(8, **i32)
            self.run = make_run(self.rows, cap)
            self.run.n_rows_dev = self.n_dev.data_ptr()
            e = self.ends
            self.spec = _lib.NutsSpec(
                n_rows=cap, n_rows_dev=self.n_dev.data_ptr(), rows=self.rows.data_ptr(), ring=self.RING,
                lead=self.LEAD,
                qf=self.qf.data_ptr(), fp=self.fp.data_ptr(), eLq=e[0].data_ptr(), eLp=e[1].data_ptr(),
                eLg=e[2].data_ptr(), eRq=e[3].data_ptr(), eRp=e[4].data_ptr(), eRg=e[5].data_ptr(),
                iw=self.iw.data_ptr(), ring_g=self.ring_g.data_ptr(), ring_tag=self.ring_tag.data_ptr(),
                avail=self.avail.data_ptr(), ack=self.ack.data_ptr(), qf_book=self.qf_book.data_ptr(), bw=self.bw.data_ptr(),
                a_seq=self.a_seq.data_ptr(), dbg=self.dbg.data_ptr())
            self.rref, self.sref = ctypes.byref(self.run), ctypes.byref(self.spec)
            self.side = side_stream
            self.graph = None
            self.seq_no = 0
            self.t_enter = __import__("time").perf_counter()

        def enter(self, src_rref, src_qf):
            """Live rows of the source batch (between two ticks) -> this tail."""
            _lib.call("bjx_nuts_async_compact", stream, dref, src_rref, src_qf.data_ptr(), self.rows.data_ptr(),
                      self.qf.data_ptr(), src_work.data_ptr(), self.n_dev.data_ptr())
            _lib.call("bjx_nuts_spec_enter", stream, dref, self.rref, self.sref)
            self.side.wait_stream(torch.cuda.current_stream(dev))

        def _seq_a(self, n):
            for i in range(n):
                lp, g_ = eval_logdensity(vg, self.qf)
                _lib.call("bjx_nuts_spec_integrate", _lib.current_stream(), dref, self.rref, self.sref,
                          lp.data_ptr(), g_.data_ptr(), 1 if i == n - 1 else 0)

        def _book(self):
            self.seq_no += 1
            with torch.cuda.stream(self.side):
                _lib.call("bjx_nuts_spec_book", _lib.current_stream(), dref, self.rref, self.sref, self.seq_no,
                          self.TIMEOUT_US)

        def advance(self):
            """-> leapfrogs issued on stream A"""
            nonlocal can_record
            # the bookkeeper of a sequence is launched BEFORE the sequence: replaying a recorded sequence keeps the host
            # busy for about as long as the GPU needs to run it (every node is enqueued by the host), so a bookkeeper
            # launched behind the replay started a whole sequence late (measured: a ring of lag, 26-56 wasted
            # leapfrogs per transition end instead of 3-4)
            if self.graph is not None:
                for _ in range(self.REPS):
                    self._book()
                    self.graph.replay()
                return self.SEQ * self.REPS
            n0 = self.SEQ0 if can_record else self.SEQ
            self._book()
            self._seq_a(n0)
            if can_record:
                try:
                    cg = new_graph()
                    with record_graph(cg):
                        self._seq_a(self.SEQ)
                    self.graph = cg
                except Exception:
                    if use_graph is True:
                        raise
                    can_record = False
                    torch.cuda.synchronize()
            return n0

        def poll(self):
            with torch.cuda.stream(self.side):  # the finished-chain counter is stream B's
                return poll_done()

        def finish(self):
            torch.cuda.current_stream(dev).wait_stream(self.side)
            d = [int(v) for v in self.dbg.tolist()]
            pushed = int(self.iw[:, 5].sum().item())
            _SPEC_STATS.update(rows=self.cap, pushed=pushed, mismatches=d[0], stalls=d[1], restarts=d[2], stale=d[3],
                               timeouts=d[4], out_of_order=d[5], sequences=self.seq_no,
                               seconds=__import__("time").perf_counter() - self.t_enter,
                               book_us_per_record=(d[6] / 100.0 / d[7]) if d[7] else None, book_records=d[7])
            if d[0] or d[5]:
                raise RuntimeError(f"speculative NUTS tail: the bookkeeper's replica disagreed with the integrator "
                                   f"({d[0]} position mismatches, {d[5]} out-of-order records) -- results discarded")

    # Lagged completion polling for the tail: the finished-chain count is copied to pinned host memory
    # behind every batch of replays and the host reads the copy of the PREVIOUS batch, so the next
    # batch is always queued before the host waits -- the GPU never idles on the round trip of a
    # blocking .item() (0.5-1 us per tick of a 12 us tick).  The count only grows: a stale (smaller)
    # value is safe for the tier choice, and the run ends at most one batch late (ticks over finished
    # chains do nothing).
    lag_busy = _os.environ.get("BJX_NUTS_LAG_BUSY", "1") != "0"
    pinned = [torch.zeros(1, dtype=torch.int32).pin_memory() for _ in range(2)]
    poll_ev = [torch.cuda.Event() for _ in range(2)]
    poll = {"slot": 0, "primed": False, "last": 0}

    def poll_done():
        k = poll["slot"]
        pinned[k].copy_(n_done, non_blocking=True)
        poll_ev[k].record()
        poll["slot"] = k ^ 1
        if poll["primed"]:
            poll_ev[k ^ 1].synchronize()
            poll["last"] = int(pinned[k ^ 1][0])
        poll["primed"] = True
        return poll["last"]

    groups: list = []
    tail_ctx = spec_ctx = None
    ticks_left = max_ticks
    # ONE transition from a common start (`step`): every chain with d doublings ends at tick 2^d + 1 (start, 2^d - 1
    # leaves, deferred end), so the live count only changes there -- the host looks right behind those ticks and compacts
    # once at most 70 % of a batch is live.  Measured at C3 (tools/step_vs_run1.py, same call): 7.2 ms per transition
    # against 7.9 with the schedule of a long run (a look every 16 ticks, compaction at one half) and 7.4 with a look
    # every 8 ticks.  Runs of many transitions keep the old schedule: their chains desynchronise within a transition.
    cohort_plan, compact_frac = None, 0.5
    cohort_ok = not fused and gemm_bufs is None
    static = {"tail": None, "spec": None}  # persistent workspace: the tails (and their recorded sequences) of earlier calls

    def to_spec(src_rref, src_qf, n_live):
        nonlocal spec_ctx, ticks_left
        if persistent:  # fixed capacity: one recording serves every call
            if static["spec"] is None:
                static["spec"] = _SpecTail(max(1, min(N, spec_rows)))
            spec_ctx = static["spec"]
            spec_ctx.dbg.zero_()
            spec_ctx.t_enter = __import__("time").perf_counter()
        else:
            spec_ctx = _SpecTail(max(1, n_live))
        spec_ctx.enter(src_rref, src_qf)
        poll["last"] = max(poll["last"], N - n_live)
        ticks_left += ticks_left // 2 + 4096  # leapfrogs speculated past transition ends, ring stalls

    def execute():
        """The run itself, on the buffers set up above (a persistent workspace calls it once per transition)."""
        nonlocal groups, tail_ctx, spec_ctx, ticks_left, max_ticks, cohort_plan, compact_frac
        max_ticks = ticks_bound(cur["T"])
        cohort_plan, compact_frac = None, 0.5
        if cur["T"] == 1 and cohort_ok:
            cohort_plan, compact_frac = [(1 << d_) + 1 for d_ in range(4, max_depth + 1)], 0.7
        # Row groups: the ensemble may be ticked group by group, each advanced by a chunk of ticks before
        # the next one gets its turn (chains are independent, so the results do not depend on the
        # grouping).  One group by default -- see auto_row_block for the measurement.
        rb = auto_row_block(N, D) if row_block is None else int(row_block)
        blk = N if not rb or rb >= N else rb
        groups = []
        for s0 in range(0, N, blk):
            n_g = min(blk, N - s0)
            rows_g = None if blk >= N else torch.arange(s0, s0 + n_g, **i32)
            groups.append(_Group(rows_g, n_g, qf[s0:s0 + n_g]))
        tail_ctx = spec_ctx = None
        ticks_left = max_ticks
        poll.update(primed=False, last=0)
        _SPEC_STATS.clear()
        while ticks_left > 0:
            if spec_ctx is not None:
                ticks_left -= spec_ctx.advance()
                if N - spec_ctx.poll() == 0:
                    break
                continue
            if tail_ctx is not None:
                if spec_ok and tail_ctx.n_cur <= spec_rows:
                    to_spec(ctypes.byref(tail_ctx.run[tail_ctx.cur]), tail_ctx.qf[tail_ctx.cur], tail_ctx.n_cur)
                    tail_ctx = None
                    continue
                ticks_left -= tail_ctx.advance()
                n_active = N - poll_done()
                if n_active == 0:
                    break
                if spec_ok and n_active <= spec_rows:
                    tail_ctx.n_cur = n_active  # (an upper bound: the count is read one batch late)
                    continue
                if tail_ctx.wants_compaction(n_active):
                    tail_ctx.compact(n_active)
                continue
            n_rows = sum(g_.n_rows for g_ in groups)
            tail = n_rows <= graph_max_rows
            n_ticks = sync_every * (4 if tail else 1)
            if cohort_plan is not None:  # one transition: look at the batch where a cohort of trees has just ended
                done_ticks = max_ticks - ticks_left
                n_ticks = next((b_ - done_ticks for b_ in cohort_plan if b_ > done_ticks), n_ticks)
            for g_ in groups:
                # a multi-group schedule replays fixed-size groups for many chunks: record at once; a
                # single small group (a run that STARTS with few chains) is recorded once its batch size
                # has lasted 4 plain chunks
                rec_now = can_record and ticks_left < max_ticks and (
                    use_graph is True or len(groups) > 1 or (tail and g_.eager_chunks >= 4))
                g_.advance(n_ticks, rec_now)
            ticks_left -= n_ticks
            if lag_busy and n_rows > graph_max_rows and cohort_plan is None:
                # busy phase of a long run: the count of the PREVIOUS chunk (read behind this chunk's launches, so the
                # GPU never drains while the host waits); it only grows, so a stale value delays a compaction by one
                # chunk at most, and the exact count is read before anything is sized by it
                n_active = N - poll_done()
                if n_active <= int(n_rows * compact_frac):
                    n_active = N - int(n_done.item())
            else:
                n_active = N - int(n_done.item())  # one host sync per chunk
                poll.update(primed=False, last=N - n_active)
            if n_active == 0:
                break
            if spec_ok and n_active <= spec_rows and len(groups) == 1:
                to_spec(groups[0].rref, groups[0].qf, n_active)  # few live chains: every tick is pure latency from here on
                groups = []
                continue
            if n_active <= int(n_rows * compact_frac) and n_rows > 64:
                # drop the finished chains from the batch (device-side compaction + gather); the groups
                # are merged into one batch of the live rows
                if can_record and n_active <= graph_max_rows:
                    if persistent:
                        if static["tail"] is None:
                            static["tail"] = _Tail(min(N, graph_max_rows), sync_every, 4)
                        tail_ctx = static["tail"]
                    else:
                        tail_ctx = _Tail(n_active, sync_every, 4)
                    tail_ctx.enter(groups, n_active)
                    poll.update(primed=False, last=N - n_active)
                    groups = []
                    continue
                rows_all = torch.empty(n_rows, **i32)
                qf_all = torch.empty((n_rows, D), **f32)
                off = 0
                for g_ in groups:
                    _lib.call("bjx_nuts_async_compact", stream, dref, g_.rref, g_.qf.data_ptr(),
                              rows_all[off:].data_ptr(), qf_all[off:].data_ptr(), src_work.data_ptr(),
                              n_out.data_ptr())
                    off += int(n_out.item()) if len(groups) > 1 else n_active
                assert off == n_active, (off, n_active)
                rb = auto_row_block(n_active, D) if row_block is None else int(row_block)
                blk_now = n_active if not rb or rb >= n_active else rb
                groups = []
                for s0 in range(0, n_active, blk_now):
                    n_g = min(blk_now, n_active - s0)
                    g_ = _Group(rows_all[s0:s0 + n_g], n_g, qf_all[s0:s0 + n_g])
                    g_.logp_f, g_.gf = eval_logdensity(vg, g_.qf)  # the gathered positions, row for row
                    groups.append(g_)
        else:
            if spec_ctx is not None:
                torch.cuda.current_stream(dev).wait_stream(spec_ctx.side)
            if int(n_done.item()) != N:
                raise RuntimeError("free-running NUTS did not finish within its tick bound")
        if spec_ctx is not None:
            spec_ctx.finish()
        groups = []
        if persistent:  # the buffers are reused by the next call
            c = lambda t: None if t is None else t[cur["t0"]:].clone()  # noqa: E731
            return (HMCState(q.clone(), logp.clone(), g.clone()), c(positions),
                    NUTSRunInfo(*[c(getattr(info, f)) for f in NUTSRunInfo._fields]))
        return HMCState(q, logp, g), positions, info

    def guarded():
        """execute(); should it raise while the second stream of a speculative tail still has work queued, wait for the
        device before the frame's buffers go back to the allocator."""
        try:
            return execute()
        except BaseException:
            if spec_ctx is not None:
                try:
                    torch.cuda.synchronize(dev)
                except Exception:
                    pass
            raise

    if not persistent:
        return guarded()

    kind0, imm_shape0 = metric.kind, tuple(imm_buf.shape)

    def rerun(rng_key2, state2, step_size2, inverse_mass_matrix2, num_steps2=None):
        """One more run on this workspace: new key, state, step size and metric (``step_major`` workspaces: and any
        number of transitions up to the capacity); no allocation, no recording."""
        k0_, k1_, fold_ = bjx_random.key_spec(rng_key2)
        m2 = metrics.default_metric(inverse_mass_matrix2, N, D, dev)
        if fold_ >= 0 or m2.kind != kind0 or tuple(m2.imm.shape) != imm_shape0:
            raise ValueError("persistent free-running workspace: the key kind or the metric's shape changed")
        T2 = cur["T"] if num_steps2 is None else int(num_steps2)
        if not 1 <= T2 <= Tc or (key_layout == "step" and T2 != 1):
            raise ValueError(f"persistent free-running workspace: {T2} transitions asked of a capacity of {Tc}")
        cur.update(T=T2, t0=Tc - T2)
        q.copy_(check_batch(state2.position, "state.position"))
        logp.copy_(check_batch(state2.logdensity, "state.logdensity"))
        g.copy_(check_batch(state2.logdensity_grad, "state.logdensity_grad"))
        e2, epc2 = step_size_args(step_size2, N, dev)
        eps_buf.fill_(e2) if epc2 is None else eps_buf.copy_(epc2)
        imm_buf.copy_(m2.imm)
        if key_layout == "step":
            # the key's two words as integer fills (no host-to-device copy, hence no host synchronisation)
            step_keys[0, 0].fill_(int(np.uint32(k0_).astype(np.int32)))
            step_keys[0, 1].fill_(int(np.uint32(k1_).astype(np.int32)))
        else:
            keys2 = np.ascontiguousarray(bjx_random.split(rng_key2, T2)).view(np.int32).reshape(T2, 2)
            step_keys[Tc - T2:].copy_(torch.as_tensor(keys2))
        for b_ in (phase, n_done, rec):
            b_.zero_()
        t_done.fill_(Tc - T2)
        return guarded()

    _handle["rerun"] = rerun
    _handle["capacity"] = Tc
    _handle["work"] = {"p": p, "bufs": bufs, "rec": rec}
    # EVERY device buffer the descriptors point to must outlive this call: the kernels of later reruns reach them through
    # raw pointers, and a tensor that only this frame referenced (checkpoints, slot tables, the front momentum ...) would
    # go back to the allocator on return -- and be handed to somebody else while the workspace still writes to it
    # (found as box-independent but allocation-pattern-dependent wrong draws on the second call: NOTEBOOK.md section 16.8)
    _handle["keep"] = [q, logp, g, p, qf, t_done, phase, n_done, rec, front_p, end_list, end_count, ck_r, ck_rs, fs, is_,
                       step_keys, src_work, n_out, eps_buf, imm_buf, info, positions, bufs, dense_f, v0, gemm_bufs, metric]
    return guarded()


def _run_lockstep(step_fn, rng_key, state, num_steps, key_layout, store_positions):
    """``run`` through ``num_steps`` lockstep steps (same keys, same draws as ``run_free``)."""
    from . import random as bjx_random

    T = int(num_steps)
    N, D = state.position.shape
    dev = state.position.device
    f32 = dict(dtype=torch.float32, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)
    info = NUTSRunInfo(torch.empty((T, N), **f32), torch.empty((T, N), **f32), torch.empty((T, N), **f32),
                       torch.empty((T, N), **i32), torch.empty((T, N), **i32),
                       torch.empty((T, N), dtype=torch.bool, device=dev),
                       torch.empty((T, N), dtype=torch.bool, device=dev), None)
    positions = torch.empty((T, N, D), **f32) if store_positions else None
    if key_layout not in ("step_major", "chain_major"):
        raise ValueError("key_layout must be 'step_major' or 'chain_major'")
    keys = bjx_random.split(rng_key, T) if key_layout == "step_major" else None
    for t in range(T):
        k = keys[t] if keys is not None else bjx_random.ChainMajorKey(rng_key, t)
        state, inf = step_fn(k, state)
        if positions is not None:
            positions[t] = state.position
        info.logdensity[t] = state.logdensity
        info.acceptance_rate[t] = inf.acceptance_rate
        info.energy[t] = inf.energy
        info.num_integration_steps[t] = inf.num_integration_steps
        info.num_trajectory_expansions[t] = inf.num_trajectory_expansions
        info.is_divergent[t] = inf.is_divergent
        info.is_turning[t] = inf.is_turning
    return state, positions, info


def as_top_level_api(logdensity_fn: Callable, step_size, inverse_mass_matrix, *,
                     max_num_doublings: int = 10, divergence_threshold: int = 1000,
                     integrator=integrators.velocity_verlet, chain_offset: int = 0,
                     recompact_every: int = 16, use_graph="auto",
                     graph_sync_every: int = 4, run_use_graph="auto", dense_gemm="auto",
                     fuse_target: bool = False, step_driver: str = "auto",
                     step_spec_rows: Optional[int] = None) -> SamplingAlgorithm:
    """blackjax/mcmc/nuts.py:150-220.  Besides ``init`` / ``step`` the returned algorithm has
    ``run(rng_key, state, num_steps, *, key_layout="step_major", store_positions=True)``: the same
    ``num_steps`` transitions with free-running chains (``run_free``), which is how many-chain NUTS
    should be driven on this engine.

    ``fuse_target=True`` (``run_free``: engine-resident targets only, outside the external-callable
    contract): ``run`` evaluates the log-density inside the tick kernels, and ``step`` is ONE free-running
    transition of every chain instead of the lockstep tree -- same keys, same draws, same state and scalar
    info fields bit for bit; the ``momentum`` and ``trajectory_*_state`` fields of ``NUTSInfo`` are ``None``
    (the free-running kernels keep a trajectory's ends only while its tree grows)."""
    integrators.check_supported(integrator, allow_general=True)
    general = integrator is not integrators.velocity_verlet
    if fuse_target and general:
        raise NotImplementedError("fuse_target=True: the free-running tick kernels integrate with velocity Verlet")
    if step_driver not in ("auto", "lockstep", "free"):
        raise ValueError("step_driver must be 'auto', 'lockstep' or 'free'")
    fuse_default = bool(fuse_target)
    kernel = build_kernel(integrator, divergence_threshold, recompact_every=recompact_every,
                          use_graph=use_graph, graph_sync_every=graph_sync_every, dense_gemm=dense_gemm)
    # ``step`` as ONE free-running transition on a persistent workspace (round 5).  The lockstep tree makes every chain
    # wait at every leaf of the deepest tree of the ensemble -- two dependent launches per leaf for a handful of live
    # chains during most of a transition; the free-running tick kernels with the two-stream speculative tail spend
    # 4.9 us instead of ~7-8 us on such a leaf, and give the same draws and the same NUTSInfo bit for bit
    # (tests/test_nuts_step_free_gpu.py).  What used to make this path lose -- buffers and tail recordings set up per
    # call -- now happens once per (shape, callable, stream): run_free(_handle=...).
    free_ws: dict = {}
    free_bad: set = set()
    step_driver = _os_environ().get("BJX_NUTS_STEP_DRIVER", step_driver)

    def _free_step_key(rng_key, state, for_run=False):
        if ((step_driver == "lockstep" and not for_run) or fuse_default or use_graph is False
                or run_use_graph is False):
            return None
        if not is_capturable(logdensity_fn):
            return None
        pos = state.position
        if not (isinstance(pos, torch.Tensor) and pos.is_cuda and pos.ndim == 2 and pos.dtype == torch.float32):
            return None
        n_, d_ = pos.shape
        if n_ == 0 or d_ % 4 != 0 or d_ > 1024 or int(max_num_doublings) < 1 or key_spec(rng_key)[2] >= 0:
            return None
        m = metrics.default_metric(inverse_mass_matrix, n_, d_, pos.device)
        if m.kind != "diag" or (general and not free_running_supports(integrator, m.kind, d_)):
            return None
        k = (n_, d_, tuple(m.imm.shape), pos.device.index, torch.cuda.current_stream(pos.device).cuda_stream)
        return None if k in free_bad else k

    def _free_step_failed(wkey, state, err):
        # a recording failed, or the speculative tail's replica check / tick bound fired (first call OR a later rerun):
        # say so, drop the workspace and keep this shape on the lockstep tree driver, which restarts the transition from
        # `state` (never modified by the free-running driver before it returns)
        import warnings

        warnings.warn(f"blackjax_amd.nuts.step: the free-running driver failed for this shape ({err}); "
                      "falling back to the lockstep tree driver", RuntimeWarning, stacklevel=4)
        free_bad.add(wkey)
        free_ws.pop(wkey, None)
        torch.cuda.synchronize(state.position.device)

    def _free_step(wkey, rng_key, state):
        h = free_ws.get(wkey)
        try:
            if h is None:
                h = {}
                new_state, _, ri = run_free(
                    rng_key, state, logdensity_fn, step_size, inverse_mass_matrix, 1, max_num_doublings,
                    divergence_threshold=divergence_threshold, chain_offset=chain_offset, key_layout="step",
                    store_positions=False, use_graph=True if use_graph is True else run_use_graph,
                    integrator=integrator, keep_ends=True, spec_rows=step_spec_rows, _handle=h)
                while len(free_ws) >= 2:  # a workspace is ~20 (N, D) buffers: keep the two most recent shapes only
                    free_ws.pop(next(iter(free_ws)))
                free_ws[wkey] = h
            else:
                new_state, _, ri = h["rerun"](rng_key, state, step_size, inverse_mass_matrix)
        except RuntimeError as err:
            if step_driver == "free":
                raise
            _free_step_failed(wkey, state, err)
            return None
        w = h["work"]
        b_, rec_f = w["bufs"], w["rec"].view(torch.float32)
        c = lambda t: t.clone()  # noqa: E731
        info = NUTSInfo(
            c(w["p"]), ri.is_divergent[0], ri.is_turning[0], ri.energy[0],
            IntegratorState(c(b_["Lq"]), c(b_["Lp"]), c(rec_f[:, 29]), c(b_["Lg"])),
            IntegratorState(c(b_["Rq"]), c(b_["Rp"]), c(rec_f[:, 30]), c(b_["Rg"])),
            ri.num_trajectory_expansions[0], ri.num_integration_steps[0], ri.acceptance_rate[0])
        return new_state, info

    # ``run`` on a persistent workspace too (round 6): a run's buffers (two checkpoint stacks, ~20 (N, D) arrays) and,
    # above all, the recorded tick sequences of its tail -- one per tier of the live-row count, plus the speculative
    # tail's -- are built once per (shape, stream) and serve every later ``run`` of at most ``capacity`` transitions.
    # Measured at C3 (NOTEBOOK section 19): recordings + allocation were ~4 ms of a 205 ms run (T = 100).
    run_ws: dict = {}
    run_bad: set = set()
    RUN_WS_MAX_POSITION_BYTES = 2 << 30  # a workspace keeps its (capacity, N, D) position array: beyond this, per-call buffers

    def _run_ws_enabled():
        return _os_environ().get("BJX_NUTS_RUN_WS", "1") != "0"

    def _persistent_run(wkey, rng_key, state, T, store_positions):
        if wkey in run_bad:
            return None
        n_, d_ = state.position.shape
        h = run_ws.get(wkey)
        if h is not None and h["capacity"] < T:
            run_ws.pop(wkey)
            h = None
        cap = T if store_positions else max(T, 512)  # (capacity, N) info arrays only: 22 B per chain and transition
        if h is None and store_positions and cap * n_ * d_ * 4 > RUN_WS_MAX_POSITION_BYTES:
            return None
        try:
            if h is None:
                # a workspace is ~20 (N, D) buffers + two checkpoint stacks: the two most recent are kept, one when they are large
                ws_bytes = 4 * n_ * d_ * (20 + 2 * int(max_num_doublings))
                while len(run_ws) >= (2 if ws_bytes < (4 << 30) else 1):
                    run_ws.pop(next(iter(run_ws)))
                h = {"capacity": cap}
                out = run_free(
                    rng_key, state, logdensity_fn, step_size, inverse_mass_matrix, T, max_num_doublings,
                    divergence_threshold=divergence_threshold, chain_offset=chain_offset, key_layout="step_major",
                    store_positions=store_positions, use_graph=True if use_graph is True else run_use_graph,
                    integrator=integrator, _handle=h)
                run_ws[wkey] = h
                return out
            return h["rerun"](rng_key, state, step_size, inverse_mass_matrix, T)
        except RuntimeError as err:
            # a recording failed or the speculative tail's replica check fired: the per-call driver restarts from `state`
            import warnings

            warnings.warn(f"blackjax_amd.nuts.run: the persistent workspace failed for this shape ({err}); "
                          "falling back to per-call buffers", RuntimeWarning, stacklevel=3)
            run_bad.add(wkey)
            run_ws.pop(wkey, None)
            torch.cuda.synchronize(state.position.device)
            return None

    def init_fn(position, rng_key=None):
        del rng_key
        return init(position, logdensity_fn)

    def step_fn(rng_key, state):
        if fuse_default:
            new_state, _, ri = run_free(rng_key, state, logdensity_fn, step_size, inverse_mass_matrix, 1,
                                        max_num_doublings, divergence_threshold=divergence_threshold,
                                        chain_offset=chain_offset, key_layout="step", store_positions=False,
                                        use_graph=True if use_graph is True else run_use_graph, fuse_target=True)
            return new_state, NUTSInfo(None, ri.is_divergent[0], ri.is_turning[0], ri.energy[0], None, None,
                                       ri.num_trajectory_expansions[0], ri.num_integration_steps[0],
                                       ri.acceptance_rate[0])
        wkey = _free_step_key(rng_key, state)
        if wkey is not None:
            out = _free_step(wkey, rng_key, state)
            if out is not None:
                return out
        return kernel(rng_key, state, logdensity_fn, step_size, inverse_mass_matrix,
                      max_num_doublings, chain_offset=chain_offset)

    def run_fn(rng_key, state, num_steps: int, *, key_layout: str = "step_major",
               store_positions: bool = True, fuse_target: bool = fuse_default):
        n_, d_ = state.position.shape
        kind = metrics.default_metric(inverse_mass_matrix, n_, d_, state.position.device).kind
        if kind == "dense" and not fuse_target and (dense_gemm is True or (dense_gemm == "auto" and d_ >= 128)):
            # ONE shared dense matrix on the GEMM path (the arithmetic `step` uses for this metric: every
            # product v = M^-1 p is an fp32 MFMA GEMM over the live rows): the free-running tick kernels apply
            # a dense metric per chain in fp64 -- other roundings, and 20 x slower at 16 384 x 512 -- so the
            # run is made of lockstep steps: same keys, run(T) == T x step bit for bit.  BJX_NUTS_FREE_GEMM=1
            # (velocity Verlet): free-running ticks with every product on the GEMM instead (bjx_nuts_async_t.gemm_*:
            # the same arithmetic, so the same results)
            if not general and _os_environ().get("BJX_NUTS_FREE_GEMM", "0") != "0":
                return run_free(rng_key, state, logdensity_fn, step_size, inverse_mass_matrix, num_steps,
                                max_num_doublings, divergence_threshold=divergence_threshold,
                                chain_offset=chain_offset, key_layout=key_layout, store_positions=store_positions,
                                use_graph=True if use_graph is True else run_use_graph, dense_gemm=True)
            return _run_lockstep(step_fn, rng_key, state, num_steps, key_layout, store_positions)
        if general:
            # multi-stage integrators run free on the low-traffic tick kernels (diagonal metric, D % 4 == 0,
            # D <= 1024: a leaf lasts K ticks); other shapes take the same num_steps transitions as lockstep
            # steps (identical draws: chain c at transition t uses the same key either way)
            if not free_running_supports(integrator, kind, d_):
                _warn_lockstep_run(integrator, kind, d_)
                return _run_lockstep(step_fn, rng_key, state, num_steps, key_layout, store_positions)
        wkey = None
        if key_layout == "step_major" and not fuse_target and int(num_steps) >= 1 and _run_ws_enabled():
            wkey = _free_step_key(rng_key, state, for_run=True)
        if wkey is not None:
            out = _persistent_run(wkey + (bool(store_positions),), rng_key, state, int(num_steps), store_positions)
            if out is not None:
                return out
        return run_free(rng_key, state, logdensity_fn, step_size, inverse_mass_matrix, num_steps,
                        max_num_doublings, divergence_threshold=divergence_threshold,
                        chain_offset=chain_offset, key_layout=key_layout,
                        store_positions=store_positions,
                        use_graph=True if use_graph is True else run_use_graph, fuse_target=fuse_target,
                        integrator=integrator)

    return SamplingAlgorithm(init_fn, step_fn, run_fn)
