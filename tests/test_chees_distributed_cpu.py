"""world_size-2 gloo test (CPU) of the ONE exchange step of the engine: the pooled ChEES statistics.

Each rank holds half of the chains and computes its local sums exactly as the kernels of
include/bjx_pool.h define them (NumPy stand-in, fp64); the product's ``all_reduce_sum_`` pools them
and the product's host update (``chees.base(...).update.scalar_update``) consumes them.  The result
must equal the oracle's single-process update over the whole ensemble, for any split of the chains.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import chees as och
from oracle import prng
from oracle.fp import f32, f64


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ensemble(t, N, D):
    keys = prng.split(prng.key(1000 + t), 5)
    props = (prng.normal(keys[0], (N, D)) * f32(2.0) + f32(0.5)).astype(f32)
    moms = prng.normal(keys[1], (N, D))
    inits = prng.normal(keys[2], (N, D))
    acc = prng.uniform(keys[3], (N,))
    div = prng.uniform(keys[4], (N,)) < 0.15
    if t == 2:
        props[1, 0] = np.inf
        inits[N - 1, D - 1] = np.nan
    return props, moms, inits, acc, div


def _local_colstats(props, w, inits):
    """bjx_chees_colstats on one shard."""
    finite = np.isfinite(props)
    xs = np.where(finite, props, 0.0).astype(f64)
    ok = ~np.isnan(inits)
    D = props.shape[1]
    return np.concatenate([(w.astype(f64)[:, None] * xs).sum(0), np.where(ok, inits, 0.0).astype(f64).sum(0),
                           ok.sum(0).astype(f64), np.full(D, w.astype(f64).sum())])


def _worker(rank, world, port, N, D, T, split, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from blackjax_amd import chees as pch
        from blackjax_amd import distributed as bd
        from blackjax_amd import optim

        group = dist.group.WORLD
        lo, hi = (0, split) if rank == 0 else (split, N)
        jitter = lambda i: och.halton_sequence(i, 11)
        init, update = pch.base(jitter, lambda i: i + 1, optim.adam(0.5, b1=0, b2=0.95), 0.651, 0.5, 1000)
        state = init(0, 0.1)
        imm = (10.0 ** np.linspace(-1, 1, D)).astype(f32)
        mean, m2, count = np.zeros(D, f32), np.zeros(D, f32), f32(0.0)
        for t in range(T):
            props, moms, inits, acc, div = (a[lo:hi] for a in _ensemble(t, N, D))
            # weights -> column statistics -> all-reduce -> means
            w = np.where(div | ~np.isfinite(props).all(-1), f32(0.0), acc).astype(f32)
            stats = torch.from_numpy(_local_colstats(props, w, inits))
            bd.all_reduce_sum_(stats, group)
            s = stats.numpy()
            with np.errstate(divide="ignore", invalid="ignore"):
                pm = (s[:D].astype(f32) / (s[3 * D:].astype(f32) + f32(1e-20))).astype(f32)
                im = (s[D:2 * D].astype(f32) / s[2 * D:3 * D].astype(f32)).astype(f32)
            crit = och.criterion_given_means(props, moms, inits, pm, im, imm, True)
            scale = f32(f32(jitter(state.random_generator_arg)) * state.trajectory_length)
            nd = ~div
            with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
                tg = (scale * crit).astype(f32)
                sums = torch.tensor([(f32(1.0) / acc)[nd].astype(f64).sum(), float(nd.sum()),
                                     (acc[nd].astype(f64) * tg[nd].astype(f64)).sum(),
                                     (acc[nd] + f32(1e-20)).astype(f32).astype(f64).sum()], dtype=torch.float64)
            bd.all_reduce_sum_(sums, group)
            state = update.scalar_update(state, sums.numpy())
            # pooled diagonal moment block: two all-reduces (sum + count, then centred squares)
            colsum = torch.from_numpy(np.concatenate([moms.astype(f64).sum(0), [float(hi - lo)]]))
            bd.all_reduce_sum_(colsum, group)
            n_b = f32(colsum[D].item())
            mean_b = (colsum[:D].numpy() / f64(n_b)).astype(f32)
            c2 = torch.from_numpy(((moms - mean_b).astype(f32).astype(f64) ** 2).sum(0))
            bd.all_reduce_sum_(c2, group)
            n_ab = f32(count + n_b)
            delta = (mean_b - mean).astype(f32)
            with np.errstate(invalid="ignore", over="ignore"):
                new_mean = (mean + (delta * f32(n_b / n_ab)).astype(f32)).astype(f32)
                cross = ((delta * delta).astype(f32) * f32(f32(count * n_b) / n_ab)).astype(f32)
                m2 = ((m2 + c2.numpy().astype(f32)).astype(f32) + cross).astype(f32)
            mean, count = new_mean, n_ab
        if rank == 0:
            q.put((tuple(state[:4]), tuple(state.da_state), tuple(state.optim_state), state.step, mean, m2,
                   float(count)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("split", [12, 5])
def test_two_rank_pooled_chees_update_equals_single_process(split):
    N, D, T = 24, 6, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, N, D, T, split, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single-process oracle over the whole ensemble
    jitter = lambda i: och.halton_sequence(i, 11)
    init, update = och.base(jitter, lambda i: i + 1, och.Adam(0.5, b1=0, b2=0.95), 0.651, 0.5, 1000)
    state = init(0, 0.1)
    imm = (10.0 ** np.linspace(-1, 1, D)).astype(f32)
    blk = och.MomentBlock(f32(0.0), np.zeros(D, f32), np.zeros(D, f32))
    for t in range(T):
        props, moms, inits, acc, div = _ensemble(t, N, D)
        state = update(state, props, moms, inits, acc, div, imm)
        blk = och.cgl_update_batch(blk, moms)
    head, da, opt, step, mean, m2, count = got
    np.testing.assert_allclose(head, tuple(state[:4]), rtol=1e-6)
    np.testing.assert_allclose(da[0], state.da_state.log_step_size, rtol=1e-6)
    np.testing.assert_allclose(np.asarray(opt, f64), np.asarray(tuple(state.optim_state), f64), rtol=1e-5)
    assert step == state.step == T + 1
    assert count == float(blk.count) == N * T
    with np.errstate(invalid="ignore"):
        np.testing.assert_allclose(mean, blk.mean, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(m2, blk.m2, rtol=1e-5)
