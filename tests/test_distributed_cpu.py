"""world_size-2 gloo tests (CPU) of the multi-GPU plumbing: chain sharding, per-chain key
derivation by global index, gather into global order, pooled moments."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from blackjax_amd import distributed as bd
from blackjax_amd import random as brandom


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shard = bd.shard_chains(total)
        # per-chain keys computed locally from the GLOBAL chain index (what the kernels do)
        keys = torch.as_tensor(brandom.split(brandom.key(7), shard.count, offset=shard.offset).astype(np.int64))
        gathered = bd.all_gather_chains(keys, shard)
        # per-chain "draws": deterministic function of the global index
        idx = torch.arange(shard.offset, shard.offset + shard.count, dtype=torch.float64)
        draws = torch.stack([idx, idx**2, torch.sin(idx)], 1)
        pooled = bd.all_reduce_moments(bd.moment_block(draws))
        all_draws = bd.all_gather_chains(draws, shard)
        if rank == 0:
            q.put((shard, gathered.numpy(), pooled.n.item(), pooled.mean.numpy(), pooled.m2.numpy(),
                   all_draws.numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [10, 7])
def test_two_rank_sharding_and_gather(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    shard0, keys, n, mean, m2, all_draws = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # global order == single-process split
    ref = brandom.split(brandom.key(7), total).astype(np.int64)
    assert np.array_equal(keys, ref)
    assert shard0.offset == 0 and shard0.count == (total + 1) // 2
    idx = np.arange(total, dtype=np.float64)
    full = np.stack([idx, idx**2, np.sin(idx)], 1)
    assert np.array_equal(all_draws, full)
    assert n == total
    np.testing.assert_allclose(mean, full.mean(0), rtol=1e-12)
    np.testing.assert_allclose(m2, ((full - full.mean(0)) ** 2).sum(0), rtol=1e-10)


def test_shard_partition_covers_range():
    for total in (0, 1, 5, 8, 65536 * 8 + 3):
        for world in (1, 2, 3, 8):
            shards = [bd.shard_chains(total, r, world) for r in range(world)]
            assert shards[0].offset == 0
            for a, b in zip(shards, shards[1:]):
                assert a.offset + a.count == b.offset
            assert shards[-1].offset + shards[-1].count == total
            assert max(s.count for s in shards) - min(s.count for s in shards) <= 1


def test_merge_moment_blocks_matches_direct():
    x = torch.randn(37, 5, dtype=torch.float64)
    a, b = bd.moment_block(x[:10]), bd.moment_block(x[10:])
    m = bd.merge_moment_blocks(a, b)
    d = bd.moment_block(x)
    torch.testing.assert_close(m.mean, d.mean)
    torch.testing.assert_close(m.m2, d.m2)
