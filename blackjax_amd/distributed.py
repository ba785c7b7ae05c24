"""Multi-GPU plumbing: chains shard embarrassingly, one process per GPU.

The reference's only multi-device pattern is one-chain-per-device ``shard_map``
(docs/examples/howto_sample_multiple_chains.md:201-234, tests/test_multidevice/test_multichain.py:
36-99).  Here each rank owns a contiguous block of the global chain range and passes its first
global chain index as ``chain_offset``; per-chain keys depend only on that global index, so the
sampling loop and per-chain warmup need NO collective and results do not depend on the number of
ranks.  RCCL (``torch.distributed`` backend ``nccl``; ``gloo`` in the CPU tests) is used only to
gather retained draws and to merge adaptation / summary statistics at the end.
"""
from __future__ import annotations

from typing import NamedTuple

import torch
import torch.distributed as dist

__all__ = ["ChainShard", "shard_chains", "all_gather_chains", "all_reduce_sum_", "MomentBlock", "moment_block",
           "merge_moment_blocks", "all_reduce_moments"]


# World-size-1 shortcuts skip the collective altogether.  Setting this to True keeps the collective
# calls even then -- used by tests/test_rccl_gpu.py to run every exchange of this module through RCCL
# on a one-GPU box (a 1-rank communicator still initialises RCCL and launches its kernels).
FORCE_COLLECTIVES = False


def _single(group) -> bool:
    return not dist.is_initialized() or (dist.get_world_size(group) == 1 and not FORCE_COLLECTIVES)


class ChainShard(NamedTuple):
    offset: int  # first global chain index owned by this rank  (-> chain_offset=)
    count: int  # number of chains owned by this rank
    total: int


def shard_chains(n_chains_total: int, rank: int | None = None, world_size: int | None = None) -> ChainShard:
    """Contiguous block partition of ``n_chains_total`` chains (earlier ranks get the remainder)."""
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    base, rem = divmod(int(n_chains_total), world_size)
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return ChainShard(offset, count, int(n_chains_total))


def all_gather_chains(local: torch.Tensor, shard: ChainShard, group=None) -> torch.Tensor:
    """All-gather per-chain data (chain axis 0) from every rank into global chain order.
    Handles ragged shards by padding to the largest block (one collective)."""
    if _single(group):
        return local
    world = dist.get_world_size(group)
    counts = [shard_chains(shard.total, r, world).count for r in range(world)]
    mx = max(counts)
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + local.shape[1:])], 0)
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad.contiguous(), group=group)
    return torch.cat([o[:c] for o, c in zip(out, counts)], 0)


def all_reduce_sum_(buf: torch.Tensor, group=None) -> torch.Tensor:
    """In-place all-reduce(sum) of a statistics buffer over the ranks of ``group``; a no-op for a
    single process (``group is None`` means "this process only", NOT the default group, so that
    rank-local runs never issue a collective by accident).  This is the exchange step of the pooled
    (cross-chain) warmup, include/bjx_pool.h."""
    if group is None or _single(group):
        return buf
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf


class MomentBlock(NamedTuple):
    """Mergeable first/second moments over chains (model: the CGL merge of
    blackjax/adaptation/metric_buffers.py:334-394)."""

    n: torch.Tensor  # () float64
    mean: torch.Tensor  # (D,) float64
    m2: torch.Tensor  # (D,) float64  sum of squared deviations


def moment_block(x: torch.Tensor) -> MomentBlock:
    """Moments over axis 0 of ``x`` (chains or pooled draws), accumulated in fp64."""
    xd = x.double()
    n = torch.tensor(float(x.shape[0]), dtype=torch.float64, device=x.device)
    mean = xd.mean(0) if x.shape[0] else torch.zeros(x.shape[1:], dtype=torch.float64, device=x.device)
    m2 = ((xd - mean) ** 2).sum(0)
    return MomentBlock(n, mean, m2)


def merge_moment_blocks(a: MomentBlock, b: MomentBlock) -> MomentBlock:
    """Chan-Golub-LeVeque pairwise merge."""
    n = a.n + b.n
    safe = torch.clamp(n, min=1.0)
    delta = b.mean - a.mean
    mean = a.mean + delta * (b.n / safe)
    m2 = a.m2 + b.m2 + delta * delta * (a.n * b.n / safe)
    return MomentBlock(n, mean, m2)


def all_reduce_moments(local: MomentBlock, group=None) -> MomentBlock:
    """Pool moment blocks over ranks with ONE all-reduce(sum) of the sufficient statistics
    ``(n, n*mean, m2 + n*mean^2)``."""
    if _single(group):
        return local
    d = local.mean.numel()
    buf = torch.cat([local.n.reshape(1), (local.n * local.mean).reshape(-1),
                     (local.m2 + local.n * local.mean**2).reshape(-1)])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    n = buf[0]
    mean = buf[1:1 + d] / torch.clamp(n, min=1.0)
    m2 = buf[1 + d:] - n * mean**2
    return MomentBlock(n, mean.reshape(local.mean.shape), m2.reshape(local.m2.shape))
