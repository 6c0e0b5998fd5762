"""Frame / view sharding across the GPUs of one node (one process per GPU, ``torch.distributed``).

The AdaIN path has no cross-image state: every frame of a video
(reference video/utils.py:327-350) and every camera view of the 3DGS guide-image precompute
(reference Style_3DGS/train.py:86-115) is an independent forward pass.  So the batch is cut into
contiguous blocks by frame index, weights and the style statistics are replicated, and the only
collective is one gather of the finished (uint8) frames — RCCL over xGMI when the tensors are on
GPUs (backend "nccl"), gloo on CPU tensors in the tests.  No all-reduce anywhere.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, world_size, rank):
    """Contiguous block [lo, hi) of ``n_items`` for ``rank``; the first ``n_items % world_size`` ranks get
    one extra item (300 views over 8 GPUs -> 38,38,38,38,37,37,37,37)."""
    if world_size < 1 or not (0 <= rank < world_size) or n_items < 0:
        raise ValueError(f"bad shard request n={n_items} world={world_size} rank={rank}")
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_counts(n_items, world_size):
    return [shard_range(n_items, world_size, r)[1] - shard_range(n_items, world_size, r)[0] for r in range(world_size)]


def gather_frames(local, n_items, dst=0, group=None):
    """Gathers the per-rank blocks ``local`` [count_r, ...] (same trailing shape and dtype on every rank) to
    ``dst`` in frame order; returns the [n_items, ...] tensor on ``dst`` and None elsewhere.  Ragged counts
    are padded to the largest block so that a single all_gather (one ring pass over xGMI) suffices."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = shard_counts(n_items, world)
    if local.shape[0] != counts[rank]:
        raise ValueError(f"rank {rank}: expected {counts[rank]} frames, got {local.shape[0]}")
    mx = max(counts)
    if mx == 0:
        return local if rank == dst else None
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))])
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad.contiguous(), group=group)
    if rank != dst:
        return None
    return torch.cat([o[:c] for o, c in zip(out, counts)])
