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


def device_transport(tensor, group=None):
    """Name of the transport a collective on ``tensor`` will use in ``group``: "rccl" for the ``nccl`` backend (RCCL on ROCm:
    xGMI between the GPUs of a node), otherwise the backend's own name ("gloo").  Decided from the process group's
    backend table, before any collective runs, so every rank reaches the same answer without a try/except around a
    collective (a per-rank fallback inside one would deadlock the ranks that did not fail)."""
    pg = group if group is not None else dist.distributed_c10d._get_default_group()
    try:
        name = pg._get_backend(tensor.device).name()
    except Exception:
        cfg = str(dist.get_backend(pg))
        table = dict(kv.split(":") for kv in cfg.split(",") if ":" in kv)
        name = table.get(tensor.device.type, cfg)
    name = str(name).lower()
    return "rccl" if name == "nccl" else name


def gather_frames(local, n_items, dst=0, group=None, async_op=False):
    """Gathers the per-rank blocks ``local`` [count_r, ...] (same trailing shape and dtype on every rank) to
    ``dst`` in frame order; returns the [n_items, ...] tensor on ``dst`` and None elsewhere.  Ragged counts
    are padded to the largest block so that ONE gather suffices — the only collective of the whole path
    (SURVEY.md 8(e)): every peer sends its block straight to ``dst`` (over its own xGMI link when the tensors are
    on GPUs); nothing is received anywhere else.  With ``async_op`` the call returns a ``finish()`` closure instead:
    the gather then overlaps whatever the caller enqueues next, and ``finish()`` waits and returns the result."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = shard_counts(n_items, world)
    if local.shape[0] != counts[rank]:
        raise ValueError(f"rank {rank}: expected {counts[rank]} frames, got {local.shape[0]}")
    mx = max(counts)
    if mx == 0:
        res = local if rank == dst else None
        return (lambda: res) if async_op else res
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))])
    pad = pad.contiguous()
    out = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    work = dist.gather(pad, out, dst=dst, group=group, async_op=async_op)

    def finish():
        if work is not None:
            work.wait()
        if rank != dst:
            return None
        return torch.cat([o[:c] for o, c in zip(out, counts)])

    return finish if async_op else finish()
