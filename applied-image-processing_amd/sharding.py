"""Frame / view sharding across the GPUs of one node (one process per GPU, ``torch.distributed``).

The AdaIN path has no cross-image state: every frame of a video
(reference video/utils.py:327-350) and every camera view of the 3DGS guide-image precompute
(reference Style_3DGS/train.py:86-115) is an independent forward pass.  So the batch is cut into
contiguous blocks by frame index, weights and the style statistics are replicated, and the only
collective on the data path is one gather of the finished (uint8) frames — RCCL over xGMI when the
tensors are on GPUs (backend "nccl"), gloo on CPU tensors in the tests.  No all-reduce of data anywhere.

Collective calls are counted per kind in ``CALLS`` (tests assert "one device gather per job" on it).
"""
import torch
import torch.distributed as dist

# collective calls issued through this module since import / ``reset_calls()``: {"gather": device-data gathers,
# "agree": one-word status all_reduces, "barrier": host barriers}
CALLS = {"gather": 0, "agree": 0, "barrier": 0}


def reset_calls():
    for k in CALLS:
        CALLS[k] = 0


def dist_on():
    return dist.is_available() and dist.is_initialized()


def rank_world(group=None):
    if dist_on():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


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


def chunk_bounds(count, chunks):
    """[0, count) cut into ``chunks`` contiguous pieces (the first ``count % chunks`` one longer; pieces may be empty)."""
    return [shard_range(count, chunks, c) for c in range(chunks)]


def backend_table(group=None):
    """{device type: backend name} of the process group, from the public backend-config string ("cpu:gloo,cuda:nccl")."""
    cfg = str(dist.get_backend_config(group)) if hasattr(dist, "get_backend_config") else str(dist.get_backend(group))
    if ":" not in cfg:                                   # a bare backend name serves every device type it supports
        return {"cpu": cfg, "cuda": cfg} if cfg.lower() != "nccl" else {"cuda": cfg}
    return dict(kv.split(":") for kv in cfg.split(",") if ":" in kv)


def device_transport(tensor, group=None):
    """Name of the transport a collective on ``tensor`` will use in ``group``: "rccl" for the ``nccl`` backend (RCCL on ROCm:
    xGMI between the GPUs of a node), otherwise the backend's own name ("gloo").  Decided from the process group's
    backend table, before any collective runs, so every rank reaches the same answer without a try/except around a
    collective (a per-rank fallback inside one would deadlock the ranks that did not fail)."""
    name = backend_table(group).get(tensor.device.type)
    if name is None:
        raise RuntimeError(f"the process group has no backend for {tensor.device.type!r} tensors ({backend_table(group)})")
    name = str(name).lower()
    return "rccl" if name == "nccl" else name


def rank_census(mine, device=None, group=None):
    """What a multi-rank result line needs to PROVE that the device transport saw ``world`` ranks on ``world`` devices: every rank's
    descriptor ``mine`` (a dict with at least "host" and "uuid" or "device") collected with all_gather_object, and the result of ONE
    all_reduce(SUM) of a 1 per rank over the transport that will carry the frames (a device-resident 1 over RCCL when ``device`` is a
    GPU of an RCCL group, a host word otherwise) - equal to the world size iff every rank took part.  Identical on every rank."""
    world = dist.get_world_size(group)
    everyone = [None] * world
    dist.all_gather_object(everyone, mine, group=group)
    transport = None
    one = torch.ones(1, dtype=torch.int32)
    if device is not None and torch.device(device).type == "cuda":
        transport = device_transport(torch.empty(0, dtype=torch.uint8, device=device), group)
    if transport == "rccl":
        one = one.to(device)                  # device tensor: rides the cuda backend (RCCL)
    dist.all_reduce(one, group=group)
    return {"world": world, "devices": everyone, "distinct_devices": len({(d.get("host"), d.get("uuid") or d.get("device")) for d in everyone}),
            "transport": transport, "allreduce_of_ones": int(one.item())}


def census_problems(census, shared_devices_allowed=False):
    """Why a multi-rank line must NOT be reported as a multi-GPU measurement (empty list = it may): the all_reduce of ones did not
    see every rank, or two ranks sit on one device (allowed only in a labelled single-GPU rehearsal).  The decision uses what every
    rank holds after ``rank_census``, so all ranks reach it together (bench.py exits non-zero on every rank)."""
    out = []
    world = census["world"]
    if census["allreduce_of_ones"] != world:
        out.append(f"all_reduce of ones over the frame transport gave {census['allreduce_of_ones']}, not the world size {world}")
    if census["distinct_devices"] != world and not shared_devices_allowed:
        out.append(f"{world} ranks on {census['distinct_devices']} distinct device(s): one process per GPU is the contract")
    return out


def _status_tensor(value, device, group):
    """One word for a host-side rendezvous: a CPU tensor when the group has a CPU backend (it then rides gloo and never
    touches the GPUs), else a device tensor (a pure ``nccl`` group)."""
    if "cpu" in backend_table(group):
        return torch.tensor([value], dtype=torch.int32)
    return torch.tensor([value], dtype=torch.int32, device=device)


def host_barrier(group=None, device=None):
    """Rendezvous of the ranks: an all_reduce of one word."""
    if dist_on():
        CALLS["barrier"] += 1
        dist.all_reduce(_status_tensor(0, device, group), group=group)


def agree(ok, group=None, device=None):
    """True iff every rank passes ``ok`` = True: ONE one-word all_reduce (MIN).  The job drivers call it once per job, after a
    rank's own block is finished (or has failed) and before the gather, so that an error on one rank raises on EVERY rank
    instead of leaving the others waiting inside the collective."""
    if not dist_on():
        return bool(ok)
    CALLS["agree"] += 1
    t = _status_tensor(1 if ok else 0, device, group)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()) == 1)


def agree_min(value, group=None, device=None):
    """The smallest ``value`` (a small int) over the ranks: one one-word all_reduce (MIN).  E.g. 2 = go, 1 = cancelled, 0 = error."""
    if not dist_on():
        return int(value)
    CALLS["agree"] += 1
    t = _status_tensor(int(value), device, group)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return int(t.item())


def agree_geometry(ok, shapes, group=None, device=None):
    """The job's one status word: every rank passes whether its own block finished (``ok``) and the set of shapes (H, W, C) of
    its finished frames (empty if its block is); ONE all_reduce (MAX over [fail, H, W, C, -H, -W, -C], a rank contributing the
    element-wise max and min of its shapes) tells every rank ``(all_ok, shape | None, uniform)``: whether every rank finished,
    the frame shape (so that a rank without frames can take part in the gather) and whether it is the same for every frame of
    every rank.  Errors and size mismatches thus raise on EVERY rank before the gather instead of leaving the other ranks waiting
    inside it."""
    shapes = [tuple(int(v) for v in s_) for s_ in shapes]
    big = 1 << 40
    if shapes:
        mx = [max(s_[d] for s_ in shapes) for d in range(3)]
        mn = [min(s_[d] for s_ in shapes) for d in range(3)]
    else:
        mx, mn = [0, 0, 0], [big, big, big]
    if not dist_on():
        return bool(ok), (tuple(mx) if shapes else None), mx == mn or not shapes
    CALLS["agree"] += 1
    t = torch.tensor([0 if ok else 1] + mx + [-d for d in mn], dtype=torch.int64)
    if "cpu" not in backend_table(group):
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    v = [int(x) for x in t.tolist()]
    mx, mn = v[1:4], [-x for x in v[4:7]]
    if mx == [0, 0, 0]:
        return v[0] == 0, None, True
    return v[0] == 0, tuple(mx), mx == mn


def gather_frames(local, n_items, dst=0, group=None, async_op=False, counts=None):
    """Gathers the per-rank blocks ``local`` [count_r, ...] (same trailing shape and dtype on every rank) to
    ``dst`` in rank order; returns the [sum(counts), ...] tensor on ``dst`` and None elsewhere.  ``counts`` are the
    per-rank block lengths (default: ``shard_counts(n_items, world)``, i.e. the result is in frame order).  Ragged counts
    are padded to the largest block so that ONE gather suffices — the only collective of the whole path
    (SURVEY.md 8(e)): every peer sends its block straight to ``dst`` (over its own xGMI link when the tensors are
    on GPUs); nothing is received anywhere else.  With ``async_op`` the call returns a ``finish()`` closure instead:
    the gather then overlaps whatever the caller enqueues next, and ``finish()`` waits and returns the result."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if counts is None:
        counts = shard_counts(n_items, world)
    if len(counts) != world:
        raise ValueError(f"gather_frames: {len(counts)} block lengths for {world} ranks")
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
    # gloo gathers CPU tensors only: device blocks then travel as host copies (single-GPU rehearsals of the multi-rank path and
    # the CPU tests; a production group carries device tensors over RCCL - callers that must not degrade pass
    # require_transport="rccl" to the job drivers, bench.py refuses to report otherwise)
    via_host = pad.is_cuda and device_transport(pad, group) != "rccl"
    dev = pad.device
    if via_host:
        pad = pad.cpu()
    # the peers' blocks land in consecutive slices of ONE buffer: with equal block lengths (the usual case) that buffer IS the
    # result in frame order and nothing is copied on dst
    buf = pad.new_empty((world * mx,) + tuple(pad.shape[1:])) if rank == dst else None
    out = [buf[r * mx:(r + 1) * mx] for r in range(world)] if rank == dst else None
    CALLS["gather"] += 1
    work = dist.gather(pad, out, dst=dst, group=group, async_op=async_op)

    def finish():
        if work is not None:
            work.wait()
        if rank != dst:
            return None
        res = buf if all(c == mx for c in counts) else torch.cat([o[:c] for o, c in zip(out, counts)])
        return res.to(dev) if via_host else res

    return finish if async_op else finish()
