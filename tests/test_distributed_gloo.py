"""world_size-2 gloo test of the multi-GPU driver logic on CPU: frame-block sharding, replicated style
statistics, the single gather at the end, and the barrier + max-over-ranks timing of bench.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import applied_image_processing_amd.sharding as sh


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_stylize(frames):
    # stands in for the per-frame GPU forward: deterministic, frame-local
    return (frames.float() * 3 + 1).to(torch.uint8)


def _worker(rank, world, port, n_frames, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = sh.shard_range(n_frames, world, rank)
        frames = torch.arange(n_frames * 6, dtype=torch.float32).view(n_frames, 1, 2, 3) % 50
        local = _fake_stylize(frames[lo:hi])
        out = sh.gather_frames(local, n_frames, dst=0)
        # bench.py timing contract: barrier on both sides, MAX over ranks
        t = torch.tensor([0.5 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            q.put((out.clone(), float(t)))
        else:
            assert out is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [7, 8, 1])
def test_shard_and_gather_world2(n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    out, tmax = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    frames = torch.arange(n_frames * 6, dtype=torch.float32).view(n_frames, 1, 2, 3) % 50
    assert torch.equal(out, _fake_stylize(frames))
    assert tmax == 1.5
