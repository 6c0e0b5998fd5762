"""Sharded job drivers: the reference's two batch callers of the AdaIN path, cut across the GPUs of one node.

* ``video_style_transfer_sharded`` — reference video/utils.py:297-369 (``apply_style_transfer_multi_ada`` / ``_ada``): one
  ``adain_inference`` per frame (each reloads the weights and re-encodes the style), INTER_AREA resize to the target
  resolution, then the flow-warp / blend recurrence over the frames and one file per frame.
* ``precompute_guides_sharded`` — reference Style_3DGS/train.py:86-115: one ``adain_inference`` per training camera with
  the mask ``gt_image_np > 0`` and ``<model_path>/stylized/<image_name>.jpg`` as output.

Both sit on ``stylize_frames_sharded``: the frame index is cut into contiguous blocks (``sharding.shard_range``), every
rank (one process per GPU) runs the batched engine on its block against replicated weights and style statistics, and the
finished uint8 frames meet on ``dst`` in ONE gather (RCCL over xGMI when the process group's CUDA backend is ``nccl``).
There is no other collective on the data path.  What is sequential in the reference stays on ``dst`` after the gather: the
frame-to-frame warp / blend recurrence and (by default) the file writes.

The engine is passed in (``engine.AdaINEngine`` on a GPU); the drivers only use its methods, so the multi-process logic is
exercised on CPU under ``gloo`` with a stand-in engine (tests/test_distributed_gloo.py).
"""
import time
from pathlib import Path

import torch
import torch.distributed as dist

from . import sharding as sh


def style_schedule(n_frames, n_styles):
    """Style index per frame as the reference switches styles through a video (video/utils.py:311-337):
    ``frames_per_style = max(1, n_frames // n_styles)``; the index advances at every frame i > 0 with
    ``i % frames_per_style == 0`` and stops at the last style."""
    if n_styles < 1:
        raise ValueError("No style images found in the style directory.")      # video/utils.py:308
    per = max(1, n_frames // n_styles)
    out, idx = [], 0
    for i in range(n_frames):
        if i > 0 and i % per == 0:
            idx = min(idx + 1, n_styles - 1)
        out.append(idx)
    return out


def _dist_on(group):
    return dist.is_available() and dist.is_initialized()


def _rank_world(group):
    if _dist_on(group):
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def host_barrier(group=None):
    """Host-side rendezvous of the ranks (an all_reduce of one CPU word: it rides the CPU backend of a
    "cpu:gloo,cuda:nccl" group and never touches the GPUs)."""
    if _dist_on(group):
        dist.all_reduce(torch.zeros(1), group=group)


def stylize_frames_sharded(engine, frames, styles, *, style_of=None, alpha=0.5, depth_maps=None, depth_offset=0.15,
                           depth_prominence=20, masks=None, post=None, sub_batch=4, group=None, dst=0, gather=True,
                           require_transport=None, style_cache=None):
    """Stylises ``frames`` (a sequence of [3,h,w] float tensors in [0,1], all one size, indexed lazily: a rank only ever
    touches its own block) and returns ``(frames_u8, info)``: the uint8 frames [n,H,W,3] in frame order on rank ``dst``
    (None on the other ranks, the local block when ``gather=False``) and a dict with the shard, timings and the transport.

    styles          one style tensor [1|.,3,hs,ws] or a list of them; ``style_of[i]`` picks the style of frame i
                    (``style_schedule``); every style image is encoded once per rank and call, its 2 x 512 statistics
                    are kept for the call (pass a dict as ``style_cache`` to keep them across calls with the SAME styles).
    depth_maps      optional sequence of [h0,w0] proximity maps, one per frame -> depth-aware blend (test.py:52-71) with
                    ``depth_offset`` / ``depth_prominence``; otherwise the ``alpha`` blend (test.py:74-81).
    masks           optional sequence of [1|3,hm,wm] masks -> content-mask composite (test.py:222-236).
    post            optional ``f(u8_block) -> u8_block`` applied per sub-batch on the owning rank BEFORE the gather
                    (frame-local work such as the INTER_AREA resize, so the gather moves the small frames).
    require_transport   e.g. "rccl": raise before any work if the gather would use another transport.
    """
    rank, world = _rank_world(group)
    n = len(frames)
    lo, hi = sh.shard_range(n, world, rank)
    style_list = list(styles) if isinstance(styles, (list, tuple)) else [styles]
    if style_of is None:
        style_of = [0] * n
    if len(style_of) != n:
        raise ValueError("style_of needs one style index per frame")
    dev = engine.device
    transport = None
    if gather and world > 1:
        transport = sh.device_transport(torch.empty(0, dtype=torch.uint8, device=dev), group)
        if require_transport and transport != require_transport:
            raise RuntimeError(f"final gather would run over {transport!r}, not {require_transport!r}")

    t0 = time.perf_counter()
    blocks = []
    cur_style = None
    stats = style_cache if style_cache is not None else {}
    i = lo
    while i < hi:
        j = i + 1
        while j < hi and j - i < sub_batch and style_of[j] == style_of[i]:
            j += 1
        if style_of[i] != cur_style:
            cur_style = style_of[i]
            if cur_style not in stats:
                stats[cur_style] = engine.set_style(style_list[cur_style]).style_stats()
            engine.use_style_stats(stats[cur_style])
        content = torch.stack([frames[k][:3] for k in range(i, j)]).to(dev, torch.float32)
        if depth_maps is not None:
            out = engine.stylize_depth(content, [depth_maps[k].to(dev, torch.float32) for k in range(i, j)], depth_offset,
                                       depth_prominence)
        else:
            out = engine.stylize(content, alpha)
        if masks is not None:
            m = torch.stack([torch.as_tensor(masks[k]).float() for k in range(i, j)]).to(dev)
            out = engine.composite(content, out, m)
        u8 = engine.to_u8(out)
        blocks.append(post(u8) if post is not None else u8)
        i = j
    if blocks:
        local = torch.cat(blocks) if len(blocks) > 1 else blocks[0]
    else:                      # a rank without frames still takes part in the gather; it learns the frame shape from dst's peers
        local = None
    info = {"rank": rank, "world": world, "shard": (lo, hi), "transport": transport}
    engine.synchronize()
    info["compute_s"] = time.perf_counter() - t0
    if not gather or world == 1:
        if local is None:
            local = torch.empty((0,), dtype=torch.uint8, device=dev)
        info["gather_s"] = 0.0
        return local, info
    # frame geometry is identical on every rank that has frames; ranks with none (n < world) get it from rank 0's block
    shape = torch.tensor(list(local.shape[1:]) if local is not None else [0, 0, 0], dtype=torch.int64)
    shapes = [torch.zeros_like(shape) for _ in range(world)]
    dist.all_gather(shapes, shape, group=group)                  # host metadata (CPU backend), 24 bytes per rank
    geom = next((tuple(int(v) for v in s_) for s_ in shapes if int(s_.sum()) > 0), None)
    if geom is None:
        return (torch.empty((0,), dtype=torch.uint8, device=dev) if rank == dst else None), info
    for s_ in shapes:
        if int(s_.sum()) > 0 and tuple(int(v) for v in s_) != geom:
            raise ValueError("stylize_frames_sharded: the gather needs one frame size on every rank")
    if local is None:
        local = torch.empty((0,) + geom, dtype=torch.uint8, device=dev)
    g0 = time.perf_counter()
    out = sh.gather_frames(local, n, dst=dst, group=group)
    engine.synchronize()
    info["gather_s"] = time.perf_counter() - g0
    return out, info


def video_style_transfer_sharded(engine, frames, styles, *, flows=None, target_resolution=None, blend_alpha=0.7, depth_maps=None,
                                 offset=0.30, prominence=20, alpha=0.5, sub_batch=4, group=None, dst=0, require_transport=None):
    """The video caller (reference video/utils.py:297-369) over a frame list: per-frame AdaIN sharded over the ranks (styles
    switching through the clip as ``style_schedule`` says when several are given; depth-aware when ``depth_maps`` are
    given, which is how the reference runs it: ``use_depth=True``, offset 0.30, prominence 20), ``cv2.resize(...,
    target_resolution, INTER_AREA)`` of every stylised frame on its own rank (:352-353; ``target_resolution`` =
    (width, height)), ONE gather, then on ``dst`` the recurrence ``frame_i = blend(frame_i, warp(result_{i-1}, flow_{i-1}),
    blend_alpha)`` (:355-368) over ``flows`` [n-1,2,H,W] (prev -> current, estimated by the caller: the optical-flow
    estimator is OpenCV's and stays outside).  Returns ``(frames_u8 on dst | None, info)``."""
    n = len(frames)
    style_list = list(styles) if isinstance(styles, (list, tuple)) else [styles]
    post = None
    if target_resolution is not None:
        post = lambda u8: engine.resize_area_u8(u8, target_resolution)
    out, info = stylize_frames_sharded(engine, frames, style_list, style_of=style_schedule(n, len(style_list)), alpha=alpha,
                                       depth_maps=depth_maps, depth_offset=offset, depth_prominence=prominence, post=post,
                                       sub_batch=sub_batch, group=group, dst=dst, require_transport=require_transport)
    if out is not None and flows is not None and n > 1:
        t0 = time.perf_counter()
        out = engine.temporal_blend(out, flows.to(out.device, torch.float32), blend_alpha)
        engine.synchronize()
        info["temporal_blend_s"] = time.perf_counter() - t0
    return out, info


def precompute_guides_sharded(engine, views, names, output_dir, style, *, masks=None, content_size=512, crop=False, alpha=0.5,
                              depth_maps=None, depth_offset=0.5, depth_prominence=20, save_ext=".jpg", sub_batch=4, group=None,
                              dst=0, write="dst", require_transport=None):
    """The guide-image precompute of the reference's Style_3DGS/train.py:86-115 over all training views, sharded: every view
    is resized as ``adain_inference(content_size=...)`` resizes it (test.py:190-200), stylised, composited with its mask
    (``gt_image_np > 0``, train.py:97) and saved as ``<output_dir>/<name><save_ext>`` — the reference's naming, so the guide
    loss (train.py:208-221) reads the files back unchanged.  ``write="dst"``: the uint8 views are gathered and rank ``dst``
    writes every file (views must then share one size); ``write="local"``: every rank writes its own block and nothing is
    gathered.  Every rank returns the full {name: Path} map once all files exist."""
    from PIL import Image

    from .AdaIN.test import test_transform

    if write not in ("dst", "local"):
        raise ValueError("write must be 'dst' or 'local'")
    rank, world = _rank_world(group)
    out_dir = Path(output_dir)
    out_dir.mkdir(exist_ok=True, parents=True)
    tf = test_transform(content_size, crop)

    class _Views:                         # lazy: a rank only opens and resizes the views of its own block
        def __len__(self):
            return len(views)

        def __getitem__(self, k):
            v = views[k]
            if isinstance(v, (str, Path)):
                v = Image.open(str(v))
            return tf(v) if not isinstance(v, torch.Tensor) else v

    frames = _Views()
    paths = {nm: out_dir / f"{nm}{save_ext}" for nm in names}
    u8, info = stylize_frames_sharded(engine, frames, style, alpha=alpha, depth_maps=depth_maps, depth_offset=depth_offset,
                                      depth_prominence=depth_prominence, masks=masks, sub_batch=sub_batch, group=group, dst=dst,
                                      gather=(write == "dst"), require_transport=require_transport)
    lo, hi = info["shard"]
    t0 = time.perf_counter()
    if write == "dst":
        if rank == dst:
            arr = u8.cpu().numpy()
            for k, nm in enumerate(names):
                Image.fromarray(arr[k]).save(str(paths[nm]))
    else:
        arr = u8.cpu().numpy()
        for k in range(lo, hi):
            Image.fromarray(arr[k - lo]).save(str(paths[names[k]]))
    info["write_s"] = time.perf_counter() - t0
    host_barrier(group)                   # every file exists when any rank returns
    return paths, info
