"""Video style transfer on the MI355X AdaIN path — the caller of ``adain_inference`` in the reference's video/utils.py
(SURVEY.md 8(f) 3): ``apply_style_transfer_ada`` (:244-295) and ``apply_style_transfer_multi_ada`` (:297-372) with their
parameters, defaults, file naming and per-frame semantics:

    frame -> adain_inference(content_size=256, use_depth=True, depth_offset=offset, depth_prominence=prominence)
          -> cv2.resize(target_resolution, INTER_AREA)
          -> blend(stylized, warp(previous result, flow(previous frame -> frame)), alpha)     (from the second frame on)
          -> <output_dir>/<frame file name>

What runs where: the per-frame AdaIN forwards are independent and go through ``jobs.video_style_transfer_sharded`` (one rank,
or every rank of an initialised ``torch.distributed`` group: contiguous frame blocks, one gather); the INTER_AREA resize, the
flow warp and the blend are GPU pixel kernels; the frame-to-frame recurrence and the file writes run on rank 0.

What stays with the caller, because the reference gets it from libraries this package does not depend on:
  * the proximity maps: ``use_depth=True`` needs the depth provider (``AdaIN.test.set_depth_provider``; the reference pulls
    MiDaS through torch.hub per frame) — or pass ``depth_maps=``;
  * the optical flow: ``set_flow_provider(fn)`` with ``fn(prev_frame_path, frame_path, target_resolution, method) ->
    [2,H,W] float32`` (x then y displacement at the target resolution, what ``estimate_optical_flow`` returns for the two
    resized frames, :75-86, :322-358) — OpenCV's Farnebäck / DualTV-L1 estimators are the caller's.
One deliberate difference: the reference writes every stylised frame as a JPEG into a temporary directory and reads it back
(:261-273); here the uint8 frames stay in memory unless ``intermediate_jpeg=True`` re-creates that lossy round trip.
"""
import io
import os
from pathlib import Path

import numpy as np
import torch
from PIL import Image

from . import jobs
from .AdaIN import test as adain_test

_flow_provider = None
_IMAGE_EXT = (".jpg", ".jpeg", ".png")


def set_flow_provider(fn):
    """``fn(prev_frame_path, frame_path, target_resolution, method) -> [2,H,W] float32`` (numpy or torch).  None clears it."""
    global _flow_provider
    _flow_provider = fn


def estimate_optical_flow(prev_frame_path, frame_path, target_resolution, method="farneback"):
    if _flow_provider is None:
        raise RuntimeError("no optical-flow provider: call video.set_flow_provider(fn) (the reference uses OpenCV's Farnebäck / "
                           "DualTV-L1 estimators here, which this package does not depend on)")
    f = _flow_provider(prev_frame_path, frame_path, target_resolution, method)
    return torch.as_tensor(np.asarray(f) if not isinstance(f, torch.Tensor) else f, dtype=torch.float32)


def normalize_image(image):
    """uint8 -> float32 in [0,1] before blending (video/utils.py:217-221)."""
    return image.astype(np.float32) / 255.0 if image.dtype == np.uint8 else image


def blend_images(stylized, warped, alpha):
    """The reference's host blend (video/utils.py:223-229), numpy float32: what ``adain_warp_blend_u8`` computes after its warp."""
    blended = alpha * normalize_image(stylized) + (1 - alpha) * normalize_image(warped)
    return np.clip(blended * 255, 0, 255).astype(np.uint8)


def _frame_files(content_dir):
    return sorted(f for f in os.listdir(content_dir) if f.lower().endswith(_IMAGE_EXT))


def _jpeg_roundtrip(u8):
    out = np.empty_like(u8)
    for i, fr in enumerate(u8):
        buf = io.BytesIO()
        Image.fromarray(fr).save(buf, format="JPEG")          # PIL defaults, as torchvision's save_image(".jpg") uses them
        out[i] = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB"))
    return out


def _run(content_dir, style_paths, output_dir, flow_method, alpha, target_resolution, cancel_flag, offset, prominence, engine,
         vgg_str, decoder_str, depth_maps, intermediate_jpeg, group):
    from . import sharding as sh
    from .engine import AdaINEngine

    os.makedirs(output_dir, exist_ok=True)
    names = _frame_files(content_dir)
    rank, world = sh.rank_world(group)
    dev = engine.device if engine is not None else (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None)
    # Everything that can stop the job is settled BEFORE any rank starts computing, with one status word, so that no rank is
    # ever left waiting in a collective for a peer that has returned or raised: cancellation (the flag is a per-process Event),
    # and the optical-flow provider the rank-0 recurrence will need from the second frame on.
    cancelled = cancel_flag is not None and cancel_flag.is_set()
    need_flow = rank == 0 and len(names) > 1 and _flow_provider is None
    go = sh.agree_min(0 if need_flow else 1 if cancelled else 2, group, dev)
    if go == 0:
        if need_flow:
            estimate_optical_flow(None, None, None)            # raises the "no optical-flow provider" error
        raise RuntimeError("video style transfer: rank 0 has no optical-flow provider (video.set_flow_provider)")
    if go == 1:                                                # cancelled on some rank: every rank stops, as the reference loop does
        print("Stopping style transfer...")
        return None
    if len(names) == 0:                                        # an empty directory: the reference's loop is a no-op
        return Path(output_dir)
    if engine is None:
        engine = AdaINEngine(torch.load(vgg_str, map_location="cpu"), torch.load(decoder_str, map_location="cpu"))
    tf = adain_test.test_transform_u8(256, False)              # adain_inference(content_size=256) of the reference loop
    on_gpu = torch.device(engine.device).type == "cuda"
    stf = adain_test.test_transform(512, False)                # its default style_size

    class Frames:                                             # lazily: a rank opens only the frames of its own block
        def __len__(self):
            return len(names)

        def __getitem__(self, k):
            img = Image.open(os.path.join(content_dir, names[k])).convert("RGB")
            if on_gpu:                                        # Resize(256) on the device, PIL's bytes exactly (csrc/resample.hip)
                return adain_test.device_transform_u8(img, 256, False, engine.device)[0]
            return tf(img)

    class Depths:
        def __len__(self):
            return len(names)

        def __getitem__(self, k):
            if depth_maps is not None:
                return torch.as_tensor(depth_maps[k], dtype=torch.float32)
            return torch.as_tensor(adain_test.midas_depth_map_est(Image.open(os.path.join(content_dir, names[k]))), dtype=torch.float32)

    styles = [stf(Image.open(p).convert("RGB")).unsqueeze(0) for p in style_paths]
    # stylise (sharded), resize to the target resolution on the owning rank, gather; the recurrence needs the flows: rank 0 only
    post = None
    if intermediate_jpeg:
        post = lambda u8: torch.from_numpy(_jpeg_roundtrip(u8.cpu().numpy())).to(u8.device)
    frames_u8, info = jobs.stylize_frames_sharded(
        engine, Frames(), styles, style_of=jobs.style_schedule(len(names), len(styles)), depth_maps=Depths(), depth_offset=offset,
        depth_prominence=prominence,
        post=(lambda u8: engine.resize_area_u8(post(u8) if post else u8, target_resolution)) if target_resolution is not None else post,
        out_hw=(int(target_resolution[1]), int(target_resolution[0])) if target_resolution is not None else None,
        group=group)
    err = None
    if rank == 0:
        # the frame-to-frame recurrence (video/utils.py:355-368) on the gathered frames; every frame is written by a worker
        # thread behind an asynchronous device -> host copy while the next frame's warp / blend is already running
        sink = jobs.FileSink(engine.device)
        try:
            n, h, w, _ = frames_u8.shape
            prev = None
            for i, name in enumerate(names):
                if cancel_flag is not None and cancel_flag.is_set():
                    print("Stopping style transfer...")
                    break
                cur = frames_u8[i]
                if prev is not None:
                    flow = estimate_optical_flow(os.path.join(content_dir, names[i - 1]), os.path.join(content_dir, name), (w, h), flow_method)
                    cur = engine.warp_blend_u8(cur.contiguous(), prev, flow.to(cur.device), alpha)
                sink.write(cur.unsqueeze(0), [os.path.join(output_dir, name)])
                print(f"Stylized and saved: {os.path.join(output_dir, name)}")
                prev = cur
        except Exception as e:
            err = e
        try:
            sink.close()
        except Exception as e:
            err = err or e
    ok = sh.agree(err is None, group, engine.device)             # the other ranks leave with rank 0 - or raise with it
    if err is not None:
        raise err
    if not ok:
        raise RuntimeError("video style transfer: the post-pass failed on rank 0")
    return Path(output_dir) if rank == 0 else None


def apply_style_transfer_ada(content_dir, style_image_path, output_dir, flow_method="farneback", alpha=0.7, target_resolution=None,
                             cancel_flag=None, offset=0.30, prominence=20, *, engine=None,
                             vgg_str="Style_3DGS/AdaIN/models/vgg_normalised.pth", decoder_str="Style_3DGS/AdaIN/models/decoder.pth",
                             depth_maps=None, intermediate_jpeg=False, group=None):
    """One style for the whole clip (video/utils.py:244-295); keyword-only extras: a ready ``engine``, checkpoint paths,
    precomputed ``depth_maps``, the reference's lossy intermediate JPEG, a process group."""
    return _run(content_dir, [style_image_path], output_dir, flow_method, alpha, target_resolution, cancel_flag, offset, prominence,
                engine, vgg_str, decoder_str, depth_maps, intermediate_jpeg, group)


def apply_style_transfer_multi_ada(content_dir, style_dir, output_dir, flow_method="farneback", alpha=0.7, target_resolution=None,
                                   cancel_flag=None, offset=0.30, prominence=20, *, engine=None,
                                   vgg_str="Style_3DGS/AdaIN/models/vgg_normalised.pth",
                                   decoder_str="Style_3DGS/AdaIN/models/decoder.pth", depth_maps=None, intermediate_jpeg=False,
                                   group=None):
    """The styles of ``style_dir`` (sorted) switch through the clip every ``frames // styles`` frames (video/utils.py:297-372)."""
    style_images = sorted(os.listdir(style_dir))
    if len(style_images) == 0:
        raise ValueError("No style images found in the style directory.")
    return _run(content_dir, [os.path.join(style_dir, s) for s in style_images], output_dir, flow_method, alpha, target_resolution,
                cancel_flag, offset, prominence, engine, vgg_str, decoder_str, depth_maps, intermediate_jpeg, group)
