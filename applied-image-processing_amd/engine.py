"""Batched AdaIN engine: many frames / views, one style — the shape of the reference's two batch callers
(video/utils.py:327-350 per-frame loop; Style_3DGS/train.py:86-115 per-view loop), which call
``adain_inference`` once per image and re-encode the same style every time.

The engine keeps the packed weights and the style's channel statistics (2 x 512 floats) resident in
HBM and runs encoder -> fused AdaIN/blend -> decoder [-> mask composite] [-> uint8] for a batch of
frames that is already on the GPU.  Sub-batches bound the activation workspace (the full-resolution
activations are 256 B per pixel and frame).  Everything is launched on the current HIP stream.
"""
import torch

from . import runtime as rt


class AdaINEngine:
    def __init__(self, vgg_state_dict, decoder_state_dict, device=None):
        if not torch.cuda.is_available():
            raise rt.AdainHipError("AdaINEngine needs a GPU (no CPU fallback)")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.enc = rt.pack_encoder(vgg_state_dict, self.device)
        self.dec = rt.pack_decoder(decoder_state_dict, self.device)
        self.s_mean = self.s_std = None

    def synchronize(self):
        torch.cuda.synchronize(self.device)

    @staticmethod
    def abi_calls():
        """C-ABI compute calls this process has made so far (the job drivers report the difference per job)."""
        return rt.ABI_CALLS[0]

    def mark(self):
        """A time stamp on the current stream (a HIP event); ``elapsed(a, b)`` gives the seconds between two of them once the
        work in between has finished.  The job drivers time their phases with these instead of synchronising per phase."""
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream(self.device))
        return e

    def elapsed(self, a, b):
        b.synchronize()
        return a.elapsed_time(b) * 1e-3

    def style_stats(self):
        """The current style's channel statistics (mean, std), each [1,512] on the GPU: what ``set_style`` computed."""
        return self.s_mean, self.s_std

    def use_style_stats(self, stats):
        """Switches to statistics obtained earlier from ``style_stats`` (the job drivers move between a few styles through a
        clip, video/utils.py:335-337: each style image is encoded once, its 2 x 512 statistics are kept by the job)."""
        self.s_mean, self.s_std = stats
        return self

    def set_style(self, style):
        """style [1,3,hs,ws] (or [1,4,...]: the alpha channel is dropped as in test.py:60-61)."""
        if style.shape[1] == 4:
            style = style[:, :3]
        f = rt.encode(style.to(self.device, torch.float32).contiguous(), self.enc)
        self.s_mean, self.s_std = rt.mean_std(f, True)
        return self

    def features(self, images):
        return self._encode(images.to(self.device))

    def _encode(self, content):
        """float NCHW [n,3,h,w] in [0,1], or decoded frames uint8 NHWC [n,h,w,3] (ToTensor then runs inside the first layer's
        kernel: same features bit for bit, a quarter of the bytes)."""
        if content.dtype == torch.uint8:
            return rt.encode_u8(content.contiguous(), self.enc)
        return rt.encode(content.to(torch.float32).contiguous(), self.enc)

    @staticmethod
    def frame_size(content):
        """(h, w) of a content batch in either form (float NCHW / uint8 NHWC)."""
        return tuple(content.shape[1:3]) if content.dtype == torch.uint8 else tuple(content.shape[-2:])

    def stylize(self, content, alpha=0.5, pmap=None):
        """content [n,3,h,w] float (or [n,h,w,3] uint8) on the GPU -> stylised [n,3,8*hc,8*wc].  ``pmap`` [1|n,1,hc,wc] switches
        to the depth-aware blend (test.py:70); otherwise the alpha blend (test.py:80)."""
        assert 0.0 <= alpha <= 1.0
        if self.s_mean is None:
            raise rt.AdainHipError("set_style() first")
        f = self._encode(content)
        c_mean, c_std = rt.mean_std(f, True)
        if pmap is not None:
            g = rt.blend_pmap(f, True, c_mean, c_std, self.s_mean, self.s_std, pmap)
        else:
            g = rt.blend_alpha(f, True, c_mean, c_std, self.s_mean, self.s_std, alpha)
        return rt.decode(g, self.dec)

    def stylize_u8(self, frames_u8, alpha=0.5, depth_maps=None, offset=0.15, prominence=20, masks=None, out=None):
        """A sub-batch of decoded frames uint8 [n,h,w,3] -> finished uint8 frames [n,H,W,3] in ONE call of the C ABI
        (``adain_stylize_u8``): what ``stylize`` / ``stylize_depth`` -> ``composite`` -> ``to_u8`` give, byte for byte, with one
        Python -> C transition per sub-batch instead of eight (the job drivers' launching thread is what eight ranks share)."""
        if self.s_mean is None:
            raise rt.AdainHipError("set_style() first")
        assert 0.0 <= alpha <= 1.0 and 0.0 <= offset <= 1.0
        if depth_maps is not None:
            depth_maps = [d.to(self.device, torch.float32) for d in depth_maps]
        if masks is not None:
            masks = masks.to(self.device)
            if masks.dtype not in (torch.uint8, torch.bool, torch.float32):
                masks = masks.float()
        return rt.stylize_u8(frames_u8.to(self.device), self.enc, self.dec, self.s_mean, self.s_std, alpha, depth_maps, offset, prominence,
                             masks, out)

    def stylize_depth(self, content, depth_maps, offset=0.15, prominence=20):
        """Depth-aware path for a batch: ``depth_maps`` is a list of [h0,w0] GPU tensors, one per frame."""
        assert 0.0 <= offset <= 1.0
        h, w = self.frame_size(content)
        hc, wc = rt.encoded_size(h, w)
        p = torch.cat([rt.strength_map(d, hc, wc, offset, prominence) for d in depth_maps])
        return self.stylize(content, pmap=p)

    def composite(self, content, stylized, masks):
        """masks [n|1, 1|3, hm, wm] float -> content*(1-m) + resize(stylized)*m (test.py:222-236)."""
        content = content.to(self.device)
        content = rt.u8_to_f32(content.contiguous()) if content.dtype == torch.uint8 else content.to(torch.float32).contiguous()
        size = tuple(content.shape[-2:])
        # F.interpolate to the size a tensor already has is the identity for both modes (nearest: index i -> i; bilinear with
        # align_corners=False: source coordinate i exactly, weight 0 on the neighbour), so equal sizes skip the two resize passes
        # (24 B per pixel each): the usual case - frames whose sides are multiples of 8, masks made from the frame itself
        m = masks.to(self.device, torch.float32).contiguous()
        if tuple(m.shape[-2:]) != size:
            m = rt.resize_nearest(m, size)
        s = stylized if tuple(stylized.shape[-2:]) == size else rt.resize_bilinear(stylized, size)
        return rt.mask_composite(content, s.contiguous(), m)

    def to_u8(self, images, out=None):
        return rt.quantize_u8(images, out)

    def resize_area_u8(self, frames_u8, dsize):
        """cv2.resize(frame, dsize, interpolation=cv2.INTER_AREA) per frame (video/utils.py:352-353); dsize = (width, height)."""
        return rt.resize_area_u8(frames_u8, dsize)

    def warp_blend_u8(self, cur, prev, flow, alpha=0.7):
        """One step of the video recurrence: blend(cur, warp(prev, flow), alpha) on uint8 HWC frames (video/utils.py:89-105, :223-229)."""
        return rt.warp_blend_u8(cur, prev, flow, alpha)

    def temporal_blend(self, frames_u8, flows, alpha=0.7):
        return temporal_blend(frames_u8, flows, alpha)


def precompute_guides(engine, views, names, output_dir, masks=None, content_size=512, crop=False, alpha=0.5,
                      depth_maps=None, depth_offset=0.5, depth_prominence=20, save_ext=".jpg", sub_batch=8):
    """Batched counterpart of the guide-image loop of the reference's Style_3DGS/train.py:86-115: every view
    is resized like ``adain_inference(content_size=...)`` does (test.py:190-200), stylised against the engine's
    current style, composited with its mask (``gt_image_np > 0``, train.py:97) and written to
    ``<output_dir>/<name><save_ext>`` — the same file naming, so the guide loss (train.py:208-221) reads it back
    unchanged.  ``views`` are PIL images (or paths); same-sized views are processed ``sub_batch`` at a time.
    Returns {name: Path}."""
    from pathlib import Path

    import numpy as np
    from PIL import Image

    from .AdaIN.test import save_image, test_transform

    out_dir = Path(output_dir)
    out_dir.mkdir(exist_ok=True, parents=True)
    tf = test_transform(content_size, crop)
    tensors = []
    for v in views:
        if isinstance(v, (str, Path)):
            v = Image.open(str(v))
        t = tf(v)
        tensors.append(t[:3] if t.shape[0] == 4 else t)
    paths = {}
    i = 0
    while i < len(tensors):
        j = i + 1
        while j < len(tensors) and j - i < sub_batch and tensors[j].shape == tensors[i].shape:
            j += 1
        content = torch.stack(tensors[i:j]).to(engine.device)
        if depth_maps is not None:
            out = engine.stylize_depth(content, [d.to(engine.device, torch.float32) for d in depth_maps[i:j]], depth_offset,
                                       depth_prominence)
        else:
            out = engine.stylize(content, alpha)
        for k in range(i, j):
            img = out[k - i:k - i + 1]
            if masks is not None and masks[k] is not None:
                m = masks[k]
                m = m if isinstance(m, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(m))
                img = engine.composite(content[k - i:k - i + 1], img, m.float().unsqueeze(0))
            p = out_dir / f"{names[k]}{save_ext}"
            save_image(img, str(p))
            paths[names[k]] = p
        i = j
    return paths


def pooled_style_embedding(style_f):
    """[1,512,h,w] relu4_1 features -> [1,512] (reference Style_3DGS/train.py:80-84: adaptive_avg_pool2d + view)."""
    return style_f.float().mean(dim=(2, 3)).view(style_f.shape[0], style_f.shape[1])


def temporal_blend(frames_u8, flows, alpha=0.7):
    """Temporal-consistency post-pass of the video caller (reference video/utils.py:352-369): frame 0 is kept,
    frame i becomes blend(frame_i, warp(result_{i-1}, flow_{i-1}), alpha).  ``frames_u8`` [n,h,w,c] uint8 stylised
    frames already at the target resolution, ``flows`` [n-1,2,h,w] float32 (prev -> current, caller-supplied: the
    optical-flow estimator stays outside), both on the GPU.  The recurrence is sequential over frames, so it runs
    on one rank over the gathered frames; each step is one pixel kernel."""
    n = frames_u8.shape[0]
    if flows.shape[0] != max(n - 1, 0):
        raise rt.AdainHipError("temporal_blend: need one flow field per consecutive frame pair")
    out = torch.empty_like(frames_u8)
    if n:
        out[0].copy_(frames_u8[0])
    for i in range(1, n):             # one launch per frame, written straight into the result block (no intermediate, no copy kernel)
        rt.warp_blend_u8(frames_u8[i], out[i - 1], flows[i - 1], alpha, out=out[i])
    return out


class GraphedStylize:
    """The whole ``engine.stylize`` pass (about 30 kernel launches) captured once into a hipGraph and replayed with one
    launch per batch — for the reference's real-world sizes (256 or 512 pixel frames, video/utils.py:264,
    test.py:160) the per-kernel launch overhead of the eager path is a visible share of a 1-2 ms forward.
    Every C-ABI entry point only enqueues work on the given stream and neither allocates nor synchronises, which is
    what makes the capture legal.  The style must be set before capture (its statistics are baked into the graph's
    input buffers by reference, so ``engine.set_style`` followed by a new capture is needed to change it).  The graph
    holds raw addresses of the engine's packed weights and style statistics: the object keeps those tensors alive
    (``_keep``), so a later ``engine.set_style`` cannot hand their memory back to the allocator under a live graph."""

    def __init__(self, engine, n, h, w, alpha=0.5, to_u8=False):
        self.engine = engine
        self._keep = (engine.s_mean, engine.s_std, engine.enc, engine.dec)
        self.static_in = torch.zeros((n, 3, h, w), dtype=torch.float32, device=engine.device)
        side = torch.cuda.Stream(engine.device)
        side.wait_stream(torch.cuda.current_stream(engine.device))
        with torch.cuda.stream(side):            # warm-up outside the capture (lazy library state); the activation workspaces are
                                                 # keyed by stream, so the capture below allocates its own from the graph's pool
            for _ in range(2):
                out = engine.stylize(self.static_in, alpha)
                if to_u8:
                    out = engine.to_u8(out)
        torch.cuda.current_stream(engine.device).wait_stream(side)
        torch.cuda.synchronize(engine.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            out = engine.stylize(self.static_in, alpha)
            self.static_out = engine.to_u8(out) if to_u8 else out

    def __call__(self, content):
        """content [n,3,h,w] (GPU or pinned host tensor) -> the graph's output buffer (overwritten by the next call)."""
        self.static_in.copy_(content, non_blocking=True)
        self.graph.replay()
        return self.static_out
