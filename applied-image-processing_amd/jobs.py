"""Sharded job drivers: the reference's two batch callers of the AdaIN path, cut across the GPUs of one node.

* ``video_style_transfer_sharded`` — reference video/utils.py:297-369 (``apply_style_transfer_multi_ada`` / ``_ada``): one
  ``adain_inference`` per frame (each reloads the weights and re-encodes the style), INTER_AREA resize to the target
  resolution, then the flow-warp / blend recurrence over the frames and one file per frame.
* ``precompute_guides_sharded`` — reference Style_3DGS/train.py:86-115: one ``adain_inference`` per training camera with
  the mask ``gt_image_np > 0`` and ``<model_path>/stylized/<image_name>.jpg`` as output.

Both sit on ``stylize_frames_sharded``: the frame index is cut into contiguous blocks (``sharding.shard_range``), every
rank (one process per GPU) runs the batched engine on its block against replicated weights and style statistics, and the
finished uint8 frames meet on ``dst`` in ONE gather (RCCL over xGMI when the process group's CUDA backend is ``nccl``).
There is no other collective on the data path, and nothing inside the frame loop synchronises the device or talks to another
rank: one one-word status all_reduce per job (``sharding.agree_geometry``: every rank learns whether all blocks finished and
what a finished frame looks like) precedes the gather, one device synchronisation ends the job.  What is sequential in the
reference stays on ``dst`` after the gather: the frame-to-frame warp / blend recurrence and (by default) the file writes.

Host side of a job (the reference goes file -> PIL -> tensor -> device per frame, test.py:190-204): ``FrameFeeder`` decodes /
fetches the frames of the next sub-batches on a worker thread, stages them in pinned buffers and uploads them on a copy stream
while the compute stream works on the current sub-batch.  Decoded frames travel as uint8 HWC (3 bytes per pixel; ToTensor runs
on the device, bit-identical to the host's ``/ 255``); ``FileSink`` brings finished uint8 frames back with asynchronous copies
and writes the image files on worker threads.

The engine is passed in (``engine.AdaINEngine`` on a GPU); the drivers only use its methods, so the multi-process logic is
exercised on CPU under ``gloo`` with a stand-in engine (tests/test_distributed_gloo.py).
"""
import queue
import threading
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
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


def _rank_world(group):
    return sh.rank_world(group)


def host_barrier(group=None, device=None):
    """Host-side rendezvous of the ranks (see ``sharding.host_barrier``)."""
    sh.host_barrier(group, device)


# ---------------------------------------------------------------------------------------------------------------------------------
# frames on their way to the device
# ---------------------------------------------------------------------------------------------------------------------------------
def as_frame(x):
    """One frame in the form the drivers move it, and its kind: ``"u8"`` = a decoded frame uint8 [h,w,3] (an RGB PIL image,
    a numpy array or a tensor: what the reference holds BEFORE ToTensor, test.py:190-200; ToTensor then runs on the device) or
    ``"f32"`` = an already transformed float tensor [3,h,w] in [0,1] (a fourth channel is dropped)."""
    if not isinstance(x, (torch.Tensor, np.ndarray)):          # a PIL image
        if getattr(x, "mode", None) == "RGB":
            x = np.array(x)                                   # (a writable copy: np.asarray of a PIL image is read-only)
        else:
            from .AdaIN.test import _to_tensor

            x = _to_tensor(x)
    if isinstance(x, np.ndarray):
        x = np.ascontiguousarray(x)
        x = torch.from_numpy(x if x.flags.writeable else x.copy())
    if x.dtype == torch.uint8:
        if x.dim() != 3 or x.shape[2] != 3:
            raise ValueError(f"a uint8 frame must be [h,w,3], got {tuple(x.shape)}")
        return x, "u8"
    if x.dim() != 3 or x.shape[0] not in (3, 4):
        raise ValueError(f"a float frame must be [3,h,w] (or [4,h,w]), got {tuple(x.shape)}")
    return x[:3].to(torch.float32), "f32"


# sub_batch=None: frames per sub-batch chosen from the frame size, about three megapixels of content per sub-batch.  Measured on
# device-resident frames (tools/probes/small_frame_probe.py, frames/s at sub-batches of 1 / 2 / 4 / 8 / 16 / 32): 256 x 456 - the
# reference's own video shape, content_size = 256 - 1587 / 2164 / 2431 / 2570 / 2590 / 2579; 512 x 912: 611 / 646 / 670 / 671 / 659 /
# 651; 1080p: 154 / 153 / 151 / 148 / 146 / 145.  Small frames need company to fill the chip (a 256 x 456 frame is 960 tile items
# for 512 resident workgroups); large ones lose to the L2-miss traffic that grows faster than the batch.
AUTO_SUB_BATCH_PIXELS = 3.0e6
MAX_AUTO_SUB_BATCH = 32
# Frames of 1.5 - 3 megapixels (1080p, 1200 x 1600, 1408 x 1408): since round 4 the C schedules run the big layers of a batch of WIDE
# frames frame by frame and only its relu4-level layers over the whole batch (csrc/api.hip, BIG_FRAME_WIDTH / BIG_LAYER_ROUNDS), so a
# sub-batch no longer costs the mid-network layers anything and fills the last round of conv4_1 / dec1: per step 331.1 / 331.8 / 334.9
# Mpixels/s at 1 / 2 / 4 frames of 1080p, 324.1 / 326.1 / 328.2 at 1200 x 1600 (same box); narrower frames of that size always gained
# from a batch (1280 x 1280: 335.5 / 337.5 at 2 / 4).  Larger frames have no layer left that a batch would help.
LARGE_FRAME_PIXELS = (1.5e6, 3.0e6)
LARGE_FRAME_SUB_BATCH = 4
MAX_QUEUED_BATCHES = 6             # sub-batches enqueued ahead of the device (stylize_frames_sharded)


def auto_sub_batch(h, w):
    if LARGE_FRAME_PIXELS[0] <= h * w < LARGE_FRAME_PIXELS[1]:
        return LARGE_FRAME_SUB_BATCH
    return int(max(1, min(MAX_AUTO_SUB_BATCH, round(AUTO_SUB_BATCH_PIXELS / max(1, h * w)))))


def sleep_wait(event, period=2e-4):
    """Host wait for a HIP event that SLEEPS.  ``event.synchronize()`` - and ``torch.cuda.synchronize`` - spin a core for as long as
    the GPU is busy on this ROCm build, also for events created with ``blocking=True`` (measured: tools/probes/thread_cpu_probe.py:
    the waiting thread's CPU time equals the wall time); with eight ranks on one host that is eight cores doing nothing.  Polling
    ``query()`` with a short sleep costs nothing and at most ``period`` of latency, which none of the waits below is sensitive to."""
    while not event.query():
        time.sleep(period)


class _Batch:
    __slots__ = ("i", "j", "content", "depth", "mask", "ready", "slot")

    def __init__(self, i, j, content, depth=None, mask=None, ready=None, slot=None):
        self.i, self.j, self.content, self.depth, self.mask, self.ready, self.slot = i, j, content, depth, mask, ready, slot


class FrameFeeder:
    """Feeds ``frames[lo:hi]`` to the compute loop as sub-batches that are already on the device.

    A feeder thread cuts sub-batches (at most ``sub_batch`` frames of one style, one size and one kind), copies them into a
    pinned staging slot and uploads the slot on a dedicated copy stream; the per-frame proximity maps and masks of the sub-batch
    ride along.  The frames themselves are fetched (``frames[k]`` may decode a file: it must be safe to call for different k
    from several threads) and copied into the pinned slot by a small pool of ``workers`` threads, a window of frames ahead of
    the cutter: PIL's decoders and numpy's memcpy release the interpreter lock, and ONE thread copying pageable frames into
    pinned memory was measured at 0.7-1.1 GB/s on the GPU box - a 1080p job needs 1 GB/s.  ``depth`` slots are in flight: the
    upload of sub-batch k+1 (and the decode of k+2 ...) overlaps the kernels of sub-batch k.  The consumer waits on the
    sub-batch's HIP event on ITS stream (no host synchronisation) and calls ``release`` once its last kernel reading the slot
    is enqueued.

    ``frames`` may offer ``block(i, j)`` -> a ready [j-i, ...] tensor of frames i..j-1 of one size (device-resident frame
    stores hand out views: no staging, no copy).  On a CPU engine (tests) the same thread and cutting logic run without
    pinned memory or streams."""

    def __init__(self, frames, lo, hi, style_of, sub_batch, device, depth_maps=None, masks=None, depth=4, cuts=(), workers=4):
        self.frames, self.lo, self.hi, self.style_of = frames, lo, hi, style_of
        self.auto = sub_batch is None          # frames per sub-batch from the frame size (auto_sub_batch), per run of equal-sized frames
        self.sub_batch = MAX_AUTO_SUB_BATCH if self.auto else max(1, int(sub_batch))
        self.cuts = set(cuts)              # frame indices at which a sub-batch must end (chunk borders of a chunked gather)
        self.device = torch.device(device)
        self.depth_maps, self.masks = depth_maps, masks
        self.cuda = self.device.type == "cuda"
        self.nslots = max(2, int(depth))
        self.q = queue.Queue()
        self.free = threading.Semaphore(self.nslots)
        self.stop = threading.Event()
        self.h2d_bytes = 0
        self.fetch_s = 0.0
        # where the worker thread's time goes (seconds): waiting for a free slot (= the consumer is the bottleneck, as it should
        # be), waiting for a slot's previous upload to leave its pinned buffer, copying frames into pinned memory
        self.stats = {"wait_slot_s": 0.0, "wait_h2d_s": 0.0, "stage_s": 0.0, "batches": 0}
        if self.cuda:
            self.copy_stream = torch.cuda.Stream(self.device)
            self.compute_stream = torch.cuda.current_stream(self.device)
            self.pinned, self.dev = {}, {}         # (slot, "content" | "mask" | "depth") -> staging buffers
            self.h2d_done = [None] * self.nslots
            self.consumed = [None] * self.nslots
        self.pool = ThreadPoolExecutor(max_workers=max(1, int(workers)), thread_name_prefix="adain-frame-fetch")
        self.window = max(2 * (8 if self.auto else self.sub_batch), 2 * max(1, int(workers)))      # frames fetched ahead of the cutter
        self.thread = threading.Thread(target=self._run, name="adain-frame-feeder", daemon=True)
        self.thread.start()

    # ---- worker thread ----------------------------------------------------------------------------------------------------------
    def _cut(self):
        """Yields (i, j, frames-or-block, kind) for consecutive sub-batches of [lo, hi)."""
        k, carry = self.lo, None
        block = getattr(self.frames, "block", None)
        ahead, nxt = [], self.lo               # futures of frames nxt - len(ahead) .. nxt - 1, fetched by the pool in index order

        def fetch(idx):
            return as_frame(self.frames[idx])

        def get(idx):
            nonlocal nxt
            while nxt < self.hi and len(ahead) < self.window and not self.stop.is_set():
                ahead.append(self.pool.submit(fetch, nxt))
                nxt += 1
            return ahead.pop(0).result()

        def size_of(t, kd):
            return (t.shape[-3], t.shape[-2]) if kd == "u8" else (t.shape[-2], t.shape[-1])

        while k < self.hi:
            i = k
            if block is not None:
                limit = self.sub_batch
                if self.auto:
                    one = block(i, i + 1)
                    limit = auto_sub_batch(*size_of(one, "u8" if one.dtype == torch.uint8 else "f32"))
                j = i + 1
                while j < self.hi and j - i < limit and self.style_of[j] == self.style_of[i] and j not in self.cuts:
                    j += 1
                t = block(i, j)
                yield i, j, t, ("u8" if t.dtype == torch.uint8 else "f32")
                k = j
                continue
            items, kind = [], None
            limit = self.sub_batch
            while k < self.hi and len(items) < limit and self.style_of[k] == self.style_of[i] and not (items and k in self.cuts):
                fr, kd = carry if carry is not None else get(k)
                carry = None
                if not items and self.auto:
                    limit = auto_sub_batch(*size_of(fr, kd))
                if items and (kd != kind or fr.shape != items[0].shape or fr.device != items[0].device):
                    carry = (fr, kd)          # another size / kind: it opens the next sub-batch
                    break
                items.append(fr)
                kind = kd
                k += 1
            yield i, k, items, kind

    def _upload(self, host):
        """A one-off upload (shapes that do not fit the slot buffers): pinned copy + asynchronous H2D on the copy stream."""
        t = host if host.is_pinned() else host.pin_memory()
        d = t.to(self.device, non_blocking=True)
        d.record_stream(self.compute_stream)             # allocated under the copy stream, read by the compute stream
        self.h2d_bytes += host.numel() * host.element_size()
        return d

    def _stage(self, s, name, items):
        """Host tensors ``items`` (one shape, one dtype) -> the slot's pinned buffer ``name`` (one memcpy each, no intermediate
        stack) -> the slot's device buffer, asynchronously on the copy stream (the caller holds the stream context and has made
        the stream wait for the slot's previous consumers).  Returns the device view [len(items), ...]."""
        nb = len(items)
        # slot capacity: the fixed sub-batch size, or - automatic sub-batches - this batch's own length (constant over a run of
        # equal-sized frames; the buffers are re-made when a longer batch or another frame size comes)
        shape = (nb if self.auto else self.sub_batch,) + tuple(items[0].shape)
        key = (s, name)
        pin = self.pinned.get(key)
        if pin is None or tuple(pin.shape[1:]) != shape[1:] or pin.shape[0] < nb or pin.dtype != items[0].dtype:
            pin = self.pinned[key] = torch.empty(shape, dtype=items[0].dtype, pin_memory=True)
            self.dev[key] = torch.empty(shape, dtype=items[0].dtype, device=self.device)
        pin_np = pin.numpy()

        def put(q_, it):
            np.copyto(pin_np[q_], it.numpy())              # a plain memcpy outside the interpreter lock

        if nb > 1:
            list(self.pool.map(put, range(nb), items))
        else:
            put(0, items[0])
        self.dev[key][:nb].copy_(pin[:nb], non_blocking=True)
        self.h2d_bytes += nb * items[0].numel() * items[0].element_size()
        return self.dev[key][:nb]

    def _extras(self, i, j):
        """The sub-batch's proximity maps (float32 [h0,w0] each) and masks ([1|3,hm,wm] each; bool / uint8 / float) as host or
        device tensors, still one per frame."""
        depth = mask = None
        if self.depth_maps is not None:
            depth = [torch.as_tensor(self.depth_maps[k], dtype=torch.float32) for k in range(i, j)]
        if self.masks is not None:
            ms = [torch.as_tensor(self.masks[k]) for k in range(i, j)]
            mask = [m if m.dtype in (torch.uint8, torch.bool, torch.float32) else m.float() for m in ms]
        return depth, mask

    @staticmethod
    def _uniform(ts):
        return all(t.shape == ts[0].shape and t.dtype == ts[0].dtype and not t.is_cuda for t in ts)

    def _run(self):
        cpu0 = time.thread_time()
        try:
            self._run_inner()
        finally:
            self.stats["feeder_thread_cpu_s"] = time.thread_time() - cpu0

    def _run_inner(self):
        try:
            if self.cuda:
                torch.cuda.set_device(self.device)
            b = 0
            for i, j, items, kind in self._cut():
                if self.stop.is_set():
                    return
                t0 = time.perf_counter()
                depth, mask = self._extras(i, j)
                t1 = time.perf_counter()
                self.fetch_s += t1 - t0
                self.free.acquire()
                t2 = time.perf_counter()
                self.stats["wait_slot_s"] += t2 - t1
                self.stats["batches"] += 1
                if self.stop.is_set():
                    return
                s = b % self.nslots
                b += 1
                ready = None
                if not self.cuda:
                    content = items if isinstance(items, torch.Tensor) else torch.stack(items)
                    if mask is not None and all(m.shape == mask[0].shape for m in mask):
                        mask = torch.stack(mask)
                else:
                    if self.h2d_done[s] is not None:
                        sleep_wait(self.h2d_done[s])                     # the slot's previous uploads have left its pinned buffers
                    t3 = time.perf_counter()
                    self.stats["wait_h2d_s"] += t3 - t2
                    # device tensors handed over by the stores (a mask or depth map computed lazily in __getitem__, a frame
                    # made on the GPU) were produced on the fetching threads' current stream, not on the copy stream: whatever
                    # reads them below - and the consumer, through `ready` - is ordered behind an event recorded there now
                    on_device = ((isinstance(items, torch.Tensor) and items.is_cuda) or (isinstance(items, list) and items and items[0].is_cuda)
                                 or any(t.is_cuda for t in (depth or [])) or any(t.is_cuda for t in (mask or [])))
                    fetched = None
                    if on_device:
                        fetched = torch.cuda.Event()
                        fetched.record(torch.cuda.current_stream(self.device))
                    with torch.cuda.stream(self.copy_stream):
                        if self.consumed[s] is not None:
                            self.copy_stream.wait_event(self.consumed[s])      # the kernels that read this slot have finished
                        staged = False                                       # something was copied out of the slot's pinned buffers
                        ordered = fetched is not None                        # `ready` must carry an ordering on to the consumer
                        if fetched is not None:
                            self.copy_stream.wait_event(fetched)
                        if isinstance(items, torch.Tensor):              # a ready block (device-resident store)
                            content = items
                        elif items[0].is_cuda:
                            content = None                               # stacked by the consumer on its own stream
                        else:
                            content = self._stage(s, "content", items)
                            staged = True
                        if depth is not None and not all(d.is_cuda for d in depth):
                            if self._uniform(depth):
                                dd = self._stage(s, "depth", depth)
                                depth = [dd[k] for k in range(len(depth))]
                            else:
                                depth = [d if d.is_cuda else self._upload(d) for d in depth]
                            staged = True
                        if mask is not None:
                            if all(m.is_cuda for m in mask):
                                mask = torch.stack(mask)
                                mask.record_stream(self.compute_stream)
                            elif self._uniform(mask):
                                mask = self._stage(s, "mask", mask)
                            else:                                        # masks of several sizes: one upload each, composited per frame
                                mask = [m if m.is_cuda else self._upload(m) for m in mask]
                            staged = True
                        if staged or ordered:
                            ready = torch.cuda.Event()
                            ready.record(self.copy_stream)
                            if staged:                                       # only pinned buffers need the host to wait before reuse
                                self.h2d_done[s] = ready
                    self.stats["stage_s"] += time.perf_counter() - t3
                    if content is None:
                        content = items                                  # list of device tensors
                self.q.put(_Batch(i, j, content, depth, mask, ready, s))
            self.q.put(None)
        except BaseException as e:                                       # re-raised by the consumer
            self.q.put(e)

    # ---- consumer ---------------------------------------------------------------------------------------------------------------
    def __iter__(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            if item.ready is not None:
                torch.cuda.current_stream(self.device).wait_event(item.ready)
            if self.cuda:
                # Device tensors made on the fetch threads (device_transform_u8's resized frames, lazily computed depth maps) belong to
                # THEIR stream's allocator pool: once this batch drops them the block may be handed to the next fetch at once, while the
                # consumer's kernels - on another stream whenever the caller runs the job under `torch.cuda.stream(...)` - are still
                # queued.  Telling the allocator who reads them closes that (round-5 advisor finding).
                cur = torch.cuda.current_stream(self.device)
                for group in (item.content, item.depth, item.mask):
                    if isinstance(group, list):
                        for t in group:
                            if isinstance(t, torch.Tensor) and t.is_cuda:
                                t.record_stream(cur)
            if isinstance(item.content, list):                           # frames that already were device tensors
                item.content = torch.stack(item.content)
            yield item

    def release(self, batch):
        """The consumer has enqueued the last kernel that reads ``batch``'s slot."""
        if self.cuda and batch.slot is not None:
            e = torch.cuda.Event()
            e.record(torch.cuda.current_stream(self.device))
            self.consumed[batch.slot] = e
        self.free.release()

    def close(self):
        self.stop.set()
        for _ in range(self.nslots + 1):
            self.free.release()
        self.thread.join(30)
        self.pool.shutdown(wait=True, cancel_futures=True)


class HostCopier:
    """Asynchronous device -> pinned-host copies on a stream of their own: a finished block leaves the device while the next
    sub-batch's kernels run.  ``finish()`` waits for all of them."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.bytes = 0
        if self.cuda:
            self.stream = torch.cuda.Stream(self.device)

    def copy(self, dst_host, src):
        """dst_host.copy_(src) once the work enqueued so far on the current stream has finished."""
        self.bytes += src.numel() * src.element_size()
        if not self.cuda:
            dst_host.copy_(src)
            return
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            dst_host.copy_(src, non_blocking=True)
            src.record_stream(self.stream)

    def finish(self):
        if self.cuda:
            done = torch.cuda.Event()
            done.record(self.stream)
            sleep_wait(done)


class FileSink:
    """Writes finished uint8 frames to image files off the compute path: the device -> host copy of a sub-batch runs on its own
    stream into a pinned buffer, a worker thread waits for it and encodes / saves the files (PIL), so the next sub-batch's
    kernels are already running.  ``close()`` waits for every file and re-raises the first error."""

    def __init__(self, device, workers=4, max_in_flight=None):
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="adain-file-sink")
        self.futures = []
        self.d2h_bytes = 0
        # back-pressure: at most `max_in_flight` blocks (default 2 per writer) sit in pinned memory waiting for their encoder;
        # write() blocks when the writers fall behind instead of pinning the whole job.  The pinned buffers are reused.
        self.max_in_flight = max(1, int(max_in_flight if max_in_flight is not None else 2 * workers))
        self.slots = threading.Semaphore(self.max_in_flight)
        self.spare = {}                    # block shape -> idle pinned buffers
        self.spare_lock = threading.Lock()
        self.wait_s = 0.0                  # time write() spent blocked on the writers
        if self.cuda:
            self.stream = torch.cuda.Stream(self.device)

    def _pinned(self, shape):
        with self.spare_lock:
            idle = self.spare.get(shape)
            if idle:
                return idle.pop()
        return torch.empty(shape, dtype=torch.uint8, pin_memory=True)

    @staticmethod
    def _save(arr, path):
        from PIL import Image

        Image.fromarray(arr[:, :, 0] if arr.shape[2] == 1 else arr).save(str(path))

    def write(self, u8_block, paths):
        """u8_block [k,h,w,c] on the engine's device (finished on the current stream); paths: k file paths."""
        t0 = time.perf_counter()
        self.slots.acquire()
        self.wait_s += time.perf_counter() - t0
        shape = tuple(u8_block.shape)
        if not self.cuda:
            host, done = u8_block, None
        else:
            cur = torch.cuda.current_stream(self.device)
            ready = torch.cuda.Event()
            ready.record(cur)
            host = self._pinned(shape)
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ready)
                host.copy_(u8_block, non_blocking=True)
                u8_block.record_stream(self.stream)
                done = torch.cuda.Event()
                done.record(self.stream)
            self.d2h_bytes += u8_block.numel()

        def job():
            try:
                if done is not None:
                    sleep_wait(done)                    # (a writer thread must not spin a core while the copy is on its way)
                arr = host.numpy()
                for k, p in enumerate(paths):
                    self._save(arr[k], p)
            finally:
                if done is not None:
                    with self.spare_lock:
                        self.spare.setdefault(shape, []).append(host)
                self.slots.release()

        self.futures.append(self.pool.submit(job))

    def close(self):
        err = None
        for f in self.futures:
            try:
                f.result()
            except BaseException as e:      # keep waiting for the others, report the first
                err = err or e
        self.pool.shutdown(wait=True)
        self.futures = []
        if err is not None:
            raise err


# ---------------------------------------------------------------------------------------------------------------------------------
def _mark(engine):
    m = getattr(engine, "mark", None)
    return m() if m is not None else time.perf_counter()


def _elapsed(engine, a, b):
    f = getattr(engine, "elapsed", None)
    return f(a, b) if f is not None else b - a


def stylize_frames_sharded(engine, frames, styles, *, style_of=None, alpha=0.5, depth_maps=None, depth_offset=0.15,
                           depth_prominence=20, masks=None, post=None, sub_batch=None, group=None, dst=0, gather=True,
                           require_transport=None, style_cache=None, out_hw=None, gather_chunks=1, agree=True, sink=None,
                           prefetch=4, host_out=None, fetch_workers=4):
    """Stylises ``frames`` (a sequence indexed lazily: a rank only ever touches its own block; an element is a decoded
    frame uint8 [h,w,3] / RGB PIL image, or a float tensor [3,h,w] in [0,1]) and returns ``(frames_u8, info)``: the uint8
    frames [n,H,W,3] in frame order on rank ``dst`` (None on the other ranks; with ``gather=False`` the local block — a
    list of per-size blocks if the frames have several sizes, None if a ``sink`` took them) and a dict with the shard, the
    timings and the transport.

    styles          one style tensor [1|.,3,hs,ws] or a list of them; ``style_of[i]`` picks the style of frame i
                    (``style_schedule``); every style image is encoded once per rank and call, its 2 x 512 statistics
                    are kept for the call (pass a dict as ``style_cache`` to keep them across calls with the SAME styles).
    depth_maps      optional sequence of [h0,w0] proximity maps, one per frame -> depth-aware blend (test.py:52-71) with
                    ``depth_offset`` / ``depth_prominence``; otherwise the ``alpha`` blend (test.py:74-81).
    masks           optional sequence of [1|3,hm,wm] masks -> content-mask composite (test.py:222-236).
    sub_batch       frames per sub-batch; None (default) = chosen from the frame size, about three megapixels per sub-batch
                    (``auto_sub_batch``: 26 frames of 256 x 456, 6 of 512 x 912; 4 of 1080p or 1200 x 1600 - inside the measured optima).
    post            optional ``f(u8_block) -> u8_block`` applied per sub-batch on the owning rank BEFORE the gather
                    (frame-local work such as the INTER_AREA resize, so the gather moves the small frames).
    sink            optional ``f(i, j, u8_block)`` called with every finished sub-batch (frames i..j-1) on the owning rank.
    require_transport   e.g. "rccl": raise before any work if the gather would use another transport.
    out_hw          (H, W) of a finished frame, if the caller knows it: lets ``agree=False`` jobs and chunked gathers run with
                    ranks whose block is empty.
    gather_chunks   1: ONE gather when every rank has finished (after the status word: an error anywhere raises everywhere).
                    k > 1 (needs ``out_hw``: ValueError without it): the block is gathered in k pieces, pieces 0 .. k-2
                    issued asynchronously as soon as they are finished so that they overlap the rest of the compute, then the
                    status word, then the last piece.  Every rank issues every piece whatever happens to its block - a rank that
                    fails (or whose frames are not ``out_hw``) sends zero-filled pieces - so the collective sequence is the same
                    on every rank and the status word still raises everywhere.
    host_out        a (pinned) uint8 host tensor [n,H,W,3]: finished frames are also copied into it, asynchronously on a copy
                    stream as they become available — per sub-batch on a single rank (or with ``gather=False``: every rank fills
                    the rows of its own block), per gathered piece on ``dst`` otherwise — and are all there when the call returns.
    agree           False skips the per-job status word (per-step benchmark mode: every rank must have frames or ``out_hw``).

    The gather needs one frame size over the whole job; with ``gather=False`` sizes may differ from frame to frame (a
    sub-batch ends where the size changes).  Nothing in the frame loop synchronises the device or communicates."""
    rank, world = _rank_world(group)
    n = len(frames)
    lo, hi = sh.shard_range(n, world, rank)
    style_list = list(styles) if isinstance(styles, (list, tuple)) else [styles]
    if style_of is None:
        style_of = [0] * n
    if len(style_of) != n:
        raise ValueError("style_of needs one style index per frame")
    dev = engine.device
    gathering = gather and world > 1
    transport = None
    if int(gather_chunks) > 1 and out_hw is None:
        # Pieces are issued inside the frame loop, BEFORE the status word: a rank that fails must still take part in every one of
        # them (with zero-filled blocks), so it has to know what a finished frame looks like without having finished one.  Asking
        # for pieces without out_hw is an error, decided from the arguments alone: the same on every rank and for every world size
        # (round 4 silently ran such a job with one end gather: no overlap and no signal).
        raise ValueError("gather_chunks > 1 needs out_hw (the size of a finished frame): pieces are gathered before the ranks agree")
    chunks = max(1, int(gather_chunks)) if gathering else 1
    if gathering:
        transport = sh.device_transport(torch.empty(0, dtype=torch.uint8, device=dev), group)
        if require_transport and transport != require_transport:
            raise RuntimeError(f"final gather would run over {transport!r}, not {require_transport!r}")
        if not agree and n < world and out_hw is None:        # the same decision on every rank
            raise ValueError("a job with fewer frames than ranks needs out_hw for unagreed gathers")
    counts = sh.shard_counts(n, world)
    # chunk c of rank r = frames [lo_r + a, lo_r + b) with (a, b) = chunk_bounds(counts[r], chunks)[c]
    my_chunks = sh.chunk_bounds(hi - lo, chunks)

    info = {"rank": rank, "world": world, "shard": (lo, hi), "transport": transport, "gathers": 0}
    t_host0 = time.perf_counter()
    cpu0_thread, cpu0_proc = time.thread_time(), time.process_time()
    abi0 = getattr(engine, "abi_calls", lambda: 0)()
    m0 = _mark(engine)
    blocks, shapes = [], set()             # finished sub-batches (i, j, u8) of this rank, their frame shapes
    pending, landed = [], []               # chunked gather: (chunk, counts, finish closure) in flight; (first frame, block) arrived on dst
    next_chunk = 0
    err = None
    copier = HostCopier(dev) if host_out is not None else None
    feeder = FrameFeeder(frames, lo, hi, style_of, sub_batch, dev, depth_maps, masks, depth=prefetch,
                         cuts=[lo + b for (_, b) in my_chunks] if chunks > 1 else (), workers=fetch_workers)

    def chunk_ready(c, done_upto):
        return lo + my_chunks[c][1] <= done_upto

    def issue_chunk(c, zeros=False):
        """Piece c of this rank's block joins the gather; ``zeros``: this rank has failed - it still takes part, with a zero-filled
        block of the agreed geometry, so that its peers (already inside the collective) are not left waiting; the status word
        after the last in-loop piece then raises on every rank."""
        a, b = my_chunks[c]
        part = [] if zeros else [u8 for (i, j, u8) in blocks if lo + a <= i and j <= lo + b]
        if part:
            local_c = torch.cat(part) if len(part) > 1 else part[0]
        else:
            local_c = torch.zeros((b - a if zeros else 0,) + tuple(out_hw) + (3,), dtype=torch.uint8, device=dev)
        cnts = [sh.chunk_bounds(cr, chunks)[c][1] - sh.chunk_bounds(cr, chunks)[c][0] for cr in counts]
        info["gathers"] += 1
        land_pending()                     # the previous piece has long arrived: hand it on before queueing the next
        pending.append((c, cnts, sh.gather_frames(local_c, n, dst=dst, group=group, async_op=True, counts=cnts)))

    def land_pending():
        """Waits (on the stream) for the gathered pieces issued so far and files them on dst: into the job's result and, piece
        by piece, into ``host_out``."""
        while pending:
            c, cnts, fin = pending.pop(0)
            got = fin()
            if rank != dst:
                continue
            at = 0
            for r in range(world):
                r_lo = sh.shard_range(n, world, r)[0]
                a = sh.chunk_bounds(counts[r], chunks)[c][0]
                if cnts[r]:
                    landed.append((r_lo + a, got[at:at + cnts[r]]))
                    if copier is not None:
                        copier.copy(host_out[r_lo + a:r_lo + a + cnts[r]], got[at:at + cnts[r]])
                at += cnts[r]

    # At most MAX_QUEUED_BATCHES sub-batches are queued on the device beyond the one that is running: enough that the GPU never
    # waits for the host (a sub-batch is about three megapixels: >= 5 ms of kernels), few enough that this thread sleeps in
    # sleep_wait instead of spinning inside hipLaunchKernel on a full HIP queue.
    queued = []
    on_gpu = dev.type == "cuda"
    try:
        cur_style = None
        stats = style_cache if style_cache is not None else {}
        for batch in feeder:
            if on_gpu and len(queued) >= MAX_QUEUED_BATCHES:
                sleep_wait(queued.pop(0))
            i, j = batch.i, batch.j
            if style_of[i] != cur_style:
                cur_style = style_of[i]
                if cur_style not in stats:
                    stats[cur_style] = engine.set_style(style_list[cur_style]).style_stats()
                engine.use_style_stats(stats[cur_style])
            content = batch.content
            one_call = getattr(engine, "stylize_u8", None)
            if (one_call is not None and content.dtype == torch.uint8 and content.dim() == 4 and content.shape[-1] == 3
                    and not isinstance(batch.mask, list)):
                # decoded RGB frames with (at most) one mask tensor for the sub-batch: the whole chain in one C-ABI call
                u8 = one_call(content, alpha=alpha, depth_maps=batch.depth, offset=depth_offset, prominence=depth_prominence, masks=batch.mask)
                out = None
            elif batch.depth is not None:
                out = engine.stylize_depth(content, [d.to(dev, torch.float32) for d in batch.depth], depth_offset, depth_prominence)
            else:
                out = engine.stylize(content, alpha)
            if out is not None:
                if isinstance(batch.mask, list):       # masks of different sizes inside one sub-batch: composite frame by frame
                    out = torch.cat([engine.composite(content[k:k + 1], out[k:k + 1], m.to(dev).float().unsqueeze(0))
                                     for k, m in enumerate(batch.mask)])
                elif batch.mask is not None:
                    out = engine.composite(content, out, batch.mask.to(dev).float())
                u8 = engine.to_u8(out)
            feeder.release(batch)
            if on_gpu:
                e = torch.cuda.Event()
                e.record(torch.cuda.current_stream(dev))
                queued.append(e)
            if post is not None:
                u8 = post(u8)
            shapes.add(tuple(u8.shape[1:]))
            if sink is not None:
                sink(i, j, u8)
            if copier is not None and not gathering:
                copier.copy(host_out[i:j], u8)
            if gather or sink is None:
                blocks.append((i, j, u8))
            if chunks > 1:
                if tuple(u8.shape[1:3]) != tuple(out_hw):
                    raise ValueError(f"stylize_frames_sharded: out_hw {tuple(out_hw)} but a finished frame is {tuple(u8.shape[1:3])}")
                while next_chunk < chunks - 1 and chunk_ready(next_chunk, j):
                    issue_chunk(next_chunk)
                    next_chunk += 1
    except Exception as e:                 # agreed on below: every rank raises, none is left waiting in the gather
        err = e
    finally:
        feeder.close()
    if err is not None:                    # a failed job: its kernels may still read the feeder's slot buffers - let them finish before
        try:                               # the buffers go back to the allocator on the way out
            engine.synchronize()
        except Exception:
            pass
    if chunks > 1:
        # every rank issues every in-loop piece, whatever happened to its block: the collective sequence stays identical
        # (pieces 0 .. k-2, status word, last piece); a failed rank sends zeros and the status word raises everywhere
        try:
            while next_chunk < chunks - 1:
                issue_chunk(next_chunk, zeros=err is not None)
                next_chunk += 1
        except Exception as e:
            err = err or e
    info["h2d_bytes"] = feeder.h2d_bytes
    info["fetch_s"] = feeder.fetch_s
    info["feeder"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in feeder.stats.items()}
    m1 = _mark(engine)
    info["enqueue_s"] = time.perf_counter() - t_host0      # host WALL time to fetch, upload and launch the whole block (includes
                                                           # time blocked on a full HIP queue or an empty feeder queue)
    # CPU time, which is what eight ranks sharing one host compete for: of the launching thread alone (kernel launches through
    # the C ABI, tensor bookkeeping) and of the whole process (+ feeder, fetch pool, sink writers, the HIP runtime's own threads)
    info["host_cpu_s"] = time.thread_time() - cpu0_thread
    info["process_cpu_s"] = time.process_time() - cpu0_proc
    info["abi_calls"] = getattr(engine, "abi_calls", lambda: 0)() - abi0

    def finish_times(m2=None):
        if on_gpu:                          # sleep until the device has finished this job's work, then the (now immediate) synchronisations
            e = torch.cuda.Event()
            e.record(torch.cuda.current_stream(dev))
            sleep_wait(e)
        if copier is not None:
            copier.finish()
            info["d2h_bytes"] = copier.bytes
        engine.synchronize()
        info["compute_s"] = _elapsed(engine, m0, m1)
        info["gather_s"] = _elapsed(engine, m1, m2) if m2 is not None else 0.0

    if not gathering:
        if err is not None:
            raise err
        finish_times()
        if sink is not None and not gather:
            return None, info
        if len(shapes) > 1:
            if gather:
                raise ValueError("stylize_frames_sharded: the gather needs one frame size over the whole job")
            return [u8 for (_, _, u8) in blocks], info
        if blocks:
            local = torch.cat([u8 for (_, _, u8) in blocks]) if len(blocks) > 1 else blocks[0][2]
        else:
            geom = tuple(out_hw) + (3,) if out_hw is not None else (0, 0, 3)
            local = torch.empty((0,) + geom, dtype=torch.uint8, device=dev)
        return local, info

    # ---- the job's status word, then its gather ------------------------------------------------------------------------------------
    geom = next(iter(shapes)) if len(shapes) == 1 else None
    def drop_pending():
        """An abandoned chunked job: the pieces already issued are complete collectives (every rank took part) - wait for them
        so that nothing of this job is left in flight on the communicator, and drop the data."""
        while pending:
            try:
                pending.pop(0)[2]()
            except Exception:
                pass

    if agree:
        ok, geom_all, uniform = sh.agree_geometry(err is None, shapes, group, dev)
        if err is not None or not ok:
            drop_pending()
        if err is not None:
            raise err
        if not ok:
            raise RuntimeError("stylize_frames_sharded: another rank failed; job abandoned before the gather")
        if not uniform:
            raise ValueError("stylize_frames_sharded: the gather needs one frame size on every rank")
        if geom_all is not None and out_hw is not None and tuple(out_hw) != tuple(geom_all[:2]):
            raise ValueError(f"stylize_frames_sharded: out_hw {tuple(out_hw)} but the finished frames are {tuple(geom_all[:2])}")
        geom = geom_all
    else:
        if err is not None:
            drop_pending()
            raise err
        if len(shapes) > 1:
            raise ValueError("stylize_frames_sharded: the gather needs one frame size over the whole job")
        if geom is None and out_hw is not None:
            geom = tuple(out_hw) + (3,)
    if geom is None:                       # an empty job: nothing to gather anywhere
        finish_times()
        return (torch.empty((0, 0, 0, 3), dtype=torch.uint8, device=dev) if rank == dst else None), info
    if chunks == 1:
        if blocks:
            local = torch.cat([u8 for (_, _, u8) in blocks]) if len(blocks) > 1 else blocks[0][2]
        else:
            local = torch.empty((0,) + tuple(geom), dtype=torch.uint8, device=dev)
        info["gathers"] += 1
        out = sh.gather_frames(local, n, dst=dst, group=group)
        if copier is not None and rank == dst:
            copier.copy(host_out, out)
    else:
        while next_chunk < chunks:
            issue_chunk(next_chunk)
            next_chunk += 1
        land_pending()
        out = None
        if rank == dst:
            out = torch.empty((n,) + tuple(geom), dtype=torch.uint8, device=dev)
            for first, blk in landed:
                out[first:first + blk.shape[0]].copy_(blk)
    m2 = _mark(engine)
    finish_times(m2)
    return out, info


def video_style_transfer_sharded(engine, frames, styles, *, flows=None, target_resolution=None, blend_alpha=0.7, depth_maps=None,
                                 offset=0.30, prominence=20, alpha=0.5, sub_batch=None, group=None, dst=0, require_transport=None,
                                 gather_chunks=1):
    """The video caller (reference video/utils.py:297-369) over a frame list: per-frame AdaIN sharded over the ranks (styles
    switching through the clip as ``style_schedule`` says when several are given; depth-aware when ``depth_maps`` are
    given, which is how the reference runs it: ``use_depth=True``, offset 0.30, prominence 20), ``cv2.resize(...,
    target_resolution, INTER_AREA)`` of every stylised frame on its own rank (:352-353; ``target_resolution`` =
    (width, height)), ONE gather, then on ``dst`` the recurrence ``frame_i = blend(frame_i, warp(result_{i-1}, flow_{i-1}),
    blend_alpha)`` (:355-368) over ``flows`` [n-1,2,H,W] (prev -> current, estimated by the caller: the optical-flow
    estimator is OpenCV's and stays outside).  Returns ``(frames_u8 on dst | None, info)``."""
    n = len(frames)
    style_list = list(styles) if isinstance(styles, (list, tuple)) else [styles]
    post = out_hw = None
    if target_resolution is not None:
        post = lambda u8: engine.resize_area_u8(u8, target_resolution)
        out_hw = (int(target_resolution[1]), int(target_resolution[0]))
    out, info = stylize_frames_sharded(engine, frames, style_list, style_of=style_schedule(n, len(style_list)), alpha=alpha,
                                       depth_maps=depth_maps, depth_offset=offset, depth_prominence=prominence, post=post,
                                       sub_batch=sub_batch, group=group, dst=dst, require_transport=require_transport,
                                       out_hw=out_hw, gather_chunks=gather_chunks)
    if out is not None and flows is not None and n > 1:
        t0 = time.perf_counter()
        out = engine.temporal_blend(out, flows.to(out.device, torch.float32), blend_alpha)
        engine.synchronize()
        info["temporal_blend_s"] = time.perf_counter() - t0
    return out, info


def precompute_guides_sharded(engine, views, names, output_dir, style, *, masks=None, content_size=512, crop=False, alpha=0.5,
                              depth_maps=None, depth_offset=0.5, depth_prominence=20, save_ext=".jpg", sub_batch=None, group=None,
                              dst=0, write="dst", require_transport=None, writers=4):
    """The guide-image precompute of the reference's Style_3DGS/train.py:86-115 over all training views, sharded: every view
    is resized as ``adain_inference(content_size=...)`` resizes it (test.py:190-200), stylised, composited with its mask
    (``gt_image_np > 0``, train.py:97) and saved as ``<output_dir>/<name><save_ext>`` — the reference's naming, so the guide
    loss (train.py:208-221) reads the files back unchanged.  ``write="dst"``: the uint8 views are gathered and rank ``dst``
    writes every file (views must then share one size); ``write="local"``: every rank writes its own block as it is
    finished — nothing is gathered and the views may have any mix of sizes.  The views are decoded / resized on a worker
    thread ahead of the kernels and travel to the device as uint8; the files are encoded and written by ``writers`` threads
    behind them.  Every rank returns the full {name: Path} map once all files exist (an error on any rank raises on all)."""
    from PIL import Image

    from .AdaIN.test import device_transform_u8, test_transform_u8

    if write not in ("dst", "local"):
        raise ValueError("write must be 'dst' or 'local'")
    rank, world = _rank_world(group)
    out_dir = Path(output_dir)
    out_dir.mkdir(exist_ok=True, parents=True)
    tf = test_transform_u8(content_size, crop)

    class _Views:                         # lazy: a rank only opens and resizes the views of its own block
        def __len__(self):
            return len(views)

        def __getitem__(self, k):
            v = views[k]
            if isinstance(v, (str, Path)):
                v = Image.open(str(v))
            if isinstance(v, (torch.Tensor, np.ndarray)):
                return v
            if on_gpu:                    # RGB: the decoded bytes go up, Resize [+ CenterCrop] run on the device (PIL's bytes exactly)
                d = device_transform_u8(v, content_size, crop, engine.device)
                if d is not None:
                    return d[0]
            return tf(v)

    on_gpu = torch.device(engine.device).type == "cuda"
    names = list(names)
    paths = {nm: out_dir / f"{nm}{save_ext}" for nm in names}
    sink = FileSink(engine.device, workers=writers)
    err = None
    info = {}
    try:
        if write == "local":
            _, info = stylize_frames_sharded(engine, _Views(), style, alpha=alpha, depth_maps=depth_maps, depth_offset=depth_offset,
                                             depth_prominence=depth_prominence, masks=masks, sub_batch=sub_batch, group=group, dst=dst,
                                             gather=False, sink=lambda i, j, u8: sink.write(u8, [paths[names[k]] for k in range(i, j)]))
        else:
            u8, info = stylize_frames_sharded(engine, _Views(), style, alpha=alpha, depth_maps=depth_maps, depth_offset=depth_offset,
                                              depth_prominence=depth_prominence, masks=masks, sub_batch=sub_batch, group=group, dst=dst,
                                              gather=True, require_transport=require_transport)
            if rank == dst and len(names):
                step = max(1, sub_batch or auto_sub_batch(*u8.shape[1:3]))
                for a in range(0, len(names), step):
                    sink.write(u8[a:a + step], [paths[nm] for nm in names[a:a + step]])
    except Exception as e:
        err = e
    t0 = time.perf_counter()
    try:
        sink.close()
    except Exception as e:
        err = err or e
    info["write_s"] = time.perf_counter() - t0
    info["d2h_bytes"] = sink.d2h_bytes
    # every file exists when any rank returns - or every rank raises
    all_ok = sh.agree(err is None, group, engine.device)
    if err is not None:
        raise err
    if not all_ok:
        raise RuntimeError("precompute_guides_sharded: another rank failed; the guide set is incomplete")
    return paths, info


# ---------------------------------------------------------------------------------------------------------------------------------
# timed job loop (bench.py --job; driven on CPU by tests/test_distributed_gloo.py)
# ---------------------------------------------------------------------------------------------------------------------------------
def run_timed_jobs(job, steps, warmup, *, barrier, group=None):
    """bench.py's contract around whole jobs: ``warmup`` untimed jobs, ``barrier()``, EXACTLY ``steps`` jobs, ``barrier()``;
    returns (seconds as the MAX over the ranks, the last job's result, per-rank info of the last job).  ``job()`` runs one
    whole sharded job (``stylize_frames_sharded`` ...) and returns ``(result, info)``; ``barrier()`` must leave this rank's
    device idle and every other rank's too."""
    res = info = None
    for _ in range(warmup):
        res, info = job()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        res, info = job()
    barrier()
    dt = time.perf_counter() - t0
    if isinstance(info, dict):
        info["t0"], info["t1"] = t0, t0 + dt          # host clock at both ends of the timed region (bench.py's telemetry window)
    if sh.dist_on() and dist.get_world_size(group) > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        if "cpu" not in sh.backend_table(group):
            t = t.to(info["device"]) if info and "device" in info else t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        dt = float(t.item())
    return dt, res, info


# ---------------------------------------------------------------------------------------------------------------------------------
# timed per-step loop (bench.py's default mode with more than one rank; driven on CPU by tests/test_distributed_gloo.py)
# ---------------------------------------------------------------------------------------------------------------------------------
GATHER_MODES = ("end", "overlap")
MAX_END_GATHER_BYTES = 64 << 30        # what rank 0 may be asked to hold for one end-of-region gather


def run_timed_steps(step, steps, warmup, *, barrier, block_shape, device, mode="end", gather=True, group=None, dst=0,
                    mark=None, elapsed=None):
    """bench.py's contract around single steps: (with several ranks: ``barrier()``,) ``warmup`` untimed steps, ``barrier()``, EXACTLY
    ``steps`` steps + the path's one collective, ``barrier()``.  Returns ``(seconds as the MAX over the ranks, gathered frames on dst | None, info)``.

    ``step(out)`` enqueues one pass of the hot path over this rank's ``b`` resident frames and leaves the finished uint8 frames
    in ``out`` [b,H,W,3] (``block_shape`` = (b, H, W, 3)) - or returns another tensor of that shape if it cannot write in place.
    Every step is a job of ``world * b`` frames cut into contiguous per-rank blocks (weak scaling); its frames meet on ``dst``:

    mode "end"      (default) every rank keeps the frames of its ``steps`` steps in HBM ([steps * b, H, W, 3] uint8) and the timed
                    region ENDS with ONE gather of the whole block - the region is one ``steps * world * b``-frame job in frame
                    order, and no transport kernel ever runs beside the compute kernels (which fill every CU and cannot
                    rebalance around a late start: tools/probes/notes/dynamic_tile_scheduling.md).
    mode "overlap"  one asynchronous gather per step, at most two in flight: step k's gather overlaps step k+1's kernels.

    ``gather=False`` (or a single process) runs the steps alone.  ``mark()`` / ``elapsed(a, b)`` are the engine's stream time stamps
    (HIP events); the defaults use the host clock (CPU tests).  ``info`` = {"mode", "gathers", "compute_ms", "gather_ms",
    "gather_bytes"}: compute_ms from the first step to the last kernel of the last step, gather_ms from there to the arrival of
    the last gathered frame on this rank's stream (mode "overlap": the gathers of all but the last two steps lie inside compute_ms)."""
    if mode not in GATHER_MODES:
        raise ValueError(f"gather mode must be one of {GATHER_MODES}, got {mode!r}")
    if steps < 1 or warmup < 0:
        raise ValueError(f"run_timed_steps: steps must be >= 1 and warmup >= 0, got {steps} / {warmup}")
    rank, world = _rank_world(group)
    gathering = bool(gather) and sh.dist_on()
    b = int(block_shape[0])
    frame = tuple(int(v) for v in block_shape[1:])
    mark = mark or time.perf_counter
    elapsed = elapsed or (lambda a_, b_: b_ - a_)
    info = {"mode": mode if gathering else None, "gathers": 0, "gather_bytes": 0}
    n_end = steps * b
    if gathering and mode == "end" and n_end * world * int(np.prod(frame)) > MAX_END_GATHER_BYTES:
        raise ValueError(f"--gather end would collect {n_end * world * int(np.prod(frame)) / 2**30:.1f} GiB on rank {dst}: "
                         "use fewer steps or the overlapped gather")
    keep = torch.empty((max(n_end, b),) + frame, dtype=torch.uint8, device=device) if gathering and mode == "end" else None
    # without a kept block a step writes into one of three rotating buffers: at most two gathers are in flight, so the buffer a
    # step overwrites is the one whose gather was waited for (on the stream) a step ago
    scratch = [torch.empty((b,) + frame, dtype=torch.uint8, device=device) for _ in range(3)] if keep is None else None
    pending, got = [], [None]

    def one(k):
        slot = keep[(k % steps) * b:(k % steps + 1) * b] if keep is not None else scratch[k % 3]
        out = step(slot)
        if out is not None and out is not slot:
            if tuple(out.shape) != tuple(slot.shape):
                raise ValueError(f"step returned {tuple(out.shape)}, expected {tuple(slot.shape)}")
            slot.copy_(out)
        if gathering and mode == "overlap":
            info["gathers"] += 1
            info["gather_bytes"] += slot.numel()
            pending.append(sh.gather_frames(slot, world * b, dst=dst, group=group, async_op=True, counts=[b] * world))
            if len(pending) > 2:
                got[0] = pending.pop(0)()

    def finish():
        while pending:
            got[0] = pending.pop(0)()
        if gathering and mode == "end":
            info["gathers"] += 1
            info["gather_bytes"] += n_end * int(np.prod(frame))
            got[0] = sh.gather_frames(keep[:n_end], world * n_end, dst=dst, group=group, counts=[n_end] * world)

    # warm-up, all outside the timing: first - once - the end-of-region collective in the very shape the timed region issues it
    # (communicator set-up, the receive buffer's first allocation), THEN the steps, so that nothing but the barrier lies between
    # the last warm-up step and the first timed one (the chip loses its clock within milliseconds of idling: DESIGN.md section 6)
    if gathering and mode == "end":
        finish()
    if sh.dist_on():
        barrier()                         # the ranks start their warm-up steps together, so none idles long at the barrier behind them
    for k in range(warmup):
        one(k)
    while pending:                        # mode "overlap": the warm-up steps' own gathers
        got[0] = pending.pop(0)()
    got[0] = None
    info["gathers"], info["gather_bytes"] = 0, 0
    barrier()
    t0 = time.perf_counter()
    m0 = mark()
    for k in range(steps):
        one(k)
    m1 = mark()
    finish()
    m2 = mark()
    barrier()
    dt = time.perf_counter() - t0
    info["t0"], info["t1"] = t0, t0 + dt          # host clock (time.perf_counter) at both ends of the timed region
    info["compute_ms"] = elapsed(m0, m1) * 1e3
    info["gather_ms"] = elapsed(m1, m2) * 1e3
    info["local_s"] = dt
    if sh.dist_on() and dist.get_world_size(group) > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        if "cpu" not in sh.backend_table(group):
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        dt = float(t.item())
    return dt, got[0], info
