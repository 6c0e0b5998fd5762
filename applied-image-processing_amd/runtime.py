"""ctypes binding of ``libadain_hip.so`` (the C ABI declared in include/adain_hip.h).

PyTorch is used only for device memory, streams and (elsewhere) ``torch.distributed``: every
compute call below hands raw device pointers (``tensor.data_ptr()``) and the current HIP stream to a
hand-written gfx950 kernel.  There is NO fallback: if the shared library is missing or the tensors
are not on a GPU, these functions raise.
"""
import ctypes
import os
import threading

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
# The product library.  Nothing in the environment can swap it: the diagnostic build (libadain_hip_diag.so, build.py --diag) is
# loaded only by an explicit ``use_library(DIAG_LIB_PATH)`` call (tools/_diag.py).
LIB_PATH = os.path.join(_PKG, "libadain_hip.so")
DIAG_LIB_PATH = os.path.join(_PKG, "libadain_hip_diag.so")

SRC_DIRECT, SRC_UP2X, SRC_POOL2 = 0, 1, 2
SCHEDULE_BATCH, SCHEDULE_LATENCY = 0, 1      # ADAIN_SCHEDULE_*: see adain_set_schedule in include/adain_hip.h
ABI_VERSION = 4          # ADAIN_ABI_VERSION of include/adain_hip.h this binding was written against

_c_int, _c_float, _c_size_t, _c_void_p = ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p
_PP = ctypes.POINTER(ctypes.c_void_p)

# name -> (restype, argtypes); mirrors include/adain_hip.h one to one
SIGNATURES = {
    "adain_abi_version": (_c_int, []),
    "adain_last_error": (ctypes.c_char_p, []),
    "adain_set_schedule": (_c_int, [_c_int]),
    "adain_get_schedule": (_c_int, []),
    "adain_encoder_packed_floats": (_c_size_t, []),
    "adain_decoder_packed_floats": (_c_size_t, []),
    "adain_encoder_pack": (_c_int, [_PP, _PP, _c_void_p, _c_void_p]),
    "adain_decoder_pack": (_c_int, [_PP, _PP, _c_void_p, _c_void_p]),
    "adain_encoded_size": (None, [_c_int, _c_int, ctypes.POINTER(_c_int), ctypes.POINTER(_c_int)]),
    "adain_encode_workspace_bytes": (_c_size_t, [_c_int, _c_int, _c_int]),
    "adain_encode": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_size_t, _c_int, _c_int, _c_int, _PP, _c_void_p]),
    "adain_encode_u8": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_size_t, _c_int, _c_int, _c_int, _PP, _c_void_p]),
    "adain_encode_relu1_1": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p]),
    "adain_encode_multi_workspace_bytes": (_c_size_t, [_c_int, ctypes.POINTER(_c_int), ctypes.POINTER(_c_int), ctypes.POINTER(_c_int)]),
    "adain_encode_multi": (_c_int, [_c_int, _PP, _PP, ctypes.POINTER(_c_int), ctypes.POINTER(_c_int), ctypes.POINTER(_c_int), _c_void_p,
                                    _c_void_p, _c_size_t, _PP, _c_void_p]),
    "adain_decode_workspace_bytes": (_c_size_t, [_c_int, _c_int, _c_int]),
    "adain_decode": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_size_t, _c_int, _c_int, _c_int, _PP, _c_void_p]),
    "adain_mean_std_workspace_bytes": (_c_size_t, [_c_int, _c_int, _c_int, _c_int]),
    "adain_mean_std": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_int, _c_float, _c_void_p, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "adain_blend_alpha": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int,
                                   _c_float, _c_float, _c_void_p, _c_void_p]),
    "adain_blend_pmap": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int,
                                  _c_void_p, _c_int, _c_void_p, _c_void_p]),
    "adain_strength_map_workspace_bytes": (_c_size_t, [_c_int, _c_int]),
    "adain_strength_map": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_int, _c_float, _c_float, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "adain_resize_bilinear": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_void_p]),
    "adain_resize_nearest": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_void_p]),
    "adain_mask_composite": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_void_p, _c_int, _c_int, _c_int, _c_void_p]),
    "adain_quantize_u8": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p]),
    "adain_u8_to_f32": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p]),
    "adain_warp_blend_u8": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_float, _c_float, _c_void_p]),
    "adain_resize_area_u8": (_c_int, [_c_void_p, _c_void_p] + [_c_int] * 6 + [_c_void_p]),
    "adain_resize_pil_bilinear_u8_workspace_bytes": (_c_size_t, [_c_int] * 4),
    "adain_resize_pil_bilinear_u8": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p] + [_c_int] * 6 + [_c_void_p, _c_size_t, _c_void_p]),
    "adain_stylize_u8_workspace_bytes": (_c_size_t, [_c_int] * 9),
    "adain_stylize_u8_out_size": (None, [_c_int, _c_int, _c_int, ctypes.POINTER(_c_int), ctypes.POINTER(_c_int)]),
    "adain_stylize_u8": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_float, _c_float, _PP,
                                  ctypes.POINTER(_c_int), ctypes.POINTER(_c_int), _c_float, _c_float, _c_void_p, _c_int, _c_int, _c_int, _c_int,
                                  _c_int, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "adain_nhwc_to_nchw": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p]),
    "adain_nchw_to_nhwc": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p]),
    "adain_conv3x3_wino4_packed_floats": (_c_size_t, [_c_int, _c_int]),
    "adain_conv3x3_wino4_pack": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_void_p]),
    "adain_conv3x3_wino": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p] + [_c_int] * 11 + [_c_void_p]),
    "adain_conv3x3_wino4_split_workspace_bytes": (_c_size_t, [_c_int] * 5),
    "adain_conv3x3_wino4_split": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p] + [_c_int] * 10 + [_c_void_p, _c_size_t, _c_void_p]),
}

# entry points of include/adain_hip_diag.h: exported by the diagnostic library only
DIAG_SIGNATURES = {
    "adain_debug_set_conv_stamp_buffer": (_c_int, [_c_void_p]),
}

_lib = None
_lock = threading.Lock()


class AdainHipError(RuntimeError):
    pass


def lib():
    """Loads the shared library once.  Raises if it has not been built (``__graft_entry__.build()``)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise AdainHipError(
                        f"{LIB_PATH} is missing: build it with `python applied-image-processing_amd/build.py` "
                        "(there is no CPU / PyTorch fallback for the AdaIN path)"
                    )
                l = ctypes.CDLL(LIB_PATH)
                sigs = dict(SIGNATURES)
                if hasattr(l, "adain_debug_set_conv_stamp_buffer"):      # the diagnostic build: both headers
                    sigs.update(DIAG_SIGNATURES)
                for name, (res, args) in sigs.items():
                    f = getattr(l, name)
                    f.restype, f.argtypes = res, args
                if l.adain_abi_version() != ABI_VERSION:
                    raise AdainHipError(f"{LIB_PATH}: ABI version {l.adain_abi_version()}, this binding needs {ABI_VERSION} "
                                        "(rebuild: python applied-image-processing_amd/build.py)")
                _lib = l
    return _lib


def use_library(path):
    """Switches the process to another build of the library (the diagnostic build, ``DIAG_LIB_PATH``; ``LIB_PATH`` switches
    back).  Only tools/ call this: the product path always runs ``LIB_PATH``."""
    global _lib, LIB_PATH
    with _lock:
        _lib, LIB_PATH = None, path
    return lib()


def is_diag():
    return hasattr(lib(), "adain_debug_set_conv_stamp_buffer")


ABI_CALLS = [0]          # compute calls made through the C ABI by this process (jobs report it per frame)


def _check(rc, what):
    ABI_CALLS[0] += 1
    if rc != 0:
        raise AdainHipError(f"{what} failed ({rc}): {lib().adain_last_error().decode()}")


class schedule:
    """``with runtime.schedule(runtime.SCHEDULE_LATENCY): ...`` - the launch schedule of the calling thread for the C-ABI calls
    inside (adain_set_schedule: under LATENCY a generic 3x3 layer whose launch leaves compute units without a tile is split along
    cin; results then differ from the batch schedule's in the last bits and depend on the launch's batch size).  Restored on exit."""

    def __init__(self, value):
        self.value = int(value)

    def __enter__(self):
        self.prev = set_schedule(self.value)
        return self

    def __exit__(self, *exc):
        set_schedule(self.prev)
        return False


def set_schedule(value):
    """Sets the calling thread's schedule; returns the previous one."""
    prev = lib().adain_set_schedule(int(value))
    if prev < 0:
        raise AdainHipError(f"adain_set_schedule failed ({prev}): {lib().adain_last_error().decode()}")
    return prev


def get_schedule():
    return lib().adain_get_schedule()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise AdainHipError(f"{name}: expected a GPU tensor (the AdaIN path has no CPU fallback)")
    if t.dtype != dtype:
        raise AdainHipError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return t.contiguous()


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return ctypes.cast(arr, _PP), arr


# --- workspaces: one growing scratch buffer per (device, stream, tag) -----------------------------------------
# (keyed by stream so that independent frames can be in flight on different HIP streams)
_ws = {}


def workspace(device, tag, nbytes):
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream, tag)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = None
        _ws.pop(key, None)
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


def free_workspaces():
    _ws.clear()


# --- weights ------------------------------------------------------------------------------------------------
ENC_KEYS = [0, 2, 5, 9, 12, 16, 19, 22, 25, 29]
DEC_KEYS = [1, 5, 8, 11, 14, 18, 21, 25, 28]


def pack_encoder(state_dict, device):
    """state_dict with reference keys ("0.weight", "2.weight", ... ) -> packed device buffer."""
    ws = [_dev(state_dict[f"{k}.weight"].to(device=device, dtype=torch.float32), "weight") for k in ENC_KEYS]
    bs = [_dev(state_dict[f"{k}.bias"].to(device=device, dtype=torch.float32), "bias") for k in ENC_KEYS]
    packed = torch.zeros(lib().adain_encoder_packed_floats(), dtype=torch.float32, device=device)
    wp, _k1 = _ptr_array(ws)
    bp, _k2 = _ptr_array(bs)
    with torch.cuda.device(device):
        _check(lib().adain_encoder_pack(wp, bp, packed.data_ptr(), _stream()), "adain_encoder_pack")
        torch.cuda.current_stream().synchronize()   # the source tensors may be freed after return
    return packed


def pack_decoder(state_dict, device):
    ws = [_dev(state_dict[f"{k}.weight"].to(device=device, dtype=torch.float32), "weight") for k in DEC_KEYS]
    bs = [_dev(state_dict[f"{k}.bias"].to(device=device, dtype=torch.float32), "bias") for k in DEC_KEYS]
    packed = torch.zeros(lib().adain_decoder_packed_floats(), dtype=torch.float32, device=device)
    wp, _k1 = _ptr_array(ws)
    bp, _k2 = _ptr_array(bs)
    with torch.cuda.device(device):
        _check(lib().adain_decoder_pack(wp, bp, packed.data_ptr(), _stream()), "adain_decoder_pack")
        torch.cuda.current_stream().synchronize()
    return packed


# --- encoder / decoder -----------------------------------------------------------------------------------
def encoded_size(h, w):
    hc, wc = ctypes.c_int(), ctypes.c_int()
    lib().adain_encoded_size(h, w, ctypes.byref(hc), ctypes.byref(wc))
    return hc.value, wc.value


def _event_array(events):
    if events is None:
        return None, None
    arr = (ctypes.c_void_p * len(events))(*[e.cuda_event for e in events])
    return ctypes.cast(arr, _PP), arr


def encode(image, packed, events=None):
    """image NCHW [n,3,h,w] -> relu4_1 features NHWC [n,hc,wc,512]."""
    image = _dev(image, "image")
    if image.dim() != 4 or image.shape[1] != 3:
        raise AdainHipError(f"encode: expected [n,3,h,w], got {tuple(image.shape)}")
    n, _, h, w = image.shape
    hc, wc = encoded_size(h, w)
    feat = torch.empty((n, hc, wc, 512), dtype=torch.float32, device=image.device)
    nbytes = lib().adain_encode_workspace_bytes(n, h, w)
    ws = workspace(image.device, "conv", nbytes)
    ev, _keep = _event_array(events)
    with torch.cuda.device(image.device):
        _check(lib().adain_encode(image.data_ptr(), feat.data_ptr(), packed.data_ptr(), ws.data_ptr(), ws.numel(), n, h, w, ev,
                                  _stream()), "adain_encode")
    return feat


def encode_u8(frames_u8, packed, events=None):
    """Decoded frames HWC uint8 [n,h,w,3] -> relu4_1 features NHWC [n,hc,wc,512]; ToTensor (v / 255) happens inside the first
    layer's kernel: bit-identical to ``encode(u8_to_f32(frames_u8))``."""
    x = _dev(frames_u8, "frames", torch.uint8)
    if x.dim() != 4 or x.shape[3] != 3:
        raise AdainHipError(f"encode_u8: expected uint8 [n,h,w,3], got {tuple(x.shape)}")
    n, h, w, _ = x.shape
    hc, wc = encoded_size(h, w)
    feat = torch.empty((n, hc, wc, 512), dtype=torch.float32, device=x.device)
    ws = workspace(x.device, "conv", lib().adain_encode_workspace_bytes(n, h, w))
    ev, _keep = _event_array(events)
    with torch.cuda.device(x.device):
        _check(lib().adain_encode_u8(x.data_ptr(), feat.data_ptr(), packed.data_ptr(), ws.data_ptr(), ws.numel(), n, h, w, ev,
                                     _stream()), "adain_encode_u8")
    return feat


def encode_relu1_1(image, packed):
    """vgg[:4] (conv0 -> pad -> conv1_1 -> relu, net.py:39-42) as the one folded layer the encoder starts with: image NCHW float
    [n,3,h,w] or decoded uint8 frames [n,h,w,3] -> relu1_1 NHWC [n,h,w,64]."""
    u8 = isinstance(image, torch.Tensor) and image.dtype == torch.uint8
    x = _dev(image, "image", torch.uint8 if u8 else torch.float32)
    if x.dim() != 4 or (x.shape[3] if u8 else x.shape[1]) != 3:
        raise AdainHipError(f"encode_relu1_1: expected float [n,3,h,w] or uint8 [n,h,w,3], got {tuple(x.shape)}")
    n, h, w = (x.shape[0], x.shape[1], x.shape[2]) if u8 else (x.shape[0], x.shape[2], x.shape[3])
    out = torch.empty((n, h, w, 64), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib().adain_encode_relu1_1(x.data_ptr(), 1 if u8 else 0, out.data_ptr(), packed.data_ptr(), n, h, w, _stream()), "adain_encode_relu1_1")
    return out


def encode_multi(images, packed, events=None):
    """Several image batches [n_i,3,h_i,w_i] of different sizes through the encoder in one pass (the content batch and the
    style image of one style_transfer call) -> list of relu4_1 features NHWC; bit-identical to ``encode`` per batch."""
    images = [_dev(x, "image") for x in images]
    for x in images:
        if x.dim() != 4 or x.shape[1] != 3:
            raise AdainHipError(f"encode: expected [n,3,h,w], got {tuple(x.shape)}")
        if x.device != images[0].device:
            raise AdainHipError("encode_multi: all image batches must be on one device")
    k = len(images)
    IntArr = _c_int * k
    n, h, w = IntArr(*[x.shape[0] for x in images]), IntArr(*[x.shape[2] for x in images]), IntArr(*[x.shape[3] for x in images])
    feats = []
    for x in images:
        hc, wc = encoded_size(x.shape[2], x.shape[3])
        feats.append(torch.empty((x.shape[0], hc, wc, 512), dtype=torch.float32, device=x.device))
    nbytes = lib().adain_encode_multi_workspace_bytes(k, n, h, w)
    if nbytes == 0:
        raise AdainHipError(f"encode_multi: 1..4 image batches per call, got {k}")
    ws = workspace(images[0].device, "conv", nbytes)
    ip, _k1 = _ptr_array(images)
    fp, _k2 = _ptr_array(feats)
    ev, _keep = _event_array(events)
    with torch.cuda.device(images[0].device):
        _check(lib().adain_encode_multi(k, ip, fp, n, h, w, packed.data_ptr(), ws.data_ptr(), ws.numel(), ev, _stream()), "adain_encode_multi")
    return feats


def decode(feat, packed, events=None):
    """features NHWC [n,hc,wc,512] -> image NCHW [n,3,8hc,8wc]."""
    feat = _dev(feat, "feat")
    if feat.dim() != 4 or feat.shape[3] != 512:
        raise AdainHipError(f"decode: expected NHWC [n,hc,wc,512], got {tuple(feat.shape)}")
    n, hc, wc, _ = feat.shape
    img = torch.empty((n, 3, 8 * hc, 8 * wc), dtype=torch.float32, device=feat.device)
    nbytes = lib().adain_decode_workspace_bytes(n, hc, wc)
    ws = workspace(feat.device, "conv", nbytes)
    ev, _keep = _event_array(events)
    with torch.cuda.device(feat.device):
        _check(lib().adain_decode(feat.data_ptr(), img.data_ptr(), packed.data_ptr(), ws.data_ptr(), ws.numel(), n, hc, wc, ev,
                                  _stream()), "adain_decode")
    return img


# --- statistics and blend ------------------------------------------------------------------------------
def mean_std(feat, nhwc, eps=1e-5):
    """feat NHWC [n,h,w,c] (nhwc=True) or NCHW [n,c,h,w] -> (mean [n,c], std [n,c])."""
    feat = _dev(feat, "feat")
    assert feat.dim() == 4
    if nhwc:
        n, h, w, c = feat.shape
    else:
        n, c, h, w = feat.shape
    mean = torch.empty((n, c), dtype=torch.float32, device=feat.device)
    std = torch.empty_like(mean)
    nbytes = lib().adain_mean_std_workspace_bytes(int(nhwc), n, c, h * w)
    ws = workspace(feat.device, "stats", nbytes)
    with torch.cuda.device(feat.device):
        _check(lib().adain_mean_std(feat.data_ptr(), int(nhwc), n, c, h * w, eps, mean.data_ptr(), std.data_ptr(), ws.data_ptr(),
                                    ws.numel(), _stream()), "adain_mean_std")
    return mean, std


def _blend_dims(x, nhwc):
    if nhwc:
        n, h, w, c = x.shape
    else:
        n, c, h, w = x.shape
    return n, c, h * w


def blend_alpha(x, nhwc, c_mean, c_std, s_mean, s_std, alpha):
    """AdaIN(x) * alpha + x * (1 - alpha);  alpha = 1 gives plain adaptive_instance_normalization."""
    x = _dev(x, "content_feat")
    n, c, hw = _blend_dims(x, nhwc)
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _check(lib().adain_blend_alpha(x.data_ptr(), int(nhwc), n, c, hw, c_mean.data_ptr(), c_std.data_ptr(), s_mean.data_ptr(),
                                       s_std.data_ptr(), s_mean.shape[0], float(alpha), float(1 - alpha), out.data_ptr(), _stream()),
               "adain_blend_alpha")
    return out


def blend_pmap(x, nhwc, c_mean, c_std, s_mean, s_std, pmap):
    """AdaIN(x) * (1 - P) + x * P with P [pn, hc, wc] (pn in {1, n})."""
    x = _dev(x, "content_feat")
    pmap = _dev(pmap, "pmap")
    n, c, hw = _blend_dims(x, nhwc)
    pn = pmap.numel() // hw
    if pn * hw != pmap.numel():
        raise AdainHipError("blend_pmap: strength map size does not match the feature map")
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _check(lib().adain_blend_pmap(x.data_ptr(), int(nhwc), n, c, hw, c_mean.data_ptr(), c_std.data_ptr(), s_mean.data_ptr(),
                                      s_std.data_ptr(), s_mean.shape[0], pmap.data_ptr(), pn, out.data_ptr(), _stream()),
               "adain_blend_pmap")
    return out


def strength_map(depth, hc, wc, offset, prominence):
    """depth [h0,w0] -> P [1,1,hc,wc]."""
    depth = _dev(depth, "depth_map")
    if depth.dim() != 2:
        raise AdainHipError(f"strength_map: expected a 2-D depth map, got {tuple(depth.shape)}")
    h0, w0 = depth.shape
    p = torch.empty((1, 1, hc, wc), dtype=torch.float32, device=depth.device)
    ws = workspace(depth.device, "pmap", lib().adain_strength_map_workspace_bytes(hc, wc))
    with torch.cuda.device(depth.device):
        _check(lib().adain_strength_map(depth.data_ptr(), h0, w0, hc, wc, float(offset), float(prominence), p.data_ptr(),
                                        ws.data_ptr(), ws.numel(), _stream()), "adain_strength_map")
    return p


# --- pixel kernels -----------------------------------------------------------------------------------------
def _resize(fn, name, x, size):
    x = _dev(x, name)
    assert x.dim() == 4
    n, c, hi, wi = x.shape
    ho, wo = size
    out = torch.empty((n, c, ho, wo), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _check(fn(x.data_ptr(), out.data_ptr(), n * c, hi, wi, ho, wo, _stream()), name)
    return out


def resize_bilinear(x, size):
    return _resize(lib().adain_resize_bilinear, "adain_resize_bilinear", x, size)


def resize_nearest(x, size):
    return _resize(lib().adain_resize_nearest, "adain_resize_nearest", x, size)


def mask_composite(content, stylized, mask):
    """content, stylized NCHW [n,c,h,w]; mask [mn,mc,h,w] float -> content*(1-m) + stylized*m."""
    content, stylized, mask = _dev(content, "content"), _dev(stylized, "stylized"), _dev(mask, "mask")
    n, c, h, w = content.shape
    if stylized.shape != content.shape or mask.shape[-2:] != content.shape[-2:]:
        raise AdainHipError("mask_composite: shape mismatch")
    out = torch.empty_like(content)
    with torch.cuda.device(content.device):
        _check(lib().adain_mask_composite(content.data_ptr(), stylized.data_ptr(), mask.data_ptr(), mask.shape[1], mask.shape[0],
                                          out.data_ptr(), n, c, h * w, _stream()), "adain_mask_composite")
    return out


def quantize_u8(img, out=None):
    """NCHW float [n,c,h,w] -> NHWC uint8 [n,h,w,c] (x*255 + 0.5, clamp, truncate); ``out``: a contiguous uint8 [n,h,w,c] GPU
    tensor to write into (a slice of a job's frame block)."""
    img = _dev(img, "image")
    n, c, h, w = img.shape
    if out is None:
        out = torch.empty((n, h, w, c), dtype=torch.uint8, device=img.device)
    elif (not out.is_cuda or out.dtype != torch.uint8 or tuple(out.shape) != (n, h, w, c) or not out.is_contiguous()
          or out.device != img.device):
        raise AdainHipError(f"quantize_u8: out must be a contiguous uint8 {(n, h, w, c)} tensor on {img.device}")
    with torch.cuda.device(img.device):
        _check(lib().adain_quantize_u8(img.data_ptr(), out.data_ptr(), n, c, h, w, _stream()), "adain_quantize_u8")
    return out


def stylize_u8(frames_u8, enc_packed, dec_packed, s_mean, s_std, alpha=0.5, depth_maps=None, depth_offset=0.15, depth_prominence=20,
               mask=None, out=None):
    """One sub-batch of decoded frames through the whole path in ONE call of the C ABI (``adain_stylize_u8``: ToTensor + encoder,
    statistics, AdaIN blend - the alpha form, or the depth-aware form when ``depth_maps`` (one [h0,w0] float GPU tensor per frame)
    are given - decoder, mask composite, uint8 quantiser); the bytes the separate calls give.  frames_u8 uint8 [n,h,w,3]; s_mean /
    s_std [1,512]; mask [1|n, 1|3, hm, wm] uint8 / bool / float32 on the GPU.  Returns uint8 [n,oh,ow,3] (``out`` if given)."""
    x = _dev(frames_u8, "frames", torch.uint8)
    if x.dim() != 4 or x.shape[3] != 3:
        raise AdainHipError(f"stylize_u8: expected uint8 [n,h,w,3], got {tuple(x.shape)}")
    n, h, w, _ = x.shape
    dev = x.device
    s_mean, s_std = _dev(s_mean, "s_mean"), _dev(s_std, "s_std")
    if s_mean.numel() != 512 or s_std.numel() != 512:
        raise AdainHipError("stylize_u8: the style statistics must be [1,512] each (one style per call)")
    mn = mc = mh = mw = 0
    m_float, m_ptr = 0, None
    if mask is not None:
        if not isinstance(mask, torch.Tensor) or not mask.is_cuda or mask.dim() != 4:
            raise AdainHipError("stylize_u8: mask must be a GPU tensor [1|n, 1|3, hm, wm]")
        if mask.dtype == torch.float32:
            m_float = 1
        elif mask.dtype not in (torch.uint8, torch.bool):
            raise AdainHipError(f"stylize_u8: mask dtype {mask.dtype} (uint8, bool or float32)")
        mask = mask.contiguous()
        mn, mc, mh, mw = mask.shape
        m_ptr = mask.data_ptr()
    dp = dh = dw = None
    if depth_maps is not None:
        depth_maps = [_dev(d, "depth_map") for d in depth_maps]
        if len(depth_maps) != n or any(d.dim() != 2 for d in depth_maps):
            raise AdainHipError(f"stylize_u8: need {n} depth maps [h0,w0], one per frame")
        dp, _keep = _ptr_array(depth_maps)
        dh = (_c_int * n)(*[d.shape[0] for d in depth_maps])
        dw = (_c_int * n)(*[d.shape[1] for d in depth_maps])
    oh, ow = ctypes.c_int(), ctypes.c_int()
    L = lib()
    L.adain_stylize_u8_out_size(h, w, int(mask is not None), ctypes.byref(oh), ctypes.byref(ow))
    shape = (n, oh.value, ow.value, 3)
    if out is None:
        out = torch.empty(shape, dtype=torch.uint8, device=dev)
    elif not out.is_cuda or out.dtype != torch.uint8 or tuple(out.shape) != shape or not out.is_contiguous() or out.device != dev:
        raise AdainHipError(f"stylize_u8: out must be a contiguous uint8 {shape} tensor on {dev}")
    nbytes = L.adain_stylize_u8_workspace_bytes(n, h, w, int(depth_maps is not None), mn, mc, mh, mw, m_float)
    ws = workspace(dev, "stylize", nbytes)
    with torch.cuda.device(dev):
        _check(L.adain_stylize_u8(x.data_ptr(), n, h, w, enc_packed.data_ptr(), dec_packed.data_ptr(), s_mean.data_ptr(), s_std.data_ptr(),
                                  float(alpha), float(1 - alpha), dp, dh, dw, float(depth_offset), float(depth_prominence), m_ptr, m_float,
                                  mn, mc, mh, mw, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "adain_stylize_u8")
    return out


def u8_to_f32(frames_u8):
    """torchvision ToTensor on the device: NHWC uint8 [n,h,w,c] -> NCHW float [n,c,h,w] = v / 255 (bit for bit the host's)."""
    x = _dev(frames_u8, "frames", torch.uint8)
    if x.dim() != 4:
        raise AdainHipError(f"u8_to_f32: expected uint8 [n,h,w,c], got {tuple(x.shape)}")
    n, h, w, c = x.shape
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib().adain_u8_to_f32(x.data_ptr(), out.data_ptr(), n, c, h, w, _stream()), "adain_u8_to_f32")
    return out


def warp_blend_u8(cur, prev, flow, alpha, out=None):
    """Video post-pass: cur, prev uint8 [h,w,c]; flow float32 [2,h,w] -> blended uint8 [h,w,c] (written into ``out`` when given: a
    contiguous uint8 [h,w,c] GPU tensor that is neither ``cur`` nor ``prev`` - a row of the clip's result block)."""
    cur, prev, flow = _dev(cur, "cur", torch.uint8), _dev(prev, "prev", torch.uint8), _dev(flow, "flow")
    h, w, c = cur.shape
    if prev.shape != cur.shape or tuple(flow.shape) != (2, h, w):
        raise AdainHipError("warp_blend_u8: shape mismatch")
    if out is None:
        out = torch.empty_like(cur)
    elif (not out.is_cuda or out.dtype != torch.uint8 or out.shape != cur.shape or not out.is_contiguous() or out.device != cur.device
          or out.data_ptr() in (cur.data_ptr(), prev.data_ptr())):
        raise AdainHipError(f"warp_blend_u8: out must be a contiguous uint8 {tuple(cur.shape)} tensor on {cur.device}, distinct from cur and prev")
    with torch.cuda.device(cur.device):
        _check(lib().adain_warp_blend_u8(cur.data_ptr(), prev.data_ptr(), flow.data_ptr(), out.data_ptr(), h, w, c, float(alpha),
                                         float(1 - alpha), _stream()), "adain_warp_blend_u8")
    return out


def resize_area_u8(frames, dsize):
    """cv2.resize(frame, dsize, interpolation=cv2.INTER_AREA) on uint8 HWC frames: [h,w,c] or a batch [n,h,w,c];
    ``dsize`` = (width, height) as in cv2 (reference video/utils.py:352-353): the true-area branch when both axes shrink
    or stay, OpenCV's fixed-point linear emulation when one is enlarged."""
    frames = _dev(frames, "frames", torch.uint8)
    single = frames.dim() == 3
    if single:
        frames = frames.unsqueeze(0)
    if frames.dim() != 4:
        raise AdainHipError(f"resize_area_u8: expected [h,w,c] or [n,h,w,c] uint8, got {tuple(frames.shape)}")
    n, hi, wi, c = frames.shape
    wo, ho = int(dsize[0]), int(dsize[1])
    out = torch.empty((n, ho, wo, c), dtype=torch.uint8, device=frames.device)
    with torch.cuda.device(frames.device):
        _check(lib().adain_resize_area_u8(frames.data_ptr(), out.data_ptr(), n, hi, wi, c, ho, wo, _stream()), "adain_resize_area_u8")
    return out[0] if single else out


_pil_lock = threading.Lock()


def resize_pil_bilinear_u8(frames, size, crop=None, out=None):
    """``PIL.Image.resize(size, BILINEAR)`` of uint8 images on the device, bit for bit (test.py:16-24's Resize on a PIL image).
    frames: uint8 [n,h,w,3] (packed RGB) or [n,h,w,4] (Pillow's RGBX storage); ``size`` = (width, height) as PIL takes it;
    ``crop`` = (top, left, height, width) window of the result (CenterCrop), default all of it.  Returns packed RGB uint8 [n,ch,cw,3]."""
    x = _dev(frames, "frames", torch.uint8)
    if x.dim() != 4 or x.shape[3] not in (3, 4):
        raise AdainHipError(f"resize_pil_bilinear_u8: expected uint8 [n,h,w,3|4], got {tuple(x.shape)}")
    n, hi, wi, pix = x.shape
    wo, ho = int(size[0]), int(size[1])
    y0, x0, ch, cw = (0, 0, ho, wo) if crop is None else [int(v) for v in crop]
    if out is None:
        out = torch.empty((n, max(ch, 0), max(cw, 0), 3), dtype=torch.uint8, device=x.device)
    elif tuple(out.shape) != (n, ch, cw, 3) or out.dtype != torch.uint8 or not out.is_contiguous() or out.device != x.device:
        raise AdainHipError(f"resize_pil_bilinear_u8: out must be a contiguous uint8 [{n},{ch},{cw},3] on {x.device}")
    # the tap tables live in the stream's workspace between the call's two launches: calls from several threads (the job feeders'
    # fetch pool) on one stream must not interleave
    with _pil_lock, torch.cuda.device(x.device):
        ws = workspace(x.device, "pil", lib().adain_resize_pil_bilinear_u8_workspace_bytes(hi, wi, ho, wo))
        _check(lib().adain_resize_pil_bilinear_u8(x.data_ptr(), pix, n, hi, wi, out.data_ptr(), ho, wo, y0, x0, ch, cw, ws.data_ptr(), ws.numel(),
                                                  _stream()), "adain_resize_pil_bilinear_u8")
    return out


def nhwc_to_nchw(x):
    x = _dev(x, "x")
    n, h, w, c = x.shape
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib().adain_nhwc_to_nchw(x.data_ptr(), out.data_ptr(), n, c, h * w, _stream()), "adain_nhwc_to_nchw")
    return out


def nchw_to_nhwc(x):
    x = _dev(x, "x")
    n, c, h, w = x.shape
    out = torch.empty((n, h, w, c), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib().adain_nchw_to_nhwc(x.data_ptr(), out.data_ptr(), n, c, h * w, _stream()), "adain_nchw_to_nhwc")
    return out


# --- single conv layer (tests / profiling) -------------------------------------------------------------------------------------
def conv3x3_wino_pack(w_oihw, form=5):
    """Packed transformed weights U = G4 g G2^T for conv3x3_wino / conv3x3_wino4_split (24 floats per (cin, cout) pair)."""
    if form != 5:
        raise AdainHipError(f"conv3x3_wino_pack: form {form} is retired; the library runs form 5, F(4,3) x F(2,3)")
    w = _dev(w_oihw, "weight")
    cout, cin = w.shape[:2]
    packed = torch.empty(lib().adain_conv3x3_wino4_packed_floats(cin, cout), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        _check(lib().adain_conv3x3_wino4_pack(w.data_ptr(), packed.data_ptr(), cin, cout, _stream()), "adain_conv3x3_wino4_pack")
    return packed


def conv3x3_wino(x_nhwc, packed_w, bias, cout, src_mode=SRC_DIRECT, relu=True, pool_out=False, m_tiles=5):
    """One generic 3x3 layer, ReflectionPad2d(1) + Conv2d [+ ReLU] [+ fused ceil-mode pool] on NHWC, in the form the schedules run
    (``m_tiles`` = the C ABI's `form`, 5 = F(4,3) x F(2,3); weights packed by conv3x3_wino_pack)."""
    x = _dev(x_nhwc, "x")
    n, hs, ws_, cin = x.shape
    h, w = (2 * hs, 2 * ws_) if src_mode == SRC_UP2X else (hs, ws_)
    oh, ow = ((h + 1) // 2, (w + 1) // 2) if pool_out else (h, w)
    out = torch.empty((n, oh, ow, cout), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib().adain_conv3x3_wino(x.data_ptr(), out.data_ptr(), packed_w.data_ptr(), bias.data_ptr(), n, h, w, hs, ws_, cin, cout,
                                        src_mode, int(relu), int(pool_out), int(m_tiles), _stream()), "adain_conv3x3_wino")
    return out


def conv3x3_wino4_split_bytes(n, h, w, cin, cout):
    """Bytes of partial-sum slabs a launch of this layer needs to be split along cin; 0: it would not be split."""
    return lib().adain_conv3x3_wino4_split_workspace_bytes(int(n), int(h), int(w), int(cin), int(cout))


def conv3x3_wino4_split(x_nhwc, packed_w, bias, cout, src_mode=SRC_DIRECT, relu=True, pool_out=False):
    """The F(4,3) x F(2,3) layer as the latency schedule runs it: split along cin when the launch is smaller than the chip
    (adain_conv3x3_wino4_split; weights packed by conv3x3_wino_pack(form=5))."""
    x = _dev(x_nhwc, "x")
    n, hs, ws_, cin = x.shape
    h, w = (2 * hs, 2 * ws_) if src_mode == SRC_UP2X else (hs, ws_)
    oh, ow = ((h + 1) // 2, (w + 1) // 2) if pool_out else (h, w)
    out = torch.empty((n, oh, ow, cout), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        nbytes = conv3x3_wino4_split_bytes(n, h, w, cin, cout)
        ws = workspace(x.device, "conv_split", nbytes) if nbytes else None
        _check(lib().adain_conv3x3_wino4_split(x.data_ptr(), out.data_ptr(), packed_w.data_ptr(), bias.data_ptr(), n, h, w, hs, ws_, cin, cout,
                                               src_mode, int(relu), int(pool_out), ws.data_ptr() if ws is not None else None, nbytes, _stream()),
               "adain_conv3x3_wino4_split")
    return out
