"""GPU tests of ``adain_resize_pil_bilinear_u8`` (round 5): test_transform's Resize step (reference test.py:16-24, torchvision
Resize(size) on a PIL image = ``PIL.Image.resize(size, BILINEAR)``) on the device must give Pillow's bytes - the Pillow installed
next to the tests IS the reference implementation here (the oracle for this row is the library the reference calls).  Bit-exact:
``torch.equal`` over the callers' own size pairs and 220 random ones, shrinking and enlarging, packed RGB and Pillow's RGBX
storage, batches, crop windows (CenterCrop).  Run with ``-m gpu``."""
import numpy as np
import pytest
import torch
from PIL import Image

import applied_image_processing_amd.synth as synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rt():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import applied_image_processing_amd.runtime as rt

    rt.lib()
    return rt


def picture(seed, h, w):
    """uint8 RGB [h,w,3]: noise over a smooth ramp with saturated patches (clip8 at both ends gets exercised)."""
    a = synth.image(seed, 1, h, w)[0].transpose(1, 2, 0)
    yy, xx = np.mgrid[:h, :w].astype(np.float32)
    ramp = ((yy / max(h - 1, 1) + xx / max(w - 1, 1)) * 0.5)[..., None]
    v = np.where(a > 0.9, 1.0, np.where(a < 0.1, 0.0, 0.6 * a + 0.4 * ramp))
    return (v * 255).astype(np.uint8)


def pil_resize(a, size):
    return np.asarray(Image.fromarray(a).resize(size, Image.BILINEAR))


NAMED = [((700, 933), (341, 256)),     # the sample style image at style_size 256: 933 x 700 -> 341 x 256
         ((270, 480), (455, 256)),     # a video frame at content_size 256 (video/utils.py:264)
         ((800, 800), (512, 512)),     # a 3DGS training view at img_size 512 (train.py:494)
         ((512, 512), (256, 256)),     # brad_pitt.jpg at 256
         ((700, 933), (682, 512)),     # the style at the API default style_size 512
         ((64, 48), (200, 123)),       # enlarging
         ((31, 57), (31, 20)),         # one axis unchanged
         ((1080, 1920), (455, 256))]   # a 1080p frame down to the video caller's size: 9-tap rows


@pytest.mark.parametrize("hw,size", NAMED)
def test_named_sizes_are_pillows_bytes(rt, hw, size):
    a = picture(5, *hw)
    want = pil_resize(a, size)
    got = rt.resize_pil_bilinear_u8(torch.from_numpy(a[None]).cuda(), size)
    assert got.shape == (1, size[1], size[0], 3)
    assert torch.equal(got[0].cpu(), torch.from_numpy(want.copy()))
    # Pillow's own 4-byte storage (Image.tobytes("raw", "RGBX")) as the source
    x4 = np.frombuffer(Image.fromarray(a).tobytes("raw", "RGBX"), np.uint8).reshape(hw[0], hw[1], 4)
    assert torch.equal(rt.resize_pil_bilinear_u8(torch.from_numpy(x4[None].copy()).cuda(), size)[0].cpu(), torch.from_numpy(want.copy()))


def test_220_random_size_pairs(rt):
    g = np.random.default_rng(20261005)
    bad = []
    for case in range(220):
        hi, wi = int(g.integers(1, 400)), int(g.integers(1, 400))
        if case % 3 == 0:      # shrink, up to 12 x
            ho, wo = max(1, int(hi / g.uniform(1.0, 12.0))), max(1, int(wi / g.uniform(1.0, 12.0)))
        elif case % 3 == 1:    # enlarge
            ho, wo = int(hi * g.uniform(1.0, 4.0)) + 1, int(wi * g.uniform(1.0, 4.0)) + 1
        else:                  # anything
            ho, wo = int(g.integers(1, 500)), int(g.integers(1, 500))
        a = picture(100 + case, hi, wi)
        got = rt.resize_pil_bilinear_u8(torch.from_numpy(a[None]).cuda(), (wo, ho))[0].cpu().numpy()
        if not np.array_equal(got, pil_resize(a, (wo, ho))):
            bad.append((hi, wi, ho, wo, int(np.abs(got.astype(int) - pil_resize(a, (wo, ho)).astype(int)).max())))
    assert not bad, bad[:10]


def test_batches_and_crop_windows(rt):
    frames = np.stack([picture(300 + k, 120, 200) for k in range(3)])
    full = np.stack([pil_resize(f, (107, 64)) for f in frames])
    x = torch.from_numpy(frames).cuda()
    assert torch.equal(rt.resize_pil_bilinear_u8(x, (107, 64)).cpu(), torch.from_numpy(full))
    # CenterCrop(64) of the 64 x 107 result (torchvision: top = round((h - s) / 2), left = round((w - s) / 2) = round(21.5) = 22)
    left = int(round((107 - 64) / 2.0))
    got = rt.resize_pil_bilinear_u8(x, (107, 64), crop=(0, left, 64, 64))
    assert torch.equal(got.cpu(), torch.from_numpy(full[:, :, left:left + 64].copy()))
    out = torch.zeros((3, 10, 7, 3), dtype=torch.uint8, device="cuda")
    assert rt.resize_pil_bilinear_u8(x, (107, 64), crop=(50, 100, 10, 7), out=out) is out
    assert torch.equal(out.cpu(), torch.from_numpy(full[:, 50:60, 100:107].copy()))
    with pytest.raises(rt.AdainHipError):
        rt.resize_pil_bilinear_u8(x, (107, 64), crop=(60, 0, 10, 10))             # window outside the result
    with pytest.raises(rt.AdainHipError):
        rt.resize_pil_bilinear_u8(x.cpu(), (107, 64))                              # no CPU fallback
    with pytest.raises(rt.AdainHipError):
        rt.resize_pil_bilinear_u8(x[..., :2].contiguous(), (107, 64))              # RGB or RGBX only


def test_a_large_photograph_sized_frame(rt):
    a = picture(9, 2160, 3840)                                                     # a 4K frame to the guide loop's 512: 15-tap rows
    got = rt.resize_pil_bilinear_u8(torch.from_numpy(a[None]).cuda(), (910, 512))[0].cpu().numpy()
    assert np.array_equal(got, pil_resize(a, (910, 512)))


def test_device_transform_is_test_transform_u8(rt):
    """``device_transform_u8`` (upload + Resize [+ CenterCrop] on the device) against ``test_transform_u8`` (PIL on the host, the
    restatement of test.py:16-24): the same bytes for every (size, crop) the callers use; images Pillow resizes through other code
    (RGBA: premultiplied alpha; L) and crops torchvision would pad stay on the host path."""
    from applied_image_processing_amd.AdaIN import test as t

    dev = torch.device("cuda:0")
    for (h, w) in [(270, 480), (700, 933), (800, 800), (96, 64)]:
        img = Image.fromarray(picture(400 + h, h, w))
        for size, crop in [(256, False), (256, True), (512, False), (0, False), (64, True), (h if h < w else w, False)]:
            want = t.test_transform_u8(size, crop)(img)
            got = t.device_transform_u8(img, size, crop, dev)
            if crop and min(want.shape[:2]) < size:
                continue
            assert got is not None and got.dtype == torch.uint8 and tuple(got.shape) == (1,) + want.shape, (h, w, size, crop)
            assert np.array_equal(got[0].cpu().numpy(), want), (h, w, size, crop)
    small = Image.fromarray(picture(7, 40, 60))
    # size 0 = "keep the size" (test.py:17: `if size != 0`): the uploaded bytes themselves, no resize launch
    whole = t.device_transform_u8(small, 0, False, dev)
    assert whole is not None and tuple(whole.shape) == (1, 40, 60, 3) and np.array_equal(whole[0].cpu().numpy(), np.asarray(small))
    # a shrink factor beyond 256 is refused by the C ABI (with a message) and left to Pillow on the host by the transform
    wide = Image.fromarray(picture(8, 3, 3000))
    with pytest.raises(rt.AdainHipError, match="shrink factor"):
        rt.resize_pil_bilinear_u8(torch.from_numpy(np.asarray(wide)).cuda()[None], (10, 3))
    assert t.device_transform_u8(Image.fromarray(picture(9, 300, 6000)), 1, False, dev) is None          # 300 x: the host path's
    assert t.device_transform_u8(small.convert("RGBA"), 32, False, dev) is None
    assert t.device_transform_u8(small.convert("L"), 32, False, dev) is None


def test_device_transform_from_several_threads(rt):
    """The job feeders fetch (decode + transform) frames on a thread pool: per-thread staging buffers, one lock around a resize's two
    launches - concurrent calls with different sizes must each give PIL's bytes."""
    from concurrent.futures import ThreadPoolExecutor

    from applied_image_processing_amd.AdaIN import test as t

    dev = torch.device("cuda:0")
    imgs = [Image.fromarray(picture(500 + k, 100 + 7 * k, 140 + 11 * (k % 5))) for k in range(24)]
    want = [t.test_transform_u8(64 + (k % 3) * 8, False)(im) for k, im in enumerate(imgs)]
    with ThreadPoolExecutor(max_workers=4) as ex:
        got = list(ex.map(lambda kv: t.device_transform_u8(kv[1], 64 + (kv[0] % 3) * 8, False, dev), enumerate(imgs)))
    torch.cuda.synchronize()
    for k in range(len(imgs)):
        assert np.array_equal(got[k][0].cpu().numpy(), want[k]), k
