#!/usr/bin/env python3
"""Launches every kernel of the product library at least once, through the public entry points, at shapes that select each launch
form - so that ONE `rocprofv3 --kernel-trace` of this script names every kernel a product entry point can launch
(profiles/r03_all_kernels.md).  Prints the entry points it called; no timing of its own.

    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d <out> -- python3 <repo>/tools/all_kernels.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import numpy as np
import torch

import applied_image_processing_amd.runtime as rt
import applied_image_processing_amd.synth as synth
from applied_image_processing_amd.engine import AdaINEngine, temporal_blend

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
vgg_sd, dec_sd = synth.to_torch(synth.vgg_state_dict(0, full=False)), synth.to_torch(synth.decoder_state_dict(0))
eng = AdaINEngine(vgg_sd, dec_sd, dev)                       # pack_conv_first / pack_conv_last / pack_wino4
called = ["adain_encoder_pack", "adain_decoder_pack"]


def u8(seed, n, h, w, c=3):
    return T((synth.image(seed, n, h, w, c=c).transpose(0, 2, 3, 1) * 255).astype(np.uint8)).to(dev)


for rep in range(3):
    # encoder / decoder: persistent 8 x 32 geometry (1024^2), the 16 x 16 geometry (1200 x 1600 maps), one-tile launches (small
    # style), up-sampling layers, fused pools; float and uint8 first layer
    c = T(synth.image(3, 1, 1024, 1024)).to(dev)
    s = T(synth.image(4, 1, 512, 512)).to(dev)
    cf, sf = rt.encode_multi([c, s], eng.enc)
    sm, ss = rt.mean_std(sf, True)                           # mean_std_nhwc_partial + mean_std_finalize
    cm, cs = rt.mean_std(cf, True)
    g = rt.blend_alpha(cf, True, cm, cs, sm, ss, 0.5)        # adain_blend_kernel<nhwc>
    out = rt.decode(g, eng.dec)
    frames = u8(7, 2, 1200, 1600)
    f8 = rt.encode_u8(frames, eng.enc)                       # conv_first_kernel<true>; 300 x 400 / 150 x 200 maps: 16 x 16 tiles
    eng.use_style_stats((sm, ss))
    o8 = eng.stylize(frames, 0.5)
    small = rt.encode(T(synth.image(5, 1, 72, 104)).to(dev), eng.enc)        # one-tile launches
    # NCHW forms of the statistics / blend / layout kernels
    nchw = rt.nhwc_to_nchw(cf)                               # transpose_kernel
    back = rt.nchw_to_nhwc(nchw)
    m2, s2 = rt.mean_std(nchw, False)                        # mean_std_nchw_kernel
    g2 = rt.blend_alpha(nchw, False, m2, s2, sm, ss, 1.0)    # adain_blend_kernel<nchw>
    # depth-aware blend: bicubic_minmax + strength_sum + strength_apply, P-map blend
    depth = T(synth.smooth_depth(6, 1024, 1024)).to(dev)
    p = rt.strength_map(depth, 128, 128, 0.15, 20)
    g3 = rt.blend_pmap(cf, True, cm, cs, sm, ss, p)
    # mask composite + save_image quantiser + ToTensor
    f32 = rt.u8_to_f32(frames)                               # u8_to_f32_rgb4_kernel
    odd = rt.u8_to_f32(u8(8, 1, 37, 53, c=4))                # u8_to_f32_kernel (generic)
    mask = (f32 > 0.3).float()
    rs = rt.resize_bilinear(o8, (1200, 1600))
    mk = rt.resize_nearest(mask, (1200, 1600))
    comp = rt.mask_composite(f32, rs, mk)
    q = rt.quantize_u8(comp)                                 # quantize_u8_rgb4_kernel
    q1 = rt.quantize_u8(T(synth.image(9, 1, 33, 47, c=1)).to(dev))       # quantize_u8_kernel (generic)
    rs2 = rt.resize_bilinear(T(synth.image(10, 1, 48, 72)).to(dev), (45, 67))
    # video post-pass: INTER_AREA forms and the warp + blend
    v = u8(11, 4, 1080, 1920)
    a2 = rt.resize_area_u8(v, (960, 540))                    # resize_area2x2_rgb4_kernel
    at = rt.resize_area_u8(v, (1280, 720))                   # resize_area_tab_u8_kernel
    ab = rt.resize_area_u8(v, (640, 360))                    # integer 3 x 3 box: resize_area_u8_kernel
    al = rt.resize_area_u8(u8(12, 2, 256, 456), (512, 288))  # resize_area_linear_u8_kernel (an enlarged axis)
    a1 = rt.resize_area_u8(u8(13, 1, 90, 121, c=4), (57, 31))
    flows = T(np.stack([synth.uniform_sym(14 + i, (2, 1080, 1920), 2.0) for i in range(3)])).to(dev)
    tb = temporal_blend(v, flows, 0.7)                       # warp_blend_u8_rgb4_kernel
    wb = rt.warp_blend_u8(u8(15, 1, 40, 56, c=1)[0], u8(16, 1, 40, 56, c=1)[0], T(synth.uniform_sym(17, (2, 40, 56), 2.0)).to(dev), 0.7)
    # one sub-batch in one call: the fused composite + quantise tail (uint8 / float masks, vector and generic forms) and the general tail
    eng.stylize_u8(frames, 0.5, masks=(frames > 0).permute(0, 3, 1, 2).contiguous())          # composite_quantize_u8_kernel<uint8_t, true>
    eng.stylize_u8(frames, 0.5, masks=(f32 > 0.3).float())                                     # composite_quantize_u8_kernel<float, true>
    small_u8 = u8(18, 1, 72, 104)
    eng.stylize_u8(small_u8, 0.5, masks=(small_u8[:, ::2, ::2] > 40).permute(0, 3, 1, 2).contiguous())    # mask_to_f32 + the resize / composite chain
    odd_u8 = u8(19, 1, 45, 67)
    eng.stylize_u8(odd_u8, 0.5, masks=(odd_u8 > 40).permute(0, 3, 1, 2).contiguous())
    eng.stylize_u8(frames, depth_maps=[depth[:600, :800].contiguous(), depth], offset=0.3)     # depth-aware form
    # test_transform's Resize on the device (Pillow-exact): packed RGB and Pillow's 4-byte storage, with a CenterCrop window
    r3 = rt.resize_pil_bilinear_u8(u8(22, 1, 700, 933), (682, 512))                           # pil_coeffs_kernel + pil_resize_kernel<3>
    r4 = rt.resize_pil_bilinear_u8(u8(23, 2, 270, 480, c=4), (455, 256), crop=(0, 100, 256, 256))    # pil_resize_kernel<4>
    first = rt.encode_relu1_1(r3, eng.enc)                                                    # conv_first_kernel<true> alone
    torch.cuda.synchronize()
called += ["adain_resize_pil_bilinear_u8", "adain_encode_relu1_1", "adain_stylize_u8", "adain_encode_multi", "adain_encode_u8", "adain_encode", "adain_decode", "adain_mean_std", "adain_blend_alpha", "adain_blend_pmap",
           "adain_strength_map", "adain_u8_to_f32", "adain_resize_bilinear", "adain_resize_nearest", "adain_mask_composite", "adain_quantize_u8",
           "adain_resize_area_u8", "adain_warp_blend_u8", "adain_nhwc_to_nchw", "adain_nchw_to_nhwc"]
# the single-layer entry point (BIG descriptors are exercised by tests/test_gpu_parity.py::test_conv_tensors_above_two_gib)
x = T(synth.uniform_sym(20, (1, 64, 64, 64), 1.0)).to(dev)
w = T(synth.uniform_sym(21, (64, 64, 3, 3), 0.1)).to(dev)
y = rt.conv3x3_wino(x, rt.conv3x3_wino_pack(w, 5), torch.zeros(64, device=dev), 64, rt.SRC_UP2X, True, False, 5)
# round 6: the latency schedule on one small frame (cin-split one-tile launches + splitk_combine_kernel) and the split layer alone
with rt.schedule(rt.SCHEDULE_LATENCY):
    eng.stylize_u8(u8(24, 1, 256, 456), 0.5)
xs = T(synth.uniform_sym(25, (1, 32, 57, 512), 1.0)).to(dev)
ws = T(synth.uniform_sym(26, (256, 512, 3, 3), 0.02)).to(dev)
ys = rt.conv3x3_wino4_split(xs, rt.conv3x3_wino_pack(ws, 5), torch.zeros(256, device=dev), 256, rt.SRC_DIRECT, True, True)       # with the fused pool
torch.cuda.synchronize()
called += ["adain_conv3x3_wino4_pack", "adain_conv3x3_wino", "adain_set_schedule", "adain_conv3x3_wino4_split"]
print("entry points called:", ", ".join(sorted(set(called))))
