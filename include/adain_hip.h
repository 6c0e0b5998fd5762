/* adain_hip.h — C ABI of the MI355X (gfx950) AdaIN style-transfer inference path.
 *
 * The reference (Ayushkuruvilla/Applied-Image-Processing) has no FFI: its hot path is plain Python
 * over torch ops (Style_3DGS/AdaIN/{function,net,test}.py).  These entry points are what a binding
 * for that path would call instead of torch; each cites the reference code it replaces.  The
 * Python host side (the modules under applied-image-processing_amd/AdaIN/) binds them with ctypes and keeps the
 * reference's function names, argument meaning and error behaviour.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless named *_host; fp32 everywhere;
 *   - the caller allocates every buffer, including workspaces sized by the *_bytes queries;
 *   - images are NCHW (the reference's tensor layout); activations between the encoder and the
 *     decoder are NHWC ("channels last"): [n][h][w][c];
 *   - `stream` is a hipStream_t (NULL = the default stream); every call only enqueues work;
 *   - return value 0 = ok, negative = error (ADAIN_E*); adain_last_error() gives the text of the
 *     calling thread's last error.  No exceptions cross the ABI.  No call allocates, frees or
 *     synchronises, so every call may be captured into a hipGraph.
 */
#ifndef ADAIN_HIP_H
#define ADAIN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: adain_encode_u8, adain_u8_to_f32 and adain_stylize_u8* added; the direct / F(2x2,3x3) single-layer entry points moved to
 * the diagnostic library (include/adain_hip_diag.h); adain_conv3x3_wino accepts form 5 only.  Version 1 was never frozen.
 * 3: adain_encode_relu1_1 and the uint8 Pillow-exact resize (adain_resize_pil_bilinear_u8*) added; nothing removed or changed.
 * 4: adain_set_schedule / adain_get_schedule and adain_conv3x3_wino4_split* added; the encoder / decoder / stylize workspaces grow by
 *    the partial-sum slabs of the latency schedule (at most 8 MB; callers that size them with the *_bytes queries need no change). */
#define ADAIN_ABI_VERSION 4
#define ADAIN_OK 0
#define ADAIN_EINVAL (-1)  /* bad argument / unsupported shape */
#define ADAIN_ELAUNCH (-2) /* HIP reported a launch error */

#define ADAIN_SRC_DIRECT 0 /* conv input = source tensor */
#define ADAIN_SRC_UP2X 1   /* conv input = nearest 2x upsample of the source (net.py:10,23,30) */
#define ADAIN_SRC_POOL2 2  /* conv input = MaxPool2d(2,2,ceil_mode=True) of the source (net.py:46,53,66) */

typedef void* adain_stream_t; /* hipStream_t */

/* The shared library is built with -fvisibility=hidden: these entry points are all it exports. */
#ifndef ADAIN_API
#define ADAIN_API __attribute__((visibility("default")))
#endif

ADAIN_API int adain_abi_version(void);
ADAIN_API const char* adain_last_error(void);

/* ---- launch schedule of the CALLING THREAD (enqueue-time state, like the error text; not captured by, and harmless to, hipGraphs) ----
 * The reference's callers work at two operating points: loops that style ONE small frame per call (video/utils.py:261-270,341-350:
 * adain_inference(content_size=256) per video frame; test.py:160: content_size 512 by default) and batch jobs.
 *   ADAIN_SCHEDULE_BATCH (default): every generic 3x3 layer accumulates its input channels in one chain.  A frame's result does not
 *     depend on the batch it is in: single-frame calls, sub-batches of any size and any sharding over GPUs give the same bits.
 *   ADAIN_SCHEDULE_LATENCY: a layer whose launch has fewer tiles than the device has compute units (relu4-level layers of a single
 *     256-class frame: 64-128 tiles for 256 compute units, each walking 256-512 input channels alone) is split along cin over
 *     2-8 workgroups per tile; their output-transformed partial sums are added in a FIXED order by a second kernel (no atomics:
 *     the same call gives the same bits every time).  Results differ from the batch schedule's in the last bits (shorter
 *     accumulation chains: slightly closer to the exact sum) and, unlike it, depend on the launch's batch size.
 * adain_set_schedule returns the previous value (>= 0) or ADAIN_EINVAL. */
#define ADAIN_SCHEDULE_BATCH 0
#define ADAIN_SCHEDULE_LATENCY 1
ADAIN_API int adain_set_schedule(int schedule);
ADAIN_API int adain_get_schedule(void);

/* ---- weights: pack a reference state_dict once (net.vgg / net.decoder, net.py:6-92) ----------------
 * w[i] / b[i] are the OIHW weight and bias tensors of the i-th conv in module order
 * (encoder: state_dict keys 0,2,5,9,12,16,19,22,25,29 ; decoder: 1,5,8,11,14,18,21,25,28).
 * `packed` receives MFMA-fragment-ordered weights + biases; its size is the *_floats query. */
ADAIN_API size_t adain_encoder_packed_floats(void);
ADAIN_API size_t adain_decoder_packed_floats(void);
ADAIN_API int adain_encoder_pack(const float* const* w_host_array_of_dev_ptrs, const float* const* b_host_array_of_dev_ptrs,
                       float* packed, adain_stream_t stream);
ADAIN_API int adain_decoder_pack(const float* const* w_host_array_of_dev_ptrs, const float* const* b_host_array_of_dev_ptrs,
                       float* packed, adain_stream_t stream);

/* ---- encoder: vgg[:31](x), conv0 .. relu4_1 (net.py:38-69; test.py:57,63,76-77,185) -----------------
 * image NCHW [n][3][h][w] -> feat NHWC [n][hc][wc][512], hc = ceil(ceil(ceil(h/2)/2)/2) (same for w).
 * layer_events: optional host array of 11 hipEvent_t recorded before layer 0 and after each of the
 * 10 conv launches (profiling only; NULL in production). */
ADAIN_API void adain_encoded_size(int h, int w, int* hc, int* wc);
ADAIN_API size_t adain_encode_workspace_bytes(int n, int h, int w);
ADAIN_API int adain_encode(const float* image_nchw, float* feat_nhwc, const float* packed, void* workspace,
                 size_t workspace_bytes, int n, int h, int w, void* const* layer_events, adain_stream_t stream);

/* adain_encode on a decoded frame as the reference holds it before ToTensor (test.py:16-24, :190-204): image HWC uint8
 * [n][h][w][3].  The first layer's kernel applies ToTensor itself (float(v) / 255, correctly rounded), so the features are
 * bit-identical to adain_encode(ToTensor(image)) while the frame crosses PCIe and HBM as 3 bytes per pixel instead of 12. */
ADAIN_API int adain_encode_u8(const uint8_t* image_nhwc_u8, float* feat_nhwc, const float* packed, void* workspace,
                    size_t workspace_bytes, int n, int h, int w, void* const* layer_events, adain_stream_t stream);

/* The encoder's FIRST fused layer alone: vgg[:4] = conv0 (1x1, the checkpoint's Caffe-style preprocessing: x255, channel swap,
 * mean subtraction) -> ReflectionPad -> conv1_1 -> ReLU (net.py:39-42), which this library computes as ONE layer with conv0 folded
 * into conv1_1's weights and bias.  image: NCHW float [n][3][h][w], or (is_u8) HWC uint8 [n][h][w][3] with ToTensor inside;
 * relu1_1: NHWC [n][h][w][64].  For parity checks of the fold at trained-checkpoint magnitudes (tests/golden/case_g.npz) and as
 * the first of the four taps AdaIN's own loss uses (net.py:110-117, enc_1); adain_encode* run the same kernel as their first launch. */
ADAIN_API int adain_encode_relu1_1(const void* image, int is_u8, float* relu1_1_nhwc, const float* packed, int n, int h, int w,
                         adain_stream_t stream);

/* The same encoder over `count` (1..4) image batches of different sizes in ONE pass: the content batch and the style image
 * of a style_transfer call go through the same vgg (test.py:57,63 / :76-77).  Results are bit-identical to one adain_encode
 * per batch; every generic 3x3 layer is a single launch whose tile list covers all batches, so the small style-branch layers
 * ride in the content launches instead of under-filling the chip on their own.  images[i] NCHW [n[i]][3][h[i]][w[i]] ->
 * feats[i] NHWC.  The pointer arrays and n / h / w are HOST arrays.  layer_events as for adain_encode. */
ADAIN_API size_t adain_encode_multi_workspace_bytes(int count, const int* n, const int* h, const int* w);
ADAIN_API int adain_encode_multi(int count, const float* const* images_nchw, float* const* feats_nhwc, const int* n, const int* h,
                       const int* w, const float* packed, void* workspace, size_t workspace_bytes,
                       void* const* layer_events, adain_stream_t stream);

/* ---- decoder: net.decoder(feat) (net.py:6-36; test.py:71,81) ----------------------------------------
 * feat NHWC [n][hc][wc][512] -> image NCHW [n][3][8hc][8wc].  layer_events: 10 events (before + 9 convs). */
ADAIN_API size_t adain_decode_workspace_bytes(int n, int hc, int wc);
ADAIN_API int adain_decode(const float* feat_nhwc, float* image_nchw, const float* packed, void* workspace,
                 size_t workspace_bytes, int n, int hc, int wc, void* const* layer_events, adain_stream_t stream);

/* ---- calc_mean_std (function.py:4-12): per (n, c) mean and sqrt(unbiased var + eps) over h*w ---------
 * nhwc != 0: feat is [n][hw][c]; nhwc == 0: feat is [n][c][hw].  mean/std: [n][c]. */
ADAIN_API size_t adain_mean_std_workspace_bytes(int nhwc, int n, int c, int hw);
ADAIN_API int adain_mean_std(const float* feat, int nhwc, int n, int c, int hw, float eps, float* mean, float* std_out,
                   void* workspace, size_t workspace_bytes, adain_stream_t stream);

/* ---- adaptive_instance_normalization + blend (function.py:15-23; test.py:69-70, 79-80) -----------------
 * t = (x - c_mean)/c_std * s_std + s_mean ;
 *   alpha form: out = t*alpha + x*one_minus_alpha          (style_transfer_simple; plain AdaIN = 1, 0)
 *   pmap  form: out = t*(1 - P) + x*P, P [pmap_n][hw]      (style_transfer, depth-aware)
 * style_n and pmap_n are 1 (broadcast over the batch) or n. */
ADAIN_API int adain_blend_alpha(const float* content_feat, int nhwc, int n, int c, int hw, const float* c_mean,
                      const float* c_std, const float* s_mean, const float* s_std, int style_n, float alpha,
                      float one_minus_alpha, float* out, adain_stream_t stream);
ADAIN_API int adain_blend_pmap(const float* content_feat, int nhwc, int n, int c, int hw, const float* c_mean,
                     const float* c_std, const float* s_mean, const float* s_std, int style_n, const float* pmap,
                     int pmap_n, float* out, adain_stream_t stream);

/* ---- compute_stylization_strength_map (test.py:119-150) ------------------------------------------------
 * depth [h0][w0] -> pmap [hc][wc]: bicubic resize, min-max normalise, minus mean, sigmoid(prominence*P),
 * clamp(max = 1 - offset); an exactly constant resized map gives zeros (test.py:141-143). */
ADAIN_API size_t adain_strength_map_workspace_bytes(int hc, int wc);
ADAIN_API int adain_strength_map(const float* depth, int h0, int w0, int hc, int wc, float offset, float prominence,
                       float* pmap, void* workspace, size_t workspace_bytes, adain_stream_t stream);

/* ---- content-mask composite of adain_inference (test.py:222-236) ---------------------------------------
 * resize_*: `planes` independent [hi][wi] planes -> [ho][wo]; bilinear = F.interpolate(mode="bilinear",
 * align_corners=False), nearest = F.interpolate(mode="nearest").
 * mask_composite: out = content*(1-m) + stylized*m on NCHW [n][c][hw]; mask [mask_n][mask_c][hw],
 * mask_c in {1, c}, mask_n in {1, n}. */
ADAIN_API int adain_resize_bilinear(const float* in, float* out, int planes, int hi, int wi, int ho, int wo,
                          adain_stream_t stream);
ADAIN_API int adain_resize_nearest(const float* in, float* out, int planes, int hi, int wi, int ho, int wo,
                         adain_stream_t stream);
ADAIN_API int adain_mask_composite(const float* content, const float* stylized, const float* mask, int mask_c, int mask_n,
                         float* out, int n, int c, int hw, adain_stream_t stream);

/* ---- torchvision save_image quantiser (test.py:243-244): NCHW float -> NHWC u8, x*255+0.5 clamped ------ */
ADAIN_API int adain_quantize_u8(const float* image_nchw, uint8_t* out_nhwc, int n, int c, int h, int w,
                      adain_stream_t stream);

/* ---- torchvision ToTensor (test.py:22): NHWC u8 [n][h][w][c] -> NCHW float [n][c][h][w], float(v) / 255 correctly rounded
 * (bit for bit `tensor.float() / 255`); what the mask composite needs of a frame that was uploaded as uint8 ------------------ */
ADAIN_API int adain_u8_to_f32(const uint8_t* in_nhwc, float* out_nchw, int n, int c, int h, int w, adain_stream_t stream);

/* ---- video post-pass (reference video/utils.py:89-105 warp_image + :223-229 blend_images) -------------------
 * HWC uint8 frames [h][w][c]; flow [2][h][w] (x then y displacement, as estimate_optical_flow returns it,
 * video/utils.py:75-86).  out = u8(clip((alpha*cur/255 + one_minus_alpha*warp(prev)/255)*255, 0, 255)), warp =
 * cv2.remap(INTER_LINEAR, BORDER_REFLECT) in OpenCV's uint8 fixed point (map rounded to 1/32 px, 2^15-scaled weights,
 * (sum + 2^14) >> 15).  The optical-flow estimate itself stays with the caller (OpenCV). */
ADAIN_API int adain_warp_blend_u8(const uint8_t* cur_u8, const uint8_t* prev_u8, const float* flow, uint8_t* out_u8, int h, int w,
                        int c, float alpha, float one_minus_alpha, adain_stream_t stream);

/* cv2.resize(frames_u8, (wo, ho), interpolation=cv2.INTER_AREA) of the same post-pass (reference video/utils.py:352-353) on
 * n HWC uint8 frames [n][hi][wi][c] -> [n][ho][wo][c].  The true-area branch of OpenCV's resize (both axes shrink or keep
 * their size): equal sizes copy; integer scales box-average in int with round-half-even ((a+b+c+d+2)>>2 for 2x2); other
 * scales use resizeArea_'s float tap tables in OpenCV's accumulation order.  With an axis enlarged OpenCV leaves that branch
 * and so does this call: the 11-bit fixed-point linear pass with "area mode" coefficients (an integer enlargement replicates
 * pixels). */
ADAIN_API int adain_resize_area_u8(const uint8_t* in_u8, uint8_t* out_u8, int n, int hi, int wi, int c, int ho, int wo,
                         adain_stream_t stream);

/* ---- test_transform's Resize [+ CenterCrop] on the device (test.py:16-24, applied at :190-204; video/utils.py:341-350) --------
 * PIL.Image.resize((wo, ho), BILINEAR) of uint8 RGB images, bit for bit (Pillow's ImagingResample: separable triangle filter whose
 * support grows with the shrink factor, double-precision taps converted to 22-bit fixed point, a horizontal pass into a uint8
 * intermediate, then a vertical pass, clip8) - what torchvision's Resize(size) runs on a PIL image.  in: [n][hi][wi][pixel_bytes]
 * with pixel_bytes = 3 (packed RGB) or 4 (Pillow's RGBX storage, 4th byte ignored); out: packed RGB [n][crop_h][crop_w][3] = the
 * window (crop_y0, crop_x0, crop_h, crop_w) of the ho x wo result (CenterCrop(size): top = round((ho - size) / 2), left likewise;
 * no crop: 0, 0, ho, wo).  The result feeds adain_encode_u8 / adain_stylize_u8 directly.  Workspace: the tap tables of both axes. */
ADAIN_API size_t adain_resize_pil_bilinear_u8_workspace_bytes(int hi, int wi, int ho, int wo);
ADAIN_API int adain_resize_pil_bilinear_u8(const uint8_t* in_u8, int pixel_bytes, int n, int hi, int wi, uint8_t* out_rgb_u8, int ho, int wo,
                                 int crop_y0, int crop_x0, int crop_h, int crop_w, void* workspace, size_t workspace_bytes,
                                 adain_stream_t stream);

/* ---- one sub-batch of the reference's batch callers in ONE call ---------------------------------------------------------------
 * What adain_inference does between `Image.open` and `save_image` for n decoded frames of one size and one style whose
 * statistics are already known (test.py:203-244 per frame; the callers loop over frames with the same style,
 * video/utils.py:341-350 and Style_3DGS/train.py:101):
 *     ToTensor + vgg(content)                     adain_encode_u8        test.py:203-204, :57 / :76
 *     calc_mean_std(content_f)                    adain_mean_std         function.py:4-12
 *     depth_maps == NULL:  AdaIN*alpha + content_f*one_minus_alpha       test.py:79-80    (adain_blend_alpha; the caller passes
 *                          float(1 - alpha) computed in double, as the reference's Python scalar is)
 *     depth_maps != NULL:  P = strength map of depth_maps[i] [depth_h[i]][depth_w[i]] per frame (adain_strength_map,
 *                          test.py:66-67), AdaIN*(1-P) + content_f*P     test.py:69-70    (adain_blend_pmap); alpha unused
 *     decoder(feat)                               adain_decode           test.py:71 / :81
 *     mask != NULL:  content*(1-m) + out*m with m = nearest(mask.float()) and out = bilinear(out), both to the frame's size
 *                                                                        test.py:222-236  (adain_resize_* + adain_mask_composite)
 *     x*255 + 0.5, clamp, uint8 HWC                adain_quantize_u8      test.py:243-244
 * Every stage runs the kernel of the entry point named beside it with the same arguments, so `out_u8` holds the bytes that
 * sequence of calls gives; the tails are fused with the same arithmetic: without a mask the decoder's last layer quantises its
 * pixels itself (no float image in HBM, no quantiser launch); when decoder output and frame share one size (sides that are multiples of 8) the composite's passes
 * and the quantiser run as ONE kernel with the same arithmetic - reading the mask in place when it has the frame's size too (a
 * mask made from the frame itself), sampling it with the nearest resize's index map when it has another (the guide loop: a view
 * resized to content_size with its mask at the view's own size, Style_3DGS/train.py:97-101).
 * frames HWC uint8 [n][h][w][3]; s_mean / s_std [512]: the style's statistics (adain_encode + adain_mean_std of the style
 * image, once per style); depth_maps / depth_h / depth_w: HOST arrays of n device pointers / sizes; mask [mask_n][mask_c]
 * [mask_h][mask_w], mask_n in {1, n}, mask_c in {1, 3}, uint8 / bool bytes (mask_is_float == 0) or float; out_u8 HWC uint8
 * [n][oh][ow][3] with (oh, ow) = adain_stylize_u8_out_size: the frame's size with a mask, 8hc x 8wc without.  One workspace
 * (adain_stylize_u8_workspace_bytes) holds every intermediate.  21-27 kernel launches, no allocation, no synchronisation. */
ADAIN_API size_t adain_stylize_u8_workspace_bytes(int n, int h, int w, int use_depth, int mask_n, int mask_c, int mask_h, int mask_w,
                                        int mask_is_float);
ADAIN_API void adain_stylize_u8_out_size(int h, int w, int has_mask, int* oh, int* ow);
ADAIN_API int adain_stylize_u8(const uint8_t* frames_nhwc_u8, int n, int h, int w, const float* enc_packed, const float* dec_packed,
                     const float* s_mean, const float* s_std, float alpha, float one_minus_alpha,
                     const float* const* depth_maps_host_array_of_dev_ptrs,
                     const int* depth_h_host, const int* depth_w_host, float depth_offset, float depth_prominence, const void* mask,
                     int mask_is_float, int mask_n, int mask_c, int mask_h, int mask_w, uint8_t* out_u8, void* workspace,
                     size_t workspace_bytes, adain_stream_t stream);

/* ---- layout changes at the boundary ([n][c][hw] <-> [n][hw][c]) ------------------------------------------ */
ADAIN_API int adain_nhwc_to_nchw(const float* in, float* out, int n, int c, int hw, adain_stream_t stream);
ADAIN_API int adain_nchw_to_nhwc(const float* in, float* out, int n, int c, int hw, adain_stream_t stream);

/* ---- one generic 3x3 layer in the form the schedules run (unit tests, profiling) ---------------------------------
 * ReflectionPad2d(1) + Conv2d(cin, cout, 3) [+ ReLU] on NHWC as Winograd F(4,3) x F(2,3): 4 x 2 output tiles, 3 multiplies per
 * output instead of 9 (fp32 rounding error ~5x the direct form's, ~1e-6 relative per layer).  The nearest-2x upsample of the
 * producer may be fused into the input gather (src_mode ADAIN_SRC_UP2X) and the ceil-mode max-pool behind the layer into the
 * epilogue (pool_out != 0: out is [n][ceil(h/2)][ceil(w/2)][cout]).  (h, w) = conv output size before any output pool;
 * (hs, ws) = source size.  Weights packed by adain_conv3x3_wino4_pack, 24 floats per (cin, cout) pair; cin % 16 == 0,
 * cout % 32 == 0; persistent kernel when the launch has >= 2 tiles per resident workgroup.  `form` must be 5 (the other forms -
 * F(2x2,3x3) and the direct implicit GEMM - live in the diagnostic library only, include/adain_hip_diag.h). */
ADAIN_API size_t adain_conv3x3_wino4_packed_floats(int cin, int cout);
ADAIN_API int adain_conv3x3_wino4_pack(const float* w_oihw, float* packed, int cin, int cout, adain_stream_t stream);
ADAIN_API int adain_conv3x3_wino(const float* in_nhwc, float* out_nhwc, const float* packed_w, const float* bias, int n, int h,
                       int w, int hs, int ws, int cin, int cout, int src_mode, int relu, int pool_out, int form,
                       adain_stream_t stream);

/* The same layer (form 5) as ADAIN_SCHEDULE_LATENCY runs it, whatever the calling thread's schedule: split along cin when the launch
 * has fewer tiles than compute units.  *_workspace_bytes = the partial-sum slabs that launch needs (S x n x h x w x cout floats;
 * 0: it would not be split, and adain_conv3x3_wino4_split then is adain_conv3x3_wino).  For unit tests and the per-layer error
 * probe (tools/probes/wino_error_gpu.py). */
ADAIN_API size_t adain_conv3x3_wino4_split_workspace_bytes(int n, int h, int w, int cin, int cout);
ADAIN_API int adain_conv3x3_wino4_split(const float* in_nhwc, float* out_nhwc, const float* packed_w, const float* bias, int n, int h,
                              int w, int hs, int ws, int cin, int cout, int src_mode, int relu, int pool_out, void* workspace,
                              size_t workspace_bytes, adain_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ADAIN_HIP_H */
