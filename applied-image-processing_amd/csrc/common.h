// Shared declarations for the gfx950 AdaIN kernels (internal; the public C ABI is include/adain_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

namespace adain {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

// Source-gather modes of the 3x3 convolution's input (fused into the LDS staging):
//   DIRECT : conceptual input == source tensor
//   UP2X   : conceptual input == nearest 2x upsample of the source   (decoder, net.py:10,23,30)
//   POOL2  : conceptual input == MaxPool2d(2,2,ceil_mode=True) of the source (encoder, net.py:46,53,66) - the schedules fuse the pool into
//            the PRODUCER's epilogue instead (ConvArgs::pool_out); the consumer-side form went with the direct kernels (docs/HISTORY.md)
enum SrcMode { SRC_DIRECT = 0, SRC_UP2X = 1, SRC_POOL2 = 2 };

struct ConvArgs {
    const float* in;    // NHWC source  [n][Hs][Ws][cin]
    float* out;         // NHWC output  [n][H][W][cout]
    const float* wpk;   // packed weights (pack_wino4_kernel layout)
    const float* bias;  // [cout]
    int n, H, W;        // output (== conceptual input) spatial size
    int Hs, Ws;         // source spatial size
    int cin, cout;
    int relu;
    int pool_out;       // 1: write only MaxPool2d(2,2,ceil_mode=True)(output), [n][ceil(H/2)][ceil(W/2)][cout]
    int xcd_order;      // 1: XCD-aware block -> tile order (speed only)
    int tiles_x, tiles_y;
    unsigned long long* dbg;   // diagnostic builds only (clock stamps); nullptr in production
    // cin split of the F(4,3) x F(2,3) one-tile form (set by launch_conv3x3_wino4 only): workgroups per (tile, channel tile),
    // input channels per workgroup, floats between the partial-sum slabs that `out` then points at
    int ksplit, cin_sub;
    size_t slab_stride;
};

// Partial-sum slabs for a cin-split launch of the F(4,3) x F(2,3) kernel (caller-owned workspace; slab == nullptr: never split).
struct SplitWs {
    float* slab;
    size_t floats;
};

// One (source, output) tensor pair of a multi-segment launch of the persistent F(4,3) x F(2,3) kernel: the segments of one
// launch share the layer (weights, bias, cin, cout, source mode, ReLU, output pool) and differ in size and addresses.
struct ConvSeg {
    const float* in;    // NHWC source  [n][Hs][Ws][cin]
    float* out;         // NHWC output  [n][H][W][cout]   (pooled size when the layer's pool_out is set)
    int n, H, W, Hs, Ws;
    int tiles_x, tiles_y;
    int item0;          // first item (tile x channel tile) of this segment in the launch's list
};
constexpr int MAX_CONV_SEGS = 4;
struct ConvSegs {
    int count;
    int ctg;            // channel tiles per group of the persistent walk (divides cout / 32): see conv3x3_wino4_kernel
    int stagger;        // start delay of the second half of the persistent grid in units of 64 cycles
    int pad_;
    ConvSeg s[MAX_CONV_SEGS];
};

// Tuning and diagnostic switches exist only in the diagnostic build (-DADAIN_DIAG: libadain_hip_diag.so, loaded by tools/
// through ADAIN_HIP_LIB): the product library takes no behaviour from the environment and holds no timing-only kernels.
inline int tune_env(const char* name, int dflt) {
#ifdef ADAIN_DIAG
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

// thread-local error text for adain_last_error()
void set_error(const char* fmt, ...);

// launchers (conv_edge.hip)
int launch_pack_conv_first(const float* w0, const float* b0, const float* w1, const float* b1, float* packed,
                           float* bias_out, hipStream_t s);
int launch_pack_conv_last(const float* w, float* packed, hipStream_t s);
// Compute units of the current device (cached per device id; 0 on failure).  Sizes the persistent kernels' grids.
inline int device_cu_count() {
    static int cached[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (!cached[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        cached[dev] = n;
    }
    return cached[dev];
}

// conv_wino4.hip: F(4,3) x F(2,3) form (its own packed-weight layout, 24 floats per weight pair)
int launch_pack_wino4(const float* w_oihw, float* packed, int cin, int cout, hipStream_t s);
// split: workspace for the cin split of a launch too small to give every compute unit a tile (see wino4_split_floats); without
// one the launch is never split
int launch_conv3x3_wino4(const ConvArgs& a, int src_mode, hipStream_t s, SplitWs split = SplitWs{nullptr, 0});
// the same layer over `count` tensor pairs (sizes / addresses from segs[i]: in, out, n, H, W, Hs, Ws; the rest from `layer`):
// one persistent launch whose tile list covers every segment when there is enough work, one launch per segment otherwise
// (split[i]: segment i's slab workspace for that case, or nullptr)
int launch_conv3x3_wino4_multi(const ConvArgs& layer, const ConvSeg* segs, int count, int src_mode, hipStream_t s,
                               const SplitWs* split = nullptr);
// floats of slab workspace a launch of this layer over n images of H x W (conv output size) needs to be split; 0: it would not be
size_t wino4_split_floats(int n, int H, int W, int cin, int cout);
double wino4_rounds_per_image(int H, int W, int cout);      // persistent-grid rounds one image of a layer is worth (schedules, api.hip)
// img: NCHW float [n][3][H][W], or (u8 != 0) HWC uint8 [n][H][W][3] converted as ToTensor does (v / 255)
int launch_conv_first(const void* img, int u8, float* out_nhwc, const float* packed, const float* bias, int n, int H,
                      int W, hipStream_t s);
// out_u8 != nullptr: the image is written as save_image's uint8 HWC [n][H][W][3] instead of NCHW float (out_nchw is then unused)
int launch_conv_last(const float* in_nhwc, float* out_nchw, const float* packed, const float* bias, int n, int H,
                     int W, hipStream_t s, uint8_t* out_u8 = nullptr);

// stats.hip
int launch_mean_std(const float* feat, int nhwc, int n, int c, int hw, float eps, float* mean, float* std_,
                    void* workspace, size_t ws_bytes, hipStream_t s);
size_t mean_std_workspace_bytes(int nhwc, int n, int c, int hw);
int launch_adain_blend_ex(const float* content, int nhwc, int n, int c, int hw, const float* c_mean,
                          const float* c_std, const float* s_mean, const float* s_std, int style_n, float alpha,
                          float one_minus_alpha, const float* pmap, int pmap_n, float* out, hipStream_t s);

// pixel.hip
int launch_strength_map(const float* depth, int h0, int w0, int hc, int wc, float offset, float prominence,
                        float* pmap, void* workspace, size_t ws_bytes, hipStream_t s);
size_t strength_map_workspace_bytes(int hc, int wc);
int launch_resize_bilinear(const float* in, float* out, int planes, int hi, int wi, int ho, int wo, hipStream_t s);
int launch_resize_nearest(const float* in, float* out, int planes, int hi, int wi, int ho, int wo, hipStream_t s);
int launch_mask_composite(const float* content, const float* stylized, const float* mask, int mask_c, int mask_n,
                          float* out, int n, int c, int hw, hipStream_t s);
int launch_warp_blend_u8(const uint8_t* cur, const uint8_t* prev, const float* flow, uint8_t* out, int h, int w, int c,
                         float alpha, float one_minus_alpha, hipStream_t s);
int launch_quantize_u8(const float* in_nchw, uint8_t* out_nhwc, int n, int c, int h, int w, hipStream_t s);
int launch_u8_to_f32(const uint8_t* in_nhwc, float* out_nchw, int n, int c, int h, int w, hipStream_t s);
int launch_resize_area_u8(const uint8_t* in, uint8_t* out, int n, int hi, int wi, int c, int ho, int wo, hipStream_t s);
int launch_composite_quantize_u8(const uint8_t* content_nhwc, const float* stylized_nchw, const void* mask, int mask_is_float,
                                 int mask_c, int mask_n, uint8_t* out_nhwc, int n, int hw, hipStream_t s);
int launch_composite_quantize_u8_nearest(const uint8_t* content_nhwc, const float* stylized_nchw, const void* mask, int mask_is_float,
                                         int mask_c, int mask_n, int mh, int mw, uint8_t* out_nhwc, int n, int h, int w, hipStream_t s);
int launch_mask_to_f32(const uint8_t* in, float* out, size_t total, hipStream_t s);
int launch_nhwc_to_nchw(const float* in, float* out, int n, int c, int hw, hipStream_t s);
int launch_nchw_to_nhwc(const float* in, float* out, int n, int c, int hw, hipStream_t s);

// resample.hip: PIL.Image.resize(size, BILINEAR) on uint8 RGB, bit-exact (Pillow's ImagingResample)
size_t resize_pil_workspace_bytes(int hi, int wi, int ho, int wo);
int launch_resize_pil_bilinear_u8(const uint8_t* in, int pixel_bytes, int n, int hi, int wi, uint8_t* out, int ho, int wo, int y0, int x0, int ch,
                                  int cw, void* workspace, size_t ws_bytes, hipStream_t s);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return -2;
    }
    return 0;
}

}  // namespace adain
