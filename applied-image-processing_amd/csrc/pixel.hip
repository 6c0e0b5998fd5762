// Pixel kernels around the conv path (HBM-bound, coalesced along x).
//
//   strength map     compute_stylization_strength_map (Style_3DGS/AdaIN/test.py:119-150): bicubic
//                    resize (A = -0.75, align_corners=False) -> min/max -> normalise -> subtract
//                    mean -> sigmoid(prominence * P) -> clamp(max = 1 - offset).  The reference's host
//                    branch `if max_val > min_val` (test.py:138) is taken on the device: no host sync.
//   resize_bilinear / resize_nearest / mask_composite   the content-mask composite of
//                    adain_inference (test.py:222-236).
//   quantize_u8      torchvision save_image's quantiser (test.py:243-244): x*255 + 0.5, clamp, u8.
//   nhwc <-> nchw    layout changes at the boundary (relu4_1 features are NCHW for callers).
//
// The resize index/weight arithmetic follows ATen's CPU kernels (UpSampleKernel.cpp:
// area_pixel_compute_source_index, get_cubic_upsample_coefficients, nearest_neighbor_compute_source_index)
// in fp32, including the accumulation order (inner sum over x, outer over y).
#include "common.h"

namespace adain {

#pragma clang fp contract(off)

__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

__device__ __forceinline__ void cubic_coeffs(float t, float w[4]) {
    const float A = -0.75f;
    w[0] = cubic2(t + 1.f, A);
    w[1] = cubic1(t, A);
    w[2] = cubic1(1.f - t, A);
    w[3] = cubic2(2.f - t, A);
}

// K1: bicubic resize + per-block min/max partials
__global__ __launch_bounds__(256) void bicubic_minmax_kernel(const float* __restrict__ in, int h0, int w0, int hc, int wc,
                                                             float* __restrict__ out, float* __restrict__ part) {
    const float sy = (float)h0 / (float)hc, sx = (float)w0 / (float)wc;
    const int total = hc * wc;
    float lo = INFINITY, hi = -INFINITY;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int oy = i / wc, ox = i - oy * wc;
        const float ry = sy * ((float)oy + 0.5f) - 0.5f, rx = sx * ((float)ox + 0.5f) - 0.5f;
        const float fy = floorf(ry), fx = floorf(rx);
        const int iy = (int)fy, ix = (int)fx;
        float wy[4], wx[4];
        cubic_coeffs(ry - fy, wy);
        cubic_coeffs(rx - fx, wx);
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int yy = min(max(iy - 1 + a, 0), h0 - 1);
            float row = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int xx = min(max(ix - 1 + b, 0), w0 - 1);
                row += wx[b] * in[(size_t)yy * w0 + xx];
            }
            acc += wy[a] * row;
        }
        out[i] = acc;
        lo = fminf(lo, acc);
        hi = fmaxf(hi, acc);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_down(lo, off, 64));
        hi = fmaxf(hi, __shfl_down(hi, off, 64));
    }
    __shared__ float sh[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sh[wave * 2] = lo; sh[wave * 2 + 1] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { lo = fminf(lo, sh[w * 2]); hi = fmaxf(hi, sh[w * 2 + 1]); }
        part[blockIdx.x * 2] = lo;
        part[blockIdx.x * 2 + 1] = hi;
    }
}

// K2 (one block): global min/max, mean of the normalised map (fp64 sum, fixed order), final map in place.
__global__ __launch_bounds__(1024) void strength_finish_kernel(float* __restrict__ p, int total, const float* __restrict__ part,
                                                               int nparts, float offset, float prominence) {
    __shared__ float shf[2];
    __shared__ double shd[16];
    const int tid = threadIdx.x;
    if (tid == 0) {
        float lo = INFINITY, hi = -INFINITY;
        for (int i = 0; i < nparts; ++i) { lo = fminf(lo, part[i * 2]); hi = fmaxf(hi, part[i * 2 + 1]); }
        shf[0] = lo; shf[1] = hi;
    }
    __syncthreads();
    const float lo = shf[0], hi = shf[1];
    if (!(hi > lo)) {   // constant map -> zeros (test.py:141-143)
        for (int i = tid; i < total; i += 1024) p[i] = 0.f;
        return;
    }
    const float range = hi - lo;
    double s = 0;
    for (int i = tid; i < total; i += 1024) s += (double)((p[i] - lo) / range);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((tid & 63) == 0) shd[tid >> 6] = s;
    __syncthreads();
    double ts = 0;
    for (int w = 0; w < 16; ++w) ts += shd[w];
    const float mean = (float)(ts / total);
    const float cap = 1.0f - offset;
    for (int i = tid; i < total; i += 1024) {
        const float v = (p[i] - lo) / range - mean;
        const float sg = 1.0f / (1.0f + expf(-prominence * v));
        p[i] = fminf(sg, cap);
    }
}

constexpr int SM_BLOCKS = 64;
size_t strength_map_workspace_bytes(int, int) { return SM_BLOCKS * 2 * sizeof(float); }

int launch_strength_map(const float* depth, int h0, int w0, int hc, int wc, float offset, float prominence, float* pmap,
                        void* workspace, size_t ws_bytes, hipStream_t s) {
    if (h0 < 1 || w0 < 1 || hc < 1 || wc < 1) { set_error("strength_map: bad shape"); return -1; }
    if (!workspace || ws_bytes < strength_map_workspace_bytes(hc, wc)) { set_error("strength_map: workspace too small"); return -1; }
    const int total = hc * wc;
    int blocks = (total + 255) / 256;
    if (blocks > SM_BLOCKS) blocks = SM_BLOCKS;
    hipLaunchKernelGGL(bicubic_minmax_kernel, dim3(blocks), dim3(256), 0, s, depth, h0, w0, hc, wc, pmap, (float*)workspace);
    hipLaunchKernelGGL(strength_finish_kernel, dim3(1), dim3(1024), 0, s, pmap, total, (const float*)workspace, blocks, offset, prominence);
    return check_launch("strength_map");
}

// ---- bilinear / nearest resize over `planes` independent [hi][wi] planes -------------------------------
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, int hi,
                                                              int wi, int ho, int wo, size_t total) {
    const float sy = (float)hi / (float)ho, sx = (float)wi / (float)wo;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % wo);
        const size_t r = i / wo;
        const int oy = (int)(r % ho);
        const size_t pl = r / ho;
        const float ry = fmaxf(sy * ((float)oy + 0.5f) - 0.5f, 0.f), rx = fmaxf(sx * ((float)ox + 0.5f) - 0.5f, 0.f);
        const int y0 = min((int)ry, hi - 1), x0 = min((int)rx, wi - 1);
        const int y1 = y0 + (y0 < hi - 1 ? 1 : 0), x1 = x0 + (x0 < wi - 1 ? 1 : 0);
        const float ly1 = fminf(fmaxf(ry - (float)y0, 0.f), 1.f), lx1 = fminf(fmaxf(rx - (float)x0, 0.f), 1.f);
        const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const float* __restrict__ p = in + pl * (size_t)hi * wi;
        const float top = lx0 * p[(size_t)y0 * wi + x0] + lx1 * p[(size_t)y0 * wi + x1];
        const float bot = lx0 * p[(size_t)y1 * wi + x0] + lx1 * p[(size_t)y1 * wi + x1];
        out[i] = ly0 * top + ly1 * bot;
    }
}

__global__ __launch_bounds__(256) void resize_nearest_kernel(const float* __restrict__ in, float* __restrict__ out, int hi,
                                                             int wi, int ho, int wo, size_t total) {
    const float sy = (float)hi / (float)ho, sx = (float)wi / (float)wo;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % wo);
        const size_t r = i / wo;
        const int oy = (int)(r % ho);
        const size_t pl = r / ho;
        const int y = min((int)floorf((float)oy * sy), hi - 1), x = min((int)floorf((float)ox * sx), wi - 1);
        out[i] = in[pl * (size_t)hi * wi + (size_t)y * wi + x];
    }
}

static unsigned grid_for(size_t total) {
    const size_t b = (total + 255) / 256;
    return (unsigned)(b < 8192 ? (b ? b : 1) : 8192);
}

int launch_resize_bilinear(const float* in, float* out, int planes, int hi, int wi, int ho, int wo, hipStream_t s) {
    if (planes < 1 || hi < 1 || wi < 1 || ho < 1 || wo < 1) { set_error("resize_bilinear: bad shape"); return -1; }
    const size_t total = (size_t)planes * ho * wo;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, out, hi, wi, ho, wo, total);
    return check_launch("resize_bilinear");
}

int launch_resize_nearest(const float* in, float* out, int planes, int hi, int wi, int ho, int wo, hipStream_t s) {
    if (planes < 1 || hi < 1 || wi < 1 || ho < 1 || wo < 1) { set_error("resize_nearest: bad shape"); return -1; }
    const size_t total = (size_t)planes * ho * wo;
    hipLaunchKernelGGL(resize_nearest_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, out, hi, wi, ho, wo, total);
    return check_launch("resize_nearest");
}

// out = content * (1 - m) + stylized * m   (test.py:236); mask has 1 or c channels, batch 1 or n
__global__ __launch_bounds__(256) void mask_composite_kernel(const float* __restrict__ content, const float* __restrict__ sty,
                                                             const float* __restrict__ mask, int mask_c, int mask_n,
                                                             float* __restrict__ out, int c, int hw, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int pix = (int)(i % hw);
        const size_t r = i / hw;
        const int ch = (int)(r % c);
        const int img = (int)(r / c);
        const float m = mask[((size_t)(mask_n == 1 ? 0 : img) * mask_c + (mask_c == 1 ? 0 : ch)) * hw + pix];
        out[i] = content[i] * (1.0f - m) + sty[i] * m;
    }
}

int launch_mask_composite(const float* content, const float* stylized, const float* mask, int mask_c, int mask_n, float* out,
                          int n, int c, int hw, hipStream_t s) {
    if (n < 1 || c < 1 || hw < 1) { set_error("mask_composite: bad shape"); return -1; }
    if (mask_c != 1 && mask_c != c) { set_error("mask_composite: mask channels %d must be 1 or %d", mask_c, c); return -1; }
    if (mask_n != 1 && mask_n != n) { set_error("mask_composite: mask batch %d must be 1 or %d", mask_n, n); return -1; }
    const size_t total = (size_t)n * c * hw;
    hipLaunchKernelGGL(mask_composite_kernel, dim3(grid_for(total)), dim3(256), 0, s, content, stylized, mask, mask_c, mask_n, out, c, hw, total);
    return check_launch("mask_composite");
}

// NCHW float -> NHWC u8, x*255 + 0.5 clamped to [0,255] then truncated (torchvision save_image)
__global__ __launch_bounds__(256) void quantize_u8_kernel(const float* __restrict__ in, uint8_t* __restrict__ out, int c, int hw,
                                                          size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % c);
        const size_t r = i / c;
        const int pix = (int)(r % hw);
        const size_t img = r / hw;
        float v = in[(img * c + ch) * (size_t)hw + pix] * 255.0f + 0.5f;
        v = fminf(fmaxf(v, 0.f), 255.f);
        out[i] = (uint8_t)v;
    }
}

int launch_quantize_u8(const float* in, uint8_t* out, int n, int c, int h, int w, hipStream_t s) {
    if (n < 1 || c < 1 || h < 1 || w < 1) { set_error("quantize_u8: bad shape"); return -1; }
    const size_t total = (size_t)n * c * h * w;
    hipLaunchKernelGGL(quantize_u8_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, out, c, h * w, total);
    return check_launch("quantize_u8");
}

// ---- video post-pass (reference video/utils.py:89-105 warp_image, :223-229 blend_images) -------------------
// out = u8( clip( (alpha * cur/255 + (1 - alpha) * warp(prev)/255) * 255, 0, 255 ) ), HWC uint8 frames;
// warp(prev)(y, x) = bilinear sample of prev at (x + flow[0][y][x], y + flow[1][y][x]) with cv2.BORDER_REFLECT
// (fedcba|abcdefgh|hgfedcb), in cv2.remap's own uint8 fixed-point arithmetic (below).  cv2 is not installed in the build
// image, so the fixed-point restatement is checked against the oracle only (parity unpinned against OpenCV itself).
__device__ __forceinline__ int reflect_border(int v, int n) {
    const int p = 2 * n;      // BORDER_REFLECT has period 2n: ...cba|abc...xyz|zyx...
    v %= p;
    if (v < 0) v += p;
    return v < n ? v : p - 1 - v;
}

// cv2.remap(uint8, float maps, INTER_LINEAR) is fixed point (OpenCV imgwarp: INTER_BITS = 5, INTER_REMAP_COEF_BITS = 15):
// the map is rounded half-to-even to 1/32 pixel, the four weights are (32-fx)(32-fy)*32 ... (they sum to 2^15; the one
// saturated entry {32767,0,0,1} at fx = fy = 0 cannot change a rounded result) and the pixel is
// (sum(S*w) + 2^14) >> 15.  Integer from the map rounding on, so the warp is bit-exact against the oracle; the blend that
// follows is the reference's float32 numpy expression evaluated without contraction.
__device__ __forceinline__ int clamp_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void warp_blend_u8_kernel(const uint8_t* __restrict__ cur, const uint8_t* __restrict__ prev,
                                                            const float* __restrict__ flow, uint8_t* __restrict__ out, int h, int w,
                                                            int c, float alpha, float one_minus_alpha) {
    const int total = h * w;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int y = i / w, x = i - y * w;
        const float mx = (float)x + flow[i], my = (float)y + flow[(size_t)total + i];
        const int ix = __float2int_rn(mx * 32.0f), iy = __float2int_rn(my * 32.0f);
        const int sx = clamp_short(ix >> 5), sy = clamp_short(iy >> 5), fx = ix & 31, fy = iy & 31;
        const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
        const int x0 = reflect_border(sx, w), x1 = reflect_border(sx + 1, w);
        const int y0 = reflect_border(sy, h), y1 = reflect_border(sy + 1, h);
        for (int ch = 0; ch < c; ++ch) {
            const int p00 = prev[((size_t)y0 * w + x0) * c + ch], p01 = prev[((size_t)y0 * w + x1) * c + ch];
            const int p10 = prev[((size_t)y1 * w + x0) * c + ch], p11 = prev[((size_t)y1 * w + x1) * c + ch];
            const int wv = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15;          // <= 255 by construction
            const float a = alpha * ((float)cur[(size_t)i * c + ch] / 255.0f);
            const float bq = one_minus_alpha * ((float)wv / 255.0f);
            const float b = (a + bq) * 255.0f;
            out[(size_t)i * c + ch] = (uint8_t)fminf(fmaxf(b, 0.f), 255.f);
        }
    }
}

int launch_warp_blend_u8(const uint8_t* cur, const uint8_t* prev, const float* flow, uint8_t* out, int h, int w, int c, float alpha,
                         float one_minus_alpha, hipStream_t s) {
    if (h < 1 || w < 1 || c < 1) { set_error("warp_blend_u8: bad shape"); return -1; }
    hipLaunchKernelGGL(warp_blend_u8_kernel, dim3(grid_for((size_t)h * w)), dim3(256), 0, s, cur, prev, flow, out, h, w, c, alpha,
                       one_minus_alpha);
    return check_launch("warp_blend_u8");
}

// ---- layout transposes through a 32x33 LDS tile: [n][R][C] -> [n][C][R] ------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C) {
    __shared__ float tile[32][33];
    const int img = blockIdx.y;
    const int tiles_c = (C + 31) / 32;
    const int c0 = (blockIdx.x % tiles_c) * 32, r0 = (blockIdx.x / tiles_c) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* __restrict__ src = in + (size_t)img * R * C;
    float* __restrict__ dst = out + (size_t)img * R * C;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + k * 8, cc = c0 + tx;
        if (r < R && cc < C) tile[ty + k * 8][tx] = src[(size_t)r * C + cc];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int cc = c0 + ty + k * 8, r = r0 + tx;
        if (r < R && cc < C) dst[(size_t)cc * R + r] = tile[tx][ty + k * 8];
    }
}

static int launch_transpose(const float* in, float* out, int n, int R, int C, hipStream_t s, const char* what) {
    if (n < 1 || R < 1 || C < 1) { set_error("%s: bad shape", what); return -1; }
    const size_t tiles = (size_t)((C + 31) / 32) * ((R + 31) / 32);
    if (tiles > 0x7fffffffULL || n > 65535) { set_error("%s: grid too large", what); return -1; }
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)tiles, n), dim3(256), 0, s, in, out, R, C);
    return check_launch(what);
}

int launch_nhwc_to_nchw(const float* in, float* out, int n, int c, int hw, hipStream_t s) {
    return launch_transpose(in, out, n, hw, c, s, "nhwc_to_nchw");
}
int launch_nchw_to_nhwc(const float* in, float* out, int n, int c, int hw, hipStream_t s) {
    return launch_transpose(in, out, n, c, hw, s, "nchw_to_nhwc");
}

}  // namespace adain
