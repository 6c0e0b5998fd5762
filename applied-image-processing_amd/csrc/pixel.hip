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
#include "device_utils.h"

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

// K2 / K3: global min/max from K1's partials (every block re-reduces the <= 64 pairs in the same fixed order), then
//   K2: per-block fp64 partial sums of the normalised map (fixed slice per block, fixed combine order);
//   K3: mean = (sum of the partials in index order) / total, final map in place.
// Both are multi-block: at a 256 x 256 map the former single 1024-thread block was a serial 15-20 us tail per frame.
__device__ __forceinline__ void strength_minmax(const float* __restrict__ part, int nparts, float* shf) {
    if (threadIdx.x == 0) {
        float lo = INFINITY, hi = -INFINITY;
        for (int i = 0; i < nparts; ++i) { lo = fminf(lo, part[i * 2]); hi = fmaxf(hi, part[i * 2 + 1]); }
        shf[0] = lo; shf[1] = hi;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void strength_sum_kernel(const float* __restrict__ p, int total, const float* __restrict__ part,
                                                           int nparts, double* __restrict__ psum) {
    __shared__ float shf[2];
    __shared__ double shd[4];
    strength_minmax(part, nparts, shf);
    const float lo = shf[0], hi = shf[1];
    double s = 0;
    if (hi > lo) {
        const float range = hi - lo;
        // block b owns the contiguous slice [b * per, (b + 1) * per): the partition depends only on (total, gridDim)
        const int per = (total + gridDim.x - 1) / gridDim.x;
        const int i0 = blockIdx.x * per, i1 = min(i0 + per, total);
        for (int i = i0 + threadIdx.x; i < i1; i += 256) s += (double)((p[i] - lo) / range);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) psum[blockIdx.x] = (shd[0] + shd[1]) + (shd[2] + shd[3]);
}

__global__ __launch_bounds__(256) void strength_apply_kernel(float* __restrict__ p, int total, const float* __restrict__ part,
                                                             int nparts, const double* __restrict__ psum, int nsum, float offset,
                                                             float prominence) {
    __shared__ float shf[3];
    strength_minmax(part, nparts, shf);
    if (threadIdx.x == 0) {
        double ts = 0;
        for (int i = 0; i < nsum; ++i) ts += psum[i];
        shf[2] = (float)(ts / total);
    }
    __syncthreads();
    const float lo = shf[0], hi = shf[1], mean = shf[2];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    if (!(hi > lo)) {   // constant map -> zeros (test.py:141-143)
        p[i] = 0.f;
        return;
    }
    const float range = hi - lo, cap = 1.0f - offset;
    const float v = (p[i] - lo) / range - mean;
    const float sg = 1.0f / (1.0f + expf(-prominence * v));
    p[i] = fminf(sg, cap);
}

constexpr int SM_BLOCKS = 64;
size_t strength_map_workspace_bytes(int, int) { return SM_BLOCKS * 2 * sizeof(float) + SM_BLOCKS * sizeof(double); }

int launch_strength_map(const float* depth, int h0, int w0, int hc, int wc, float offset, float prominence, float* pmap,
                        void* workspace, size_t ws_bytes, hipStream_t s) {
    if (h0 < 1 || w0 < 1 || hc < 1 || wc < 1) { set_error("strength_map: bad shape"); return -1; }
    if ((size_t)hc * wc >= 0x7fffffffULL || (size_t)h0 * w0 >= 0x7fffffffULL) { set_error("strength_map: map too large"); return -1; }
    if (!workspace || ws_bytes < strength_map_workspace_bytes(hc, wc)) { set_error("strength_map: workspace too small"); return -1; }
    const int total = hc * wc;
    int blocks = (total + 255) / 256;
    if (blocks > SM_BLOCKS) blocks = SM_BLOCKS;
    float* part = (float*)workspace;
    double* psum = (double*)(part + SM_BLOCKS * 2);
    hipLaunchKernelGGL(bicubic_minmax_kernel, dim3(blocks), dim3(256), 0, s, depth, h0, w0, hc, wc, pmap, part);
    hipLaunchKernelGGL(strength_sum_kernel, dim3(blocks), dim3(256), 0, s, pmap, total, part, blocks, psum);
    hipLaunchKernelGGL(strength_apply_kernel, dim3((total + 255) / 256), dim3(256), 0, s, pmap, total, part, blocks, psum, blocks, offset,
                       prominence);
    return check_launch("strength_map");
}

// ---- bilinear / nearest resize over `planes` independent [hi][wi] planes -------------------------------
// Launch geometry of the row kernels: block = 64 x 4 threads, a thread owns 4 consecutive output pixels of one row
// (blockIdx.y = row group, blockIdx.z = plane): no per-thread division, 32-bit offsets inside a plane, one b128 store per
// thread when the row length is a multiple of 4 (scalar stores otherwise).  The index / weight arithmetic is ATen's.
__device__ __forceinline__ void store_quad(float* __restrict__ row, int x, int wo, bool vec, const f32x4 v) {
    if (vec) {
        *(f32x4*)(row + x) = v;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (x + k < wo) row[x + k] = v[k];
    }
}

// SAMEW: wi == wo, so the x scale is exactly 1, rx == ox, x0 == ox and lx1 == 0: the four outputs of a thread read the b128 at
// their own columns plus one more element (the x1 taps still enter the sum with weight 0, as in ATen, so a NaN / Inf neighbour
// propagates identically).
template <bool SAMEW>
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, int hi,
                                                              int wi, int ho, int wo, int vec) {
    const int x = (blockIdx.x * 64 + threadIdx.x) * 4, oy = blockIdx.y * 4 + threadIdx.y;
    if (x >= wo || oy >= ho) return;
    const float sy = (float)hi / (float)ho, sx = (float)wi / (float)wo;
    const float ry = fmaxf(sy * ((float)oy + 0.5f) - 0.5f, 0.f);
    const int y0 = min((int)ry, hi - 1), y1 = y0 + (y0 < hi - 1 ? 1 : 0);
    const float ly1 = fminf(fmaxf(ry - (float)y0, 0.f), 1.f), ly0 = 1.f - ly1;
    const float* __restrict__ p0 = in + (size_t)blockIdx.z * hi * wi + (unsigned)(y0 * wi);
    const float* __restrict__ p1 = in + (size_t)blockIdx.z * hi * wi + (unsigned)(y1 * wi);
    f32x4 r;
    if (SAMEW) {
        const f32x4 a0 = *(const f32x4*)(p0 + x), a1 = *(const f32x4*)(p1 + x);
        const int xe = min(x + 4, wi - 1);
        const float e0 = p0[xe], e1 = p1[xe];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float n0 = k < 3 ? a0[k < 3 ? k + 1 : 3] : e0, n1 = k < 3 ? a1[k < 3 ? k + 1 : 3] : e1;
            const float top = 1.f * a0[k] + 0.f * n0;
            const float bot = 1.f * a1[k] + 0.f * n1;
            r[k] = ly0 * top + ly1 * bot;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ox = min(x + k, wo - 1);
            const float rx = fmaxf(sx * ((float)ox + 0.5f) - 0.5f, 0.f);
            const int x0 = min((int)rx, wi - 1), x1 = x0 + (x0 < wi - 1 ? 1 : 0);
            const float lx1 = fminf(fmaxf(rx - (float)x0, 0.f), 1.f), lx0 = 1.f - lx1;
            const float top = lx0 * p0[x0] + lx1 * p0[x1];
            const float bot = lx0 * p1[x0] + lx1 * p1[x1];
            r[k] = ly0 * top + ly1 * bot;
        }
    }
    store_quad(out + ((size_t)blockIdx.z * ho + oy) * wo, x, wo, vec, r);
}

__global__ __launch_bounds__(256) void resize_nearest_kernel(const float* __restrict__ in, float* __restrict__ out, int hi,
                                                             int wi, int ho, int wo, int vec) {
    const int x = (blockIdx.x * 64 + threadIdx.x) * 4, oy = blockIdx.y * 4 + threadIdx.y;
    if (x >= wo || oy >= ho) return;
    const float sy = (float)hi / (float)ho, sx = (float)wi / (float)wo;
    const int y = min((int)floorf((float)oy * sy), hi - 1);
    const float* __restrict__ p = in + (size_t)blockIdx.z * hi * wi + (unsigned)(y * wi);
    f32x4 r;
    if (wi == wo && vec) {                 // same width: source column == output column, one b128 load
        r = *(const f32x4*)(p + x);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = p[min((int)floorf((float)min(x + k, wo - 1) * sx), wi - 1)];
    }
    store_quad(out + ((size_t)blockIdx.z * ho + oy) * wo, x, wo, vec, r);
}

static unsigned grid_for(size_t total) {
    const size_t b = (total + 255) / 256;
    return (unsigned)(b < 8192 ? (b ? b : 1) : 8192);
}

static int row_grid(const char* what, int planes, int ho, int wo, dim3* g) {
    const long long gx = ((long long)wo + 255) / 256, gy = ((long long)ho + 3) / 4;
    if (gy > 65535 || planes > 65535) { set_error("%s: more than 262140 rows or 65535 planes per call", what); return -1; }
    if ((size_t)ho * wo >= 0x7fffffffULL) { set_error("%s: a plane must stay below 2^31 pixels", what); return -1; }
    *g = dim3((unsigned)gx, (unsigned)gy, (unsigned)planes);
    return 0;
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

int launch_resize_bilinear(const float* in, float* out, int planes, int hi, int wi, int ho, int wo, hipStream_t s) {
    if (planes < 1 || hi < 1 || wi < 1 || ho < 1 || wo < 1) { set_error("resize_bilinear: bad shape"); return -1; }
    if ((size_t)hi * wi >= 0x7fffffffULL) { set_error("resize_bilinear: a plane must stay below 2^31 pixels"); return -1; }
    dim3 g;
    if (row_grid("resize_bilinear", planes, ho, wo, &g)) return -1;
    const int vec = (int)(wo % 4 == 0 && aligned16(out));
    if (wi == wo && vec && aligned16(in)) hipLaunchKernelGGL(resize_bilinear_kernel<true>, g, dim3(64, 4), 0, s, in, out, hi, wi, ho, wo, vec);
    else hipLaunchKernelGGL(resize_bilinear_kernel<false>, g, dim3(64, 4), 0, s, in, out, hi, wi, ho, wo, vec);
    return check_launch("resize_bilinear");
}

int launch_resize_nearest(const float* in, float* out, int planes, int hi, int wi, int ho, int wo, hipStream_t s) {
    if (planes < 1 || hi < 1 || wi < 1 || ho < 1 || wo < 1) { set_error("resize_nearest: bad shape"); return -1; }
    if ((size_t)hi * wi >= 0x7fffffffULL) { set_error("resize_nearest: a plane must stay below 2^31 pixels"); return -1; }
    dim3 g;
    if (row_grid("resize_nearest", planes, ho, wo, &g)) return -1;
    hipLaunchKernelGGL(resize_nearest_kernel, g, dim3(64, 4), 0, s, in, out, hi, wi, ho, wo,
                       (int)(wo % 4 == 0 && aligned16(out) && aligned16(in)));
    return check_launch("resize_nearest");
}

// out = content * (1 - m) + stylized * m   (test.py:236); mask has 1 or c channels, batch 1 or n.
// blockIdx.y = plane (img * c + ch): the mask plane is block-uniform; VEC: one b128 per operand per thread.
template <bool VEC>
__global__ __launch_bounds__(256) void mask_composite_kernel(const float* __restrict__ content, const float* __restrict__ sty,
                                                             const float* __restrict__ mask, int mask_c, int mask_n,
                                                             float* __restrict__ out, int c, int hw) {
    const unsigned plane = blockIdx.y, img = plane / (unsigned)c, ch = plane - img * (unsigned)c;
    const size_t base = (size_t)plane * hw;
    const float* __restrict__ mp = mask + ((size_t)(mask_n == 1 ? 0u : img) * mask_c + (mask_c == 1 ? 0u : ch)) * hw;
    const int i = (blockIdx.x * 256 + threadIdx.x) * (VEC ? 4 : 1);
    if (i >= hw) return;
    if (VEC) {
        const f32x4 m = *(const f32x4*)(mp + i), a = *(const f32x4*)(content + base + i), b = *(const f32x4*)(sty + base + i);
        f32x4 r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = a[k] * (1.0f - m[k]) + b[k] * m[k];
        *(f32x4*)(out + base + i) = r;
    } else {
        const float m = mp[i];
        out[base + i] = content[base + i] * (1.0f - m) + sty[base + i] * m;
    }
}

int launch_mask_composite(const float* content, const float* stylized, const float* mask, int mask_c, int mask_n, float* out,
                          int n, int c, int hw, hipStream_t s) {
    if (n < 1 || c < 1 || hw < 1) { set_error("mask_composite: bad shape"); return -1; }
    if (mask_c != 1 && mask_c != c) { set_error("mask_composite: mask channels %d must be 1 or %d", mask_c, c); return -1; }
    if (mask_n != 1 && mask_n != n) { set_error("mask_composite: mask batch %d must be 1 or %d", mask_n, n); return -1; }
    if ((long long)n * c > 65535) { set_error("mask_composite: more than 65535 planes per call"); return -1; }
    const bool vec = hw % 4 == 0 && aligned16(content) && aligned16(stylized) && aligned16(mask) && aligned16(out);
    const unsigned gx = (unsigned)(((size_t)hw / (vec ? 4 : 1) + 255) / 256);
    if (vec)
        hipLaunchKernelGGL(mask_composite_kernel<true>, dim3(gx, n * c), dim3(256), 0, s, content, stylized, mask, mask_c, mask_n, out, c, hw);
    else
        hipLaunchKernelGGL(mask_composite_kernel<false>, dim3(gx, n * c), dim3(256), 0, s, content, stylized, mask, mask_c, mask_n, out, c, hw);
    return check_launch("mask_composite");
}

// NCHW float -> NHWC u8, x*255 + 0.5 clamped to [0,255] then truncated (torchvision save_image)
__device__ __forceinline__ unsigned quant1(float x) {
    float v = x * 255.0f + 0.5f;
    v = fminf(fmaxf(v, 0.f), 255.f);
    return (unsigned)v;
}

// c == 3, hw % 4 == 0: a thread owns 4 pixels = three b128 plane loads -> 12 packed bytes (one 96-bit store); blockIdx.y = image
__global__ __launch_bounds__(256) void quantize_u8_rgb4_kernel(const float* __restrict__ in, uint8_t* __restrict__ out, int hw) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q * 4 >= hw) return;
    const float* __restrict__ p = in + (size_t)blockIdx.y * 3 * hw + q * 4;
    const f32x4 r = *(const f32x4*)p, g = *(const f32x4*)(p + hw), b = *(const f32x4*)(p + 2 * (size_t)hw);
    unsigned by[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) { by[3 * k] = quant1(r[k]); by[3 * k + 1] = quant1(g[k]); by[3 * k + 2] = quant1(b[k]); }
    using u32x3 = __attribute__((ext_vector_type(3))) unsigned;
    u32x3 w;
#pragma unroll
    for (int d = 0; d < 3; ++d) w[d] = by[4 * d] | (by[4 * d + 1] << 8) | (by[4 * d + 2] << 16) | (by[4 * d + 3] << 24);
    *(u32x3*)(out + ((size_t)blockIdx.y * hw + q * 4) * 3) = w;
}

__global__ __launch_bounds__(256) void quantize_u8_kernel(const float* __restrict__ in, uint8_t* __restrict__ out, int c, int hw) {
    // generic channel count: a thread owns one pixel of image blockIdx.y
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= hw) return;
    const float* __restrict__ p = in + (size_t)blockIdx.y * c * hw + pix;
    uint8_t* __restrict__ o = out + ((size_t)blockIdx.y * hw + pix) * c;
    for (int ch = 0; ch < c; ++ch) o[ch] = (uint8_t)quant1(p[(size_t)ch * hw]);
}

int launch_quantize_u8(const float* in, uint8_t* out, int n, int c, int h, int w, hipStream_t s) {
    if (n < 1 || c < 1 || h < 1 || w < 1) { set_error("quantize_u8: bad shape"); return -1; }
    if ((size_t)h * w * c >= 0x7fffffffULL || n > 65535) { set_error("quantize_u8: image too large (2^31 elements) or batch > 65535"); return -1; }
    const int hw = h * w;
    if (c == 3 && hw % 4 == 0 && aligned16(in) && ((uintptr_t)out & 3) == 0)
        hipLaunchKernelGGL(quantize_u8_rgb4_kernel, dim3((hw / 4 + 255) / 256, n), dim3(256), 0, s, in, out, hw);
    else
        hipLaunchKernelGGL(quantize_u8_kernel, dim3((hw + 255) / 256, n), dim3(256), 0, s, in, out, c, hw);
    return check_launch("quantize_u8");
}

// torchvision ToTensor on the device (reference test.py:22 via :203-204): NHWC u8 [n][hw][c] -> NCHW float [n][c][hw], float(v) / 255
// with a correctly rounded division, i.e. bit for bit what `tensor.float() / 255` gives on the host.  The job drivers upload
// frames as uint8 (3 bytes per pixel across PCIe instead of 12) and convert here (or inside conv_first_kernel<true>).
// c == 3, hw % 4 == 0: a thread owns 4 pixels = one 96-bit load -> three b128 plane stores; blockIdx.y = image
__global__ __launch_bounds__(256) void u8_to_f32_rgb4_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int hw) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q * 4 >= hw) return;
    using u32x3 = __attribute__((ext_vector_type(3))) unsigned;
    const u32x3 w = *(const u32x3*)(in + ((size_t)blockIdx.y * hw + q * 4) * 3);
    f32x4 pl[3];
#pragma unroll
    for (int k = 0; k < 12; ++k) pl[k % 3][k / 3] = __fdiv_rn((float)((w[k >> 2] >> (8 * (k & 3))) & 255u), 255.0f);
    float* __restrict__ p = out + (size_t)blockIdx.y * 3 * hw + q * 4;
    *(f32x4*)p = pl[0];
    *(f32x4*)(p + hw) = pl[1];
    *(f32x4*)(p + 2 * (size_t)hw) = pl[2];
}

__global__ __launch_bounds__(256) void u8_to_f32_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int c, int hw) {
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= hw) return;
    const uint8_t* __restrict__ p = in + ((size_t)blockIdx.y * hw + pix) * c;
    float* __restrict__ o = out + (size_t)blockIdx.y * c * hw + pix;
    for (int ch = 0; ch < c; ++ch) o[(size_t)ch * hw] = __fdiv_rn((float)p[ch], 255.0f);
}

int launch_u8_to_f32(const uint8_t* in, float* out, int n, int c, int h, int w, hipStream_t s) {
    if (n < 1 || c < 1 || h < 1 || w < 1) { set_error("u8_to_f32: bad shape"); return -1; }
    if ((size_t)h * w * c >= 0x7fffffffULL || n > 65535) { set_error("u8_to_f32: image too large (2^31 elements) or batch > 65535"); return -1; }
    const int hw = h * w;
    if (c == 3 && hw % 4 == 0 && aligned16(out) && ((uintptr_t)in & 3) == 0)
        hipLaunchKernelGGL(u8_to_f32_rgb4_kernel, dim3((hw / 4 + 255) / 256, n), dim3(256), 0, s, in, out, hw);
    else
        hipLaunchKernelGGL(u8_to_f32_kernel, dim3((hw + 255) / 256, n), dim3(256), 0, s, in, out, c, hw);
    return check_launch("u8_to_f32");
}

// ---- the tail of a masked frame in one pass (adain_stylize_u8): ToTensor(content) -> content*(1-m) + stylized*m -> save_image's
// quantiser (test.py:203-204, :236, :243-244) when mask, stylised frame and content already share one size.  The same
// arithmetic, operation by operation, as u8_to_f32 -> mask.float() -> mask_composite -> quantize_u8 (this file compiles with
// fp contraction off), so the bytes are identical to that sequence while a pixel moves 3 + 12 + 3|12 + 3 bytes instead of ~80.
// content NHWC u8 [n][hw][3]; sty NCHW f32 [n][3][hw]; mask [mask_n][mask_c][hw] of MaskT (uint8 / bool bytes -> float(v), or
// float); out NHWC u8 [n][hw][3].  blockIdx.y = image.
template <typename MaskT>
__device__ __forceinline__ float mask_val(const MaskT* __restrict__ p, size_t i) { return (float)p[i]; }

template <typename MaskT, bool VEC4>
__global__ __launch_bounds__(256) void composite_quantize_u8_kernel(const uint8_t* __restrict__ content, const float* __restrict__ sty,
                                                                    const MaskT* __restrict__ mask, int mask_c, int mask_n,
                                                                    uint8_t* __restrict__ out, int hw) {
    const unsigned img = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;
    const MaskT* __restrict__ mp = mask + (size_t)(mask_n == 1 ? 0u : img) * mask_c * hw;
    const size_t mstep = mask_c == 1 ? 0 : (size_t)hw;
    const float* __restrict__ sp = sty + (size_t)img * 3 * hw;
    if (VEC4) {
        if (q * 4 >= hw) return;
        using u32x3 = __attribute__((ext_vector_type(3))) unsigned;
        const u32x3 cw = *(const u32x3*)(content + ((size_t)img * hw + q * 4) * 3);
        unsigned by[12];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const f32x4 b = *(const f32x4*)(sp + (size_t)ch * hw + q * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = 3 * k + ch;
                const float a = __fdiv_rn((float)((cw[e >> 2] >> (8 * (e & 3))) & 255u), 255.0f);
                const float m = mask_val(mp + ch * mstep, (size_t)q * 4 + k);
                by[e] = quant1(a * (1.0f - m) + b[k] * m);
            }
        }
        u32x3 w;
#pragma unroll
        for (int d = 0; d < 3; ++d) w[d] = by[4 * d] | (by[4 * d + 1] << 8) | (by[4 * d + 2] << 16) | (by[4 * d + 3] << 24);
        *(u32x3*)(out + ((size_t)img * hw + q * 4) * 3) = w;
    } else {
        if (q >= hw) return;
        const uint8_t* __restrict__ cp = content + ((size_t)img * hw + q) * 3;
        uint8_t* __restrict__ o = out + ((size_t)img * hw + q) * 3;
        for (int ch = 0; ch < 3; ++ch) {
            const float a = __fdiv_rn((float)cp[ch], 255.0f), m = mask_val(mp + ch * mstep, (size_t)q);
            o[ch] = (uint8_t)quant1(a * (1.0f - m) + sp[(size_t)ch * hw + q] * m);
        }
    }
}

int launch_composite_quantize_u8(const uint8_t* content, const float* stylized, const void* mask, int mask_is_float, int mask_c,
                                 int mask_n, uint8_t* out, int n, int hw, hipStream_t s) {
    if (n < 1 || hw < 1) { set_error("composite_quantize_u8: bad shape"); return -1; }
    if (mask_c != 1 && mask_c != 3) { set_error("composite_quantize_u8: mask channels %d must be 1 or 3", mask_c); return -1; }
    if (mask_n != 1 && mask_n != n) { set_error("composite_quantize_u8: mask batch %d must be 1 or %d", mask_n, n); return -1; }
    if ((size_t)hw * 3 >= 0x7fffffffULL || n > 65535) { set_error("composite_quantize_u8: frame too large or batch > 65535"); return -1; }
    const bool vec = hw % 4 == 0 && aligned16(stylized) && ((uintptr_t)content & 3) == 0 && ((uintptr_t)out & 3) == 0;
    const dim3 g((unsigned)(((size_t)hw / (vec ? 4 : 1) + 255) / 256), n);
    if (mask_is_float) {
        if (vec) hipLaunchKernelGGL((composite_quantize_u8_kernel<float, true>), g, dim3(256), 0, s, content, stylized, (const float*)mask, mask_c, mask_n, out, hw);
        else hipLaunchKernelGGL((composite_quantize_u8_kernel<float, false>), g, dim3(256), 0, s, content, stylized, (const float*)mask, mask_c, mask_n, out, hw);
    } else {
        if (vec) hipLaunchKernelGGL((composite_quantize_u8_kernel<uint8_t, true>), g, dim3(256), 0, s, content, stylized, (const uint8_t*)mask, mask_c, mask_n, out, hw);
        else hipLaunchKernelGGL((composite_quantize_u8_kernel<uint8_t, false>), g, dim3(256), 0, s, content, stylized, (const uint8_t*)mask, mask_c, mask_n, out, hw);
    }
    return check_launch("composite_quantize_u8");
}

// The same tail when only the MASK has another size than the frame (the 3DGS guide loop: the view is resized to content_size, its
// mask arrives at the view's own size, Style_3DGS/train.py:97-101): F.interpolate(mask.float(), size, mode="nearest") is an index
// map - source row min(floor(oy * mh / h), mh - 1), column likewise, in float as ATen computes it (resize_nearest_kernel above) -
// so the mask is sampled in place and the five passes (u8_to_f32, mask_to_f32, resize_nearest, mask_composite, quantize_u8) run
// as one.  A thread owns 4 consecutive pixels of row oy (blockIdx.y * 4 + threadIdx.y) of image blockIdx.z.
template <typename MaskT>
__global__ __launch_bounds__(256) void composite_quantize_u8_nearest_kernel(const uint8_t* __restrict__ content, const float* __restrict__ sty,
                                                                            const MaskT* __restrict__ mask, int mask_c, int mask_n, int mh,
                                                                            int mw, uint8_t* __restrict__ out, int h, int w, int vec) {
    const int x = (blockIdx.x * 64 + threadIdx.x) * 4, oy = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || oy >= h) return;
    const unsigned img = blockIdx.z;
    const float sy = (float)mh / (float)h, sx = (float)mw / (float)w;
    const int my = min((int)floorf((float)oy * sy), mh - 1);
    const size_t mplane = (size_t)mh * mw;
    const MaskT* __restrict__ mrow = mask + (size_t)(mask_n == 1 ? 0u : img) * mask_c * mplane + (size_t)my * mw;
    const size_t mstep = mask_c == 1 ? 0 : mplane;
    const size_t hw = (size_t)h * w, pix0 = (size_t)oy * w + x;
    const float* __restrict__ sp = sty + (size_t)img * 3 * hw + pix0;
    const uint8_t* __restrict__ cp = content + ((size_t)img * hw + pix0) * 3;
    uint8_t* __restrict__ op = out + ((size_t)img * hw + pix0) * 3;
    const int np = min(4, w - x);
    int mx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) mx[k] = min((int)floorf((float)min(x + k, w - 1) * sx), mw - 1);
    unsigned by[12];
    if (vec) {            // w % 4 == 0: the thread's 12 content bytes are one aligned 96-bit load, its stylised values three b128
        using u32x3 = __attribute__((ext_vector_type(3))) unsigned;
        const u32x3 cw = *(const u32x3*)cp;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const f32x4 b = *(const f32x4*)(sp + (size_t)ch * hw);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = 3 * k + ch;
                const float a = __fdiv_rn((float)((cw[e >> 2] >> (8 * (e & 3))) & 255u), 255.0f);
                const float m = (float)mrow[ch * mstep + mx[k]];
                by[e] = quant1(a * (1.0f - m) + b[k] * m);
            }
        }
        u32x3 wv;
#pragma unroll
        for (int d = 0; d < 3; ++d) wv[d] = by[4 * d] | (by[4 * d + 1] << 8) | (by[4 * d + 2] << 16) | (by[4 * d + 3] << 24);
        *(u32x3*)op = wv;
    } else {
        for (int k = 0; k < np; ++k)
            for (int ch = 0; ch < 3; ++ch) {
                const float a = __fdiv_rn((float)cp[3 * k + ch], 255.0f), m = (float)mrow[ch * mstep + mx[k]];
                op[3 * k + ch] = (uint8_t)quant1(a * (1.0f - m) + sp[(size_t)ch * hw + k] * m);
            }
    }
}

int launch_composite_quantize_u8_nearest(const uint8_t* content, const float* stylized, const void* mask, int mask_is_float, int mask_c,
                                         int mask_n, int mh, int mw, uint8_t* out, int n, int h, int w, hipStream_t s) {
    if (n < 1 || h < 1 || w < 1 || mh < 1 || mw < 1) { set_error("composite_quantize_u8_nearest: bad shape"); return -1; }
    if (mask_c != 1 && mask_c != 3) { set_error("composite_quantize_u8_nearest: mask channels %d must be 1 or 3", mask_c); return -1; }
    if (mask_n != 1 && mask_n != n) { set_error("composite_quantize_u8_nearest: mask batch %d must be 1 or %d", mask_n, n); return -1; }
    if ((size_t)h * w * 3 >= 0x7fffffffULL || (size_t)mh * mw >= 0x7fffffffULL) { set_error("composite_quantize_u8_nearest: frame or mask too large"); return -1; }
    dim3 g;
    if (row_grid("composite_quantize_u8_nearest", n, h, w, &g)) return -1;
    const int vec = (int)(w % 4 == 0 && aligned16(stylized) && ((uintptr_t)content & 3) == 0 && ((uintptr_t)out & 3) == 0);
    if (mask_is_float)
        hipLaunchKernelGGL(composite_quantize_u8_nearest_kernel<float>, g, dim3(64, 4), 0, s, content, stylized, (const float*)mask, mask_c, mask_n, mh, mw, out, h, w, vec);
    else
        hipLaunchKernelGGL(composite_quantize_u8_nearest_kernel<uint8_t>, g, dim3(64, 4), 0, s, content, stylized, (const uint8_t*)mask, mask_c, mask_n, mh, mw, out, h, w, vec);
    return check_launch("composite_quantize_u8_nearest");
}

// mask.float() of a uint8 / bool mask (test.py:224-226): float(v), element by element
__global__ __launch_bounds__(256) void mask_to_f32_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < total) out[i] = (float)in[i];
}

int launch_mask_to_f32(const uint8_t* in, float* out, size_t total, hipStream_t s) {
    if (total < 1 || total >= ((size_t)1 << 39)) { set_error("mask_to_f32: bad size"); return -1; }
    hipLaunchKernelGGL(mask_to_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, total);
    return check_launch("mask_to_f32");
}

// ---- video post-pass (reference video/utils.py:89-105 warp_image, :223-229 blend_images) -------------------
// out = u8( clip( (alpha * cur/255 + (1 - alpha) * warp(prev)/255) * 255, 0, 255 ) ), HWC uint8 frames;
// warp(prev)(y, x) = bilinear sample of prev at (x + flow[0][y][x], y + flow[1][y][x]) with cv2.BORDER_REFLECT
// (fedcba|abcdefgh|hgfedcb), in cv2.remap's own uint8 fixed-point arithmetic (below).  cv2 is not installed in the build
// image, so the fixed-point restatement is checked against the oracle only (parity unpinned against OpenCV itself).
__device__ __attribute__((noinline)) int reflect_border_far(int v, int n) {
    const int p = 2 * n;      // BORDER_REFLECT has period 2n: ...cba|abc...xyz|zyx...
    v %= p;
    if (v < 0) v += p;
    return v < n ? v : p - 1 - v;
}
__device__ __forceinline__ int reflect_border(int v, int n) {
    if ((unsigned)v < (unsigned)n) return v;                      // inside the frame: almost every tap
    const int once = v < 0 ? -1 - v : 2 * n - 1 - v;              // one reflection covers displacements up to a frame size
    if (__builtin_expect((unsigned)once < (unsigned)n, 1)) return once;
    return reflect_border_far(v, n);                              // a displacement beyond a whole frame: out of line (integer modulo)
}

// cv2.remap(uint8, float maps, INTER_LINEAR) is fixed point (OpenCV imgwarp: INTER_BITS = 5, INTER_REMAP_COEF_BITS = 15):
// the map is rounded half-to-even to 1/32 pixel, the four weights are (32-fx)(32-fy)*32 ... (they sum to 2^15; the one
// saturated entry {32767,0,0,1} at fx = fy = 0 cannot change a rounded result) and the pixel is
// (sum(S*w) + 2^14) >> 15.  Integer from the map rounding on, so the warp is bit-exact against the oracle; the blend that
// follows is the reference's float32 numpy expression evaluated without contraction.
__device__ __forceinline__ int clamp_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

#pragma clang fp contract(off)
struct WarpTap { int o00, o01, o10, o11, w00, w01, w10, w11; };      // pixel offsets into prev (x c) and 2^15-scaled weights

__device__ __forceinline__ WarpTap warp_tap(int x, int y, float fx_, float fy_, int h, int w, int c) {
    const float mx = (float)x + fx_, my = (float)y + fy_;
    const int ix = __float2int_rn(mx * 32.0f), iy = __float2int_rn(my * 32.0f);
    const int sx = clamp_short(ix >> 5), sy = clamp_short(iy >> 5), fx = ix & 31, fy = iy & 31;
    const int x0 = reflect_border(sx, w), x1 = reflect_border(sx + 1, w);
    const int y0 = reflect_border(sy, h), y1 = reflect_border(sy + 1, h);
    WarpTap t;
    t.w00 = (32 - fx) * (32 - fy) * 32; t.w01 = fx * (32 - fy) * 32; t.w10 = (32 - fx) * fy * 32; t.w11 = fx * fy * 32;
    t.o00 = (y0 * w + x0) * c; t.o01 = (y0 * w + x1) * c; t.o10 = (y1 * w + x0) * c; t.o11 = (y1 * w + x1) * c;
    return t;
}

// v / 255.0f for an integer-valued v in [0, 2^24), correctly rounded (= numpy's float32 division): one Newton step on
// v * RN(1/255) with the exact residual.  v / 255 is never within 2^-32 (relative) of a rounding boundary (255 does not
// divide a power of two), while the corrected quotient is within 2^-47 of the exact one, so the final rounding is the exact
// quotient's.  Replaces the ~10-instruction IEEE division sequence, six times per pixel.
__device__ __forceinline__ float div255(float v) {
    const float r = 1.0f / 255.0f;
    const float q = v * r;
    return __builtin_fmaf(__builtin_fmaf(-q, 255.0f, v), r, q);
}

__device__ __forceinline__ unsigned blend1(int wv, unsigned cur, float alpha, float one_minus_alpha) {
    const float a = alpha * div255((float)cur);
    const float bq = one_minus_alpha * div255((float)wv);
    const float b = (a + bq) * 255.0f;
    return (unsigned)fminf(fmaxf(b, 0.f), 255.f);
}

__device__ __forceinline__ unsigned warp_blend1(const uint8_t* __restrict__ prev, const WarpTap& t, int ch, unsigned cur, float alpha,
                                                float one_minus_alpha) {
    const int wv = ((int)prev[t.o00 + ch] * t.w00 + (int)prev[t.o01 + ch] * t.w01 + (int)prev[t.o10 + ch] * t.w10 +
                    (int)prev[t.o11 + ch] * t.w11 + (1 << 14)) >> 15;          // <= 255 by construction
    return blend1(wv, cur, alpha, one_minus_alpha);
}

// c == 3, h*w % 4 == 0: a thread owns 4 consecutive pixels (flat index): two b128 flow loads, 12 current bytes as three
// dwords, 12 output bytes as one 96-bit store.  The two taps of a row are neighbouring pixels (BORDER_REFLECT is continuous:
// |x1 - x0| <= 1), i.e. 6 consecutive bytes of the previous frame: ONE bounds-checked 8-byte buffer load at the left one of
// the two (2 gathers per pixel instead of 12 byte loads).
__global__ __launch_bounds__(256) void warp_blend_u8_rgb4_kernel(const uint8_t* __restrict__ cur, const uint8_t* __restrict__ prev,
                                                                 const float* __restrict__ flow, uint8_t* __restrict__ out, int h, int w,
                                                                 float alpha, float one_minus_alpha) {
    const int total = h * w;
    const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= total) return;
    const f32x4 fx = *(const f32x4*)(flow + i), fy = *(const f32x4*)(flow + total + i);
    using u32x3 = __attribute__((ext_vector_type(3))) unsigned;
    using u32x2 = __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned;
    const u32x3 cw = *(const u32x3*)(cur + (size_t)i * 3);
    const rsrc_t pr = make_rsrc(prev, (unsigned)total * 3u);
    const int nbytes = total * 3;
    // 8 bytes of the previous frame at byte offset `off` (the 6 that matter are inside the frame); the last two pixels of the
    // frame would read past its end: those assemble their bytes one by one (a buffer load that crosses the end returns zeros)
    auto load8 = [&](int off) -> unsigned long long {
        if (__builtin_expect(off + 8 <= nbytes, 1)) {
            const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(pr, off, 0, 0);
            return (unsigned long long)r[0] | ((unsigned long long)r[1] << 32);
        }
        unsigned long long v = 0;
        for (int j = 0; j < 6 && off + j < nbytes; ++j) v |= (unsigned long long)prev[off + j] << (8 * j);
        return v;
    };
    int y = i / w, x = i - y * w;
    unsigned by[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float mx = (float)x + fx[k], my = (float)y + fy[k];
        const int ix = __float2int_rn(mx * 32.0f), iy = __float2int_rn(my * 32.0f);
        const int sx = clamp_short(ix >> 5), sy = clamp_short(iy >> 5), qx = ix & 31, qy = iy & 31;
        const int x0 = reflect_border(sx, w), x1 = reflect_border(sx + 1, w);
        const int y0 = reflect_border(sy, h), y1 = reflect_border(sy + 1, h);
        const int w00 = (32 - qx) * (32 - qy) * 32, w01 = qx * (32 - qy) * 32, w10 = (32 - qx) * qy * 32, w11 = qx * qy * 32;
        const int xm = min(x0, x1);
        // byte shifts of tap 0 / tap 1 inside the 8 loaded bytes: the left pixel sits at 0, its right neighbour at 24 bits
        const int s0 = x0 == xm ? 0 : 24, s1 = x1 == xm ? 0 : 24;
        const unsigned long long v0 = load8((y0 * w + xm) * 3), v1 = load8((y1 * w + xm) * 3);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const int p00 = (int)((v0 >> (s0 + 8 * ch)) & 255u), p01 = (int)((v0 >> (s1 + 8 * ch)) & 255u);
            const int p10 = (int)((v1 >> (s0 + 8 * ch)) & 255u), p11 = (int)((v1 >> (s1 + 8 * ch)) & 255u);
            const int wv = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15;
            const int b = 3 * k + ch;
            by[b] = blend1(wv, (cw[b >> 2] >> (8 * (b & 3))) & 255u, alpha, one_minus_alpha);
        }
        if (++x == w) { x = 0; ++y; }
    }
    u32x3 o;
#pragma unroll
    for (int d = 0; d < 3; ++d) o[d] = by[4 * d] | (by[4 * d + 1] << 8) | (by[4 * d + 2] << 16) | (by[4 * d + 3] << 24);
    *(u32x3*)(out + (size_t)i * 3) = o;
}

__global__ __launch_bounds__(256) void warp_blend_u8_kernel(const uint8_t* __restrict__ cur, const uint8_t* __restrict__ prev,
                                                            const float* __restrict__ flow, uint8_t* __restrict__ out, int h, int w,
                                                            int c, float alpha, float one_minus_alpha) {
    const int total = h * w;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int y = i / w, x = i - y * w;
    const WarpTap t = warp_tap(x, y, flow[i], flow[total + i], h, w, c);
    for (int ch = 0; ch < c; ++ch)
        out[(size_t)i * c + ch] = (uint8_t)warp_blend1(prev, t, ch, cur[(size_t)i * c + ch], alpha, one_minus_alpha);
}

int launch_warp_blend_u8(const uint8_t* cur, const uint8_t* prev, const float* flow, uint8_t* out, int h, int w, int c, float alpha,
                         float one_minus_alpha, hipStream_t s) {
    if (h < 1 || w < 1 || c < 1) { set_error("warp_blend_u8: bad shape"); return -1; }
    if ((size_t)h * w * c >= 0x7fffffffULL) { set_error("warp_blend_u8: frame too large (2^31 bytes)"); return -1; }
    const int total = h * w;
    if (c == 3 && total % 4 == 0 && aligned16(flow) && ((uintptr_t)cur & 3) == 0 && ((uintptr_t)out & 3) == 0)
        hipLaunchKernelGGL(warp_blend_u8_rgb4_kernel, dim3((total / 4 + 255) / 256), dim3(256), 0, s, cur, prev, flow, out, h, w, alpha,
                           one_minus_alpha);
    else
        hipLaunchKernelGGL(warp_blend_u8_kernel, dim3((total + 255) / 256), dim3(256), 0, s, cur, prev, flow, out, h, w, c, alpha,
                           one_minus_alpha);
    return check_launch("warp_blend_u8");
}

// ---- cv2.resize(uint8 HWC, dsize, interpolation=cv2.INTER_AREA) of the video post-pass (reference video/utils.py:352-353) ----
// OpenCV 4.x modules/imgproc/src/resize.cpp, the true-area branch (scale_x >= 1 and scale_y >= 1):
//   * equal sizes: copy;
//   * both scales integers ("is_area_fast"): box sum in int, D = saturate_cast<uchar>(sum * (1.f / area)) = round-half-even;
//     2 x 2 with 1, 3 or 4 channels takes ResizeAreaFastVec's integer form (a + b + c + d + 2) >> 2;
//   * otherwise resizeArea_<uchar, float>: per axis a tap table (computeResizeAreaTab, double arithmetic: a partial
//     first cell, whole cells weighted 1 / cellWidth, a partial last cell; partial cells below 1e-3 dropped), the row buffer
//     accumulates S * alpha over the x taps in table order in float, rows are combined sum = beta * buf, then
//     sum += beta * buf, and the result is saturate_cast<uchar> (round-half-even).  Evaluated here in exactly that order
//     without contraction; the tables are recomputed per thread in double (a handful of IEEE divisions).
// An enlarged axis (a scale < 1) leaves this branch: resize_area_linear_u8_kernel below.  Unpinned against OpenCV itself (cv2 is
// absent).
struct AreaTaps { int s0, n; float a_first, a_mid, a_last; };      // taps s0 .. s0 + n - 1; weights: first, middle ones, last

__device__ __forceinline__ AreaTaps area_taps(int d, int ssize, double scale) {
    const double fsx1 = (double)d * scale, fsx2 = fsx1 + scale;
    const double cell = fmin(scale, (double)ssize - fsx1);
    int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
    sx2 = min(sx2, ssize - 1);
    sx1 = min(sx1, sx2);
    const bool left = (double)sx1 - fsx1 > 1e-3, right = fsx2 - (double)sx2 > 1e-3;
    AreaTaps t;
    t.s0 = left ? sx1 - 1 : sx1;
    t.n = (left ? 1 : 0) + (sx2 - sx1) + (right ? 1 : 0);
    const float mid = (float)(1.0 / cell);
    t.a_mid = mid;
    t.a_first = left ? (float)(((double)sx1 - fsx1) / cell) : mid;
    t.a_last = right ? (float)(fmin(fmin(fsx2 - (double)sx2, 1.), cell) / cell) : mid;
    if (t.n == 1 && left && !right) t.a_last = t.a_first;       // a single tap: first == last
    if (t.n == 1 && right && !left) t.a_first = t.a_last;
    return t;
}

__device__ __forceinline__ uint8_t sat_u8_rne(float v) {
    const int r = __float2int_rn(v);
    return (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
}

// mode 0: general (fractional) scale; 1: integer scale box mean; 2: 2 x 2 fast form
template <int MODE>
__global__ __launch_bounds__(256) void resize_area_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int hi, int wi,
                                                             int c, int ho, int wo, double scale_x, double scale_y, int isx, int isy) {
    const int dx = blockIdx.x * 64 + threadIdx.x, dy = blockIdx.y * 4 + threadIdx.y;
    if (dx >= wo || dy >= ho) return;
    const uint8_t* __restrict__ src = in + (size_t)blockIdx.z * hi * wi * c;
    uint8_t* __restrict__ dst = out + ((size_t)blockIdx.z * ho * wo + (size_t)dy * wo + dx) * c;
    if constexpr (MODE == 0) {
        const AreaTaps tx = area_taps(dx, wi, scale_x), ty = area_taps(dy, hi, scale_y);
        for (int ch = 0; ch < c; ++ch) {
            float sum = 0.f;
            for (int j = 0; j < ty.n; ++j) {
                const float beta = j == 0 ? ty.a_first : (j == ty.n - 1 ? ty.a_last : ty.a_mid);
                const uint8_t* __restrict__ S = src + ((size_t)(ty.s0 + j) * wi + tx.s0) * c + ch;
                float buf = 0.f;
                for (int k = 0; k < tx.n; ++k) {
                    const float alpha = k == 0 ? tx.a_first : (k == tx.n - 1 ? tx.a_last : tx.a_mid);
                    buf = buf + (float)S[(size_t)k * c] * alpha;
                }
                sum = j == 0 ? beta * buf : sum + beta * buf;
            }
            dst[ch] = sat_u8_rne(sum);
        }
    } else {
        const float scale = 1.f / (float)(isx * isy);
        for (int ch = 0; ch < c; ++ch) {
            int sum = 0;
            for (int sy = 0; sy < isy; ++sy) {
                const uint8_t* __restrict__ S = src + ((size_t)(dy * isy + sy) * wi + (size_t)dx * isx) * c + ch;
                for (int sx = 0; sx < isx; ++sx) sum += S[(size_t)sx * c];
            }
            dst[ch] = MODE == 2 ? (uint8_t)((sum + 2) >> 2) : sat_u8_rne((float)sum * scale);
        }
    }
}

// The fractional form with the tap tables of a 64 x 16 output block computed once per block (80 table entries in LDS instead of
// two per pixel: the tables are a dozen double-precision divisions each, which was nearly all of the kernel's time); a thread
// owns four rows of one column.  Same arithmetic and accumulation order as resize_area_u8_kernel<0>.
template <bool RGBW>
__global__ __launch_bounds__(256) void resize_area_tab_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int hi,
                                                                 int wi, int c, int ho, int wo, double scale_x, double scale_y,
                                                                 size_t total_bytes) {
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    __shared__ AreaTaps tx_s[64], ty_s[16];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const int dx = blockIdx.x * 64 + threadIdx.x;
    if (tid < 64) tx_s[tid] = area_taps(min(blockIdx.x * 64 + tid, wo - 1), wi, scale_x);
    else if (tid < 80) ty_s[tid - 64] = area_taps(min(blockIdx.y * 16 + (tid - 64), ho - 1), hi, scale_y);
    __syncthreads();
    if (dx >= wo) return;
    const AreaTaps tx = tx_s[threadIdx.x];
    const uint8_t* __restrict__ src = in + (size_t)blockIdx.z * hi * wi * c;
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
        const int ry = threadIdx.y + 4 * r, dy = blockIdx.y * 16 + ry;
        if (dy >= ho) break;
        const AreaTaps ty = ty_s[ry];
        uint8_t* __restrict__ dst = out + ((size_t)blockIdx.z * ho * wo + (size_t)dy * wo + dx) * c;
        if constexpr (RGBW) {
            // RGB with at most 4 taps across (scale_x < 3): a row's taps are <= 12 consecutive bytes = ONE 16-byte load from the
            // 4-byte-aligned address below them, shifted into place with v_alignbyte, instead of up to 12 byte loads; the
            // arithmetic and its order are those of the loops below
            float sum[3] = {0.f, 0.f, 0.f};
            for (int j = 0; j < ty.n; ++j) {
                const float beta = j == 0 ? ty.a_first : (j == ty.n - 1 ? ty.a_last : ty.a_mid);
                const size_t byte0 = (((size_t)blockIdx.z * hi + ty.s0 + j) * wi + tx.s0) * 3, base = byte0 & ~(size_t)3;
                unsigned w0, w1, w2;
                if (base + 16 <= total_bytes) {
                    const u32x4 w = *(const u32x4*)(in + base);
                    const unsigned sh = (unsigned)(byte0 & 3);
                    w0 = __builtin_amdgcn_alignbyte(w[1], w[0], sh);
                    w1 = __builtin_amdgcn_alignbyte(w[2], w[1], sh);
                    w2 = __builtin_amdgcn_alignbyte(w[3], w[2], sh);
                } else {                                    // the last bytes of the last frame: byte by byte
                    unsigned b[12];
#pragma unroll
                    for (int i = 0; i < 12; ++i) b[i] = i < tx.n * 3 ? in[byte0 + i] : 0u;
                    w0 = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
                    w1 = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
                    w2 = b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24);
                }
                const unsigned wd[3] = {w0, w1, w2};
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    float buf = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (k < tx.n) {
                            const float alpha = k == 0 ? tx.a_first : (k == tx.n - 1 ? tx.a_last : tx.a_mid);
                            const int bi = 3 * k + ch;
                            buf = buf + (float)((wd[bi >> 2] >> (8 * (bi & 3))) & 255u) * alpha;
                        }
                    }
                    sum[ch] = j == 0 ? beta * buf : sum[ch] + beta * buf;
                }
            }
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) dst[ch] = sat_u8_rne(sum[ch]);
            continue;
        }
        for (int ch = 0; ch < c; ++ch) {
            float sum = 0.f;
            for (int j = 0; j < ty.n; ++j) {
                const float beta = j == 0 ? ty.a_first : (j == ty.n - 1 ? ty.a_last : ty.a_mid);
                const uint8_t* __restrict__ S = src + ((size_t)(ty.s0 + j) * wi + tx.s0) * c + ch;
                float buf = 0.f;
                for (int k = 0; k < tx.n; ++k) {
                    const float alpha = k == 0 ? tx.a_first : (k == tx.n - 1 ? tx.a_last : tx.a_mid);
                    buf = buf + (float)S[(size_t)k * c] * alpha;
                }
                sum = j == 0 ? beta * buf : sum + beta * buf;
            }
            dst[ch] = sat_u8_rne(sum);
        }
    }
}

// The 2 x 2 form on RGB frames whose width is a multiple of 8 (the common case: a 1080p / 720p / 456 x 256 frame halved): a
// thread owns 4 output pixels = 8 source pixels of two rows, read as 2 x 2 dwordx3 (24 bytes are 4-byte aligned when the width
// is a multiple of 4 pixels) and written as one dwordx3, instead of 48 byte loads and 12 byte stores.  Same integer arithmetic.
__global__ __launch_bounds__(256) void resize_area2x2_rgb4_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int hi,
                                                                  int wi, int ho, int wo) {
    using u32x3 = __attribute__((ext_vector_type(3))) unsigned;
    const int x4 = blockIdx.x * 64 + threadIdx.x, dy = blockIdx.y * 4 + threadIdx.y;      // x4: group of 4 output pixels
    if (x4 * 4 >= wo || dy >= ho) return;
    const uint8_t* __restrict__ r0 = in + ((size_t)blockIdx.z * hi + 2 * dy) * wi * 3 + (size_t)x4 * 24;
    const uint8_t* __restrict__ r1 = r0 + (size_t)wi * 3;
    const u32x3 a0 = *(const u32x3*)r0, a1 = *(const u32x3*)(r0 + 12), b0 = *(const u32x3*)r1, b1 = *(const u32x3*)(r1 + 12);
    const unsigned ra[6] = {a0[0], a0[1], a0[2], a1[0], a1[1], a1[2]}, rb[6] = {b0[0], b0[1], b0[2], b1[0], b1[1], b1[2]};
    auto byte_of = [](const unsigned (&w)[6], int i) { return (w[i >> 2] >> (8 * (i & 3))) & 255u; };
    unsigned o[3] = {0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const int i0 = 6 * j + ch, i1 = i0 + 3;                 // source pixels 2j and 2j + 1
            const unsigned v = (byte_of(ra, i0) + byte_of(ra, i1) + byte_of(rb, i0) + byte_of(rb, i1) + 2u) >> 2;
            const int ob = 3 * j + ch;
            o[ob >> 2] |= v << (8 * (ob & 3));
        }
    *(u32x3*)(out + (((size_t)blockIdx.z * ho + dy) * wo + (size_t)x4 * 4) * 3) = u32x3{o[0], o[1], o[2]};
}

// INTER_AREA with an ENLARGED axis: cv::resize leaves the true-area branch and runs its generic linear path in "area mode" on
// uint8 (resize.cpp: the coefficient loop with area_mode, HResizeLinear<uchar, int, short, 2048>, VResizeLinear's fixed-point
// combine).  Per output index: s = cvFloor(d * scale); f = (float)((d + 1) - (s + 1) * inv_scale), f <= 0 -> 0 else f - cvFloor(f);
// along x an index at the last source column gets f = 0 (the xmax rule); weights saturate_cast<short>((1.f - f, f) * 2048).
struct AreaLin { int s; int w0, w1; };
__device__ __forceinline__ AreaLin area_linear(int d, int ssize, double inv_scale, double scale, bool clamp_last) {
    int sx = (int)floor((double)d * scale);
    float f = (float)((double)(d + 1) - (double)(sx + 1) * inv_scale);
    f = f <= 0.f ? 0.f : f - floorf(f);
    if (clamp_last && sx >= ssize - 1) { f = 0.f; sx = ssize - 1; }
    AreaLin r;
    r.s = sx;
    r.w0 = __float2int_rn((1.f - f) * 2048.f);
    r.w1 = __float2int_rn(f * 2048.f);
    return r;
}

__global__ __launch_bounds__(256) void resize_area_linear_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int hi,
                                                                    int wi, int c, int ho, int wo, double inv_x, double inv_y) {
    const int dx = blockIdx.x * 64 + threadIdx.x, dy = blockIdx.y * 4 + threadIdx.y;
    if (dx >= wo || dy >= ho) return;
    const uint8_t* __restrict__ src = in + (size_t)blockIdx.z * hi * wi * c;
    uint8_t* __restrict__ dst = out + ((size_t)blockIdx.z * ho * wo + (size_t)dy * wo + dx) * c;
    const AreaLin ax = area_linear(dx, wi, inv_x, 1. / inv_x, true), ay = area_linear(dy, hi, inv_y, 1. / inv_y, false);
    const int x0 = ax.s, x1 = min(ax.s + 1, wi - 1), y0 = min(ay.s, hi - 1), y1 = min(ay.s + 1, hi - 1);
    for (int ch = 0; ch < c; ++ch) {
        const int r0 = (int)src[((size_t)y0 * wi + x0) * c + ch] * ax.w0 + (int)src[((size_t)y0 * wi + x1) * c + ch] * ax.w1;
        const int r1 = (int)src[((size_t)y1 * wi + x0) * c + ch] * ax.w0 + (int)src[((size_t)y1 * wi + x1) * c + ch] * ax.w1;
        dst[ch] = (uint8_t)((((ay.w0 * (r0 >> 4)) >> 16) + ((ay.w1 * (r1 >> 4)) >> 16) + 2) >> 2);
    }
}

int launch_resize_area_u8(const uint8_t* in, uint8_t* out, int n, int hi, int wi, int c, int ho, int wo, hipStream_t s) {
    if (n < 1 || hi < 1 || wi < 1 || c < 1 || ho < 1 || wo < 1) { set_error("resize_area_u8: bad shape"); return -1; }
    if ((size_t)hi * wi * c >= 0x7fffffffULL || n > 65535 || (ho + 3) / 4 > 65535) { set_error("resize_area_u8: frame or batch too large"); return -1; }
    if (ho == hi && wo == wi) {
        if (hipMemcpyAsync(out, in, (size_t)n * hi * wi * c, hipMemcpyDeviceToDevice, s) != hipSuccess) {
            set_error("resize_area_u8: copy failed: %s", hipGetErrorString(hipGetLastError()));
            return -2;
        }
        return 0;
    }
    const dim3 g((wo + 63) / 64, (ho + 3) / 4, n), b(64, 4);
    if (ho > hi || wo > wi) {       // an enlarged axis: the fixed-point linear path in area mode
        hipLaunchKernelGGL(resize_area_linear_u8_kernel, g, b, 0, s, in, out, hi, wi, c, ho, wo, (double)wo / wi, (double)ho / hi);
        return check_launch("resize_area_u8");
    }
    // cv::resize: inv_scale = dsize / ssize (double), scale = 1. / inv_scale, iscale = saturate_cast<int>(scale) (= cvRound)
    const double scale_x = 1. / ((double)wo / wi), scale_y = 1. / ((double)ho / hi);
    const int isx = (int)nearbyint(scale_x), isy = (int)nearbyint(scale_y);
    const bool fast = fabs(scale_x - isx) < 2.220446049250313e-16 && fabs(scale_y - isy) < 2.220446049250313e-16;
    const size_t total_bytes = (size_t)n * hi * wi * c;
    if (!fast && (ho + 15) / 16 <= 65535 && c == 3 && scale_x < 3.0 && ((uintptr_t)in & 3) == 0)        // at most 4 taps across
        hipLaunchKernelGGL(resize_area_tab_u8_kernel<true>, dim3((wo + 63) / 64, (ho + 15) / 16, n), b, 0, s, in, out, hi, wi, c, ho, wo, scale_x,
                           scale_y, total_bytes);
    else if (!fast && (ho + 15) / 16 <= 65535)
        hipLaunchKernelGGL(resize_area_tab_u8_kernel<false>, dim3((wo + 63) / 64, (ho + 15) / 16, n), b, 0, s, in, out, hi, wi, c, ho, wo, scale_x,
                           scale_y, total_bytes);
    else if (!fast) hipLaunchKernelGGL(resize_area_u8_kernel<0>, g, b, 0, s, in, out, hi, wi, c, ho, wo, scale_x, scale_y, 0, 0);
    else if (isx == 2 && isy == 2 && c == 3 && wi % 8 == 0 && wo * 2 == wi && ((uintptr_t)in & 3) == 0 && ((uintptr_t)out & 3) == 0)
        hipLaunchKernelGGL(resize_area2x2_rgb4_kernel, dim3((wo / 4 + 63) / 64, (ho + 3) / 4, n), b, 0, s, in, out, hi, wi, ho, wo);
    else if (isx == 2 && isy == 2 && (c == 1 || c == 3 || c == 4))
        hipLaunchKernelGGL(resize_area_u8_kernel<2>, g, b, 0, s, in, out, hi, wi, c, ho, wo, scale_x, scale_y, isx, isy);
    else hipLaunchKernelGGL(resize_area_u8_kernel<1>, g, b, 0, s, in, out, hi, wi, c, ho, wo, scale_x, scale_y, isx, isy);
    return check_launch("resize_area_u8");
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
