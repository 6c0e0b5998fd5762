// Pillow's antialiased BILINEAR resize of a uint8 RGB image, bit for bit, on gfx950 - the Resize step of the reference's
// test_transform (Style_3DGS/AdaIN/test.py:16-24: torchvision Resize(size) [+ CenterCrop(size)] on a PIL image, i.e.
// PIL.Image.resize(size, BILINEAR) = libImaging/Resample.c ImagingResample, applied at test.py:203-204 to the content frame and
// the style image of every adain_inference call, video/utils.py:341-350 once per video frame).
//
// The algorithm restated (Pillow 4.x .. 12.x, Resample.c):
//   * per axis, for output index xx: scale = in / out (double), filterscale = max(scale, 1), support = filterscale (the triangle
//     filter's support of 1, stretched when shrinking: this is the antialiasing), center = (xx + 0.5) * scale,
//     xmin = max((int)(center - support + 0.5), 0), xmax = min((int)(center + support + 0.5), in) - xmin taps,
//     w[x] = tri((x + xmin - center + 0.5) / filterscale), normalised by their sum - all in double (precompute_coeffs) -
//     then converted to 22-bit fixed point, (int)(0.5 + w * 2^22) (normalize_coeffs_8bpc);
//   * a HORIZONTAL pass into a uint8 intermediate, then a VERTICAL pass: out = clip8((2^21 + sum in * k) >> 22) each time
//     (ImagingResampleHorizontal_8bpc / ImagingResampleVertical_8bpc; clip8 clamps to 0..255).
//
// Here: one tiny kernel builds both axes' tap tables in the caller's workspace with the same double arithmetic (this file is
// compiled with fp contraction OFF: +, -, *, / on doubles round as the x86-64 build of Pillow rounds them), one kernel computes
// the output pixels - each thread one output pixel, running the horizontal pass for the rows its vertical taps need and rounding
// every intermediate value to uint8 exactly where Pillow's intermediate image does.  An optional crop window (CenterCrop) limits
// the pixels computed; the result is packed RGB, what adain_encode_u8 / adain_stylize_u8 take.  Source pixels may be packed
// RGB (numpy / decoded video frames) or 4 bytes per pixel (Pillow's own RGBX storage, the 4th byte ignored).
#include "common.h"

namespace adain {

#pragma clang fp contract(off)

constexpr int PRECISION_BITS = 32 - 8 - 2;      // Resample.c: 22

struct PilAxis {
    int in_size, out_size, ksize;
    int* bounds;      // [out_size][2]: first source index, tap count
    int* kk;          // [out_size][ksize]: fixed-point taps
};

__host__ __device__ inline int pil_ksize(int in_size, int out_size) {
    double filterscale = (double)((float)in_size - 0.f) / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    return (int)ceil(1.0 * filterscale) * 2 + 1;
}

// precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle) filter, box = the whole axis
__device__ void pil_axis_coeffs(const PilAxis& a, int xx) {
    const double scale = (double)((float)a.in_size - 0.f) / a.out_size;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 1.0 * filterscale;
    const double center = 0.0 + (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > a.in_size) xmax = a.in_size;
    xmax -= xmin;
    int* __restrict__ k = a.kk + (size_t)xx * a.ksize;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
        double t = ((x + xmin) - center + 0.5) * ss;
        if (t < 0.0) t = -t;
        ww += t < 1.0 ? 1.0 - t : 0.0;
    }
    for (int x = 0; x < a.ksize; ++x) {
        double w = 0.0;
        if (x < xmax) {
            double t = ((x + xmin) - center + 0.5) * ss;
            if (t < 0.0) t = -t;
            w = t < 1.0 ? 1.0 - t : 0.0;
            if (ww != 0.0) w /= ww;
        }
        k[x] = w < 0 ? (int)(-0.5 + w * (1 << PRECISION_BITS)) : (int)(0.5 + w * (1 << PRECISION_BITS));
    }
    a.bounds[2 * xx] = xmin;
    a.bounds[2 * xx + 1] = xmax;
}

__global__ __launch_bounds__(256) void pil_coeffs_kernel(PilAxis ax, PilAxis ay) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ax.out_size) pil_axis_coeffs(ax, i);
    else if (i - ax.out_size < ay.out_size) pil_axis_coeffs(ay, i - ax.out_size);
}

__device__ __forceinline__ int clip8(int v) {
    v >>= PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// out[n][ch][cw][3] = crop window (y0, x0, ch, cw) of the (ho x wo) resize of in[n][hi][wi][PIX]
template <int PIX>
__global__ __launch_bounds__(256) void pil_resize_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int hi, int wi, PilAxis ax,
                                                         PilAxis ay, int y0, int x0, int ch, int cw) {
    const int ox = blockIdx.x * 64 + threadIdx.x, oy = blockIdx.y * 4 + threadIdx.y;
    if (ox >= cw || oy >= ch) return;
    const int xx = ox + x0, yy = oy + y0;
    const int xmin = ax.bounds[2 * xx], xn = ax.bounds[2 * xx + 1];
    const int ymin = ay.bounds[2 * yy], yn = ay.bounds[2 * yy + 1];
    const int* __restrict__ kx = ax.kk + (size_t)xx * ax.ksize;
    const int* __restrict__ ky = ay.kk + (size_t)yy * ay.ksize;
    const uint8_t* __restrict__ src = in + ((size_t)blockIdx.z * hi + ymin) * wi * PIX + (size_t)xmin * PIX;
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int y = 0; y < yn; ++y) {
        const uint8_t* __restrict__ row = src + (size_t)y * wi * PIX;
        int h0 = 1 << (PRECISION_BITS - 1), h1 = h0, h2 = h0;
        for (int x = 0; x < xn; ++x) {
            const int k = kx[x];
            h0 += (int)row[x * PIX + 0] * k;
            h1 += (int)row[x * PIX + 1] * k;
            h2 += (int)row[x * PIX + 2] * k;
        }
        const int k = ky[y];                 // the horizontal pass's uint8 intermediate, then the vertical tap
        a0 += clip8(h0) * k;
        a1 += clip8(h1) * k;
        a2 += clip8(h2) * k;
    }
    uint8_t* __restrict__ o = out + (((size_t)blockIdx.z * ch + oy) * cw + ox) * 3;
    o[0] = (uint8_t)clip8(a0);
    o[1] = (uint8_t)clip8(a1);
    o[2] = (uint8_t)clip8(a2);
}

static size_t align_ints(size_t n) { return (n + 63) & ~(size_t)63; }

size_t resize_pil_workspace_bytes(int hi, int wi, int ho, int wo) {
    if (hi < 1 || wi < 1 || ho < 1 || wo < 1) return 0;
    return (align_ints((size_t)wo * 2) + align_ints((size_t)wo * pil_ksize(wi, wo)) + align_ints((size_t)ho * 2) + align_ints((size_t)ho * pil_ksize(hi, ho))) * sizeof(int);
}

int launch_resize_pil_bilinear_u8(const uint8_t* in, int pixel_bytes, int n, int hi, int wi, uint8_t* out, int ho, int wo, int y0, int x0, int ch,
                                  int cw, void* workspace, size_t ws_bytes, hipStream_t s) {
    if (pixel_bytes != 3 && pixel_bytes != 4) { set_error("resize_pil_bilinear_u8: source pixels of 3 (RGB) or 4 (RGBX) bytes, got %d", pixel_bytes); return -1; }
    // every side below 2^24 (Pillow's own limit on an image side is far below; keeps wo + ho, the tap-table sizes and the crop sums in int)
    if (n < 1 || hi < 1 || wi < 1 || ho < 1 || wo < 1 || hi >= (1 << 24) || wi >= (1 << 24) || ho >= (1 << 24) || wo >= (1 << 24)) {
        set_error("resize_pil_bilinear_u8: bad size %dx%d -> %dx%d (every side must be in [1, 2^24))", hi, wi, ho, wo);
        return -1;
    }
    // a shrink factor beyond 256 (a tap window of more than ~770 source pixels per output pixel and axis, recomputed per output pixel by
    // this one-pass kernel) is not a resize any caller of the path makes (test.py:16-24: 512 / 256 from camera-sized images): refused
    // rather than run for seconds
    if ((long long)hi > 256LL * ho || (long long)wi > 256LL * wo) {
        set_error("resize_pil_bilinear_u8: shrink factor above 256 (%dx%d -> %dx%d) is not supported", hi, wi, ho, wo);
        return -1;
    }
    if (y0 < 0 || x0 < 0 || ch < 1 || cw < 1 || (long long)y0 + ch > ho || (long long)x0 + cw > wo) {
        set_error("resize_pil_bilinear_u8: crop window (%d, %d, %d x %d) outside the %d x %d result", y0, x0, ch, cw, ho, wo);
        return -1;
    }
    if (!workspace || ws_bytes < resize_pil_workspace_bytes(hi, wi, ho, wo)) { set_error("resize_pil_bilinear_u8: workspace too small"); return -1; }
    if ((size_t)n > 65535 || (size_t)(ch + 3) / 4 > 65535) { set_error("resize_pil_bilinear_u8: grid too large"); return -1; }
    int* p = (int*)workspace;
    PilAxis ax{wi, wo, pil_ksize(wi, wo), p, nullptr};
    p += align_ints((size_t)wo * 2);
    ax.kk = p;
    p += align_ints((size_t)wo * ax.ksize);
    PilAxis ay{hi, ho, pil_ksize(hi, ho), p, nullptr};
    p += align_ints((size_t)ho * 2);
    ay.kk = p;
    hipLaunchKernelGGL(pil_coeffs_kernel, dim3((wo + ho + 255) / 256), dim3(256), 0, s, ax, ay);
    if (int r = check_launch("resize_pil_bilinear_u8 (tap tables)")) return r;
    const dim3 g((cw + 63) / 64, (ch + 3) / 4, n), b(64, 4);
    if (pixel_bytes == 3) hipLaunchKernelGGL(pil_resize_kernel<3>, g, b, 0, s, in, out, hi, wi, ax, ay, y0, x0, ch, cw);
    else hipLaunchKernelGGL(pil_resize_kernel<4>, g, b, 0, s, in, out, hi, wi, ax, ay, y0, x0, ch, cw);
    return check_launch("resize_pil_bilinear_u8");
}

}  // namespace adain
