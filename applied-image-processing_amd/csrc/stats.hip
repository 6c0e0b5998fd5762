// Per-channel statistics and the AdaIN feature blend (HBM-bound kernels).
//
// Replaces calc_mean_std (Style_3DGS/AdaIN/function.py:4-12), adaptive_instance_normalization
// (function.py:15-23) and the two feature blends of the reference:
//   style_transfer_simple  feat = AdaIN * alpha + content_f * (1 - alpha)      (test.py:79-80)
//   style_transfer         feat = AdaIN * (1 - P) + content_f * P              (test.py:69-70)
// Both layouts are served: NHWC (the internal activation layout, stats are column reductions over
// pixel rows) and NCHW (the reference's tensor layout, used when a caller hands in its own tensors).
//
// Sums are accumulated in fp64 (sum and sum of squares): the unbiased variance is then exact to
// fp32 rounding without a second pass over the 33.5 MB feature map, and partial sums are combined in
// a fixed order, so results are bitwise reproducible run to run.
#include "common.h"

namespace adain {

constexpr int MS_THREADS = 256;

// ---- NHWC: feat [n][hw][c] ; block (b, img) reduces pixels b, b+nblk, ... (in row groups) ----------
// thread layout: cols = c/4 thread-columns (4 channels each), rows = MS_THREADS / cols pixel rows.
__global__ __launch_bounds__(MS_THREADS) void mean_std_nhwc_partial(const float* __restrict__ feat, int c, int hw,
                                                                    int nblk, double* __restrict__ part) {
    const int cols = c >> 2;
    const int rows = MS_THREADS / cols;
    const int tid = threadIdx.x;
    const int col = tid % cols, row = tid / cols;
    const int img = blockIdx.y, b = blockIdx.x;
    const float* __restrict__ base = feat + (size_t)img * hw * c;
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    if (row < rows) {
        const int step = nblk * rows;
        auto add = [&](const f32x4 v) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double d = (double)v[k];
                s[k] += d;
                q[k] += d * d;
            }
        };
        auto at = [&](int p) { return *(const f32x4*)(base + (size_t)p * c + col * 4); };
        int p = b * rows + row;
        // four loads in flight per thread, summed in pixel order (the order of a plain loop: results unchanged bit for bit);
        // one dependent 16-byte load per iteration left the kernel latency-bound (1 workgroup per CU: 1.9 TB/s)
        for (; p + 3 * step < hw; p += 4 * step) {
            const f32x4 v0 = at(p), v1 = at(p + step), v2 = at(p + 2 * step), v3 = at(p + 3 * step);
            add(v0); add(v1); add(v2); add(v3);
        }
        for (; p < hw; p += step) add(at(p));
    }
    // combine the row groups through LDS in a fixed order
    __shared__ double sh[MS_THREADS * 8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sh[tid * 8 + k] = s[k];
        sh[tid * 8 + 4 + k] = q[k];
    }
    __syncthreads();
    if (tid < cols) {
        double ts[4] = {0, 0, 0, 0}, tq[4] = {0, 0, 0, 0};
        for (int r = 0; r < rows; ++r) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ts[k] += sh[(r * cols + tid) * 8 + k];
                tq[k] += sh[(r * cols + tid) * 8 + 4 + k];
            }
        }
        double* __restrict__ o = part + (((size_t)img * nblk + b) * c + tid * 4) * 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            o[k * 2] = ts[k];
            o[k * 2 + 1] = tq[k];
        }
    }
}

// one wave per channel: lanes stride over the partial blocks, then a fixed-order butterfly reduction
__global__ __launch_bounds__(256) void mean_std_finalize(const double* __restrict__ part, int c, int hw, int nblk, float eps,
                                                         float* __restrict__ mean, float* __restrict__ std_) {
    const int lane = threadIdx.x & 63;
    const int ch = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int img = blockIdx.y;
    if (ch >= c) return;
    double s = 0, q = 0;
    for (int b = lane; b < nblk; b += 64) {
        const double* p = part + (((size_t)img * nblk + b) * c + ch) * 2;
        s += p[0];
        q += p[1];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_xor(s, off, 64);
        q += __shfl_xor(q, off, 64);
    }
    if (lane == 0) {
        const double m = s / hw;
        // unbiased variance (torch.var default, function.py:9); hw == 1 gives 0/0 = NaN like torch
        const double var = (q - s * m) / (double)(hw - 1);
        mean[(size_t)img * c + ch] = (float)m;
        std_[(size_t)img * c + ch] = sqrtf((float)var + eps);
    }
}

// ---- NCHW: feat [n][c][hw] ; one block per (n, c) plane -------------------------------------------------
__global__ __launch_bounds__(MS_THREADS) void mean_std_nchw_kernel(const float* __restrict__ feat, int hw, float eps,
                                                                   float* __restrict__ mean, float* __restrict__ std_) {
    const size_t plane = blockIdx.x;
    const float* __restrict__ p = feat + plane * hw;
    double s = 0, q = 0;
    for (int i = threadIdx.x; i < hw; i += MS_THREADS) {
        const double d = (double)p[i];
        s += d;
        q += d * d;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off, 64);
        q += __shfl_down(q, off, 64);
    }
    __shared__ double sh[2 * (MS_THREADS / 64)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        sh[wave * 2] = s;
        sh[wave * 2 + 1] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = 0, tq = 0;
        for (int w = 0; w < MS_THREADS / 64; ++w) {
            ts += sh[w * 2];
            tq += sh[w * 2 + 1];
        }
        const double m = ts / hw;
        const double var = (tq - ts * m) / (double)(hw - 1);
        mean[plane] = (float)m;
        std_[plane] = sqrtf((float)var + eps);
    }
}

static int nhwc_blocks(int c, int hw) {
    const int rows = MS_THREADS / (c >> 2);
    int nblk = (hw + rows * 16 - 1) / (rows * 16);   // >= 16 row groups of work per block
    int cap = 256;
#ifdef ADAIN_DIAG
    static const int cap_env = tune_env("ADAIN_MS_BLOCKS", 256);
    cap = cap_env;
#endif
    if (nblk > cap) nblk = cap;
    if (nblk < 1) nblk = 1;
    return nblk;
}

size_t mean_std_workspace_bytes(int nhwc, int n, int c, int hw) {
    if (!nhwc) return 0;
    if (c < 4 || (c & 3) || (c >> 2) > MS_THREADS) return 0;
    return (size_t)n * nhwc_blocks(c, hw) * c * 2 * sizeof(double);
}

int launch_mean_std(const float* feat, int nhwc, int n, int c, int hw, float eps, float* mean, float* std_,
                    void* workspace, size_t ws_bytes, hipStream_t s) {
    if (n < 1 || c < 1 || hw < 1) { set_error("mean_std: bad shape n=%d c=%d hw=%d", n, c, hw); return -1; }
    if (nhwc) {
        if ((c & 3) || (c >> 2) > MS_THREADS) { set_error("mean_std(NHWC): c=%d must be a multiple of 4 and <= 1024", c); return -1; }
        const int nblk = nhwc_blocks(c, hw);
        if (ws_bytes < mean_std_workspace_bytes(1, n, c, hw) || !workspace) { set_error("mean_std: workspace too small"); return -1; }
        hipLaunchKernelGGL(mean_std_nhwc_partial, dim3(nblk, n), dim3(MS_THREADS), 0, s, feat, c, hw, nblk, (double*)workspace);
        hipLaunchKernelGGL(mean_std_finalize, dim3((c + 3) / 4, n), dim3(256), 0, s, (const double*)workspace, c, hw, nblk, eps, mean, std_);
    } else {
        hipLaunchKernelGGL(mean_std_nchw_kernel, dim3((unsigned)((size_t)n * c)), dim3(MS_THREADS), 0, s, feat, hw, eps, mean, std_);
    }
    return check_launch("mean_std");
}

// ---- AdaIN + blend ---------------------------------------------------------------------------------------
// out = t * w1 + x * w2 with t = (x - mu_c) / sigma_c * sigma_s + mu_s, evaluated in the reference's
// operation order without FMA contraction (function.py:21-23, test.py:70,80).
//   alpha mode (pmap == nullptr): w1 = alpha, w2 = one_minus_alpha (computed on the host in double)
//   pmap  mode                  : w1 = 1 - P[pixel], w2 = P[pixel]      (P broadcast over channels)
template <bool NHWC>
__global__ __launch_bounds__(256) void adain_blend_kernel(const float* __restrict__ x, int c, int hw,
                                                          const float* __restrict__ c_mean, const float* __restrict__ c_std,
                                                          const float* __restrict__ s_mean, const float* __restrict__ s_std,
                                                          int style_n, float alpha, float one_minus_alpha,
                                                          const float* __restrict__ pmap, int pmap_n,
                                                          float* __restrict__ out, size_t total4) {
#pragma clang fp contract(off)
    // 32-bit index arithmetic (the launcher checks total < 2^31) and, for NHWC, one b128 load per statistic: the four
    // elements of a quad are four consecutive channels of one pixel
    const unsigned per_img = (unsigned)c * (unsigned)hw;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total4; i += gridDim.x * blockDim.x) {
        const unsigned e = i * 4u;
        const f32x4 v = *(const f32x4*)(x + e);
        f32x4 r;
        const unsigned img = e / per_img;
        const unsigned rem = e - img * per_img;
        const unsigned simg = style_n == 1 ? 0u : img;
        if (NHWC) {
            const unsigned pix = rem / (unsigned)c, ch = rem - pix * (unsigned)c;
            const f32x4 mc = *(const f32x4*)(c_mean + img * c + ch), sc = *(const f32x4*)(c_std + img * c + ch);
            const f32x4 ms = *(const f32x4*)(s_mean + simg * c + ch), ss = *(const f32x4*)(s_std + simg * c + ch);
            float w1 = alpha, w2 = one_minus_alpha;
            if (pmap) {
                const float p = pmap[(pmap_n == 1 ? 0u : img) * (unsigned)hw + pix];
                w1 = 1.0f - p;
                w2 = p;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float nrm = (v[k] - mc[k]) / sc[k];
                const float t = nrm * ss[k] + ms[k];
                r[k] = t * w1 + v[k] * w2;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned ch = (rem + k) / (unsigned)hw, pix = (rem + k) - ch * (unsigned)hw;
                const float mc = c_mean[img * c + ch], sc = c_std[img * c + ch];
                const float ms = s_mean[simg * c + ch], ss = s_std[simg * c + ch];
                const float nrm = (v[k] - mc) / sc;
                const float t = nrm * ss + ms;
                float w1 = alpha, w2 = one_minus_alpha;
                if (pmap) {
                    const float p = pmap[(pmap_n == 1 ? 0u : img) * (unsigned)hw + pix];
                    w1 = 1.0f - p;
                    w2 = p;
                }
                r[k] = t * w1 + v[k] * w2;
            }
        }
        *(f32x4*)(out + e) = r;
    }
}

int launch_adain_blend_ex(const float* content, int nhwc, int n, int c, int hw, const float* c_mean, const float* c_std,
                          const float* s_mean, const float* s_std, int style_n, float alpha, float one_minus_alpha,
                          const float* pmap, int pmap_n, float* out, hipStream_t s) {
    if (n < 1 || c < 1 || hw < 1) { set_error("adain_blend: bad shape"); return -1; }
    if (style_n != 1 && style_n != n) { set_error("adain_blend: style batch %d must be 1 or %d", style_n, n); return -1; }
    if (pmap && pmap_n != 1 && pmap_n != n) { set_error("adain_blend: pmap batch %d must be 1 or %d", pmap_n, n); return -1; }
    const size_t total = (size_t)n * c * hw;
    if (nhwc ? (c & 3) : (total & 3)) { set_error("adain_blend: element count / channels must be a multiple of 4"); return -1; }
    if (total >= 0x7fffffffULL) { set_error("adain_blend: more than 2^31 elements per call"); return -1; }
    const size_t total4 = total / 4;
    const unsigned blocks = (unsigned)((total4 + 255) / 256 < 8192 ? (total4 + 255) / 256 : 8192);
    if (nhwc)
        hipLaunchKernelGGL(adain_blend_kernel<true>, dim3(blocks), dim3(256), 0, s, content, c, hw, c_mean, c_std, s_mean, s_std,
                           style_n, alpha, one_minus_alpha, pmap, pmap_n, out, total4);
    else
        hipLaunchKernelGGL(adain_blend_kernel<false>, dim3(blocks), dim3(256), 0, s, content, c, hw, c_mean, c_std, s_mean, s_std,
                           style_n, alpha, one_minus_alpha, pmap, pmap_n, out, total4);
    return check_launch("adain_blend");
}

}  // namespace adain
