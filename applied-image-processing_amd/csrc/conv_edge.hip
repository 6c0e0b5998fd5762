// gfx950 (CDNA4) kernels of the two edge layers of the AdaIN networks (reference Style_3DGS/AdaIN/net.py): the first encoder layer
// (conv0 1x1 folded into conv1_1, 3 -> 64 channels, NCHW / uint8-HWC image in, NHWC out) and the last decoder layer (64 -> 3, NHWC in,
// NCHW image out).  Both are HBM-bound (256 B per pixel on their wide side) and built around their memory side; the generic 3x3
// layers in between run in conv_wino4.hip.
#include <stdlib.h>
#include <utility>

#include "common.h"
#include "device_utils.h"

namespace adain {

constexpr int TW = 32;           // tile width == MFMA M
constexpr int HW_ = TW + 2;      // halo width

// ---------------------------------------------------------------------------------------------
// Weight packing (runs once per weight set, on the device).
// ---------------------------------------------------------------------------------------------
// conv0 (1x1, 3->3, net.py:39) folded into conv1_1 (3->64, net.py:41): a pointwise conv commutes
// with reflection padding, so W'[o][c][t] = sum_c' W1[o][c'][t] W0[c'][c] and
// b'[o] = b1[o] + sum_{c',t} W1[o][c'][t] b0[c'].  K index e = 3 tap + c for e < 27, and e = 27 multiplies a constant 1: its
// weight is the bias b'[o], so the kernel's epilogue adds nothing (K = 28 = 14 MFMAs of K 2).
// packed (A operand = weights): [2 cout tiles][14 steps][64 lanes] with cout = 32 T + lane % 32, e = 2 step + lane / 32.
__global__ void pack_conv_first_kernel(const float* __restrict__ w0, const float* __restrict__ b0,
                                       const float* __restrict__ w1, const float* __restrict__ b1,
                                       float* __restrict__ p, float* __restrict__ bias_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    auto folded_bias = [&](int o) {
        float b = b1[o];
        for (int cp = 0; cp < 3; ++cp)
            for (int t = 0; t < 9; ++t) b += w1[(o * 3 + cp) * 9 + t] * b0[cp];
        return b;
    };
    if (i < 2 * 14 * 64) {
        const int lane = i & 63, step = (i >> 6) % 14, T = i / (14 * 64);
        const int o = T * 32 + (lane & 31), e = 2 * step + (lane >> 5);
        float v;
        if (e < 27) {
            const int tap = e / 3, c = e % 3;
            v = 0.f;
            for (int cp = 0; cp < 3; ++cp) v += w1[(o * 3 + cp) * 9 + tap] * w0[cp * 3 + c];
        } else {
            v = folded_bias(o);
        }
        p[i] = v;
    }
    if (i < 64) bias_out[i] = folded_bias(i);
}

// last decoder conv (64->3, net.py:35): OIHW [3][64][3][3] -> the A operand of conv_last_kernel's 32 MFMAs,
// [j = 8 loads][lane][s = 4]: row n' = lane % 32 = tap * 3 + cout (rows 27..31 zero), input channel (2j + lane / 32) * 4 + s
__global__ void pack_conv_last_kernel(const float* __restrict__ w, float* __restrict__ p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 8 * 64 * 4) return;
    const int s = i & 3, l = (i >> 2) & 63, j = i >> 8;
    const int np = l & 31, ci = (2 * j + (l >> 5)) * 4 + s;
    p[i] = np < 27 ? w[((np % 3) * 64 + ci) * 9 + np / 3] : 0.f;
}

// ---------------------------------------------------------------------------------------------
// First layer: conv0 (1x1) folded into conv1_1 (3 -> 64), NCHW image in, NHWC out, ReLU.  HBM-bound: 12 B read and 256 B
// written per pixel, so the kernel is built around its stores.
//   * the reflect-padded 10 x 34 halo of an 8 x 32 pixel tile is staged once as three colour planes (at most 6 coalesced
//     scalar loads per thread instead of 27 gathered ones);
//   * MFMA with A = weights, B = pixels, K = 28: element e = 3 tap + colour is one dword LDS read (consecutive lanes read
//     consecutive floats of a plane: no bank conflicts), e = 27 is a constant 1 against the folded bias; 14 MFMAs per
//     32 x 32 output tile (round 2 started with K = 10 taps x [r, g, b, 0] = 40: 20 MFMAs); the weights (28 values per lane)
//     stay in registers over the tile walk; a lane ends up holding 4 consecutive channels of one pixel;
//   * each wave passes its two output rows through a private LDS row buffer so that every store instruction writes 1 KiB of
//     contiguous NHWC memory (4 pixels x 64 channels, b128 per lane): 16 store instructions per wave and tile, where the
//     accumulator layout itself would need 64 dword stores.
// ---------------------------------------------------------------------------------------------
constexpr int CF_OSTR = 68;      // floats per pixel of the row buffer: b128 writes of 8 consecutive pixels cover all 32 banks
// Persistent: a workgroup walks tiles blockIdx.x, + gridDim.x, ...; the NEXT tile's halo pixels (at most 2 per thread) are loaded
// into registers before the current tile's MFMAs and stores, so a workgroup's memory latency overlaps its own matrix work
// (one tile per workgroup ran load -> MFMAs -> store strictly in sequence: 3.2 TB/s).
// U8: the image arrives as uint8 HWC [n][H][W][3] (a decoded frame as PIL / the job drivers hold it) and the kernel applies
// torchvision's ToTensor itself, float(v) / 255 with a correctly rounded fp32 division (reference test.py:22, :203): the result is
// bit-identical to encoding the float NCHW tensor, the frame crosses PCIe and HBM as 3 instead of 12 bytes per pixel.
// Round 5 - a fifth of the vector instructions (~80 per wave and tile instead of ~500; every one costs the fp32 matrix pipe ~4 cycles
// here).  Measured effect: none at 1024 x 1024 (74.6-74.9 against 75.0-76.0 us), -3 % on the uint8 entry at 1080p - the kernel is bound
// neither by its vector instructions nor by its barriers or waits (tools/probes/notes/conv_first_round5.md: 56.7 us without a single
// store, 63.1 without an MFMA, 31.0 with neither; a start stagger of the CU's three workgroups made it slower).  Kept for what it is:
//   * halo loads through a buffer descriptor per image: 32-bit per-thread offsets, and for tiles whose halo lies inside the image
//     (92 % at 1024 x 1024) the offset is a per-thread CONSTANT + a per-tile scalar (soffset) - no reflection arithmetic;
//   * two halo images in LDS (alternating per tile; the loop body exists once per parity, so every LDS address is a per-lane
//     register + an immediate) in front of the waves' private row buffers: ONE barrier per tile instead of three, and the next
//     tile's halo is written before this tile's stores are issued;
//   * stores through a descriptor per output ROW that starts at the tile's first pixel of that row and ends at the row's end: the
//     hardware range check drops the pixels right of the image, rows below it are skipped wave-uniformly - no per-lane bounds
//     selects, offsets = two per-lane constants + immediates;
//   * the weights are consumed before the loop (their waits used to sit at their first use INSIDE it, where in every later tile
//     they waited for the previous tile's stores to be acknowledged).
constexpr int CF_HBUF = 1024;    // floats per halo image slot (3 planes x 340 = 1020)
// CFD (diagnostic library only, ADAIN_CF_DIAG; timing-only, wrong results by construction): 1 = no global stores, 2 = no MFMAs,
// 3 = neither (round 5's ablations, tools/probes/cf_diag_ab.sh: 75.7 / 56.7 / 63.1 / 31.0 us at 1024 x 1024).
// DEEP (round 6, the review's last structural A/B; diagnostic library, ADAIN_CF_DEEP=1): the halo loads run TWO tiles ahead - tile t
// starts by writing tile t+1's halo (loaded during tile t-1, long arrived) into the other LDS image and then requests tile t+2's,
// so a load has a whole tile period to arrive instead of one matrix phase; the stores are then unconditional (rows below the image
// get an empty descriptor) so that the compiler's wait for those loads can leave the 16 younger stores in flight.
template <bool U8, int CFD = 0, bool DEEP = false>
__global__ __launch_bounds__(256, 3) void conv_first_kernel(const void* __restrict__ img_any,
                                                            float* __restrict__ out, const float* __restrict__ wpk,
                                                            const float* __restrict__ bias, int H, int W, int tiles_x,
                                                            int tiles_y, int ntiles) {
    constexpr int HALO = 10 * 34;
    __shared__ __attribute__((aligned(16))) float smem_all[2 * CF_HBUF + 4 * 32 * CF_OSTR];
    static_assert(HALO * 3 <= CF_HBUF, "LDS layout");
    (void)bias;                                      // folded into K (pack_conv_first_kernel)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int tiles = tiles_x * tiles_y;
    const size_t plane = (size_t)H * W;
    const bool second = tid + 256 < HALO;          // threads 0..83 own a second halo pixel
    constexpr int EB = U8 ? 3 : 4;                 // bytes between horizontally neighbouring source elements
    // this thread's halo pixels p = tid, tid + 256: position in the 10 x 34 halo and offset from the halo's first pixel in the image
    const int hy0 = tid / 34, hx0 = tid - hy0 * 34, hy1 = (tid + 256) / 34, hx1 = tid + 256 - hy1 * 34;
    const int rel0 = (hy0 * W + hx0) * EB, rel1 = (hy1 * W + hx1) * EB;
    const unsigned img_bytes = (unsigned)min((size_t)0xfffffff0u, plane * 3 * (U8 ? 1 : 4));

    // halo pixels of tile t -> registers (raw bytes as integer bit patterns for U8: the loads stay in flight until the LDS store)
    f32x4 h0, h1;
    auto halo_load = [&](int t) {
        const int pt = t % tiles, img = t / tiles;
        const int tx0 = (pt % tiles_x) * 32, ty0 = (pt / tiles_x) * 8;
        const rsrc_t src = make_rsrc((const char*)img_any + (size_t)img * plane * 3 * (U8 ? 1 : 4), img_bytes);
        const bool interior = ty0 >= 1 && ty0 + 9 <= H && tx0 >= 1 && tx0 + 33 <= W;       // wave-uniform
        auto fetch = [&](int voff, int soff) {
            if constexpr (U8) {
                return f32x4{__uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b8(src, voff, soff, 0)),
                             __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b8(src, voff + 1, soff, 0)),
                             __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b8(src, voff + 2, soff, 0)), 0.f};
            } else {
                const int pb = (int)(plane * 4);      // (images of 2 GiB per plane are refused by the launcher)
                return f32x4{__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(src, voff, soff, 0)),
                             __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(src, voff, soff + pb, 0)),
                             __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(src, voff, soff + 2 * pb, 0)), 0.f};
            }
        };
        // DEEP: the second pixel's load is issued by EVERY thread (threads without one aim past the descriptor: the hardware returns 0
        // without touching memory) - a load under a lane mask becomes a merge with the register's old value, and for a value that is
        // carried over the loop's back edge the compiler then loads into a scratch register and waits for it at once
        constexpr int CF_OOB = 0x7ffffff0;
        if (interior) {
            const int soff = ((ty0 - 1) * W + tx0 - 1) * EB;
            h0 = fetch(rel0, soff);
            if constexpr (DEEP) h1 = fetch(second ? rel1 : CF_OOB, soff);
            else if (second) h1 = fetch(rel1, soff);
        } else {
            auto at = [&](int hy, int hx) { return (reflect1(ty0 + hy - 1, H) * W + reflect1(tx0 + hx - 1, W)) * EB; };
            h0 = fetch(at(hy0, hx0), 0);
            if constexpr (DEEP) h1 = fetch(second ? at(hy1, hx1) : CF_OOB, 0);
            else if (second) h1 = fetch(at(hy1, hx1), 0);
        }
    };
    // halo registers -> LDS image `buf` (planar [r | g | b][10][34]: the K reads walk consecutive floats)
    auto halo_to_lds = [&](float* buf) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if constexpr (U8) {                          // ToTensor: float(v) / 255, correctly rounded
                buf[c * HALO + tid] = __fdiv_rn((float)__float_as_uint(h0[c]), 255.0f);
                if (second) buf[c * HALO + tid + 256] = __fdiv_rn((float)__float_as_uint(h1[c]), 255.0f);
            } else {
                buf[c * HALO + tid] = h0[c];
                if (second) buf[c * HALO + tid + 256] = h1[c];
            }
        }
    };

    int t = blockIdx.x;
    if (t >= ntiles) return;
    halo_load(t);
    if constexpr (DEEP) {        // tile t's halo to image 0 now; the registers then carry tile t + stride's
        halo_to_lds(smem_all);
        if (t + (int)gridDim.x < ntiles) halo_load(t + gridDim.x);
    }
    // this lane's 2 x 14 weights (A operand; K index e = 2 g + lh) and the LDS index of halo element e = (tap, channel) for this
    // lane's pixel column in the wave's first row: held in registers over the tile walk
    float wf[2][14];
    int ka[14];
#pragma unroll
    for (int g = 0; g < 14; ++g) {
#pragma unroll
        for (int c = 0; c < 2; ++c) wf[c][g] = wpk[(c * 14 + g) * 64 + lane];
        const int e = min(2 * g + lh, 26), tap = e / 3;
        ka[g] = (e % 3) * HALO + (tap / 3) * 34 + tap % 3 + wave * 2 * 34 + li;
    }
#pragma unroll
    for (int g = 0; g < 14; ++g)
#pragma unroll
        for (int c = 0; c < 2; ++c) asm volatile("" ::"v"(wf[c][g]));       // arrived BEFORE the loop (see above)
    if constexpr (!DEEP) halo_to_lds(smem_all);
    float* const st = smem_all + 2 * CF_HBUF + wave * (32 * CF_OSTR);        // this wave's row buffer
    // per-lane constants of the epilogue: where this lane's four channels of pixel li go in the row buffer, which 16 bytes of the
    // row buffer it reads back (pixel 4 k + lane / 16, quad lane % 16) and where those go in the output row
    float* const st_w = st + li * CF_OSTR + 4 * lh;
    const float* const st_r = st + (lane >> 4) * CF_OSTR + (lane & 15) * 4;
    const int vo = ((lane >> 4) * 64 + (lane & 15) * 4) * 4;

    auto tile_body = [&](auto PAR) -> bool {
        constexpr int par = decltype(PAR)::value;
        const float* const halo = smem_all + par * CF_HBUF;
        const int pt = t % tiles, img = t / tiles;
        const int tx0 = (pt % tiles_x) * 32, ty0 = (pt / tiles_x) * 8;
        __syncthreads();                                 // halo image `par` is complete (and image 1 - par is free, see above)
        const int tn = t + gridDim.x;
        if constexpr (DEEP) {
            // the registers hold tile tn's halo since the previous tile: into the free image, then the request for the tile after it
            if (tn < ntiles) halo_to_lds(smem_all + (1 - par) * CF_HBUF);
            if (tn + (int)gridDim.x < ntiles) halo_load(tn + gridDim.x);
        } else {
            if (tn < ntiles) halo_load(tn);              // in flight during this tile's MFMAs
        }

        f32x16 acc[2][2];      // [channel tile][row of this wave]
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][m][r] = 0.f;
#pragma unroll
        for (int g = 0; g < 14; ++g) {
            float xf[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                xf[m] = halo[ka[g] + m * 34];
                if (g == 13) xf[m] = lh ? 1.0f : xf[m];                       // e = 27: the bias row
            }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    if constexpr (CFD == 2 || CFD == 3) acc[c][m][g & 15] += wf[c][g] * xf[m];      // timing-only: operands kept alive, no MFMA
                    else acc[c][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[c][g], xf[m], acc[c][m], 0, 0, 0);
                }
        }
        // the next tile's halo goes to the OTHER LDS image here, before this tile's stores are issued (loads and stores share one
        // in-flight counter on gfx9: consumed later, the prefetched loads would wait for every store issued in between)
        if constexpr (!DEEP)
            if (tn < ntiles) halo_to_lds(smem_all + (1 - par) * CF_HBUF);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int y = ty0 + wave * 2 + m;            // wave-uniform
            // lane (li, lh) holds channels 32 c + 8 q + 4 lh + (0..3) of pixel li in acc[c][m][4 q .. 4 q + 3]
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
#pragma unroll
                    // ReLU as ONE instruction: max over the BIT PATTERNS as signed integers (negative floats, -0 and negative NaNs are
                    // negative integers -> +0; everything else unchanged: fmaxf(x, 0) for every x).  fmaxf itself costs two v_max
                    // here - accumulators straight out of an MFMA get a canonicalising v_max(x, x) first.  (The bias came in through K.)
                    for (int e = 0; e < 4; ++e) {
                        const int xi = __float_as_int(acc[c][m][4 * q + e]);
                        v[e] = __int_as_float(xi > 0 ? xi : 0);
                    }
                    *(f32x4*)(st_w + 32 * c + 8 * q) = v;
                }
            if (DEEP || y < H) {
                // this output row from the tile's first pixel to the END OF THE ROW: pixels right of the image fall outside the
                // descriptor and are dropped by the hardware; the row buffer is private to the wave and LDS operations of one wave
                // complete in order: no barrier.  (DEEP: a row below the image gets an EMPTY descriptor instead of a branch.)
                const bool row_ok = y < H;
                const rsrc_t dst = make_rsrc(out + (((size_t)img * H + (row_ok ? y : 0)) * W + tx0) * 64, row_ok ? (unsigned)(W - tx0) * 256u : 0u);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const f32x4 v = *(const f32x4*)(st_r + 4 * k * CF_OSTR);
                    if constexpr (CFD == 1 || CFD == 3) asm volatile("" ::"v"(v));        // timing-only: no global stores
                    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), dst, vo + k * 1024, 0, 0);
                }
            }
        }
        if (tn >= ntiles) return false;
        t = tn;
        return true;
    };
    for (;;) {
        if (!tile_body(std::integral_constant<int, 0>{})) break;
        if (!tile_body(std::integral_constant<int, 1>{})) break;
    }
}

// ---------------------------------------------------------------------------------------------
// Last layer: 64 -> 3, no ReLU, NHWC in, NCHW image out (net.py:33-35).  N = 3 would waste 90 % of an MFMA tile as a
// convolution, so the layer runs as a GEMM over INPUT pixels followed by a shifted sum:
//     T[p][tap * 3 + co] = sum_ci x[p][ci] * w[co][ci][tap]          (K = 64, 27 of 32 MFMA rows used)
//     out[y][x][co]      = bias[co] + sum_tap T[(y + dy, x + dx)][tap * 3 + co]
// Every input pixel of the reflect-padded 18 x 34 halo of a 16 x 32 tile goes global memory -> registers -> B operand once
// (no input staging in LDS, no 9-fold operand re-read); the weights are the A operand, 32 registers loaded once; T meets in
// LDS (27 planes of 640 pixel slots) and each thread sums 9 x 3 values for its two pixels.  Per tile 20 x 32 MFMAs (4 waves x 5
// groups of 32 halo pixels) = 5.4 GFLOP executed at 1024 x 1024 against 268 MB read: HBM-bound.
// ---------------------------------------------------------------------------------------------
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    [&]<int... I>(std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, N>{});
}

// CLD (diagnostic library only, ADAIN_CL_DIAG): 1 = no MFMAs (memory side alone: 60-63 us at 1024 x 1024), 2 = every load of a
// group from one address (matrix side alone: 57 us), 4 = tiles in launch order (no XCD ranges: 68 us); product: 63-65 us.
// TH = tile height: 16 (612 halo pixels = 20 groups of 32, 5 per wave; T = 69 KB: two workgroups per CU; the product) or 12
// (476 pixels = 15 groups, the fourth wave takes 3; T = 52 KB: three workgroups per CU; diagnostic library, ADAIN_CL_TH=12:
// 64.3-64.5 us against 64.5-65.5 on the same box - the third workgroup buys nothing here, unlike in conv_first).
// U8OUT (round 6): the image leaves as what torchvision's save_image makes of it (reference test.py:243-244) - x * 255 + 0.5, clamped to
// [0, 255], truncated, uint8 HWC [n][H][W][3] - straight from the registers that hold the float pixel: adain_quantize_u8's arithmetic
// operation for operation (the multiply and the add round separately: __fmul_rn / __fadd_rn, this file is compiled with contraction
// on), so the bytes are those of conv_last + quantize_u8 (tests/test_gpu_stylize_u8.py: torch.equal) without the float image's
// 12 + 12 bytes per pixel of HBM traffic and without the second launch.  adain_stylize_u8 without a mask ends with it.
__device__ __forceinline__ unsigned cl_quant1(float x) {
    float v = __fadd_rn(__fmul_rn(x, 255.0f), 0.5f);
    v = fminf(fmaxf(v, 0.f), 255.f);
    return (unsigned)v;
}
template <int CLD, int TH, bool U8OUT = false>
__global__ __launch_bounds__(256, TH == 12 ? 3 : 2) void conv_last_kernel(const float* __restrict__ in, void* __restrict__ out_any,
                                                                          const float* __restrict__ wpk, const float* __restrict__ bias,
                                                                          int H, int W, int tiles_x, int tiles_y) {
    constexpr int NPX = (TH + 2) * HW_, NGRP = (NPX + 31) / 32, GPW = (NGRP + 3) / 4, SLOTS = NGRP * 32;
    __shared__ float T[27 * SLOTS];                                          // [n'][pixel slot]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 31, lh = lane >> 5;
    const int tiles = tiles_x * tiles_y;
    // the workgroups of one XCD (blockIdx % 8) take a contiguous range of the tile list: the halo rows and columns that
    // neighbouring tiles share are re-read from that XCD's L2 (speed only)
    int lid = blockIdx.x;
    if (CLD != 4 && (gridDim.x & 7) == 0) lid = (lid & 7) * (gridDim.x >> 3) + (lid >> 3);
    const int pt = lid % tiles, img = lid / tiles;
    const int tx0 = (pt % tiles_x) * TW, ty0 = (pt / tiles_x) * TH;
    // the source descriptor starts at the first row the tile's halo can touch: offsets stay inside TH + 2 rows
    const int srow0 = max(ty0 - 1, 0);
    const size_t sleft = (size_t)(H - srow0) * W * 256;
    const rsrc_t src = make_rsrc(in + ((size_t)img * H + srow0) * W * 64, sleft < 0x7ffffff0ull ? (unsigned)sleft : 0x7ffffff0u);
    const rsrc_t wsr = make_rsrc(wpk, 8 * 64 * 16);

    constexpr int PF = 2, RB = PF + 1;                            // pixel groups in flight ahead of the one being multiplied (3: 66.5 us, 2: 64.4-65.5)
    f32x4 bx[RB][8];
    // this wave's g-th group = group 4 g + wave of the tile: this lane's halo pixel, 8 x 16 bytes of its 256 (lane half lh takes
    // the odd quads)
    auto load_group = [&](auto G) {
        constexpr int g = decltype(G)::value;
        const int f = min((4 * g + wave) * 32 + px, NPX - 1);
        const int hy = (f * 241) >> 13, hx = f - hy * HW_;                   // f / 34 for f < 1024
        const int y = reflect1(ty0 + hy - 1, H), x = reflect1(tx0 + hx - 1, W);
        const int off = ((y - srow0) * W + x) * 256 + lh * 16;
#pragma unroll
        for (int j = 0; j < 8; ++j) bx[g % RB][j] = buf_load4(src, off, CLD == 2 ? 0 : j * 32);
    };
    f32x4 wq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wq[j] = buf_load4(wsr, lane * 16, j * 1024);
    __builtin_amdgcn_sched_barrier(0);       // weights and two pixel groups in flight before the first MFMA (hipcc sinks the loads otherwise)
    static_for<PF>([&](auto G) { if constexpr (decltype(G)::value < GPW) load_group(G); });
    __builtin_amdgcn_sched_barrier(0);

    static_for<GPW>([&](auto G) {
        constexpr int g = decltype(G)::value;
        if constexpr (g + PF < GPW) load_group(std::integral_constant<int, g + PF>{});   // (a group past the tile's last repeats its last pixel)
        __builtin_amdgcn_sched_barrier(0);
        if (4 * g + 3 < NGRP || 4 * g + wave < NGRP) {                        // wave-uniform, and only the last round can be short
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if constexpr (CLD == 1) acc[(j * 4 + s) & 15] += wq[j][s] * bx[g % RB][j][s];
                    else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[j][s], bx[g % RB][j][s], acc, 0, 0, 0);
                }
            // D[row n' = (r & 3) + 8 (r >> 2) + 4 lh][column = this lane's pixel] -> plane n' of T: a store instruction writes 32
            // consecutive floats per lane half, and the shifted sum below reads consecutive floats too (no bank conflicts either way)
            float* rec = T + (lh * 4) * SLOTS + (4 * g + wave) * 32 + px;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int np = (r & 3) + 8 * (r >> 2);                        // + 4 lh
                if (np + 4 < 27 || (np < 27 && lh == 0)) rec[np * SLOTS] = acc[r];
            }
        }
    });
    __syncthreads();

    const int ox = tid & 31, oy = tid >> 5;
    const float b0 = bias[0], b1 = bias[1], b2 = bias[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int yy = oy + 8 * r;
        if (yy >= TH) break;
        float o0 = b0, o1 = b1, o2 = b2;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const float* tp = T + (dy * 3 + dx) * 3 * SLOTS + (yy + dy) * HW_ + ox + dx;
                o0 += tp[0]; o1 += tp[SLOTS]; o2 += tp[2 * SLOTS];
            }
        const int y = ty0 + yy, x = tx0 + ox;
        if (y < H && x < W) {
            if constexpr (U8OUT) {
                // handled below: four neighbouring lanes' pixels leave as ONE 12-byte store (byte stores at a stride of 3 cost the
                // kernel a third of its time)
            } else {
                float* __restrict__ o = (float*)out_any + (size_t)img * 3 * H * W + (size_t)y * W + x;
                o[0] = o0;
                o[(size_t)H * W] = o1;
                o[(size_t)2 * H * W] = o2;
            }
        }
        if constexpr (U8OUT) {
            // this lane's pixel as 24 bits, then lane 4 q gathers the pixels of lanes 4 q .. 4 q + 3 (the same output row: a row is 32
            // lanes) into three dwords; W is a multiple of 8 (the decoder's output is 8 hc x 8 wc) and the launcher checked the base
            // pointer's alignment, so the 12 bytes sit on a dword boundary and a group of four pixels is inside the row or outside it
            const unsigned w0 = cl_quant1(o0) | (cl_quant1(o1) << 8) | (cl_quant1(o2) << 16);
            const unsigned w1 = __shfl_down(w0, 1), w2 = __shfl_down(w0, 2), w3 = __shfl_down(w0, 3);
            if ((ox & 3) == 0 && y < H && x < W) {
                using u32x3 = __attribute__((ext_vector_type(3))) unsigned;
                u32x3 v;
                v[0] = w0 | (w1 << 24);
                v[1] = (w1 >> 8) | (w2 << 16);
                v[2] = (w2 >> 16) | (w3 << 8);
                *(u32x3*)((uint8_t*)out_any + (((size_t)img * H + y) * W + x) * 3) = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------------------------
int launch_pack_conv_first(const float* w0, const float* b0, const float* w1, const float* b1, float* p,
                           float* bias_out, hipStream_t s) {
    hipLaunchKernelGGL(pack_conv_first_kernel, dim3(7), dim3(256), 0, s, w0, b0, w1, b1, p, bias_out);
    return check_launch("pack_conv_first");
}

int launch_pack_conv_last(const float* w, float* p, hipStream_t s) {
    hipLaunchKernelGGL(pack_conv_last_kernel, dim3(8), dim3(256), 0, s, w, p);
    return check_launch("pack_conv_last");
}

int launch_conv_first(const void* img, int u8, float* out, const float* packed, const float* bias, int n, int H, int W,
                      hipStream_t s) {
    if (H < 2 || W < 2 || n < 1) { set_error("conv_first: H, W must be >= 2, got %dx%d", H, W); return -1; }
    if ((size_t)W * 256 * 8 >= 0x7ffffff0ULL) { set_error("conv_first: eight 64-channel rows of width %d reach 2 GiB", W); return -1; }
    if ((size_t)H * W * 12 >= 0x7ffffff0ULL) { set_error("conv_first: an image of %d x %d pixels reaches 2 GiB as three float planes", H, W); return -1; }
    const int tx = (W + 31) / 32, ty = (H + 7) / 8;
    const long long ntiles = (long long)tx * ty * n;
    if (ntiles > 0x7fffffffLL) { set_error("conv_first: bad grid"); return -1; }
    const int cus = device_cu_count();
    if (cus <= 0) { set_error("conv_first: device query failed"); return -1; }
    long long per_cu = 3;                                                  // 3 workgroups per CU walk the tiles (same box: 2: 85 us, 3: 80, 4: 90)
#ifdef ADAIN_DIAG
    static const int wgs_env = tune_env("ADAIN_CF_WGS", 3);
    per_cu = wgs_env;
#endif
    const long long grid = ntiles < per_cu * cus ? ntiles : per_cu * cus;
#ifdef ADAIN_DIAG
    // timing-only ablations and the two-tiles-ahead variant (tools/probes/cf_diag_ab.sh; float entry only for the ablations)
    static const int cfd = tune_env("ADAIN_CF_DIAG", 0), deep = tune_env("ADAIN_CF_DEEP", 0);
#define CF_LAUNCH(...) hipLaunchKernelGGL((conv_first_kernel<__VA_ARGS__>), dim3((unsigned)grid), dim3(256), 0, s, img, out, packed, bias, H, W, tx, ty, (int)ntiles)
    if (deep) {
        if (u8) CF_LAUNCH(true, 0, true);
        else if (cfd == 3) CF_LAUNCH(false, 3, true);
        else CF_LAUNCH(false, 0, true);
        return check_launch("conv_first(deep)");
    }
    if (!u8 && cfd >= 1 && cfd <= 3) {
        if (cfd == 1) CF_LAUNCH(false, 1);
        else if (cfd == 2) CF_LAUNCH(false, 2);
        else CF_LAUNCH(false, 3);
        return check_launch("conv_first(diag)");
    }
#undef CF_LAUNCH
#endif
    if (u8) hipLaunchKernelGGL(conv_first_kernel<true>, dim3((unsigned)grid), dim3(256), 0, s, img, out, packed, bias, H, W, tx, ty, (int)ntiles);
    else hipLaunchKernelGGL(conv_first_kernel<false>, dim3((unsigned)grid), dim3(256), 0, s, img, out, packed, bias, H, W, tx, ty, (int)ntiles);
    return check_launch("conv_first");
}

int launch_conv_last(const float* in, float* out_f32, const float* packed, const float* bias, int n, int H, int W,
                     hipStream_t s, uint8_t* out_u8) {
    void* const out = out_u8 ? (void*)out_u8 : (void*)out_f32;
    if (H < 2 || W < 2 || n < 1) { set_error("conv_last: H, W must be >= 2, got %dx%d", H, W); return -1; }
    if ((size_t)W * 256 * 18 >= 0x7ffffff0ULL) { set_error("conv_last: eighteen 64-channel rows of width %d reach 2 GiB", W); return -1; }
    int th = 16;
#ifdef ADAIN_DIAG
    static const int th_env = tune_env("ADAIN_CL_TH", 16);
    th = th_env == 12 ? 12 : 16;
#endif
    const int tx = (W + TW - 1) / TW, ty = (H + th - 1) / th;
    if ((long long)tx * ty * n > 0x7fffffffLL) { set_error("conv_last: bad grid"); return -1; }
    const dim3 grid((unsigned)(tx * ty * n));
#ifdef ADAIN_DIAG
    static const int cld = tune_env("ADAIN_CL_DIAG", 0);
    if (cld == 1) hipLaunchKernelGGL((conv_last_kernel<1, 16>), grid, dim3(256), 0, s, in, out, packed, bias, H, W, tx, ty);
    else if (cld == 2) hipLaunchKernelGGL((conv_last_kernel<2, 16>), grid, dim3(256), 0, s, in, out, packed, bias, H, W, tx, ty);
    else if (cld == 4) hipLaunchKernelGGL((conv_last_kernel<4, 16>), grid, dim3(256), 0, s, in, out, packed, bias, H, W, tx, ty);
    else if (th == 12) hipLaunchKernelGGL((conv_last_kernel<0, 12>), grid, dim3(256), 0, s, in, out, packed, bias, H, W, tx, ty);
    else
#endif
    if (out_u8 && ((W & 3) || ((uintptr_t)out_u8 & 3))) { set_error("conv_last: the uint8 form needs W %% 4 == 0 and a 4-byte aligned image"); return -1; }
    if (out_u8) hipLaunchKernelGGL((conv_last_kernel<0, 16, true>), grid, dim3(256), 0, s, in, out, packed, bias, H, W, tx, ty);
    else hipLaunchKernelGGL((conv_last_kernel<0, 16>), grid, dim3(256), 0, s, in, out, packed, bias, H, W, tx, ty);
    return check_launch("conv_last");
}

}  // namespace adain
