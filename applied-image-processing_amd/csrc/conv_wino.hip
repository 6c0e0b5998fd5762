// Winograd F(2x2, 3x3) form of the 3x3 convolution for the K-heavy layers (cin >= 128), gfx950.
//
// The direct implicit GEMM (conv.hip) is bound by the fp32 MFMA rate; the only lever left is fewer
// multiplies.  F(2x2,3x3) computes a 2x2 output tile from a 4x4 input patch with 16 multiplies per
// (cin, cout) pair instead of 36: Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A, 2.25x fewer MFMA flops.
// In fp32 its rounding error is ~2x that of the direct form (3.9e-7 vs 1.7e-7 relative on these layers),
// far inside the path's tolerance.
//
// Mapping (one workgroup of 8 waves per CU, 128 KiB of LDS):
//   * block tile = 8 x 32 output pixels = 64 Winograd tiles (4 tile rows x 16 tile columns) x 64 channels;
//   * the 16 transform positions xi = (i, j) are 16 independent GEMMs [64 tiles x cin] x [cin x 64].  Wave
//     (mh, i) owns transform ROW i (its 4 positions j) for M-tile mh (32 tiles): 4 x 2 MFMA tiles of
//     32x32 = 128 accumulator registers; v_mfma_f32_32x32x2_f32 as in the direct kernel;
//   * A operand: per 8-channel chunk the raw reflect-padded 10 x 34 halo is staged into LDS (global ->
//     registers -> LDS, prefetched two chunks ahead), every thread transforms one column of one patch
//     (B^T d B: 32 adds) and writes V[xi][tile][8 ch] (row stride 48 B: conflict-free b128 reads);
//     the transform of chunk c+1 is interleaved with the MFMAs of chunk c; one barrier per chunk;
//   * B operand: transformed weights U = G g G^T are packed on the device once per weight set into
//     per-lane fragment order [cout/64][row i][chunk][j][n][lane][4] and streamed straight into VGPRs
//     (the two M-tile waves of a row share them through L1);
//   Measured while tuning (tools/tune_conv.py, 256->256 at 256x256): 8 waves / one block per CU 228 TF/s (algorithmic),
//   4 waves / two blocks per CU 237; without the input transform (timing-only build) 281, i.e. the fused transform
//   costs ~19 % (14 % its LDS reads + adds, 3-5 % its V stores); fetching the patches straight from global memory
//   instead of the raw-halo LDS image was 3 % SLOWER; pinning the issue order, scalar instead of packed adds and
//   reading all A fragments at chunk start are each within +-1 %.
//   * epilogue: each wave reduces its row (P = M A, two partial matrices), the 4 rows meet in LDS
//     (all 128 KiB), then every lane owns one channel and finishes Y = A^T P for 8 tiles: bias, ReLU,
//     optional 2x2 max-pool of the output (the 4 outputs of a Winograd tile ARE one pool window), stores of
//     256 contiguous bytes per pixel.
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "device_utils.h"

namespace adain {

constexpr int WKC = 8;                 // channels per chunk
constexpr int WVSTR = 12;              // floats per V row (8 channels + 4 pad)
constexpr int WHALO_W = 34;            // raw halo width (32 + 2)

// geometry by MH = number of 32-tile M-tiles per block (MH*4 waves):
//   MH = 2: 8 x 32 output pixels, 8 waves, 128 KiB LDS, one block per CU
//   MH = 1: 4 x 32 output pixels, 4 waves, 69 KiB LDS, two independent blocks per CU (their barriers overlap):
//           5-10 % faster on every config-2 layer, the default
template <int MH>
struct WinoGeo {
    static constexpr int TILES = 32 * MH;
    static constexpr int VSTAGE = 16 * TILES * WVSTR;
    static constexpr int HALO = (4 * MH + 2) * WHALO_W;
    static constexpr int RBUF = HALO * WVSTR;
    static constexpr int PEX = MH * 4 * 2 * 32 * 64;
    static constexpr int LDS = (2 * VSTAGE + 2 * RBUF) > PEX ? (2 * VSTAGE + 2 * RBUF) : PEX;
    static constexpr int NTHR = MH * 256;
    static constexpr int RITEMS = (HALO * 2 + NTHR - 1) / NTHR;
};

// OIHW [cout][cin][3][3] -> U = G g G^T packed as [cout/64][i 4][cin/8][j 4][n 2][lane 64][s 4]:
//   value = U[xi = 4 i + j][cout = 64 ct + 32 n + (lane & 31)][cin = 8 chunk + 4 (lane >> 5) + s]
__global__ void pack_wino_kernel(const float* __restrict__ w, float* __restrict__ p, int cin, int cout) {
    const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
    const size_t total = (size_t)cin * cout * 16;
    const int nch = cin / WKC;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        size_t r = idx;
        const int s = r & 3; r >>= 2;
        const int lane = r & 63; r >>= 6;
        const int n = r & 1; r >>= 1;
        const int j = r & 3; r >>= 2;
        const int chunk = r % nch; r /= nch;
        const int i = r & 3; r >>= 2;
        const int ct = (int)r;
        const int co = ct * 64 + n * 32 + (lane & 31), ci = chunk * WKC + 4 * (lane >> 5) + s;
        const float* g = w + ((size_t)co * cin + ci) * 9;
        float u = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) u += G[i][a] * g[a * 3 + b] * G[j][b];
        p[idx] = u;
    }
}

template <int MODE, int MH>
__global__ __launch_bounds__(MH * 256, 2) void conv3x3_wino_kernel(ConvArgs a) {
    using Geo = WinoGeo<MH>;
    constexpr int WTILES = Geo::TILES, WVSTAGE = Geo::VSTAGE, WHALO = Geo::HALO, WRBUF = Geo::RBUF, NTHR = Geo::NTHR;
    constexpr int RITEMS = Geo::RITEMS;
    __shared__ __attribute__((aligned(16))) float smem[Geo::LDS];
    float* const Vs = smem;                      // 2 stages
    float* const Rs = smem + 2 * WVSTAGE;        // 2 raw halo buffers

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mh = wave >> 2, wi = wave & 3;     // M-tile, transform row
    const int li = lane & 31, lh = lane >> 5;

    // block -> (pixel tile, channel tile, image), XCD-aware order as in conv.hip
    const int tiles = a.tiles_x * a.tiles_y;
    const int nct = a.cout / 64;
    int lid = blockIdx.x;
    if (a.xcd_order && (gridDim.x & 7) == 0) lid = (lid & 7) * (gridDim.x >> 3) + (lid >> 3);
    const int ct = lid % nct; lid /= nct;
    const int pt = lid % tiles;
    const int img = lid / tiles;
    const int tx0 = (pt % a.tiles_x) * 32, ty0 = (pt / a.tiles_x) * (4 * MH);
    const int nch = a.cin / WKC;

    const rsrc_t src = make_rsrc(a.in + (size_t)img * a.Hs * a.Ws * a.cin, (unsigned)a.Hs * a.Ws * a.cin * 4u);
    const rsrc_t wsr = make_rsrc(a.wpk, (unsigned)a.cin * a.cout * 64u);

    // ---- raw halo staging: HALO*2 items (pixel, quad) over the block's threads ----------------------------------
    int roff[RITEMS];
#pragma unroll
    for (int k = 0; k < RITEMS; ++k) {
        const int idx = tid + k * NTHR;
        const int hp = min(idx >> 1, WHALO - 1), q = idx & 1;
        const int hy = hp / WHALO_W, hx = hp - hy * WHALO_W;
        int y = reflect1(ty0 + hy - 1, a.H), x = reflect1(tx0 + hx - 1, a.W);
        if (MODE == SRC_UP2X) { y >>= 1; x >>= 1; }
        roff[k] = ((y * a.Ws + x) * a.cin + q * 4) * 4;
    }
    f32x4 rawreg[RITEMS];
    auto raw_load = [&](int chunk) {
#pragma unroll
        for (int k = 0; k < RITEMS; ++k) rawreg[k] = buf_load4(src, roff[k], chunk * WKC * 4);
    };
    auto raw_store = [&](float* buf) {
#pragma unroll
        for (int k = 0; k < RITEMS; ++k) {
            const int idx = tid + k * NTHR;
            if (idx < WHALO * 2) *(f32x4*)(buf + (idx >> 1) * WVSTR + (idx & 1) * 4) = rawreg[k];
        }
    };

    // ---- input transform unit of this thread: column tc of the patch of tile ut, channel quad uq ----------------
    // (tc is the same for a whole wave: readfirstlane makes that provable, so its branches below are scalar)
    const int tc = __builtin_amdgcn_readfirstlane(tid / (NTHR / 4)), ut = (tid % (NTHR / 4)) >> 1, uq = tid & 1;
    const int u_base = ((2 * (ut >> 4)) * WHALO_W + 2 * (ut & 15)) * WVSTR + uq * 4;     // patch origin in the raw buffer
    const int colA = tc == 0 ? 0 : 1, colB = tc == 3 ? 3 : 2;
    const int v_base = (tc * WTILES + ut) * WVSTR + uq * 4;                             // V[xi = 4 i + tc][ut][uq*4]
    f32x4 dA[4], dB[4], tr[4];
    auto xf_read = [&](const float* rbuf) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dA[r] = *(const f32x4*)(rbuf + u_base + (r * WHALO_W + colA) * WVSTR);
            dB[r] = *(const f32x4*)(rbuf + u_base + (r * WHALO_W + colB) * WVSTR);
        }
    };
    auto xf_rows = [&]() {      // (d B)[r][tc]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (tc == 1) tr[r] = dA[r] + dB[r];
            else if (tc == 2) tr[r] = dB[r] - dA[r];
            else tr[r] = dA[r] - dB[r];
        }
    };
    auto xf_write = [&](float* vbuf) {   // B^T (.) : rows t0 - t2, t1 + t2, t2 - t1, t1 - t3
        *(f32x4*)(vbuf + v_base + 0 * 4 * WTILES * WVSTR) = tr[0] - tr[2];
        *(f32x4*)(vbuf + v_base + 1 * 4 * WTILES * WVSTR) = tr[1] + tr[2];
        *(f32x4*)(vbuf + v_base + 2 * 4 * WTILES * WVSTR) = tr[2] - tr[1];
        *(f32x4*)(vbuf + v_base + 3 * 4 * WTILES * WVSTR) = tr[1] - tr[3];
    };

    // ---- MFMA operands ---------------------------------------------------------------------------------------------
    const int a_base = ((4 * wi) * WTILES + mh * 32 + li) * WVSTR + lh * 4;   // + j * WTILES * WVSTR
    const int wvo = lane * 16;
    int wso = ((ct * 4 + wi) * nch) * 8192;     // byte offset of this wave's record stream: [chunk][j][n] x 1 KiB
    constexpr int PF = 3, RING = 4;
    f32x4 bq[RING][2];
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int n = 0; n < 2; ++n) bq[p][n] = buf_load4(wsr, wvo, wso + p * 2048 + n * 1024);

    f32x16 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][n][r] = 0.f;

    // ---- prologue: raw chunk 0 -> V stage 0, raw chunk 1 -> raw buffer 1 ------------------------------------------
    raw_load(0);
    raw_store(Rs);
    __syncthreads();
    if (nch > 1) raw_load(1);
    xf_read(Rs);
    xf_rows();
    xf_write(Vs);
    if (nch > 1) raw_store(Rs + WRBUF);
    __syncthreads();

    for (int c = 0; c < nch; ++c) {
        const float* Vc = Vs + (c & 1) * WVSTAGE + a_base;
        float* Vn = Vs + ((c + 1) & 1) * WVSTAGE;
        const float* Rn = Rs + ((c + 1) & 1) * WRBUF;
        const bool xf = c + 1 < nch, pre = c + 2 < nch;
        if (pre) raw_load(c + 2);
        // all four A fragments of the chunk up front: their LDS reads must not queue behind the transform's LDS traffic
        f32x4 aq[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) aq[j] = *(const f32x4*)(Vc + j * WTILES * WVSTR);
        auto step = [&](auto JJ) {
            constexpr int j = decltype(JJ)::value;
#pragma unroll
            for (int n = 0; n < 2; ++n) bq[(j + PF) % RING][n] = buf_load4(wsr, wvo, wso + (j + PF) * 2048 + n * 1024);
            // the input transform of the NEXT chunk rides along with this chunk's MFMAs
            if (xf) {
                if constexpr (j == 0) xf_read(Rn);
                if constexpr (j == 1) xf_rows();
                if constexpr (j == 2) xf_write(Vn);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[j][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[j][s], bq[j % RING][n][s], acc[j][n], 0, 0, 0);
            {
                // issue order of the step: loads first (next A fragment, weight prefetch, the transform's LDS reads),
                // then the 8 MFMAs with the transform's VALU / LDS-write work dealt out between them, so that this
                // wave's non-matrix instructions issue under its own MFMAs instead of in front of them
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                // one MFMA, then one slice of the transform (step 0: a raw-patch LDS read; steps 1, 2: VALU adds; step 2
                // also the V stores).  The chunk's first MFMA must only wait for its A fragment, not for the raw reads.
#define WINO_SLOT(M)                                                                                      \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                        \
                if constexpr (j == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                  \
                if constexpr (j == 1 || j == 2) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);        \
                if constexpr (j == 2 && ((M) & 1)) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                WINO_SLOT(0) WINO_SLOT(1) WINO_SLOT(2) WINO_SLOT(3) WINO_SLOT(4) WINO_SLOT(5) WINO_SLOT(6) WINO_SLOT(7)
#undef WINO_SLOT
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
        wso += 8192;
        if (pre) raw_store(Rs + (c & 1) * WRBUF);
        __syncthreads();
    }

    // ---- epilogue: row reduction P = M A, exchange through LDS, Y = A^T P -----------------------------------------
    // P[b=0] = M0 + M1 + M2 ; P[b=1] = M1 - M2 - M3.  LDS P[mh][i][b][tile 32][co 64]
    {
        float* Pw = smem + ((mh * 4 + wi) * 2) * (32 * 64);
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int trow = (r & 3) + 8 * (r >> 2) + 4 * lh;      // tile inside the M-tile (C layout row)
                const float p0 = acc[0][n][r] + acc[1][n][r] + acc[2][n][r];
                const float p1 = acc[1][n][r] - acc[2][n][r] - acc[3][n][r];
                Pw[trow * 64 + n * 32 + li] = p0;
                Pw[32 * 64 + trow * 64 + n * 32 + li] = p1;
            }
    }
    __syncthreads();
    {
        const int co = ct * 64 + lane;
        const float bias = a.bias[co];
        const float* Pm = smem + (mh * 4 * 2) * (32 * 64);
        const int Hp = (a.H + 1) >> 1, Wp = (a.W + 1) >> 1;
#pragma unroll 2
        for (int t = 0; t < 8; ++t) {
            const int tl = wi * 8 + t;                 // tile inside the M-tile
            const int tile = mh * 32 + tl;
            const int oy = ty0 + 2 * (tile >> 4), ox = tx0 + 2 * (tile & 15);
            if (oy >= a.H || ox >= a.W) continue;
            float P[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int b = 0; b < 2; ++b) P[i][b] = Pm[((i * 2 + b) * 32 + tl) * 64 + lane];
            float y[2][2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                y[0][b] = P[0][b] + P[1][b] + P[2][b] + bias;
                y[1][b] = P[1][b] - P[2][b] - P[3][b] + bias;
            }
            if (a.relu) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int b = 0; b < 2; ++b) y[i][b] = fmaxf(y[i][b], 0.f);
            }
            const bool row1 = oy + 1 < a.H, col1 = ox + 1 < a.W;
            if (a.pool_out) {
                float v = y[0][0];
                if (col1) v = fmaxf(v, y[0][1]);
                if (row1) {
                    v = fmaxf(v, y[1][0]);
                    if (col1) v = fmaxf(v, y[1][1]);
                }
                a.out[(((size_t)img * Hp + (oy >> 1)) * Wp + (ox >> 1)) * a.cout + co] = v;
            } else {
                float* o = a.out + (((size_t)img * a.H + oy) * a.W + ox) * a.cout + co;
                o[0] = y[0][0];
                if (col1) o[a.cout] = y[0][1];
                if (row1) {
                    o[(size_t)a.W * a.cout] = y[1][0];
                    if (col1) o[(size_t)a.W * a.cout + a.cout] = y[1][1];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Second form: the transformed input never touches LDS.  Wave i (transform row i) needs, for its A operand, exactly
// V[i][0..3] of tile `lane & 31` for the 4 channels `4 * (lane >> 5) ..`, and row i of B^T d B depends on only TWO
// rows of the 4x4 patch (B^T row i has two non-zeros).  So every lane reads its 2 x 4 patch pixels (b128 each) from
// the raw halo image, combines them (16 + 16 adds) and holds the four A fragments of the next chunk in registers:
// no V image (48 KB), no V stores, no A-fragment reads, and the raw halo is staged 16 channels at a time, so the
// workgroup barrier comes once per 64 MFMAs per wave instead of once per 32.
// (The patch reads have 2-way LDS bank conflicts; a 1-bit XOR swizzle of the channel quads removes them and measured
// 3-4 % SLOWER — the extra address arithmetic costs more than the conflicts; timing-only builds: without the input
// transform +17 %, without staging + barrier +6 %, without weight loads +5 %.)
// ---------------------------------------------------------------------------------------------------------------
// DIAG 1: per-wave phase stamps; 2: + LDS padded to one block per CU; 3 / 4: + a shader-clock stamp per MFMA step of the
// first 32 workgroups (3: one block per CU, 4: two)
template <int MODE, bool PIN = true, int DIAG = 0>
__global__ __launch_bounds__(256, (DIAG == 2 || DIAG == 3) ? 1 : 2) void conv3x3_wino2_kernel(ConvArgs a) {
    constexpr int KR = 16;                  // channels per raw stage = 2 MFMA chunks of 8
    constexpr int RSTR = KR + 4;            // floats per halo pixel (80 B: conflict-free b128 for 16 distinct tiles)
    constexpr int HALO = 6 * WHALO_W;       // 6 x 34 halo of a 4 x 32 pixel tile
    constexpr int RBUF = 256 * RSTR;        // 204 halo pixels, rounded up to the 4 x 256 staging items (no predicated stores)
    constexpr int PEX = 4 * 2 * 32 * 64;    // epilogue exchange (64 KiB) >= 2 raw buffers
    constexpr int RITEMS = 4;
    static_assert(HALO * 4 <= RITEMS * 256, "staging items");
    static_assert(2 * RBUF <= PEX, "LDS layout");
    __shared__ __attribute__((aligned(16))) float smem[(DIAG == 2 || DIAG == 3) ? PEX + 8192 : (DIAG == 4 ? PEX + 4096 : PEX)];
    unsigned* const steplog = (unsigned*)(smem + PEX);     // DIAG >= 3: [wave][128] low words of s_memtime
    int nlog = 0;
    float* const Rs = smem;
    unsigned long long stamp[4] = {0, 0, 0, 0};
    if constexpr (DIAG) stamp[0] = __builtin_amdgcn_s_memrealtime();

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wi = __builtin_amdgcn_readfirstlane(tid >> 6);      // transform row of this wave
    const int li = lane & 31, lh = lane >> 5;

    const int tiles = a.tiles_x * a.tiles_y;
    const int nct = a.cout / 64;
    int lid = blockIdx.x, ct, pt, img;
    if (a.xcd_order == 1) {          // channel tile fastest inside a contiguous per-XCD range (halo reuse in L2)
        if ((gridDim.x & 7) == 0) lid = (lid & 7) * (gridDim.x >> 3) + (lid >> 3);
        ct = lid % nct; lid /= nct;
        pt = lid % tiles;
        img = lid / tiles;
    } else if (a.xcd_order == 2) {   // pixel tile fastest inside a contiguous per-XCD range (weight reuse in L2)
        if ((gridDim.x & 7) == 0) lid = (lid & 7) * (gridDim.x >> 3) + (lid >> 3);
        pt = lid % tiles; lid /= tiles;
        ct = lid % nct;
        img = lid / nct;
    } else {                         // plain: pixel tile fastest, round-robin over XCDs
        pt = lid % tiles; lid /= tiles;
        ct = lid % nct;
        img = lid / nct;
    }
    const int tx0 = (pt % a.tiles_x) * 32, ty0 = (pt / a.tiles_x) * 4;
    const int nst = a.cin / KR;

    const rsrc_t src = make_rsrc(a.in + (size_t)img * a.Hs * a.Ws * a.cin, (unsigned)a.Hs * a.Ws * a.cin * 4u);
    const rsrc_t wsr = make_rsrc(a.wpk, (unsigned)a.cin * a.cout * 64u);

    // ---- raw halo staging: HALO x 4 quads over 256 threads ---------------------------------------------------------
    int roff[RITEMS];
#pragma unroll
    for (int k = 0; k < RITEMS; ++k) {
        const int idx = tid + k * 256;
        const int hp = min(idx >> 2, HALO - 1), q = idx & 3;
        const int hy = hp / WHALO_W, hx = hp - hy * WHALO_W;
        int y = reflect1(ty0 + hy - 1, a.H), x = reflect1(tx0 + hx - 1, a.W);
        if (MODE == SRC_UP2X) { y >>= 1; x >>= 1; }
        roff[k] = ((y * a.Ws + x) * a.cin + q * 4) * 4;
    }
    f32x4 rawreg[RITEMS];
    auto raw_load = [&](int stage) {
#pragma unroll
        for (int k = 0; k < RITEMS; ++k) rawreg[k] = buf_load4(src, roff[k], stage * KR * 4);
    };
    auto raw_store = [&](float* buf) {
#pragma unroll
        for (int k = 0; k < RITEMS; ++k) {
            const int idx = tid + k * 256;          // items past the halo land in the buffer's unused tail
            *(f32x4*)(buf + (idx >> 2) * RSTR + (idx & 3) * 4) = rawreg[k];
        }
    };

    // ---- this lane's share of the input transform ------------------------------------------------------------------
    // patch of tile li starts at halo pixel (2 * (li >> 4), 2 * (li & 15)); rows (rA, rB) feed transform row wi:
    //   wi 0: d0 - d2   wi 1: d1 + d2   wi 2: d2 - d1   wi 3: d1 - d3      then across columns: e0-e2, e1+e2, e2-e1, e1-e3
    // The whole main loop is instantiated once per transform row (WI is a compile-time constant inside): with a run-time
    // `wi` the compiler turns the sign selection into scalar branches around every group of four adds, which splits the
    // MFMA steps into many basic blocks and leaves the transform outside the matrix pipe's shadow (tools/wino_probe.py:
    // 1200-1300 cycles for a step with the transform against 630 for one without, ideal 512).
    f32x16 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][n][r] = 0.f;

    auto main_loop = [&](auto WIC) {
    constexpr int WI = decltype(WIC)::value;
    constexpr int rA = WI == 0 ? 0 : 1, rB = WI == 3 ? 3 : 2;
    const int p_base = ((2 * (li >> 4)) * WHALO_W + 2 * (li & 15)) * RSTR + 4 * lh;
    const int pA = p_base + rA * WHALO_W * RSTR, pB = p_base + rB * WHALO_W * RSTR;
    f32x4 dA[4], dB[4];
    auto xf_read = [&](const float* rb) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            dA[c] = *(const f32x4*)(rb + pA + c * RSTR);
            dB[c] = *(const f32x4*)(rb + pB + c * RSTR);
        }
    };
    auto xf_make = [&](f32x4 (&out)[4]) {
        f32x4 e[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if constexpr (WI == 1) e[c] = dA[c] + dB[c];
            else if constexpr (WI == 2) e[c] = dB[c] - dA[c];
            else e[c] = dA[c] - dB[c];
        }
        out[0] = e[0] - e[2];
        out[1] = e[1] + e[2];
        out[2] = e[2] - e[1];
        out[3] = e[1] - e[3];
    };

    // ---- weights ------------------------------------------------------------------------------------------------------
    const int wvo = lane * 16;
    const int nch = a.cin / WKC;
    int wso = ((ct * 4 + wi) * nch) * 8192;
    constexpr int PF = (DIAG == 2 || DIAG == 3) ? 7 : 3, RING = PF + 1;   // the one-block-per-CU diagnostic forms have 512 registers
    f32x4 bq[RING][2];
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int n = 0; n < 2; ++n) bq[p][n] = buf_load4(wsr, wvo, wso + p * 2048 + n * 1024);

    // one chunk of 8 channels: 4 steps of 8 MFMAs on `use`; meanwhile `make` is filled from `nsrc` (next chunk's raw)
    static_assert(RING == 4 || RING == 8, "ring slots follow the step index inside a stage (2 chunks x 4 steps)");
    auto chunk = [&](const f32x4 (&use)[4], f32x4 (&make)[4], const float* nsrc, auto DOXF, auto HALF) {
        constexpr bool do_xf = decltype(DOXF)::value;
        constexpr int half = decltype(HALF)::value;
        auto step = [&](auto JJ) {
            constexpr int j = decltype(JJ)::value;
            constexpr int g = half * 4 + j;
            if constexpr (DIAG >= 3) {
                const unsigned tnow = (unsigned)__builtin_amdgcn_s_memtime();
                if (lane == 0 && nlog < 128) steplog[wi * 128 + nlog] = tnow;
                ++nlog;
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) bq[(g + PF) % RING][n] = buf_load4(wsr, wvo, wso + (j + PF) * 2048 + n * 1024);

            if constexpr (do_xf) {
                if constexpr (j == 0) xf_read(nsrc);
                if constexpr (j == 2) xf_make(make);
            }
            if constexpr (DIAG == 2 || DIAG == 3) {      // experiment: the four MFMAs of one accumulator back to back
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        acc[j][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(use[j][s], bq[g % RING][n][s], acc[j][n], 0, 0, 0);
            } else {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[j][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(use[j][s], bq[g % RING][n][s], acc[j][n], 0, 0, 0);
            }
            if constexpr (PIN) {
                // weight prefetch first, then one MFMA followed by one slice of the transform: a patch read (step 0) or
                // four adds (step 2), so the wave's non-matrix work issues in the shadow of its own MFMAs
#define W2_LOAD __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#define W2_SLOT                                                                                  \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                               \
                if constexpr (do_xf && j == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);         \
                if constexpr (do_xf && j == 2) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                W2_SLOT W2_LOAD W2_SLOT W2_SLOT W2_SLOT W2_SLOT W2_LOAD W2_SLOT W2_SLOT W2_SLOT
#undef W2_SLOT
#undef W2_LOAD
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
        wso += 8192;
    };

    // ---- prologue ---------------------------------------------------------------------------------------------------------
    f32x4 aq0[4], aq1[4];
    raw_load(0);
    raw_store(Rs);
    __syncthreads();
    raw_load(1);                                        // stages past the end read the neighbouring channels / pixels or
    xf_read(Rs);                                        // (out of range) zeros and are never consumed: no branches in the loop
    xf_make(aq0);
    if constexpr (DIAG) stamp[1] = __builtin_amdgcn_s_memrealtime();

    constexpr std::true_type XF{};
    constexpr std::integral_constant<int, 0> H0{};
    constexpr std::integral_constant<int, 1> H1{};
    for (int s = 0; s + 1 < nst; ++s) {
        const float* cur = Rs + (s & 1) * RBUF;
        float* nxt = Rs + ((s + 1) & 1) * RBUF;
        chunk(aq0, aq1, cur + 8, XF, H0);               // channels 0..7 of the stage; prepares channels 8..15
        raw_store(nxt);                                 // the next stage's halo (loaded one stage ago)
        __syncthreads();
        raw_load(s + 2);
        chunk(aq1, aq0, nxt, XF, H1);                   // channels 8..15; prepares the next stage's first chunk
    }
    chunk(aq0, aq1, Rs + ((nst - 1) & 1) * RBUF + 8, XF, H0);      // last stage: nothing left to stage or prepare
    chunk(aq1, aq0, Rs, std::false_type{}, H1);
    if constexpr (DIAG) stamp[2] = __builtin_amdgcn_s_memrealtime();
    };
    if (wi == 0) main_loop(std::integral_constant<int, 0>{});
    else if (wi == 1) main_loop(std::integral_constant<int, 1>{});
    else if (wi == 2) main_loop(std::integral_constant<int, 2>{});
    else main_loop(std::integral_constant<int, 3>{});
    __syncthreads();
    if constexpr (DIAG >= 3) {
        if (a.dbg && blockIdx.x < 32) {
            unsigned* d32 = (unsigned*)(a.dbg + (size_t)gridDim.x * 17) + blockIdx.x * 512;
            for (int i = tid; i < 512; i += 256) d32[i] = (i & 127) < nlog ? steplog[i] : 0u;
        }
        __syncthreads();
    }

    // ---- epilogue: identical to conv3x3_wino_kernel<MODE, 1> ----------------------------------------------------------------
    {
        float* Pw = smem + (wi * 2) * (32 * 64);
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int trow = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float p0 = acc[0][n][r] + acc[1][n][r] + acc[2][n][r];
                const float p1 = acc[1][n][r] - acc[2][n][r] - acc[3][n][r];
                Pw[trow * 64 + n * 32 + li] = p0;
                Pw[32 * 64 + trow * 64 + n * 32 + li] = p1;
            }
    }
    __syncthreads();
    {
        const int co = ct * 64 + lane;
        const float bias = a.bias[co];
        const int Hp = (a.H + 1) >> 1, Wp = (a.W + 1) >> 1;
#pragma unroll 2
        for (int t = 0; t < 8; ++t) {
            const int tl = wi * 8 + t;
            const int oy = ty0 + 2 * (tl >> 4), ox = tx0 + 2 * (tl & 15);
            if (oy >= a.H || ox >= a.W) continue;
            float P[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int b = 0; b < 2; ++b) P[i][b] = smem[((i * 2 + b) * 32 + tl) * 64 + lane];
            float y[2][2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                y[0][b] = P[0][b] + P[1][b] + P[2][b] + bias;
                y[1][b] = P[1][b] - P[2][b] - P[3][b] + bias;
            }
            if (a.relu) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int b = 0; b < 2; ++b) y[i][b] = fmaxf(y[i][b], 0.f);
            }
            const bool row1 = oy + 1 < a.H, col1 = ox + 1 < a.W;
            if (a.pool_out) {
                float v = y[0][0];
                if (col1) v = fmaxf(v, y[0][1]);
                if (row1) {
                    v = fmaxf(v, y[1][0]);
                    if (col1) v = fmaxf(v, y[1][1]);
                }
                a.out[(((size_t)img * Hp + (oy >> 1)) * Wp + (ox >> 1)) * a.cout + co] = v;
            } else {
                float* o = a.out + (((size_t)img * a.H + oy) * a.W + ox) * a.cout + co;
                o[0] = y[0][0];
                if (col1) o[a.cout] = y[0][1];
                if (row1) {
                    o[(size_t)a.W * a.cout] = y[1][0];
                    if (col1) o[(size_t)a.W * a.cout + a.cout] = y[1][1];
                }
            }
        }
    }
    if constexpr (DIAG) {
        stamp[3] = __builtin_amdgcn_s_memrealtime();
        if (lane == 0 && a.dbg) {     // [block][wave][4 stamps], then [block] hardware ids
            unsigned long long* d = a.dbg + ((size_t)blockIdx.x * 4 + wi) * 4;
            d[0] = stamp[0]; d[1] = stamp[1]; d[2] = stamp[2]; d[3] = stamp[3];
            if (wi == 0) a.dbg[(size_t)gridDim.x * 16 + blockIdx.x] =
                ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        }
    }
}

int launch_pack_wino(const float* w, float* p, int cin, int cout, hipStream_t s) {
    if (cin % WKC || cout % 64) { set_error("pack_wino: cin %% 8 or cout %% 64 != 0 (%d, %d)", cin, cout); return -1; }
    const size_t total = (size_t)cin * cout * 16;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_wino_kernel, dim3(blocks), dim3(256), 0, s, w, p, cin, cout);
    return check_launch("pack_wino");
}

int launch_conv3x3_wino(const ConvArgs& a0, int src_mode, int mh, hipStream_t s) {
    ConvArgs a = a0;
    if (a.cin % WKC || a.cin < WKC) { set_error("conv3x3_wino: cin %d not a multiple of 8", a.cin); return -1; }
    if (a.cout % 64) { set_error("conv3x3_wino: cout %d not a multiple of 64", a.cout); return -1; }
    if (a.H < 2 || a.W < 2 || a.n < 1) { set_error("conv3x3_wino: H, W must be >= 2, got %dx%d", a.H, a.W); return -1; }
    if ((size_t)a.Hs * a.Ws * a.cin * 4 >= 0x7fffffffULL || (size_t)a.H * a.W * a.cout * 4 >= 0x7fffffffULL) {
        set_error("conv3x3_wino: per-image source and output tensors must stay below 2 GiB (32-bit buffer offsets)");
        return -1;
    }
    if (src_mode == SRC_DIRECT) {
        if (a.Hs != a.H || a.Ws != a.W) { set_error("conv3x3_wino: direct mode needs Hs==H, Ws==W"); return -1; }
    } else if (src_mode == SRC_UP2X) {
        if (a.H != 2 * a.Hs || a.W != 2 * a.Ws) { set_error("conv3x3_wino: up2x mode needs H==2Hs, W==2Ws"); return -1; }
    } else {
        set_error("conv3x3_wino: unsupported src_mode %d", src_mode);
        return -1;
    }
    if ((mh == 4 || mh == 17) && a.cin % 16 == 0 && a.cin >= 32) {       // 17: form 4 with time stamps (tools/wino_probe.py)
        ConvArgs a4 = a0;
        if (mh == 4) a4.dbg = nullptr;
        return launch_conv3x3_wino3(a4, src_mode, s);
    }
    const int geo = mh == 2 ? 2 : 1;
    a.tiles_x = (a.W + 31) / 32;
    a.tiles_y = (a.H + 4 * geo - 1) / (4 * geo);
    const long long blocks = (long long)a.tiles_x * a.tiles_y * (a.cout / 64) * a.n;
    if (blocks <= 0 || blocks > 0x7fffffffLL) { set_error("conv3x3_wino: bad grid %lld", blocks); return -1; }
    static const int xcd_env = tune_env("ADAIN_XCD_ORDER", 1);
    a.xcd_order = xcd_env;
    const dim3 g((unsigned)blocks);
    const bool up = src_mode == SRC_UP2X;
#ifdef ADAIN_DIAG
    if (mh >= 13 && mh <= 16 && a.cin % 16 == 0 && !up) {      // stamp builds (tools/wino_probe.py), diagnostic library only
        if (mh == 13) hipLaunchKernelGGL((conv3x3_wino2_kernel<SRC_DIRECT, true, 1>), g, dim3(256), 0, s, a);
        else if (mh == 14) hipLaunchKernelGGL((conv3x3_wino2_kernel<SRC_DIRECT, true, 2>), g, dim3(256), 0, s, a);
        else if (mh == 15) hipLaunchKernelGGL((conv3x3_wino2_kernel<SRC_DIRECT, true, 3>), g, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv3x3_wino2_kernel<SRC_DIRECT, true, 4>), g, dim3(256), 0, s, a);
        return check_launch("conv3x3_wino(diag)");
    }
#endif
    if (mh == 3 && a.cin % 16 == 0) {
        if (up) hipLaunchKernelGGL((conv3x3_wino2_kernel<SRC_UP2X>), g, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv3x3_wino2_kernel<SRC_DIRECT>), g, dim3(256), 0, s, a);
    } else if (mh == 2) {
        if (up) hipLaunchKernelGGL((conv3x3_wino_kernel<SRC_UP2X, 2>), g, dim3(512), 0, s, a);
        else hipLaunchKernelGGL((conv3x3_wino_kernel<SRC_DIRECT, 2>), g, dim3(512), 0, s, a);
    } else {
        if (up) hipLaunchKernelGGL((conv3x3_wino_kernel<SRC_UP2X, 1>), g, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv3x3_wino_kernel<SRC_DIRECT, 1>), g, dim3(256), 0, s, a);
    }
    return check_launch("conv3x3_wino");
}

}  // namespace adain
