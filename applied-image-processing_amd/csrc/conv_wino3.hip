// Persistent form of the Winograd F(2x2, 3x3) convolution (conv_wino.hip, second form), gfx950.
//
// What tools/wino_probe.py shows for the one-tile-per-workgroup kernel: with two workgroups per CU the matrix pipe is
// ~93 % busy while BOTH are inside their main loops, but a workgroup spends 1.7-2.6 us before its first MFMA (first
// halo from HBM, LDS round trip, first transform) and 2.9-4.7 us after its last one (LDS exchange of the four transform
// rows, output transform, stores).  On the 256-channel layers that is 22 % of the CU time with only one wave per SIMD
// feeding the pipe; on the 64-channel layers (main loop 12 us) it is 70 %.
//
// This kernel keeps the mapping of conv3x3_wino2_kernel (block tile 4 x 32 output pixels x 64 channels, wave i = transform
// row i, A operand transformed in registers from the raw halo image in LDS, weights streamed as packed fragments, same
// packed-weight layout) and changes what surrounds the main loop:
//   * a workgroup walks a list of tiles (grid = 2 workgroups per CU; XCD x owns a contiguous range of the tile list,
//     channel tile fastest, so the 64 workgroups of an XCD share halos and weights in its L2);
//   * the next tile's first halo stage is loaded into the (idle) staging registers during the current tile's last
//     stages and its first three weight fragments replace the ring's run-off loads, so a new tile starts with one LDS
//     write + barrier + transform instead of a cold HBM round trip;
//   * MFMA operands are swapped (A = weights, B = tiles): a lane then holds 4 consecutive output channels of one tile, the
//     row exchange uses b128 LDS accesses on an XOR-swizzled image (16 + 16 per lane instead of 64 + 64 b32) and the
//     outputs leave as b128 buffer stores (4 or 1 per lane and tile instead of 16 or 4 b32);
//   * halo stores / loads of the staging pipeline are pinned into MFMA gaps like the weight loads.
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "device_utils.h"

namespace adain {

namespace {
constexpr int W3_KR = 16;                    // channels per raw stage = 2 MFMA chunks of 8
constexpr int W3_RSTR = W3_KR + 4;           // floats per halo pixel (80 B: conflict-free b128 patch reads)
constexpr int W3_HALO_W = 34;
constexpr int W3_HALO = 6 * W3_HALO_W;       // 6 x 34 halo of a 4 x 32 pixel tile
constexpr int W3_RBUF = 256 * W3_RSTR;       // rounded up to the 4 x 256 staging items (no predicated stores)
constexpr int W3_PEX = 4 * 2 * 32 * 64;      // row exchange: [row 4][p 2][tile 32][64 channels] floats = 64 KiB
static_assert(2 * W3_RBUF <= W3_PEX, "LDS layout");

// 128-bit buffer store.  The scalar offset operand is deliberately NOT exposed (always the literal 0): with an SGPR soffset
// LLVM models no hazard between a >64-bit MUBUF store and a following VALU write of its data registers and emits none of
// the wait states it emits for the literal form - and on gfx950, with two waves per SIMD queueing VMEM work, the store
// then reads data the next instruction has already overwritten (seen here: the register allocator reused the first data
// register of one store as the address of the next; the .x lane values arrived as address bit patterns).  With the literal
// form the compiler inserts its `s_nop 1` and the stores are exact.
__device__ __forceinline__ void buf_store4(rsrc_t r, f32x4 v, int voff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, 0);
}
}  // namespace

// DIAG: wave 0 of every workgroup stamps s_memrealtime at main-loop start / end and epilogue end of its first 16 tiles
// (tools/wino3_probe.py); not instantiated for the product path
template <int MODE, bool DIAG = false>
__global__ __launch_bounds__(256, 2) void conv3x3_wino3_kernel(ConvArgs a, int items, int prio_mode) {
    __shared__ __attribute__((aligned(16))) float smem[W3_PEX];
    float* const Rs = smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wi = __builtin_amdgcn_readfirstlane(tid >> 6);      // transform row of this wave
    // `wvo` (lane * 16, the weight fragments' per-lane offset) is the ONE lane constant kept in a register through the main
    // loop; every other lane-derived address (halo pixel decomposition, LDS store / patch addresses) is rebuilt from an
    // opaque copy of it where it is used (a few VALU ops per stage) - the kernel sits at the 256-register limit and the
    // compiler would otherwise hoist them all out of the tile loop and spill them.
    const int wvo = lane * 16;
    auto lane_now = [&]() {
        int w = wvo;
        asm volatile("" : "+v"(w));
        return w >> 4;
    };

    const int tiles = a.tiles_x * a.tiles_y;
    const int nct = a.cout / 64;
    const int nst = a.cin / W3_KR;
    const int nch = a.cin / 8;

    // ---- tile list of this workgroup: XCD (blockIdx & 7) owns items [lo, hi), its workgroups interleave inside it ----------
    const int xcd = blockIdx.x & 7, stride = gridDim.x >> 3;
    const int lo = (int)((long long)items * xcd / 8), hi = (int)((long long)items * (xcd + 1) / 8);
    int item = lo + (blockIdx.x >> 3);
    if (item >= hi) return;

    int ct, img, tx0, ty0;
    auto decode = [&](int it, int& c, int& im, int& x0, int& y0) {
        c = it % nct;
        const int r = it / nct;
        const int pt = r % tiles;
        im = r / tiles;
        x0 = (pt % a.tiles_x) * 32;
        y0 = (pt / a.tiles_x) * 4;
    };
    decode(item, ct, img, tx0, ty0);

    const unsigned src_bytes = (unsigned)a.Hs * a.Ws * a.cin * 4u;
    const size_t src_img = (size_t)a.Hs * a.Ws * a.cin;
    rsrc_t src = make_rsrc(a.in + img * src_img, src_bytes);
    const rsrc_t wsr = make_rsrc(a.wpk, (unsigned)a.cin * a.cout * 64u);

    // ---- raw halo staging: 204 pixels x 4 quads over 256 threads x 4 items -------------------------------------------------
    int roff[4];
    auto halo_offsets = [&](int x0, int y0) {
        const int t = lane_now() | (wi << 6);      // = threadIdx.x
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = t + k * 256;
            const int hp = min(idx >> 2, W3_HALO - 1), q = idx & 3;
            const int hy = hp / W3_HALO_W, hx = hp - hy * W3_HALO_W;
            int y = reflect1(y0 + hy - 1, a.H), x = reflect1(x0 + hx - 1, a.W);
            if (MODE == SRC_UP2X) { y >>= 1; x >>= 1; }
            roff[k] = ((y * a.Ws + x) * a.cin + q * 4) * 4;
        }
    };
    halo_offsets(tx0, ty0);
    f32x4 rawreg[4];
    auto raw_load = [&](int soff) {
#pragma unroll
        for (int k = 0; k < 4; ++k) rawreg[k] = buf_load4(src, roff[k], soff);
    };
    auto raw_store = [&](float* buf) {
        const int t = lane_now() | (wi << 6);
        const int st_base = (t >> 2) * W3_RSTR + (t & 3) * 4;      // item k: pixel (t >> 2) + 64 k -> one address + immediates
#pragma unroll
        for (int k = 0; k < 4; ++k) *(f32x4*)(buf + st_base + k * 64 * W3_RSTR) = rawreg[k];   // items past the halo: unused tail
    };

    // ---- weights: ring of 4 fragment pairs, 3 steps ahead; `wso` = byte offset of the current chunk ---------------------------
    constexpr int PF = 3, RING = 4;
    f32x4 bq[RING][2];
    int wso = ((ct * 4 + wi) * nch) * 8192;
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int n = 0; n < 2; ++n) bq[p][n] = buf_load4(wsr, wvo, wso + p * 2048 + n * 1024);

    f32x16 acc[4][2];

    // The whole tile loop is instantiated once per transform row (WI compile-time): see conv_wino.hip.
    auto run = [&](auto WIC) {
        constexpr int WI = decltype(WIC)::value;
        constexpr int rA = WI == 0 ? 0 : 1, rB = WI == 3 ? 3 : 2;
        asm volatile("; transform row %0" ::"n"(WI));      // differs per instance: keeps the four copies' code from being merged
        // the patch is read and row-combined in two halves (columns 0-1, then 2-3) so that at most 24 of its registers are
        // live at once (the kernel sits at the 256-register limit of two workgroups per CU)
        f32x4 dA[2], dB[2], e[4];
        auto xf_read = [&](const float* rb, int c0) {
            const int l = lane_now(), li = l & 31, lh = l >> 5;
            const int p_base = ((2 * (li >> 4)) * W3_HALO_W + 2 * (li & 15)) * W3_RSTR + 4 * lh;
            const int pA = p_base + rA * W3_HALO_W * W3_RSTR, pB = p_base + rB * W3_HALO_W * W3_RSTR;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                dA[c] = *(const f32x4*)(rb + pA + (c0 + c) * W3_RSTR);
                dB[c] = *(const f32x4*)(rb + pB + (c0 + c) * W3_RSTR);
            }
        };
        auto xf_rows = [&](int c0) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if constexpr (WI == 1) e[c0 + c] = dA[c] + dB[c];
                else if constexpr (WI == 2) e[c0 + c] = dB[c] - dA[c];
                else e[c0 + c] = dA[c] - dB[c];
            }
        };
        auto xf_cols = [&](f32x4 (&out)[4]) {
            out[0] = e[0] - e[2];
            out[1] = e[1] + e[2];
            out[2] = e[2] - e[1];
            out[3] = e[1] - e[3];
        };
        // The A fragments live in ONE array: the fragment of column j is dead once step j's MFMAs have issued, so the next
        // chunk's column j is written over it as soon as step j is behind (columns 0, 1 during step 2, column 2 during step 3,
        // column 3 after step 3) instead of into a second 16-register set.

        int wso_next = wso;
        int ndiag = 0, ntile = 0;
        int raw_soff = 0;
        float* store_to = Rs;

        // one chunk of 8 channels = 4 steps of 8 MFMAs.  XF: transform the next chunk's patches meanwhile; HALF: position in
        // the stage (ring slots); ST: step 3 also writes the staged halo registers to LDS; LD: step 0 also issues the halo
        // loads two stages ahead; TAIL: last chunk of a tile, the ring's look-ahead continues in the next tile's weights
        auto chunk = [&](f32x4 (&aq)[4], f32x4 (&mk)[4], const float* nsrc, auto XFC, auto HALFC, auto STC, auto LDC, auto TAILC) {
            constexpr bool do_xf = decltype(XFC)::value, st = decltype(STC)::value, ld = decltype(LDC)::value,
                           tail = decltype(TAILC)::value;
            constexpr int half = decltype(HALFC)::value;
            auto step = [&](auto JJ) {
                constexpr int j = decltype(JJ)::value;
                constexpr int g = half * 4 + j;
                {
                    const int base = (tail && j + PF >= 4) ? wso_next + (j + PF - 4) * 2048 : wso + (j + PF) * 2048;
#pragma unroll
                    for (int n = 0; n < 2; ++n) bq[(g + PF) % RING][n] = buf_load4(wsr, wvo, base + n * 1024);
                }
                if constexpr (ld && j == 0) raw_load(raw_soff);
                if constexpr (st && j == 3) raw_store(store_to);
                if constexpr (do_xf) {
                    if constexpr (j == 0) xf_read(nsrc, 0);
                    if constexpr (j == 1) { xf_rows(0); xf_read(nsrc, 2); }
                    if constexpr (j == 2) {
                        xf_rows(2);
                        mk[0] = e[0] - e[2];
                        mk[1] = e[1] + e[2];
                    }
                    if constexpr (j == 3) {
                        mk[2] = e[2] - e[1];
                        e[3] = e[1] - e[3];
                    }
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[j][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[g % RING][n][s], aq[j][s], acc[j][n], 0, 0, 0);
                if constexpr (do_xf && j == 3) mk[3] = e[3];
                // issue order: one MFMA, then one slice of the other work of this step in its shadow
#define W3_MFMA __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#define W3_WLOAD __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#define W3_EXTRA                                                                                          \
                if constexpr (do_xf && j == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);         \
                if constexpr (do_xf && j == 1) {                                                          \
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                    \
                }                                                                                         \
                if constexpr (do_xf && j == 2) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);         \
                if constexpr (do_xf && j == 3) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
#define W3_HALO_LD if constexpr (ld && j == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#define W3_HALO_ST                                                                                        \
                if constexpr (st && j == 3) {                                                             \
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);                                    \
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                    \
                }
                W3_MFMA W3_EXTRA W3_WLOAD
                W3_MFMA W3_EXTRA W3_HALO_LD W3_HALO_ST
                W3_MFMA W3_EXTRA W3_HALO_LD W3_HALO_ST
                W3_MFMA W3_EXTRA
                W3_MFMA W3_EXTRA W3_WLOAD
                W3_MFMA W3_EXTRA W3_HALO_LD W3_HALO_ST
                W3_MFMA W3_EXTRA W3_HALO_LD W3_HALO_ST
                W3_MFMA W3_EXTRA
#undef W3_MFMA
#undef W3_WLOAD
#undef W3_EXTRA
#undef W3_HALO_LD
#undef W3_HALO_ST
                __builtin_amdgcn_sched_barrier(0);
            };
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{});
            wso += 8192;
        };
        constexpr std::true_type T{};
        constexpr std::false_type F{};
        constexpr std::integral_constant<int, 0> H0{};
        constexpr std::integral_constant<int, 1> H1{};

        // ---- first tile: cold start ----------------------------------------------------------------------------------------------
        f32x4 aq[4];
        raw_load(0);
        raw_store(Rs);
        __syncthreads();
        raw_load(W3_KR * 4);
        xf_read(Rs, 0); xf_rows(0);
        xf_read(Rs, 2); xf_rows(2);
        xf_cols(aq);

        for (;;) {
            // the tile after this one (or this one again when the list is exhausted: its loads are then never consumed)
            const int nitem = item + stride < hi ? item + stride : item;
            int nct_, nimg, ntx0, nty0;
            decode(nitem, nct_, nimg, ntx0, nty0);
            wso_next = ((nct_ * 4 + wi) * nch) * 8192;

#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][n][r] = 0.f;
            unsigned long long st0 = 0, st1 = 0, sc0 = 0, sc1 = 0;
            if constexpr (DIAG) { st0 = __builtin_amdgcn_s_memrealtime(); sc0 = __builtin_amdgcn_s_memtime(); }
            // The SIMD arbiter prefers the older wave: the workgroup dispatched first runs its tiles ~1.4x faster than its CU
            // partner (tools/wino3_probe.py: 53 vs 77 us per 256-channel tile) and then sits idle at the end of the launch
            // (static tile lists).  Alternating the wave priority per tile, in opposite phase for the two halves of the
            // grid (workgroups b and b + grid/2 share a CU), evens the two out: both in their main loops 71 % -> 84 % of the
            // time.  (A start offset of half a tile period for the second half was also tried: no gain.)
            if (prio_mode) {
                if ((ntile + (int)((blockIdx.x >> 3) >= (stride >> 1))) & 1) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
            ++ntile;

            // The halo loads run two stages ahead; once the current tile's last stage has been requested (stage nst-1, issued in
            // iteration nst-3, or by the tile's entry code when nst == 2) the offsets switch to the next tile, whose first stage
            // is then what iteration nst-2 requests.  The switch sits at the loop's tail: a branch BETWEEN the two chunks of
            // a stage would split the block and let the compiler sink the transform out of the MFMA shadow.
            auto next_halo = [&]() {
                halo_offsets(ntx0, nty0);
                src = make_rsrc(a.in + nimg * src_img, src_bytes);
            };
            if (nst == 2) next_halo();
            for (int s = 0; s + 1 < nst; ++s) {
                const float* cur = Rs + (s & 1) * W3_RBUF;
                float* nxt = Rs + ((s + 1) & 1) * W3_RBUF;
                store_to = nxt;
                raw_soff = s == nst - 2 ? 0 : (s + 2) * W3_KR * 4;
                chunk(aq, aq, cur + 8, T, H0, T, F, F);          // channels 0..7; prepares 8..15; writes the next stage's halo
                __syncthreads();
                chunk(aq, aq, nxt, T, H1, F, T, F);              // channels 8..15; prepares the next stage; loads two stages ahead
                if (s == nst - 3) next_halo();
            }
            chunk(aq, aq, Rs + ((nst - 1) & 1) * W3_RBUF + 8, T, H0, F, F, F);
            chunk(aq, aq, Rs, F, H1, F, F, T);
            wso = wso_next;
            if constexpr (DIAG) { st1 = __builtin_amdgcn_s_memrealtime(); sc1 = __builtin_amdgcn_s_memtime(); }
            __syncthreads();

            // ---- epilogue -------------------------------------------------------------------------------------------------------------
            // lane (li = tile, lh): acc[j][n][r] = M[row WI][col j][channel 32 n + 8 (r >> 2) + 4 lh + (r & 3)][tile li]
            // lane-derived addresses are rebuilt from an opaque copy of the lane id: hoisted out of the tile loop they would
            // stay live (and spill) across the main loop
            int le = lane_now();
            asm volatile("" : "+v"(le));
            const int q16 = le & 15, tt = le >> 4, li_e = le & 31, lh_e = le >> 5;
            const f32x4 bias4 = *(const f32x4*)(a.bias + ct * 64 + 4 * q16);
            {
                float* Pw = smem + (WI * 2) * (32 * 64) + li_e * 64;
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        f32x4 p0, p1;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int r = rq * 4 + e;
                            p0[e] = acc[0][n][r] + acc[1][n][r] + acc[2][n][r];
                            p1[e] = acc[1][n][r] - acc[2][n][r] - acc[3][n][r];
                        }
                        const int q = n * 8 + 2 * rq + lh_e;                   // channel quad of the 64
                        const int slot = ((q ^ (li_e & 15)) << 2);             // XOR swizzle: conflict-free both ways
                        *(f32x4*)(Pw + slot) = p0;
                        *(f32x4*)(Pw + 32 * 64 + slot) = p1;
                    }
            }
            __syncthreads();
            f32x4 P[2][4][2];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int tl = wi * 8 + it * 4 + tt;
                const float* Pr = smem + tl * 64 + ((q16 ^ (tl & 15)) << 2);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int b = 0; b < 2; ++b) P[it][i][b] = *(const f32x4*)(Pr + (i * 2 + b) * (32 * 64));
            }
            __syncthreads();
            raw_store(Rs);                                   // the next tile's first halo stage (loaded during this tile)
            {
                const rsrc_t dst = make_rsrc(a.out + (size_t)img * (a.pool_out ? ((a.H + 1) >> 1) * ((a.W + 1) >> 1) : a.H * a.W) * a.cout,
                                             (unsigned)((a.pool_out ? ((a.H + 1) >> 1) * ((a.W + 1) >> 1) : a.H * a.W) * a.cout) * 4u);
                const int Wp = (a.W + 1) >> 1;
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int tl = wi * 8 + it * 4 + tt;
                    const int oy = ty0 + 2 * (tl >> 4), ox = tx0 + 2 * (tl & 15);
                    f32x4 y[2][2];
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        y[0][b] = P[it][0][b] + P[it][1][b] + P[it][2][b] + bias4;
                        y[1][b] = P[it][1][b] - P[it][2][b] - P[it][3][b] + bias4;
                    }
                    if (a.relu) {
                        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int b = 0; b < 2; ++b) y[i][b] = max4(y[i][b], z);
                    }
                    const bool in = oy < a.H && ox < a.W;
                    const bool row1 = oy + 1 < a.H, col1 = ox + 1 < a.W;
                    const int cbyte = (ct * 64 + 4 * q16) * 4;
                    if (a.pool_out) {
                        f32x4 v = y[0][0];
                        if (col1) v = max4(v, y[0][1]);
                        if (row1) {
                            v = max4(v, y[1][0]);
                            if (col1) v = max4(v, y[1][1]);
                        }
                        const int off = (((oy >> 1) * Wp + (ox >> 1)) * a.cout) * 4 + cbyte;
                        buf_store4(dst, v, in ? off : 0x7fffffff);
                    } else {
                        const int off = ((oy * a.W + ox) * a.cout) * 4 + cbyte;
                        const int rowb = a.W * a.cout * 4, colb = a.cout * 4;
                        buf_store4(dst, y[0][0], in ? off : 0x7fffffff);
                        buf_store4(dst, y[0][1], in && col1 ? off + colb : 0x7fffffff);
                        buf_store4(dst, y[1][0], in && row1 ? off + rowb : 0x7fffffff);
                        buf_store4(dst, y[1][1], in && row1 && col1 ? off + rowb + colb : 0x7fffffff);
                    }
                }
            }
            if constexpr (DIAG) {
                if (WI == 0 && a.dbg && ndiag < 16) {
                    if (lane_now() == 0) {
                        unsigned long long* d = a.dbg + ((size_t)blockIdx.x * 16 + ndiag) * 3;
                        d[0] = st0; d[1] = st1; d[2] = __builtin_amdgcn_s_memrealtime();
                        if (ndiag == 0) a.dbg[(size_t)gridDim.x * 48 + blockIdx.x] =
                            ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
                        if (ndiag == 1) a.dbg[(size_t)gridDim.x * 49 + blockIdx.x] = ((sc1 - sc0) << 32) | ((st1 - st0) & 0xffffffffu);
                    }
                }
                ++ndiag;
            }
            if (item + stride >= hi) break;
            item += stride;
            ct = nct_; img = nimg; tx0 = ntx0; ty0 = nty0;
            __syncthreads();
            raw_load(W3_KR * 4);
            xf_read(Rs, 0); xf_rows(0);
            xf_read(Rs, 2); xf_rows(2);
            xf_cols(aq);
        }
    };
    if (wi == 0) run(std::integral_constant<int, 0>{});
    else if (wi == 1) run(std::integral_constant<int, 1>{});
    else if (wi == 2) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 3>{});
}

int launch_conv3x3_wino3(const ConvArgs& a0, int src_mode, hipStream_t s) {
    ConvArgs a = a0;
    if (a.cin % W3_KR || a.cin < 2 * W3_KR) { set_error("conv3x3_wino3: cin %d must be a multiple of 16, >= 32", a.cin); return -1; }
    if (a.cout % 64) { set_error("conv3x3_wino3: cout %d not a multiple of 64", a.cout); return -1; }
    if (a.H < 2 || a.W < 2 || a.n < 1) { set_error("conv3x3_wino3: H, W must be >= 2, got %dx%d", a.H, a.W); return -1; }
    if ((size_t)a.Hs * a.Ws * a.cin * 4 >= 0x7fffffffULL || (size_t)a.H * a.W * a.cout * 4 >= 0x7fffffffULL) {
        set_error("conv3x3_wino3: per-image tensors must stay below 2 GiB");
        return -1;
    }
    if (src_mode == SRC_DIRECT) {
        if (a.Hs != a.H || a.Ws != a.W) { set_error("conv3x3_wino3: direct mode needs Hs==H, Ws==W"); return -1; }
    } else if (src_mode == SRC_UP2X) {
        if (a.H != 2 * a.Hs || a.W != 2 * a.Ws) { set_error("conv3x3_wino3: up2x mode needs H==2Hs, W==2Ws"); return -1; }
    } else {
        set_error("conv3x3_wino3: unsupported src_mode %d", src_mode);
        return -1;
    }
    a.tiles_x = (a.W + 31) / 32;
    a.tiles_y = (a.H + 3) / 4;
    const long long items = (long long)a.tiles_x * a.tiles_y * (a.cout / 64) * a.n;
    if (items <= 0 || items > 0x7fffffffLL) { set_error("conv3x3_wino3: bad tile count %lld", items); return -1; }
    const int cus = device_cu_count();
    if (cus <= 0) { set_error("conv3x3_wino3: device query failed"); return -1; }
    long long grid = 2LL * cus;                          // two workgroups per CU (launch bounds), a multiple of the 8 XCDs
    static const int grid_env = tune_env("ADAIN_W3_GRID", 0);       // debugging aid
    if (grid_env > 0) grid = grid_env;
    grid -= grid % 8;
    if (grid < 8) grid = 8;
    const long long need = ((items + 7) / 8) * 8;        // never more workgroups than tiles (rounded up to the XCD count)
    if (grid > need) grid = need;
    const bool up = src_mode == SRC_UP2X;
    static const int prio_env = tune_env("ADAIN_W3_PRIO", 1);
    const int prio = prio_env && grid == 2LL * cus;      // alternate wave priority only when every CU really holds two workgroups
#ifdef ADAIN_DIAG
    if (a.dbg && !up) {
        hipLaunchKernelGGL((conv3x3_wino3_kernel<SRC_DIRECT, true>), dim3((unsigned)grid), dim3(256), 0, s, a, (int)items, prio);
        return check_launch("conv3x3_wino3(diag)");
    }
#endif
    if (up) hipLaunchKernelGGL((conv3x3_wino3_kernel<SRC_UP2X>), dim3((unsigned)grid), dim3(256), 0, s, a, (int)items, prio);
    else hipLaunchKernelGGL((conv3x3_wino3_kernel<SRC_DIRECT>), dim3((unsigned)grid), dim3(256), 0, s, a, (int)items, prio);
    return check_launch("conv3x3_wino3");
}

}  // namespace adain
