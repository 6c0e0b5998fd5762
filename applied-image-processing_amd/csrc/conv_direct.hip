// gfx950 (CDNA4) direct implicit-GEMM 3x3 convolution kernels for the AdaIN encoder/decoder.
// DIAGNOSTIC LIBRARY ONLY (libadain_hip_diag.so, build.py --diag): the product schedules run the F(4,3) x F(2,3) kernels of
// conv_wino4.hip; this family (round 1's 0.9-of-peak direct form) stays built and tested there as the A/B baseline.
//
// Replaces the 29 torch.nn.Conv2d calls of the reference hot path (Style_3DGS/AdaIN/net.py:6-36
// decoder, :38-69 encoder up to relu4_1) together with the ReflectionPad2d(1) in front of every
// 3x3 conv, the ReLU behind it, the ceil-mode MaxPool2d(2,2) (net.py:46,53,66) and the nearest 2x
// Upsample (net.py:10,23,30): all of those are folded into the convolution's LDS staging or
// epilogue, so none of them round-trips HBM.
//
// Data layout in HBM: activations NHWC fp32; images NCHW fp32 at the two ends (reference layout).
//
// Main kernel = implicit GEMM on the exact-fp32 matrix pipe (v_mfma_f32_32x32x2_f32):
//   M = 32 consecutive output pixels of one row, N = 32 output channels, K = 9 taps x Cin.
//   * A operand: the (TH+2) x 34 reflect-padded input halo of a TH x 32 pixel tile is staged ONCE per
//     16-channel chunk into LDS and re-used by all 9 taps (a tap is just an LDS address offset, so
//     every ds_read_b128 uses one base VGPR + an immediate).  Pixel stride in LDS is 20 floats (80 B):
//     any 16 pixels distinct mod 16 then cover all 64 banks -> conflict-free b128 reads and writes.
//   * B operand: weights are pre-packed on the device into the exact per-lane fragment order
//     ([cout/32][cin/16][tap][k-group][lane][4]), so a wave fetches each fragment with ONE fully
//     coalesced 1 KiB global_load_dwordx4 straight into VGPRs (no LDS, no barrier for B).
//   * K order inside a group of 8 channels is permuted (lane half h supplies channels 4h..4h+3 of
//     the group, k-step s uses element s), identically for A and B, so each b128 feeds 4 MFMAs.
//   fp32 MFMA runs at the fp32 vector rate (64 cycles per 32x32x2), so HBM, L2 and LDS traffic are
//   far below their limits; the kernel is bound by MFMA issue.
#include <stdlib.h>
#include <utility>

#include "common.h"
#include "device_utils.h"

namespace adain {

constexpr int KC = 16;           // input channels per LDS chunk
constexpr int LSTR = KC + 4;     // LDS pixel stride in floats (80 B)
constexpr int TW = 32;           // tile width == MFMA M
constexpr int HW_ = TW + 2;      // halo width

// ---------------------------------------------------------------------------------------------
// Weight packing (runs once per weight set, on the device).
// ---------------------------------------------------------------------------------------------
// OIHW [cout][cin][3][3]  ->  [cout/32][cin/16][9][2][64 lanes][4]
//   lane = j + 32 h ; element s : W[cout = 32 T + j][cin = 16 c + 8 g + 4 h + s][tap]
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, float* __restrict__ p, int cin, int cout) {
    const size_t total = (size_t)cin * cout * 9;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int s = r & 3; r >>= 2;
        const int lane = r & 63; r >>= 6;
        const int g = r & 1; r >>= 1;
        const int tap = r % 9; r /= 9;
        const int nch = cin / KC;
        const int c = r % nch; r /= nch;
        const int T = (int)r;
        const int j = lane & 31, h = lane >> 5;
        const int co = T * 32 + j, ci = c * KC + g * 8 + h * 4 + s;
        p[i] = w[((size_t)co * cin + ci) * 9 + tap];
    }
}

// ---------------------------------------------------------------------------------------------
// Halo staging shared by the MFMA kernel and the last-layer kernel.
// One item = 4 consecutive channels (16 B) of one halo pixel; a thread owns NITEM items.
// ---------------------------------------------------------------------------------------------
template <int MODE, int TH, int NTHR>
struct HaloStager {
    static constexpr int HH = TH + 2;
    static constexpr int HALO = HH * HW_;
    static constexpr int NITEM = (HALO * 4 + NTHR - 1) / NTHR;
    static constexpr int NSRC = MODE == SRC_POOL2 ? 4 : 1;

    int soff[NITEM];              // BYTE offset of the (first) source pixel inside the image, + 16*quad
    int dxo[NITEM], dyo[NITEM];   // POOL2 only: byte offsets to the right / lower window element (0 if clipped)
    f32x4 raw[NITEM][NSRC];

    __device__ __forceinline__ void init(int tid, int ty0, int tx0, int H, int W, int Hs, int Ws, int cin) {
#pragma unroll
        for (int k = 0; k < NITEM; ++k) {
            const int idx = tid + k * NTHR;
            const int hp = min(idx >> 2, HALO - 1), q = idx & 3;   // surplus items re-read the last pixel
            const int hy = hp / HW_, hx = hp - hy * HW_;
            int y = reflect1(ty0 + hy - 1, H), x = reflect1(tx0 + hx - 1, W);
            if (MODE == SRC_UP2X) { y >>= 1; x >>= 1; }
            if (MODE == SRC_POOL2) {
                dxo[k] = (2 * x + 1 < Ws) ? cin * 4 : 0;
                dyo[k] = (2 * y + 1 < Hs) ? Ws * cin * 4 : 0;
                y *= 2; x *= 2;
            } else {
                dxo[k] = 0; dyo[k] = 0;
            }
            soff[k] = ((y * Ws + x) * cin + q * 4) * 4;
        }
    }
    // issue the global loads of one 16-channel chunk (coff_bytes = chunk * 64) into registers
    __device__ __forceinline__ void load(rsrc_t src, int coff_bytes) {
#pragma unroll
        for (int k = 0; k < NITEM; ++k) {
            raw[k][0] = buf_load4(src, soff[k], coff_bytes);
            if (MODE == SRC_POOL2) {
                raw[k][1] = buf_load4(src, soff[k] + dxo[k], coff_bytes);
                raw[k][2] = buf_load4(src, soff[k] + dyo[k], coff_bytes);
                raw[k][3] = buf_load4(src, soff[k] + dyo[k] + dxo[k], coff_bytes);
            }
        }
    }
    __device__ __forceinline__ void store(float* buf, int tid) {
#pragma unroll
        for (int k = 0; k < NITEM; ++k) {
            const int idx = tid + k * NTHR;
            f32x4 v = raw[k][0];
            if (MODE == SRC_POOL2) v = max4(max4(v, raw[k][1]), max4(raw[k][2], raw[k][3]));
            if (NITEM * NTHR == HALO * 4 || idx < HALO * 4) *(f32x4*)(buf + (idx >> 2) * LSTR + (idx & 3) * 4) = v;
        }
    }
};

// ---------------------------------------------------------------------------------------------
// 3x3 convolution, implicit GEMM on v_mfma_f32_32x32x2_f32.
//   block  = WM x WN waves; wave tile = MT rows x 32 px  by  NT x 32 channels
//   block tile = (WM*MT) rows x 32 px  by  WN*NT*32 channels
// Software pipeline per wave: weight fragments are prefetched 2 k-groups ahead (3-slot register
// ring, the stream is linear across chunks), A fragments 1 k-group ahead from LDS; the next
// chunk's halo travels HBM -> registers during the whole chunk and is written to the other LDS
// buffer just before the single barrier per chunk.
// ---------------------------------------------------------------------------------------------
template <int MODE, int WM, int WN, int MT, int NT, int LDSPAD, int PF = 2, int EXPER = 0>
__global__ __launch_bounds__(WM * WN * 64) void conv3x3_mfma_kernel(ConvArgs a) {
    constexpr int TH = WM * MT;
    constexpr int NTHR = WM * WN * 64;
    using Stager = HaloStager<MODE, TH, NTHR>;
    constexpr int BUF = Stager::HALO * LSTR;
    constexpr int NSTEP = 9 * (KC / 8);   // k-groups per chunk (18)
    constexpr int RING = PF + 1;          // weight-fragment register ring: PF k-groups in flight
    static_assert(NSTEP % RING == 0, "ring slot bookkeeping assumes NSTEP % RING == 0");
    // LDSPAD floats of unused LDS cap the number of co-resident blocks per CU (occupancy tuning knob)
    __shared__ __attribute__((aligned(16))) float smem[2 * BUF + LDSPAD];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const int tiles = a.tiles_x * a.tiles_y;
    const int nct = a.cout / (WN * NT * 32);
    int pt, ct, img;
    if (a.xcd_order) {
        // XCD-aware order (speed only): workgroups are dealt round-robin to the 8 XCDs, so block b and b + 8 share an
        // L2.  Give each XCD a contiguous range of the (pixel tile, channel tile) list with the channel tile fastest:
        // the channel tiles of one pixel tile then run back to back on ONE XCD and the halo is fetched from HBM once
        // and re-read from that L2, instead of once per channel tile.
        const int nb = gridDim.x;
        int lid = blockIdx.x;
        if ((nb & 7) == 0) lid = (lid & 7) * (nb >> 3) + (lid >> 3);
        ct = lid % nct; lid /= nct;
        pt = lid % tiles;
        img = lid / tiles;
    } else {
        int bid = blockIdx.x;
        pt = bid % tiles; bid /= tiles;
        ct = bid % nct;
        img = bid / nct;
    }
    const int tx0 = (pt % a.tiles_x) * TW, ty0 = (pt / a.tiles_x) * TH;
    const int nchunks = a.cin / KC;

    const rsrc_t src = make_rsrc(a.in + (size_t)img * a.Hs * a.Ws * a.cin, (unsigned)a.Hs * a.Ws * a.cin * 4u);
    const rsrc_t wsr = make_rsrc(a.wpk, (unsigned)a.cin * a.cout * 36u);
    unsigned long long t0c = 0, t0r = 0, tentry = 0;
    if constexpr (EXPER == 2) tentry = __builtin_amdgcn_s_memrealtime();   // diagnostic build only

    Stager st;
    st.init(tid, ty0, tx0, a.H, a.W, a.Hs, a.Ws, a.cin);

    // per-wave weight stream: one 1 KiB record per (chunk, tap, k-group), consumed in order;
    // the NT channel tiles of a wave are consecutive streams.
    const int tile_bytes = nchunks * NSTEP * 1024;
    int wso = ((ct * WN + wn) * NT) * tile_bytes;   // scalar byte offset of the current record (tile 0)
    const int wvo = lane * 16;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const int a_base = ((wm * MT) * HW_ + li) * LSTR + lh * 4;

    f32x4 bq[RING][NT];
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int n = 0; n < NT; ++n) bq[p][n] = buf_load4(wsr, wvo, wso + n * tile_bytes + p * 1024);

    st.load(src, 0);
    st.store(smem, tid);
    __syncthreads();
    if constexpr (EXPER == 2) {   // shader clock = d(memtime) / d(memrealtime) * 100 MHz
        t0c = __builtin_amdgcn_s_memtime();
        t0r = __builtin_amdgcn_s_memrealtime();
    }

    for (int c = 0; c < nchunks; ++c) {
        const float* sbuf = smem + (c & 1) * BUF + a_base;
        const bool more = c + 1 < nchunks;
        if (more) st.load(src, (c + 1) * KC * 4);
        f32x4 aq[2][MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) aq[0][m] = *(const f32x4*)(sbuf + (m * HW_) * LSTR);
#pragma unroll
        for (int k = 0; k < NSTEP; ++k) {
            if (k + 1 < NSTEP) {
                const int t = (k + 1) >> 1, g = (k + 1) & 1;
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    aq[(k + 1) & 1][m] = *(const f32x4*)(sbuf + ((m + t / 3) * HW_ + (t % 3)) * LSTR + g * 8);
            }
            // record k+2 of the stream (runs harmlessly past the tile / buffer end on the last steps:
            // the descriptor's range check returns 0 there)
#pragma unroll
            for (int n = 0; n < NT; ++n) bq[(k + PF) % RING][n] = buf_load4(wsr, wvo, wso + n * tile_bytes + (k + PF) * 1024);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[k & 1][m][s], bq[k % RING][n][s], acc[m][n], 0, 0, 0);
            // keep the prefetch distance: without this fence hipcc sinks the loads of step k+2 down to
            // their first use and waits vmcnt(0) in front of every MFMA group
            __builtin_amdgcn_sched_group_barrier(0x100, MT, 0);            // next step's A fragments (LDS) first
            __builtin_amdgcn_sched_group_barrier(0x020, NT, 0);            // then the weight prefetch
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * MT * NT, 0);   // then this step's MFMAs
            __builtin_amdgcn_sched_barrier(0);
        }
        wso += NSTEP * 1024;
        if (more) st.store(smem + ((c + 1) & 1) * BUF, tid);
        __syncthreads();
    }

    if constexpr (EXPER == 2) {
        const unsigned long long t1c = __builtin_amdgcn_s_memtime(), t1r = __builtin_amdgcn_s_memrealtime();
        if (tid == 0 && a.dbg) {
            a.dbg[6 * blockIdx.x] = t1c - t0c;
            a.dbg[6 * blockIdx.x + 1] = t1r - t0r;
            a.dbg[6 * blockIdx.x + 2] = tentry;
            a.dbg[6 * blockIdx.x + 3] = t0r;
            a.dbg[6 * blockIdx.x + 4] = t1r;
        }
        if (tid == 0 && a.dbg) a.dbg[6 * gridDim.x + blockIdx.x] =
            ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
    // epilogue: bias + ReLU, NHWC store.  C layout of the 32x32 tile: column (lane&31) = channel,
    // row = (r&3) + 8*(r>>2) + 4*(lane>>5) = pixel; each store instruction writes 2 x 128 B segments.
    if constexpr (MT % 2 == 0) {
        if (a.pool_out) {
            // fused MaxPool2d(2,2,ceil_mode=True) of this layer's output (net.py:46,53,66): the 2x2 window of
            // a pooled pixel is rows (m, m+1) x registers (r, r+1) of ONE lane, so the max is register-local and
            // only the pooled tensor [ceil(H/2)][ceil(W/2)][cout] is written.  max commutes exactly with the
            // monotone "+bias" and ReLU.  Windows clipped by the image edge use the valid elements only.
            const int Hp = (a.H + 1) >> 1, Wp = (a.W + 1) >> 1;
#pragma unroll
            for (int m = 0; m < MT; m += 2) {
                const int y = ty0 + wm * MT + m;
                if (y >= a.H) continue;
                const bool row1 = y + 1 < a.H;
                float* __restrict__ orow = a.out + ((size_t)img * Hp + (y >> 1)) * Wp * a.cout;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int co = ((ct * WN + wn) * NT + n) * 32 + li;
                    const float b = a.bias[co];
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const int x = tx0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const bool col1 = x + 1 < a.W;
                        float v = acc[m][n][r];
                        if (col1) v = fmaxf(v, acc[m][n][r + 1]);
                        if (row1) {
                            v = fmaxf(v, acc[m + 1][n][r]);
                            if (col1) v = fmaxf(v, acc[m + 1][n][r + 1]);
                        }
                        v += b;
                        if (a.relu) v = fmaxf(v, 0.f);
                        if (x < a.W) orow[(size_t)(x >> 1) * a.cout + co] = v;
                    }
                }
            }
            return;
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int y = ty0 + wm * MT + m;
        if (y >= a.H) continue;
        float* __restrict__ orow = a.out + ((size_t)img * a.H + y) * a.W * a.cout;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int co = ((ct * WN + wn) * NT + n) * 32 + li;
            const float b = a.bias[co];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int x = tx0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = acc[m][n][r] + b;
                if (a.relu) v = fmaxf(v, 0.f);
                if (x < a.W) orow[(size_t)x * a.cout + co] = v;
            }
        }
    }
    if constexpr (EXPER == 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0 && a.dbg) a.dbg[6 * blockIdx.x + 5] = __builtin_amdgcn_s_memrealtime();
    }
}

// ---------------------------------------------------------------------------------------------
// Persistent form of the same implicit GEMM.
//
// Why: in the one-tile-per-block kernel every resident block of a launch starts and ends at the same moment
// (equal work), so once per "round" all CUs run their epilogues at once: a chip-wide burst of output stores
// (tens of MB) during which no MFMA issues, then the block exit / dispatch / prologue latency.  A timeline probe
// (tools/clock_probe.py) shows main loops at ~99 % MFMA occupancy but 9-40 us of store drain + 2 us of prologue
// per 250 us round.  Here a block is resident for the whole launch and walks tiles t = b, b+G, b+2G, ...:
//   * the finished tile's accumulators STAY in registers while the next tile accumulates into a second set, and
//     are written out as a trickle (4 dwords per lane per k-group) during the next tile's first 16-channel chunk:
//     no store burst, nothing waits for the stores;
//   * the next tile's first halo chunk and its weight fragments are prefetched during the current tile's last
//     chunk: no prologue bubble; the weight stream simply continues into the next tile's stream.
// ---------------------------------------------------------------------------------------------
struct TileId { int tx0, ty0, ct, img; };

// Reads one accumulator element where it lives (the AGPR half of the register file).  Without this hipcc copies
// the whole previous-tile accumulator (64 registers) into VGPRs before the trickle starts, which costs a wave of
// occupancy; the "a" constraint keeps the value in its accumulator register until this point.
__device__ __forceinline__ float acc_read(float in_acc) {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(in_acc));
    return v;
}

__device__ __forceinline__ void buf_store1(rsrc_t r, float v, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}

// Output addressing of one tile for the trickle epilogue: a buffer descriptor over the image's output tensor, one
// per-lane byte offset (VGPR) and scalar offsets per element, so a store costs no vector address arithmetic.
// A lane whose pixel falls outside the image gets an out-of-range offset: the descriptor's range check drops it.
struct OutAddr {
    rsrc_t rs;
    int lane_off;   // bytes: (4*lh pixels) * cout + channel lane
    int xlim;       // pixels of this lane's half still inside the row: W - tx0 - 4*lh
};

template <int MT, int NT, int WN, int STEP>
__device__ __forceinline__ void trickle_store(const ConvArgs& a, const f32x16 (&acc)[MT][NT], const float (&bias)[NT],
                                              const TileId& t, const OutAddr& o, int wm) {
    // step STEP writes elements [4*STEP, 4*STEP+4): one (m, n) tile, registers r = 4q .. 4q+3 (4 consecutive pixels)
    constexpr int E = STEP * 4;
    if constexpr (E < MT * NT * 16) {
        constexpr int m = E / (NT * 16), n = (E / 16) % NT, q = (E % 16) / 4;
        const int y = t.ty0 + wm * MT + m;
        if (y < a.H) {
            const int row = ((y * a.W + t.tx0 + 8 * q) * a.cout + n * 32) * 4;   // scalar byte offset
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = acc_read(acc[m][n][4 * q + j]) + bias[n];
                if (a.relu) v = fmaxf(v, 0.f);
                buf_store1(o.rs, v, (8 * q + j < o.xlim) ? o.lane_off : 0x7fffffff, row + j * a.cout * 4);
            }
        }
    }
}

template <int MT, int NT, int WN, int STEP>
__device__ __forceinline__ void trickle_store_pooled(const ConvArgs& a, const f32x16 (&acc)[MT][NT], const float (&bias)[NT],
                                                     const TileId& t, const OutAddr& o, int wm) {
    // fused 2x2 ceil-mode max-pool of the output: step STEP writes pooled outputs [2*STEP, 2*STEP+2);
    // o.lane_off here is (2*lh pooled pixels) * cout + channel lane, o.xlim as above (unpooled pixels)
    constexpr int O = STEP * 2;
    if constexpr (MT % 2 == 0 && O < (MT / 2) * NT * 8) {
        constexpr int mp = O / (NT * 8), n = (O / 8) % NT, r0 = (O % 8) * 2;
        constexpr int m = mp * 2;
        const int Wp = (a.W + 1) >> 1;
        const int y = t.ty0 + wm * MT + m;
        if (y < a.H) {
            const bool row1 = y + 1 < a.H;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = r0 + 2 * j;                       // even register: pixel x = tx0 + xr + 4*lh
                const int xr = (r & 3) + 8 * (r >> 2);
                const bool col1 = xr + 1 < o.xlim;
                float v = acc_read(acc[m][n][r]);
                v = col1 ? fmaxf(v, acc_read(acc[m][n][r + 1])) : v;
                if (row1) {
                    v = fmaxf(v, acc_read(acc[m + 1][n][r]));
                    v = col1 ? fmaxf(v, acc_read(acc[m + 1][n][r + 1])) : v;
                }
                v += bias[n];
                if (a.relu) v = fmaxf(v, 0.f);
                const int soff = ((((y >> 1) * Wp + ((t.tx0 + xr) >> 1)) * a.cout) + n * 32) * 4;
                buf_store1(o.rs, v, (xr < o.xlim) ? o.lane_off : 0x7fffffff, soff);
            }
        }
    }
}

template <int MODE, int WM, int WN, int MT, int NT, int PF>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv3x3_persist_kernel(ConvArgs a) {
    constexpr int TH = WM * MT;
    constexpr int NTHR = WM * WN * 64;
    using Stager = HaloStager<MODE, TH, NTHR>;
    constexpr int BUF = Stager::HALO * LSTR;
    constexpr int NSTEP = 9 * (KC / 8);
    constexpr int RING = PF + 1;
    static_assert(NSTEP % RING == 0, "ring slot bookkeeping assumes NSTEP % RING == 0");
    static_assert(MT * NT * 4 <= NSTEP, "the trickle epilogue must fit into one chunk");
    __shared__ __attribute__((aligned(16))) float smem[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const int tiles = a.tiles_x * a.tiles_y;
    const int nct = a.cout / (WN * NT * 32);
    const int total = tiles * nct * a.n;
    const int G = gridDim.x;
    const int nchunks = a.cin / KC;
    const int tile_bytes = nchunks * NSTEP * 1024;
    const unsigned img_bytes = (unsigned)a.Hs * a.Ws * a.cin * 4u;
    const rsrc_t wsr = make_rsrc(a.wpk, (unsigned)a.cin * a.cout * 36u);
    const int wvo = lane * 16;
    const int a_base = ((wm * MT) * HW_ + li) * LSTR + lh * 4;

    auto decode = [&](int t) {
        TileId r;
        int pt;
        if (a.xcd_order) {          // channel tile fastest (see conv3x3_mfma_kernel)
            r.ct = t % nct;
            const int q = t / nct;
            pt = q % tiles;
            r.img = q / tiles;
        } else {
            pt = t % tiles;
            const int q = t / tiles;
            r.ct = q % nct;
            r.img = q / nct;
        }
        r.tx0 = (pt % a.tiles_x) * TW;
        r.ty0 = (pt / a.tiles_x) * TH;
        return r;
    };
    auto stream_base = [&](const TileId& t) { return ((t.ct * WN + wn) * NT) * tile_bytes; };
    auto src_of = [&](const TileId& t) { return make_rsrc(a.in + (size_t)t.img * a.Hs * a.Ws * a.cin, img_bytes); };
    auto load_bias = [&](const TileId& t, float (&b)[NT]) {
#pragma unroll
        for (int n = 0; n < NT; ++n) b[n] = a.bias[((t.ct * WN + wn) * NT + n) * 32 + li];
    };
    auto out_addr = [&](const TileId& t) {
        OutAddr o;
        const size_t opix = a.pool_out ? (size_t)((a.H + 1) >> 1) * ((a.W + 1) >> 1) : (size_t)a.H * a.W;
        o.rs = make_rsrc(a.out + (size_t)t.img * opix * a.cout, (unsigned)(opix * a.cout * 4));
        o.lane_off = (((a.pool_out ? 2 : 4) * lh) * a.cout + ((t.ct * WN + wn) * NT) * 32 + li) * 4;
        o.xlim = a.W - t.tx0 - 4 * lh;
        return o;
    };

    int t = blockIdx.x;
    if (a.xcd_order && (G & 7) == 0) t = (t & 7) * (G >> 3) + (t >> 3);   // blocks of one XCD walk neighbouring tiles
    if (t >= total) return;
    TileId cur = decode(t);

    Stager st;
    st.init(tid, cur.ty0, cur.tx0, a.H, a.W, a.Hs, a.Ws, a.cin);
    int wso = stream_base(cur);
    f32x4 bq[RING][NT];
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int n = 0; n < NT; ++n) bq[p][n] = buf_load4(wsr, wvo, wso + n * tile_bytes + p * 1024);
    st.load(src_of(cur), 0);
    st.store(smem, tid);
    __syncthreads();
    int gc = 0;   // global chunk counter: LDS buffer parity

    f32x16 accA[MT][NT], accB[MT][NT];
    float biasA[NT], biasB[NT];
    TileId prev = cur;
    bool have_prev = false;

    // One tile: accumulate into acc_cur; during chunk 0 trickle out acc_prev (the previous tile's result).
    auto process = [&](f32x16 (&acc_cur)[MT][NT], float (&bias_cur)[NT], const f32x16 (&acc_prev)[MT][NT],
                       const float (&bias_prev)[NT]) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_cur[m][n][r] = 0.f;
        load_bias(cur, bias_cur);
        const int t_next = t + G;
        const bool has_next = t_next < total;
        const TileId nxt = decode(has_next ? t_next : t);
        const int wso_next = stream_base(nxt);
        const rsrc_t src_cur = src_of(cur);
        const OutAddr oprev = out_addr(prev);
        for (int c = 0; c < nchunks; ++c) {
            const bool last = c + 1 == nchunks;
            const float* sbuf = smem + (gc & 1) * BUF + a_base;
            const bool stage = !last || has_next;
            if (!last) {
                st.load(src_cur, (c + 1) * KC * 4);
            } else if (has_next) {
                st.init(tid, nxt.ty0, nxt.tx0, a.H, a.W, a.Hs, a.Ws, a.cin);
                st.load(src_of(nxt), 0);
            }
            // weight records past this chunk: the next chunk of this tile, or the first records of the next tile
            const int wnext = last ? wso_next - NSTEP * 1024 : wso;
            const bool trickle = have_prev && c == 0;
            f32x4 aq[2][MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) aq[0][m] = *(const f32x4*)(sbuf + (m * HW_) * LSTR);
            auto step = [&](auto KK) {
                constexpr int k = decltype(KK)::value;
                if constexpr (k + 1 < NSTEP) {
                    constexpr int tp = (k + 1) >> 1, g = (k + 1) & 1;
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        aq[(k + 1) & 1][m] = *(const f32x4*)(sbuf + ((m + tp / 3) * HW_ + (tp % 3)) * LSTR + g * 8);
                }
                constexpr int pos = k + PF;
                const int wb = (pos < NSTEP ? wso : wnext) + pos * 1024;
#pragma unroll
                for (int n = 0; n < NT; ++n) bq[pos % RING][n] = buf_load4(wsr, wvo, wb + n * tile_bytes);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            acc_cur[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[k & 1][m][s], bq[k % RING][n][s], acc_cur[m][n], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, MT, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, NT, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * MT * NT, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (trickle) {
                    if (a.pool_out) trickle_store_pooled<MT, NT, WN, k>(a, acc_prev, bias_prev, prev, oprev, wm);
                    else trickle_store<MT, NT, WN, k>(a, acc_prev, bias_prev, prev, oprev, wm);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            [&]<int... K>(std::integer_sequence<int, K...>) { (step(std::integral_constant<int, K>{}), ...); }
            (std::make_integer_sequence<int, NSTEP>{});
            wso = last ? wso_next : wso + NSTEP * 1024;
            if (stage) st.store(smem + ((gc + 1) & 1) * BUF, tid);
            __syncthreads();
            ++gc;
        }
        prev = cur;
        have_prev = true;
        t = t_next;
        cur = nxt;
        return has_next;
    };

    // final flush of a block's last tile (all elements at once)
    auto flush = [&](const f32x16 (&acc)[MT][NT], const float (&bias)[NT]) {
        const OutAddr o = out_addr(prev);
        [&]<int... K>(std::integer_sequence<int, K...>) {
            ((a.pool_out ? trickle_store_pooled<MT, NT, WN, K>(a, acc, bias, prev, o, wm)
                         : trickle_store<MT, NT, WN, K>(a, acc, bias, prev, o, wm)), ...);
        }(std::make_integer_sequence<int, MT * NT * 4>{});
    };

    while (true) {
        if (!process(accA, biasA, accB, biasB)) { flush(accA, biasA); break; }
        if (!process(accB, biasB, accA, biasA)) { flush(accB, biasB); break; }
    }
}

// ---------------------------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------------------------
int launch_pack_conv3x3(const float* w, float* p, int cin, int cout, hipStream_t s) {
    if (cin % KC || cout % 32) { set_error("pack_conv3x3: cin %% 16 or cout %% 32 != 0 (%d, %d)", cin, cout); return -1; }
    const size_t total = (size_t)cin * cout * 9;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(pack_conv3x3_kernel, dim3(blocks), dim3(256), 0, s, w, p, cin, cout);
    return check_launch("pack_conv3x3");
}

template <int MODE, int WM, int WN, int MT, int NT, int LDSPAD, int PF = 2, int EXPER = 0>
static int launch_cfg(ConvArgs a, hipStream_t s) {
    constexpr int TH = WM * MT, BN = WN * NT * 32;
    if (a.cout % BN) { set_error("conv3x3: cout %d not a multiple of %d", a.cout, BN); return -1; }
    if (a.pool_out && (MT % 2)) { set_error("conv3x3: this tile variant cannot fuse the output pool"); return -1; }
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + TH - 1) / TH;
    const long long blocks = (long long)a.tiles_x * a.tiles_y * (a.cout / BN) * a.n;
    if (blocks <= 0 || blocks > 0x7fffffffLL) { set_error("conv3x3: bad grid %lld", blocks); return -1; }
    hipLaunchKernelGGL((conv3x3_mfma_kernel<MODE, WM, WN, MT, NT, LDSPAD, PF, EXPER>), dim3((unsigned)blocks), dim3(WM * WN * 64), 0, s, a);
    return check_launch("conv3x3");
}

template <int MODE, int WM, int WN, int MT, int NT, int PF>
static int launch_persist(ConvArgs a, hipStream_t s) {
    constexpr int TH = WM * MT, BN = WN * NT * 32;
    if (a.cout % BN) { set_error("conv3x3: cout %d not a multiple of %d", a.cout, BN); return -1; }
    if (a.pool_out && (MT % 2)) { set_error("conv3x3: this tile variant cannot fuse the output pool"); return -1; }
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + TH - 1) / TH;
    const long long total = (long long)a.tiles_x * a.tiles_y * (a.cout / BN) * a.n;
    if (total <= 0 || total > 0x7fffffffLL) { set_error("conv3x3: bad grid %lld", total); return -1; }
    // resident block slots of this kernel on the chip (host query, cached): the grid never exceeds them, and the
    // tiles are dealt evenly: G = ceil(total / rounds)
    static int slots = 0;
    if (!slots) {
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv3x3_persist_kernel<MODE, WM, WN, MT, NT, PF>, WM * WN * 64, 0) != hipSuccess ||
            hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || per_cu < 1) {
            set_error("conv3x3: occupancy query failed");
            return -2;
        }
        slots = per_cu * prop.multiProcessorCount;
    }
    const long long rounds = (total + slots - 1) / slots;
    const unsigned G = (unsigned)((total + rounds - 1) / rounds);
    hipLaunchKernelGGL((conv3x3_persist_kernel<MODE, WM, WN, MT, NT, PF>), dim3(G), dim3(WM * WN * 64), 0, s, a);
    return check_launch("conv3x3(persistent)");
}

// Tile variants (block = WM x WN waves, wave tile = MT rows x 32 px by NT x 32 channels, PF = weight prefetch
// depth in k-groups); measured with tools/tune_conv.py on the config-2 layer shapes:
//   0: 8 rows x 64 ch,  4 waves of 2 rows x 64 ch, PF 5   (pool-out capable default)
//   1: 8 rows x 128 ch, 8 waves (4x2) of 2 rows x 64 ch, PF 5   (+3 % on >= 128-channel layers with >= 512 blocks)
//   2: 4 rows x 64 ch,  4 waves of 1 row x 64 ch, PF 5    (best without output pool; 2x the blocks for small maps)
//   3: as 0 with PF 2 (round-1 baseline, kept for A/B runs)
//   4: 4 rows x 64 ch,  2 waves of 2 rows x 64 ch, PF 2   (pool-out on small maps)
//   5: PERSISTENT form of 3 (conv3x3_persist_kernel: resident blocks walk the tiles, trickle epilogue)
//   6: PERSISTENT form of 2: best on the low-K layers (cin <= 128), where the epilogue share is largest
//      (steady-state, config-2 shapes: conv2_1 136 vs 127 TF/s, dec8 138 vs 131, dec7 140 vs 137, dec6 142 vs 140)
//   7: 8 rows x 64 ch, 8 waves (4x2) of 2 rows x 32 ch, PF 5 (pool-out capable; best at cout = 64)
//  10: DIAGNOSTIC build of 3 with clock / timeline stamps (tools/clock_probe.py); never used by the product path
template <int MODE>
static int launch_variant(const ConvArgs& a, int variant, hipStream_t s) {
    switch (variant) {
        case 0: return launch_cfg<MODE, 4, 1, 2, 2, 0, 5, 0>(a, s);
        case 1: return launch_cfg<MODE, 4, 2, 2, 2, 0, 5, 0>(a, s);
        case 2: return launch_cfg<MODE, 4, 1, 1, 2, 0, 5, 0>(a, s);
        case 3: return launch_cfg<MODE, 4, 1, 2, 2, 0, 2, 0>(a, s);
        case 4: return launch_cfg<MODE, 2, 1, 2, 2, 0, 2, 0>(a, s);
        case 5:   // persistent, 8 rows x 64 ch, trickle epilogue
            if constexpr (MODE != SRC_POOL2) return launch_persist<MODE, 4, 1, 2, 2, 2>(a, s);
            break;
        case 6:   // persistent, 4 rows x 64 ch
            if constexpr (MODE != SRC_POOL2) return launch_persist<MODE, 4, 1, 1, 2, 5>(a, s);
            break;
        case 7:   // 8 rows x 64 ch, 8 waves (4x2) of 2 rows x 32 ch: best pool-out form at cout = 64 (conv1_2: 139 vs 135 TF/s)
            return launch_cfg<MODE, 4, 2, 2, 1, 0, 5, 0>(a, s);
#ifdef ADAIN_DIAG
        case 10: return launch_cfg<MODE, 4, 1, 2, 2, 0, 2, 2>(a, s);
#endif
    }
    set_error("conv3x3: unknown tile variant %d", variant);
    return -1;
}

int conv3x3_auto_variant(const ConvArgs& a, int src_mode) {
    const long long tiles8 = (long long)((a.W + 31) / 32) * ((a.H + 7) / 8) * a.n;
    const bool wide_ok = a.cout % 128 == 0 && tiles8 * (a.cout / 128) >= 512;
    if (a.pool_out) {
        if (wide_ok) return 1;
        return tiles8 * (a.cout / 64) >= 256 ? 7 : 4;
    }
    if (a.cin <= 128 && src_mode != SRC_POOL2) return 6;
    return wide_ok ? 1 : 2;
}

int launch_conv3x3(const ConvArgs& a, int src_mode, int variant, hipStream_t s) {
    if (a.cin % KC || a.cin < KC) { set_error("conv3x3: cin %d not a multiple of 16", a.cin); return -1; }
    if (a.H < 2 || a.W < 2 || a.n < 1) { set_error("conv3x3: H, W must be >= 2 (reflection pad), got %dx%d", a.H, a.W); return -1; }
    if ((size_t)a.Hs * a.Ws * a.cin * 4 >= 0x7fffffffULL || (size_t)a.H * a.W * a.cout * 4 >= 0x7fffffffULL) {
        set_error("conv3x3: per-image source and output tensors must stay below 2 GiB (32-bit buffer offsets)");
        return -1;
    }
    if (variant < 0) variant = conv3x3_auto_variant(a, src_mode);
    ConvArgs b = a;
    static const int xcd_env = tune_env("ADAIN_XCD_ORDER", 1);
    b.xcd_order = xcd_env;
    switch (src_mode) {
        case SRC_DIRECT:
            if (a.Hs != a.H || a.Ws != a.W) { set_error("conv3x3: direct mode needs Hs==H, Ws==W"); return -1; }
            return launch_variant<SRC_DIRECT>(b, variant, s);
        case SRC_UP2X:
            if (a.H != 2 * a.Hs || a.W != 2 * a.Ws) { set_error("conv3x3: up2x mode needs H==2Hs, W==2Ws"); return -1; }
            return launch_variant<SRC_UP2X>(b, variant, s);
        case SRC_POOL2:
            if (a.H != (a.Hs + 1) / 2 || a.W != (a.Ws + 1) / 2) { set_error("conv3x3: pool mode needs H==ceil(Hs/2), W==ceil(Ws/2)"); return -1; }
            return launch_variant<SRC_POOL2>(b, variant, s);
    }
    set_error("conv3x3: unknown src_mode %d", src_mode);
    return -1;
}

}  // namespace adain
