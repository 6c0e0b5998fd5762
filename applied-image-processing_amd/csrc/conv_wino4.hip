// Winograd F(4,3) x F(2,3) form of the 3x3 convolution, gfx950: 4 x 2 output tiles (4 rows, 2 columns) from 6 x 4 input
// patches, 24 multiplies per (cin, cout) pair and tile = 3 per output instead of 4 for F(2x2,3x3) and 9 for the direct form.
//
//   Y = A4^T [ sum_cin (G4 g G2^T) .* (B4^T d B2) ] A2          (rows: F(4,3), points 0, +-1, +-2, inf; columns: F(2,3))
//
// fp32 error of one layer ~ 1.2e-6 relative (F(2x2,3x3): 4.6e-7, direct: 2.3e-7), far inside the path's tolerance.
// F(4,3) is applied along the ROWS so that neighbouring lanes' patches stay 2 pixels apart in the LDS halo image (the
// conflict pattern of the F(2x2,3x3) kernels; 4 pixels apart would be a 4-way conflict for every 16-byte-aligned pixel stride).
//
// Mapping (that of the retired F(2x2,3x3) kernels - docs/HISTORY.md - with the roles of rows and columns swapped):
//   * block = 4 waves, tile = 32 Winograd tiles (2 tile rows x 16 tile columns = 8 x 32 output pixels) x 32 channels;
//   * the 24 transform positions xi = (r, j) are 24 GEMMs [32 tiles x cin] x [cin x 32]; wave j owns transform COLUMN j
//     (its 6 row positions r): 6 MFMA tiles of 32x32 = 96 accumulator registers (v_mfma_f32_32x32x2_f32, A = weights,
//     B = tiles, so that a lane holds 4 consecutive channels of one tile);
//   * A-side operand in registers: column j of d B2 needs two patch columns, so a lane reads 6 rows x 2 columns (b128,
//     4 channels) from the raw reflect-padded 10 x 34 halo image in LDS, combines them (24 adds) and applies B4^T down the
//     rows (48 fma/adds); fragments are updated in place (row r's fragment is dead once step r has issued);
//   * weights U = G4 g G2^T packed once as [cout/32][j][cin/8][r][lane][4], streamed into a 6-slot register ring (slot = r);
//   * the raw halo is staged 16 channels at a time (global -> registers -> LDS), one barrier per 48 MFMAs per wave;
//   * epilogue: P_j = A4^T M_j per wave (4 values), the four columns meet in LDS (64 KiB, XOR-swizzled b128), then
//     Y = P A2: bias, ReLU, optional 2x2 max-pool (a tile holds two pool windows), b128 buffer stores.
//
// Tried and measured without effect (tools/tune_conv.py, all 13 layer shapes, +-1 %): pixel-tile-fastest and grouped
// block -> tile orders (weights L2-resident), raised wave priority outside the main loop, halo loads spread differently,
// a 12-slot weight ring (512-register build).  See DESIGN.md 4(C) for what the probes say bounds the kernel.
#include <stdlib.h>
#include <type_traits>
#include <utility>

#include "common.h"
#include "device_utils.h"

namespace adain {

namespace {
#ifndef W4_GROUP
#define W4_GROUP 1                           // 1: one MFMA per scheduling region (default); 8: eight MFMAs back to back, then their work
#endif
static_assert(W4_GROUP == 1 || W4_GROUP == 8, "the burst schedule is laid out for groups of 8 (one pair of row positions)");
constexpr int W4_KR = 16;                    // channels per raw stage = 2 chunks of 8
constexpr int W4_RSTR = W4_KR + 4;           // floats per halo pixel (80 B = 5 quads: an odd number of 16-byte bank quads)
constexpr int W4_RITEMS = 6;                 // staging items per thread: at most 340 halo pixels x 4 quads over 256 threads
// Tile geometry (template parameter GEO): the 32 Winograd tiles of a workgroup sit either 2 x 16 (GEO 0: 8 x 32 output pixels, a
// 10 x 34 halo - the tuned default) or 4 x 8 (GEO 1: 16 x 16 output pixels, an 18 x 18 halo).  One kernel, two layouts: the
// launcher picks per layer the one that pads the feature map less (W = 400 pads to 416 with 32-wide tiles but not with
// 16-wide ones; the reference's own video frames, 256 x 456, give maps 228 / 114 / 57 wide).
//
// LDS image of the raw halo, DE-INTERLEAVED by column parity: plane E holds the even halo columns, plane O the odd ones, each
// [HALO_H rows][HALO_W / 2 columns] pixels of 20 floats.  The tiles of a workgroup sit 2 pixels apart, i.e. 1 plane pixel = 5
// quads: the 16 lanes that one LDS cycle of a ds_read_b128 serves ({0-3,12-15,20-27} ..., MI355X_MICROARCH.md, LDS) must read
// 16 different quads of the 64 banks.  In one interleaved image (round 1: [10][34] pixels) neighbouring tiles are 10 quads apart
// and every patch read was a two-way bank conflict (SQ_LDS_BANK_CONFLICT = 49 % of SQ_LDS_IDX_ACTIVE).
//   GEO 0: lane li = tile (row li >> 4, column li & 15).  Row stride 88 quads (17 x 5 + 3): the second tile row, 4 halo rows
//          down, lands on the same quads modulo 16 as the first, so the columns 4-11 of one row and 0-3 / 12-15 of the other
//          (the hardware's lane groups) still cover all 16.
//   GEO 1: lane li = tile (row li >> 3, column li & 7).  A lane group reads columns 0-3 of tile rows 0 and 3 and columns 4-7 of
//          rows 1 and 2 (or the complement): columns 0-3 sit on quads {0,5,10,15}, columns 4-7 on {4,9,14,3} (mod 16), and the
//          two sets together with their shifts by 8 cover all 16 - so tile rows 1 and 3 must be shifted by 8 quads and row 2 by
//          0 (mod 16): 4 S = 8, 8 S = 0, 12 S = 8 (mod 16) for a row stride of S quads, i.e. S = 2 (mod 4): 46 = 9 x 5 + 1.
// Plane O starts 4 quads (mod 8) after plane E: the two neighbouring pixels that one 8-lane group of a staging ds_write_b128
// stores cover all 32 banks.
template <int GEO>
struct W4G {
    static constexpr int TROWS_LOG = GEO == 0 ? 1 : 2;             // Winograd-tile rows of a workgroup = 2 / 4
    static constexpr int TCOLS_LOG = 5 - TROWS_LOG;                // Winograd-tile columns = 16 / 8
    static constexpr int TILE_H = 4 << TROWS_LOG, TILE_W = 2 << TCOLS_LOG;        // output pixels: 8 x 32 / 16 x 16
    static constexpr int HALO_H = TILE_H + 2, HALO_W = TILE_W + 2;
    static constexpr int HALO = HALO_H * HALO_W;                   // 340 / 324 pixels
    static constexpr int PROW = (HALO_W / 2) * W4_RSTR + (GEO == 0 ? 12 : 4);      // 352 / 184 floats per plane row
    static constexpr int PLANE = HALO_H * PROW + (GEO == 0 ? 16 : 0);              // 3536 / 3312 floats per plane
    static constexpr int TAIL = 2 * PLANE;                         // staging items past the halo land here, never read
    static_assert(TAIL + (W4_RITEMS * 64 - HALO) * W4_RSTR <= 8192, "LDS layout");
    static_assert(PROW % 4 == 0 && (PLANE / 4) % 8 == 4, "bank layout");
    static_assert(GEO != 0 || (PROW / 4) % 16 == 8, "bank layout (2 x 16 tiles)");
    static_assert(GEO != 1 || (PROW / 4) % 4 == 2, "bank layout (4 x 8 tiles)");
    static_assert(HALO * 4 <= W4_RITEMS * 256, "staging items");
};
constexpr int W4_RBUF = 8192;                          // 2 planes + tail, rounded: two buffers = the exchange area
constexpr int W4_PEX = 4 * 4 * 32 * 32;                // [column j][a][tile][32 channels] floats = 64 KiB
static_assert(2 * W4_RBUF <= W4_PEX, "LDS layout");

// 128-bit buffer store.  The scalar offset operand is deliberately NOT exposed (always the literal 0): with an SGPR soffset LLVM
// models no hazard between a >64-bit MUBUF store and a following VALU write of its data registers and emits none of the wait states
// it emits for the literal form - and on gfx950, with two waves per SIMD queueing VMEM work, the store then reads data the next
// instruction has already overwritten (seen in round 2's F(2x2,3x3) kernel: the register allocator reused the first data register of
// one store as the address of the next; the .x lane values arrived as address bit patterns).  With the literal form the compiler
// inserts its `s_nop 1` and the stores are exact.
__device__ __forceinline__ void buf_store4(rsrc_t r, f32x4 v, int voff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, 0);
}
}  // namespace

// OIHW [cout][cin][3][3] -> U = G4 g G2^T packed as [cout/32][j 4][cin/8][r 6][lane 64][s 4]:
//   value = U[r][j][cout = 32 ct + (lane & 31)][cin = 8 chunk + 4 (lane >> 5) + s]
__global__ void pack_wino4_kernel(const float* __restrict__ w, float* __restrict__ p, int cin, int cout) {
    // transformed in double and rounded once: U carries half an ulp instead of the few ulp of nine fp32 products with 1/6, 1/12, 1/24
    // (a systematic, per-weight error; runs once per weight set)
    const double G2[4][3] = {{1., 0., 0.}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0., 0., 1.}};
    const double G4[6][3] = {{0.25, 0., 0.},           {-1. / 6, -1. / 6, -1. / 6}, {-1. / 6, 1. / 6, -1. / 6},
                             {1. / 24, 1. / 12, 1. / 6}, {1. / 24, -1. / 12, 1. / 6}, {0., 0., 1.}};
    const size_t total = (size_t)cin * cout * 24;
    const int nch = cin / 8;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        size_t t = idx;
        const int s = t & 3; t >>= 2;
        const int lane = t & 63; t >>= 6;
        const int r = t % 6; t /= 6;
        const int chunk = t % nch; t /= nch;
        const int j = t & 3; t >>= 2;
        const int ct = (int)t;
        const int co = ct * 32 + (lane & 31), ci = chunk * 8 + 4 * (lane >> 5) + s;
        const float* g = w + ((size_t)co * cin + ci) * 9;
        double u = 0.;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) u += G4[r][a] * (double)g[a * 3 + b] * G2[j][b];
        p[idx] = (float)u;
    }
}

// DIAG (not the product path; selected by ADAIN_W4_DIAG when a stamp buffer is set, tools/wino4_probe.py, tools/wino4_phase_probe.py,
// tools/probes/diag_ab.py):
//   1 = a shader-clock stamp per 8-MFMA double step of the first 32 workgroups, 2 = the same with LDS padded to one workgroup per
//   CU, 3 = four s_memrealtime phase stamps per wave (entry, main-loop start / end, exit);
//   timing-only ablations of the one-tile form (wrong results by construction; A/B against the product kernel, two workgroups
//   per CU, 256->256 at 256^2 / 64->64 at 1024^2): 5 = no input transform at all (+16 %), 6 = no patch reads, arithmetic on stale
//   registers (+16 %: the LDS reads are the transform's whole cost), 9 = ds_read_b32 instead of b128 (+5 %), 7 = no stage
//   barrier (+2 %), 8 = no halo loads or stores (+8 / +10 %), 10 = no halo loads (+5 / +8 %), 11 = no halo stores (+3 %),
//   12 = (correct results) 12-slot weight ring, a whole chunk ahead (+1 %); 13 = no weight loads in the main loop, 14 = no vector-memory
//   loads at all in the main loop (round 2: prices the CU's vector-memory path); a build that sent the halo straight to LDS
//   (`buffer_load ... lds`, wrong image layout, no staging registers or ds_writes; since removed) measured +3 %.  In a bare MFMA loop neither LDS reads nor streaming
//   weight loads cost the matrix pipe anything (tools/probes/mfma_chain_probe.hip), so these are waits, not port conflicts.
// PERSIST: a workgroup walks a list of tiles (grid = 2 per CU; XCD x owns a contiguous range of the tile list, channel tile
// fastest) instead of one: the next tile's first halo stage is loaded during the current tile's last stages and its first
// weight fragments replace the ring's run-off loads, so a tile starts with an LDS write + barrier + transform instead of a cold
// HBM round trip (tools/wino4_phase_probe.py: 3.4-4.1 us of prologue per one-tile workgroup; with cin = 64 both workgroups of a
// CU are inside their main loops only 22 % of the time).  The wave priority alternates per tile between the two halves of the
// grid: the SIMD arbiter prefers the older wave - the workgroup dispatched first ran its tiles ~1.4x faster than its CU partner and
// then sat idle at the end of the launch (static tile lists); alternating the priority per tile, in opposite phase for workgroups b
// and b + grid / 2 (they share a CU), evens the two out (round 2's persistent F(2x2,3x3) kernel: both in their main loops 71 % -> 84 %
// of the time).
// SEGMENTS (persistent form only): the tile list may run over up to four (source, output) tensor pairs of different sizes
// that share the layer's weights - the content batch and the style image of one encoder pass (reference test.py:57,63 encodes
// both through the same vgg) - so that the deep style-branch layers, too small to fill the chip on their own, ride in the
// content launch's list.  A tile's geometry (H, W, pointers) comes from its segment's descriptor, re-read from the kernel
// arguments at the two places that need it (the next tile's halo offsets, the epilogue's stores).
// BIG: per-tile buffer descriptors (any image size); the default form keeps one descriptor per image (tensors below 2 GiB)
template <int MODE, int DIAG = 0, bool PERSIST = false, bool BIG = false, int GEO = 0>
__global__ __launch_bounds__(256, DIAG == 2 ? 1 : 2) void conv3x3_wino4_kernel(ConvArgs a, ConvSegs m, int items, int prio_mode) {
    using G = W4G<GEO>;
    constexpr int W4_HALO_W = G::HALO_W, W4_HALO_H = G::HALO_H, W4_HALO = G::HALO, W4_PROW = G::PROW, W4_PLANE = G::PLANE,
                  W4_TAIL = G::TAIL, TILE_W = G::TILE_W, TILE_H = G::TILE_H, TCL = G::TCOLS_LOG, TCM = (1 << G::TCOLS_LOG) - 1;
    __shared__ __attribute__((aligned(16))) float smem[DIAG == 2 ? W4_PEX + 8192 : (DIAG == 1 ? W4_PEX + 1024 : W4_PEX)];
    float* const Rs = smem;
    unsigned* const steplog = (unsigned*)(smem + W4_PEX);     // DIAG 1, 2: [wave][96] low words of s_memtime
    int nlog = 0;
    unsigned long long phase[4] = {0, 0, 0, 0};
    unsigned long long tpx[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // DIAG 4
    if constexpr (DIAG == 3) phase[0] = __builtin_amdgcn_s_memrealtime();

    const int tid = threadIdx.x;
    const int lane = tid & 63;
#ifdef W4_AGPR_ACC
    // an inline-asm AGPR operand makes the compiler select the MFMAs' AGPR-destination form: the 96 accumulator registers then
    // live in the accumulator half of the register file (experiment: -DW4_AGPR_ACC)
    { float agpr_hint = 1.0f; asm volatile("; agpr hint %0" ::"a"(agpr_hint)); }
#endif
    const int wj = __builtin_amdgcn_readfirstlane(tid >> 6);      // transform column of this wave
    // the one lane constant kept live through the main loop; other lane-derived addresses are rebuilt from an opaque copy
    const int wvo = lane * 16;
    auto lane_now = [&]() {
        int w = wvo;
        asm volatile("" : "+v"(w));
        return w >> 4;
    };

    // block -> (channel tile fastest, pixel tile, image) inside a contiguous per-XCD range (halo reuse in L2)
    int tiles = a.tiles_x * a.tiles_y;
    const int nct = a.cout / 32;
    int lid = blockIdx.x, ct, pt, img;
    // geometry of the tile being computed (gH, gW: conv output size; gWs: source row length) and of its tensors
    int gH = a.H, gW = a.W, gWs = a.Ws, gtx = a.tiles_x;
    int seg = 0;                                     // persistent form: segment of the current tile
    // CIN SPLIT (one-tile form only, launches too small to give every compute unit a tile: launch_conv3x3_wino4): workgroup
    // (tile, channel tile, ks) accumulates input channels [ks * cin_sub, (ks + 1) * cin_sub) only and writes its output-transformed
    // partial sums - no bias, no ReLU, no pool - into slab ks of a workspace; splitk_combine_kernel adds the slabs in the fixed
    // order ks = 0, 1, ... and finishes the layer.  The output transform is linear, so the split changes rounding only.
    int ks = 0;
    auto seg_of = [&](int it) {
        int si = 0;
        for (int k = 1; k < m.count; ++k)
            if (it >= m.s[k].item0) si = k;
        return si;
    };
    // persistent form: the tile list of this workgroup = items lo + slot, lo + slot + stride, ... below hi
    const int stride = gridDim.x >> 3;
    const int lo = (int)((long long)items * (blockIdx.x & 7) / 8), hi = (int)((long long)items * ((blockIdx.x & 7) + 1) / 8);
    int item = lo + (blockIdx.x >> 3);
    if constexpr (PERSIST) {
        if (item >= hi) return;
        seg = seg_of(item);
        lid = item - m.s[seg].item0;
        gH = m.s[seg].H; gW = m.s[seg].W; gWs = m.s[seg].Ws; gtx = m.s[seg].tiles_x;
        tiles = gtx * m.s[seg].tiles_y;
    } else if (a.xcd_order && (gridDim.x & 7) == 0) {
        lid = (lid & 7) * (gridDim.x >> 3) + (lid >> 3);
    }
    // Persistent walk: inside a segment the items are ordered (channel-tile GROUP, image, pixel tile, channel tile in the group),
    // group size m.ctg.  With ctg == nct (every channel tile in one group) consecutive items are the channel tiles of one pixel
    // tile: the 64 workgroups resident on an XCD share halos, but a layer whose transformed weights exceed the XCD's 4 MB L2
    // (256 -> 256: 6.3 MB) re-fetches them from the Infinity Cache once per group of 8 pixel tiles.  With a smaller group an
    // XCD's contiguous range stays inside ONE group for many pixel tiles: its weights (cin * 32 ctg * 96 B) stay in L2 and
    // the halos are fetched once per group instead.
    auto decode = [&](int li, int ntl, int nimgs, int& ct_, int& pt_, int& img_) {
        const int G = m.ctg;
        const int c = li % G, r = li / G;
        const int per = ntl * nimgs;
        const int g = r / per, pi = r - g * per;
        ct_ = g * G + c;
        img_ = pi / ntl;
        pt_ = pi - img_ * ntl;
    };
    if constexpr (PERSIST) {
        decode(lid, tiles, m.s[seg].n, ct, pt, img);
    } else if (a.xcd_order == 2) {          // pixel tile fastest: the workgroups resident on an XCD share one or two channel tiles' weights
        pt = lid % tiles; lid /= tiles;
        ct = lid % nct;
        img = lid / nct;
    } else {                         // channel tile fastest: they share halos
        ct = lid % nct; lid /= nct;
        if (a.ksplit > 1) { ks = lid % a.ksplit; lid /= a.ksplit; }      // CIN SPLIT: then the cin ranges of one (pixel tile, channel tile)
        pt = lid % tiles;
        img = lid / tiles;
    }
    int tx0 = (pt % gtx) * TILE_W, ty0 = (pt / gtx) * TILE_H;
    const int nst = (PERSIST ? a.cin : a.cin_sub) / W4_KR;      // stages of this workgroup's cin range
    const int nch = a.cin / 8;

    // BIG: per-tile buffer descriptors.  The source descriptor starts at the first source row a tile's halo can touch, the output
    // descriptor (epilogue) at the tile's first output row, so every 32-bit offset stays inside a band of 10 source / 8 output
    // rows and an image may be of any size; the default form keeps one descriptor per image (row 0), which caps a tensor at
    // 2 GiB per image and costs 13 scalar registers less (same-box A/B: 0.4 % of the kernel's time).
    auto src_row0 = [&](int y0) {
        if constexpr (!BIG) return 0;
        const int r = max(y0 - 1, 0);                  // reflection never reaches above this row (row -1 maps to row 1)
        return MODE == SRC_UP2X ? r >> 1 : r;
    };
    auto src_of = [&](const float* base, int Hs, int Ws, int im, int y0) {
        if constexpr (!BIG) {
            const size_t per = (size_t)Hs * Ws * a.cin;
            return make_rsrc(base + im * per, (unsigned)(per * 4u));
        } else {
            const size_t row = (size_t)Ws * a.cin;
            const int r0 = src_row0(y0);
            const size_t left = (size_t)(Hs - r0) * row * 4;
            return make_rsrc(base + ((size_t)im * Hs + r0) * row, left < 0x7ffffff0ull ? (unsigned)left : 0x7ffffff0u);
        }
    };
    rsrc_t src = PERSIST ? src_of(m.s[seg].in, m.s[seg].Hs, m.s[seg].Ws, img, ty0) : src_of(a.in, a.Hs, a.Ws, img, ty0);
    const rsrc_t wsr = make_rsrc(a.wpk, (unsigned)a.cin * a.cout * 96u);

    // ---- raw halo staging: 340 pixels x 4 quads over 256 threads x 6 items ---------------------------------------------------
    // Per-lane byte offsets of the six staging items + a per-tile SCALAR base (the buffer load's soffset operand).  A tile whose
    // 10 x 34 halo lies inside the image needs no reflection (92 % of the tiles at 1024 x 1024), and then item k of a lane sits at
    // tile base + an offset that depends only on the lane and the segment's row pitch: the six registers are computed once per
    // segment (roff_seg remembers for which) and an interior tile costs scalar arithmetic only - round 2 recomputed them per
    // tile, 78 vector instructions of the ~560 a tile spends outside its main loop (they cost matrix-pipe time: DESIGN.md 4).
    // Tile origins are multiples of 8 / 32, so the nearest-2x source index splits the same way: (x0 + hx - 1) >> 1 =
    // (x0 / 2 - 1) + ((hx + 1) >> 1).  Tiles at the image border take the general path (reflection, base 0).
    int roff[W4_RITEMS];
    int tbase = 0, roff_seg = -1;
    auto halo_offsets = [&](int x0, int y0, int H, int W, int Ws, int sg) {
        const int t = PERSIST ? (lane_now() | (wj << 6)) : tid;        // persistent: recomputed per tile, nothing hoisted
        const bool interior = PERSIST && y0 >= 1 && y0 + W4_HALO_H - 1 <= H && x0 >= 1 && x0 + W4_HALO_W - 1 <= W;
        [[maybe_unused]] const int r0 = src_row0(y0);  // BIG: rows are counted from the tile's source descriptor
        if (interior) {
            if (roff_seg != sg) {
#pragma unroll
                for (int k = 0; k < W4_RITEMS; ++k) {
                    const int idx = t + k * 256;
                    const int hp = min(idx >> 2, W4_HALO - 1), q = idx & 3;
                    const int hy = hp / W4_HALO_W, hx = hp - hy * W4_HALO_W;
                    const int yc = MODE == SRC_UP2X ? (hy + 1) >> 1 : hy, xc = MODE == SRC_UP2X ? (hx + 1) >> 1 : hx;
                    roff[k] = ((yc * Ws + xc) * a.cin + q * 4) * 4;
                }
                roff_seg = sg;
            }
            // source pixel of the halo's top-left corner; BIG: the tile's descriptor already starts at that row
            const int by = BIG ? 0 : (MODE == SRC_UP2X ? (y0 >> 1) - 1 : y0 - 1), bx = MODE == SRC_UP2X ? (x0 >> 1) - 1 : x0 - 1;
            tbase = ((by * Ws + bx) * a.cin) * 4;
        } else {
#pragma unroll
            for (int k = 0; k < W4_RITEMS; ++k) {
                const int idx = t + k * 256;
                const int hp = min(idx >> 2, W4_HALO - 1), q = idx & 3;
                const int hy = hp / W4_HALO_W, hx = hp - hy * W4_HALO_W;
                int y = reflect1(y0 + hy - 1, H), x = reflect1(x0 + hx - 1, W);
                if (MODE == SRC_UP2X) { y >>= 1; x >>= 1; }
                if constexpr (BIG) y -= r0;
                roff[k] = ((y * Ws + x) * a.cin + q * 4) * 4;
            }
            roff_seg = -1;
            tbase = PERSIST ? 0 : ks * a.cin_sub * 4;       // first channel of this workgroup's cin range
        }
    };
    halo_offsets(tx0, ty0, gH, gW, gWs, seg);
    f32x4 rawreg[W4_RITEMS];
    auto raw_load = [&](int soff) {
#pragma unroll
        for (int k = 0; k < W4_RITEMS; ++k) rawreg[k] = buf_load4(src, roff[k], tbase + soff);
    };
    // LDS address (in BYTES from the start of a halo buffer) of staging item k of this thread: pixel (t >> 2) + 64 k of the halo,
    // quad t & 3; depends on the thread only, computed once.  Bytes, so that a store into a buffer whose place is known at compile
    // time is the register + an immediate offset (the persistent form's stage loop: no address arithmetic at all).
    int sa[W4_RITEMS];
    {
        const int t = tid;
#pragma unroll
        for (int k = 0; k < W4_RITEMS; ++k) {
            const int hp = (t >> 2) + 64 * k, q = t & 3;
            const int hy = hp / W4_HALO_W, hx = hp - hy * W4_HALO_W;
            sa[k] = 4 * (hp < W4_HALO ? (hx & 1) * W4_PLANE + hy * W4_PROW + (hx >> 1) * W4_RSTR + q * 4
                                      : W4_TAIL + (hp - W4_HALO) * W4_RSTR + q * 4);
            asm volatile("" : "+v"(sa[k]));      // held in a register: left alone the compiler recomputes it (8 vector
        }                                        // instructions per item and stage, and every one of them costs matrix-pipe time)
    }
    auto raw_store = [&](float* buf) {
#pragma unroll
        for (int k = 0; k < W4_RITEMS; ++k) *(f32x4*)((char*)buf + sa[k]) = rawreg[k];
    };

    // ---- weights: one b128 fragment per step (row position r), ring slot = r, loaded 5 steps ahead ---------------------------
    // RING12 (experiment, DIAG 12): two chunks of weight slots - a chunk's six fragments are requested during the previous chunk
    constexpr bool RING12 = DIAG == 12;
    f32x4 bq[RING12 ? 12 : 6];
    int wso = ((ct * 4 + wj) * nch + (PERSIST ? 0 : ks * (a.cin_sub >> 3))) * 6144;
#pragma unroll
    for (int r = 0; r < (RING12 ? 6 : 4); ++r) bq[r] = buf_load4(wsr, wvo, wso + r * 1024);

    f32x16 acc[6];
    if constexpr (!PERSIST) {       // the persistent form starts every tile with MFMAs on a zero C operand
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;
    }

    // ---- epilogue (a lambda: the persistent form runs it inside the tile loop) ------------------------------------------------------
    // lane (li = tile, lh): acc[r][e16] = M[row r][column wj][channel 8 (e16 >> 2) + 4 lh + (e16 & 3)][tile li]
    auto epilogue = [&]() {
    {
        const int le = lane_now(), li = le & 31, lh = le >> 5;
        float* Pw = smem + (wj * 4) * (32 * 32) + li * 32;
        const int sw = (li >> 1) & 7;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            f32x4 P0, P1, P2, P3;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = rq * 4 + e;
                const float s1 = acc[1][k] + acc[2][k], d1 = acc[1][k] - acc[2][k];
                const float s2 = acc[3][k] + acc[4][k], d2 = acc[3][k] - acc[4][k];
                P0[e] = acc[0][k] + s1 + s2;               // A4^T = (1,1,1,1,1,0) (0,1,-1,2,-2,0) (0,1,1,4,4,0) (0,1,-1,8,-8,1)
                P1[e] = d1 + 2.f * d2;
                P2[e] = s1 + 4.f * s2;
                P3[e] = d1 + 8.f * d2 + acc[5][k];
            }
            const int slot = ((2 * rq + lh) ^ sw) << 2;    // channel quad of the 32, XOR-swizzled: conflict-free both ways
            *(f32x4*)(Pw + slot) = P0;
            *(f32x4*)(Pw + 1 * 32 * 32 + slot) = P1;
            *(f32x4*)(Pw + 2 * 32 * 32 + slot) = P2;
            *(f32x4*)(Pw + 3 * 32 * 32 + slot) = P3;
        }
    }
    __syncthreads();
    if constexpr (DIAG == 4) tpx[3] = __builtin_amdgcn_s_memtime();
    {
        const int le = lane_now(), q8 = le & 7, tt = le >> 3;
        const int tl = wj * 8 + tt;
        const float* Pr = smem + tl * 32 + ((q8 ^ ((tl >> 1) & 7)) << 2);
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
        if (PERSIST || a.ksplit <= 1) bias4 = *(const f32x4*)(a.bias + ct * 32 + 4 * q8);      // (a split launch's bias is the combine kernel's)
        const int eH = PERSIST ? m.s[seg].H : a.H, eW = PERSIST ? m.s[seg].W : a.W;
        float* const eout = PERSIST ? m.s[seg].out : a.out + (size_t)ks * a.slab_stride;         // slab_stride = 0 unless split
        const int Ho = a.pool_out ? (eH + 1) >> 1 : eH, Wo = a.pool_out ? (eW + 1) >> 1 : eW;
        rsrc_t dst;
        if constexpr (!BIG) {
            dst = make_rsrc(eout + (size_t)img * Ho * Wo * a.cout, (unsigned)(Ho * Wo * a.cout) * 4u);
        } else {                                                           // the tile's first output row is the descriptor's origin
            const int orow0 = a.pool_out ? ty0 >> 1 : ty0;
            dst = make_rsrc(eout + ((size_t)img * Ho + orow0) * Wo * a.cout, 0x7ffffff0u);      // stores are masked per lane, never range-checked
        }
        const int oy = ty0 + 4 * (tl >> TCL), ox = tx0 + 2 * (tl & TCM);
        const int ry = BIG ? 4 * (tl >> TCL) : oy;                         // row counted from the descriptor's origin
        const int cbyte = (ct * 32 + 4 * q8) * 4;
        const bool colok0 = ox < eW, colok1 = ox + 1 < eW;
        const float relu_lo = a.relu ? 0.f : -__builtin_inff();
        f32x4 y[4][2], P[4][4];
#pragma unroll
        for (int ap = 0; ap < 4; ++ap)
#pragma unroll
            for (int j = 0; j < 4; ++j) P[ap][j] = *(const f32x4*)(Pr + (j * 4 + ap) * (32 * 32));
        if constexpr (PERSIST) {
            __syncthreads();                               // every wave has its P values: the LDS image is free again
            raw_store(Rs);                                 // the next tile's first halo stage (loaded during this tile)
            if constexpr (DIAG == 4) tpx[4] = __builtin_amdgcn_s_memtime();
        }
#pragma unroll
        for (int ap = 0; ap < 4; ++ap) {
            y[ap][0] = P[ap][0] + P[ap][1] + P[ap][2] + bias4;          // A2 = columns (1,1,1,0), (0,1,-1,-1)
            y[ap][1] = P[ap][1] - P[ap][2] - P[ap][3] + bias4;
            // ReLU as ONE v_med3_f32 per element, max(y, lo) = med3(y, lo, +inf) with lo = 0 or -inf (no ReLU): fmaxf costs two
            // instructions here (a canonicalising v_max before the real one) and the branch around it a dozen register copies
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[ap][0][e] = __builtin_amdgcn_fmed3f(y[ap][0][e], relu_lo, __builtin_inff());
                y[ap][1][e] = __builtin_amdgcn_fmed3f(y[ap][1][e], relu_lo, __builtin_inff());
            }
        }
        if (a.pool_out) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {               // two pool windows per tile: rows (0,1) and (2,3)
                const int r0 = oy + 2 * h2;
                f32x4 v = y[2 * h2][0];
                if (colok1) v = max4(v, y[2 * h2][1]);
                if (r0 + 1 < eH) {
                    v = max4(v, y[2 * h2 + 1][0]);
                    if (colok1) v = max4(v, y[2 * h2 + 1][1]);
                }
                const int off = BIG ? ((((ry >> 1) + h2) * Wo + (ox >> 1)) * a.cout) * 4 + cbyte
                                    : (((r0 >> 1) * Wo + (ox >> 1)) * a.cout) * 4 + cbyte;
                buf_store4(dst, v, (r0 < eH && colok0) ? off : 0x7fffffff);
            }
        } else {
#pragma unroll
            for (int ap = 0; ap < 4; ++ap) {
                const int off = (((ry + ap) * eW + ox) * a.cout) * 4 + cbyte;
                const bool rowok = oy + ap < eH;
                buf_store4(dst, y[ap][0], (rowok && colok0) ? off : 0x7fffffff);
                buf_store4(dst, y[ap][1], (rowok && colok1) ? off + a.cout * 4 : 0x7fffffff);
            }
        }
    }
    };

    // the main loop is instantiated once per transform column (WJ a compile-time constant inside): with a run-time column the compiler
    // turns the sign selection of the column combine into scalar branches around every group of adds, which splits the MFMA steps into
    // many basic blocks and leaves the transform outside the matrix pipe's shadow (measured on round 1's F(2x2,3x3) kernel: 1200-1300
    // cycles for a step with the transform against 630 without, ideal 512)
    auto main_loop = [&](auto WJC) {
        constexpr int WJ = decltype(WJC)::value;
        constexpr int cA = WJ == 0 ? 0 : 1, cB = WJ == 3 ? 3 : 2;
        asm volatile("; transform column %0" ::"n"(WJ));
        // patch rows 0-2, then 3-5: column combine f[a] = d[a][cA] +- d[a][cB]
        f32x4 dA[3], dB[3], f[6], o4, o5;
        auto xf_cols = [&](int a0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if constexpr (WJ == 1) f[a0 + k] = dA[k] + dB[k];
                else if constexpr (WJ == 2) f[a0 + k] = dB[k] - dA[k];
                else f[a0 + k] = dA[k] - dB[k];
            }
        };
        // B4^T down the rows: (4,0,-5,0,1,0) (0,-4,-4,1,1,0) (0,4,-4,-1,1,0) (0,-2,-1,2,1,0) (0,2,-1,-2,1,0) (0,4,0,-5,0,1)
        auto rows_012 = [&](f32x4 (&q)[6]) {
            q[0] = 4.f * f[0] + (f[4] - 5.f * f[2]);
            const f32x4 t1 = f[4] - 4.f * f[2], t2 = f[3] - 4.f * f[1];
            q[1] = t1 + t2;
            q[2] = t1 - t2;
        };
        auto rows_34 = [&](f32x4 (&q)[6]) {
            const f32x4 t3 = f[4] - f[2], d31 = f[3] - f[1];
            q[3] = t3 + 2.f * d31;
            o4 = t3 - 2.f * d31;
        };
        auto rows_5 = [&]() { o5 = 4.f * f[1] + (f[5] - 5.f * f[3]); };

        f32x4 aq[6], t1, t2, t3, d31;
        int xaddr = 0;          // this lane's patch origin in the halo image (floats): computed once, held in a register
        {
            const int li = lane & 31, lh = lane >> 5;
            xaddr = (4 * (li >> TCL)) * W4_PROW + (li & TCM) * W4_RSTR + 4 * lh;
            asm volatile("" : "+v"(xaddr));
        }
        auto xf_addr = [&]() {};
        // patch pixel (row a, column c of the 6 x 4 patch) relative to xaddr: plane c & 1, plane column + (c >> 1)
        auto poff = [=](int a, int c) { return (c & 1) * W4_PLANE + a * W4_PROW + (c >> 1) * W4_RSTR; };
        auto xf_read3 = [&](const float* rb, int a0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                dA[k] = *(const f32x4*)(rb + xaddr + poff(a0 + k, cA));
                dB[k] = *(const f32x4*)(rb + xaddr + poff(a0 + k, cB));
            }
        };
        // one patch read (read index i = 0..5 of rows a0..a0+2: i / 2 = row, i & 1 = column cA / cB)
        auto xf_read1 = [&](const float* rb, int a0, auto II) {
            constexpr int i = decltype(II)::value, k = i / 2;
            if constexpr (DIAG == 9) {      // timing-only: the same number of LDS instructions, a quarter of the bytes
                if constexpr ((i & 1) == 0) dA[k][0] = *(const float*)(rb + xaddr + poff(a0 + k, cA));
                else dB[k][0] = *(const float*)(rb + xaddr + poff(a0 + k, cB));
            } else if constexpr ((i & 1) == 0) dA[k] = *(const f32x4*)(rb + xaddr + poff(a0 + k, cA));
            else dB[k] = *(const f32x4*)(rb + xaddr + poff(a0 + k, cB));
        };

        // One chunk of 8 channels = 24 MFMAs in 24 scheduling regions of ONE MFMA plus its share of the chunk's other work
        // (issued right behind it, in its 64-cycle shadow).  Regions alternate between the accumulators of row positions 2d
        // and 2d+1 (d = region / 8): four back-to-back MFMAs on one accumulator are a dependent chain (tools/wino4_probe.py:
        // 412 cycles per 4 instead of 256 + issue overheads), and left to itself the scheduler builds exactly those chains.
        // XF: transform the next chunk's patches meanwhile; ST: the last third also writes the staged halo registers to LDS;
        // LD: the halo loads two stages ahead.  Weight ring: slot = row position, refilled as soon as its MFMAs have issued.
        auto chunk = [&](const float* nsrc, auto XFC, auto STC, auto LDC, int raw_soff, float* store_to, int wnext, auto PARC,
                         auto FIRSTC) {
            constexpr bool do_xf = decltype(XFC)::value, st = decltype(STC)::value, ld = decltype(LDC)::value;
            // FIRST: the tile's first chunk - every accumulator's first MFMA takes a zero C operand (an inline constant) instead of
            // a cleared register: no 96 v_mov per tile (vector instructions cost this kernel matrix-pipe time)
            constexpr bool first = decltype(FIRSTC)::value;
            constexpr int par = RING12 ? decltype(PARC)::value : 0;
            // Schedule: GRP == 1 (default) = one MFMA per scheduling region followed by its share of the chunk's other work;
            // GRP == 8 (-DW4_GROUP=8) = the eight MFMAs of a pair of row positions back to back, then their regions' work in one
            // burst.  A plain fp32 vector instruction takes matrix-pipe time on gfx950 (tools/probes/mfma_valu_probe.hip: a bare
            // stream of fp32 MFMAs with 5 v_fma_f32 each reaches 113-119 TFLOP/s alternating 1 : 5 and 126-129 in bursts of 4 : 20 or
            // more), but in this kernel the burst form measured 0.6 % SLOWER (3.64 vs 3.62 ms per config-2 step): what counts is
            // the NUMBER of vector instructions per MFMA (see `sa`, `xaddr`), not their placement.
            constexpr int GRP = W4_GROUP;
            auto mf = [&](auto HH) {
                constexpr int h = decltype(HH)::value, m = h / 2, b = h & 1, d = m / 4, sidx = m & 3, r = 2 * d + b;
                if constexpr ((DIAG == 1 || DIAG == 2) && (h % 8) == 0) {
                    const unsigned tnow = (unsigned)__builtin_amdgcn_s_memtime();
                    if (lane_now() == 0 && nlog < 96) steplog[WJ * 96 + nlog] = tnow;
                    ++nlog;
                }
                if constexpr (first && sidx == 0) {
                    constexpr f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[6 * par + r][sidx], aq[r][sidx], zero, 0, 0, 0);
                } else {
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[6 * par + r][sidx], aq[r][sidx], acc[r], 0, 0, 0);
                }
            };
            // region (1 MFMA : work) schedule, and the burst schedule's regions for the same pieces of work
            constexpr int H_COLS0 = GRP == 1 ? 10 : 8, H_READ1 = GRP == 1 ? 11 : 9, H_COLS3 = GRP == 1 ? 20 : 16, H_ROWS = GRP == 1 ? 21 : 17;
            auto work = [&](auto HH) {
                constexpr int h = decltype(HH)::value;          // 0..23: mini-step m = h / 2 (d = m / 4, s = m % 4), b = second MFMA
                constexpr int m = h / 2, b = h & 1;
                // ---- weight ring: (this chunk) slots 4, 5 at mini-steps 0, 1; (next chunk) slots 0, 1 at 4, 5; slots 2, 3 at 8, 9 ----
                if constexpr (DIAG == 13 || DIAG == 14) {
                    // timing-only: no weight loads in the main loop (the ring keeps its first fragments)
                } else if constexpr (RING12) {
                    if constexpr (b == 0 && m < 6) bq[6 * (1 - par) + m] = buf_load4(wsr, wvo, wnext + m * 1024);
                } else if constexpr (b == 0) {
                    if constexpr (m == 0) bq[4] = buf_load4(wsr, wvo, wso + 4 * 1024);
                    if constexpr (m == 1) bq[5] = buf_load4(wsr, wvo, wso + 5 * 1024);
                    if constexpr (m == 4) bq[0] = buf_load4(wsr, wvo, wnext);
                    if constexpr (m == 5) bq[1] = buf_load4(wsr, wvo, wnext + 1024);
                    if constexpr (m == 8) bq[2] = buf_load4(wsr, wvo, wnext + 2 * 1024);
                    if constexpr (m == 9) bq[3] = buf_load4(wsr, wvo, wnext + 3 * 1024);
                }
                // ---- halo loads two stages ahead, in the first two thirds of the chunk (regions 5, 7, 9, 11, 13, 15) ----
                if constexpr (ld && DIAG != 10 && DIAG != 14 && b == 1 && h >= 5 && h <= 15) {      // DIAG 10: timing-only, no halo loads (stale stores)
                    constexpr int k = (h - 5) / 2;
                    rawreg[k] = buf_load4(src, roff[k], tbase + raw_soff);
                }
                // ---- input transform of the next chunk ----
                if constexpr (do_xf && DIAG != 5) {
                    if constexpr (h == 0) xf_addr();
                    if constexpr (h >= 1 && h <= 6 && DIAG != 6) xf_read1(nsrc, 0, std::integral_constant<int, h - 1>{});
                    if constexpr (h == H_COLS0) xf_cols(0);
                    if constexpr (h >= H_READ1 && h <= H_READ1 + 5 && DIAG != 6) xf_read1(nsrc, 3, std::integral_constant<int, h - H_READ1>{});
                    if constexpr (h == H_COLS3) xf_cols(3);
                    if constexpr (h == H_ROWS) { aq[0] = 4.f * f[0] + (f[4] - 5.f * f[2]); t1 = f[4] - 4.f * f[2]; t2 = f[3] - 4.f * f[1]; }   // fragments 0..3: dead since region 15
                    if constexpr (h == H_ROWS + 1) { aq[1] = t1 + t2; aq[2] = t1 - t2; t3 = f[4] - f[2]; d31 = f[3] - f[1]; }
                    if constexpr (h == H_ROWS + 2) { aq[3] = t3 + 2.f * d31; o4 = t3 - 2.f * d31; o5 = 4.f * f[1] + (f[5] - 5.f * f[3]); }
                }
                // ---- halo store of the stage loaded one stage ago ----
                if constexpr (st && DIAG != 8 && DIAG != 11) {
                    if constexpr (h >= 17 && h <= 22) *(f32x4*)((char*)store_to + sa[h - 17]) = rawreg[h - 17];
                }
                if constexpr (st && DIAG == 11) {      // timing-only: halo loads kept alive without the LDS stores
                    if constexpr (h >= 17 && h <= 22) asm volatile("" ::"v"(rawreg[h - 17]));
                }
                if constexpr (do_xf && h == 23) { aq[4] = o4; aq[5] = o5; }
            };
            auto group = [&](auto GG) {
                constexpr int g0 = decltype(GG)::value * GRP;
                [&]<int... I>(std::integer_sequence<int, I...>) { (mf(std::integral_constant<int, g0 + I>{}), ...); }(std::make_integer_sequence<int, GRP>{});
                if constexpr (GRP > 1) __builtin_amdgcn_sched_barrier(0);
                [&]<int... I>(std::integer_sequence<int, I...>) { (work(std::integral_constant<int, g0 + I>{}), ...); }(std::make_integer_sequence<int, GRP>{});
                if constexpr (GRP == 1) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // the MFMA first, everything else behind it
                __builtin_amdgcn_sched_barrier(0);
            };
            [&]<int... G>(std::integer_sequence<int, G...>) { (group(std::integral_constant<int, G>{}), ...); }(std::make_integer_sequence<int, 24 / GRP>{});
            wso += 6144;
        };
        constexpr std::true_type T{};
        constexpr std::false_type F{};
        constexpr std::integral_constant<int, 0> P0{};
        constexpr std::integral_constant<int, 1> P1{};

        // ---- prologue --------------------------------------------------------------------------------------------------------------
        raw_load(0);
        if constexpr (PERSIST) {
            // STAGGER: the two workgroups of a CU start together and run tiles of equal length, so their epilogues (no MFMAs for
            // ~10 k cycles) tend to coincide and the matrix pipe idles (tools/wino4_persist_probe.py: 17 % of the time at cin = 64).
            // The second-dispatched half of the grid starts m.stagger x 1024 cycles late, once, while its first loads are in flight.
            if ((int)(blockIdx.x >> 3) >= (stride >> 1))
                for (int k = 0; k < m.stagger; ++k) __builtin_amdgcn_s_sleep(16);
        }
        raw_store(Rs);
        __syncthreads();
        raw_load(W4_KR * 4);                                // stages past the end read neighbouring data or zeros, never consumed
        xf_addr();
        xf_read3(Rs, 0); xf_cols(0);
        xf_read3(Rs, 3); xf_cols(3);
        rows_012(aq); rows_34(aq); rows_5();
        aq[4] = o4; aq[5] = o5;
        if constexpr (DIAG == 3) phase[1] = __builtin_amdgcn_s_memrealtime();

        if constexpr (!PERSIST) {
            // stages in pairs (even stage: reads buffer 0, fills buffer 1; odd stage: the reverse): every LDS address of the loop is a
            // register + an immediate (see the persistent form); the last pair is its first half when nst is even
            for (int s = 0; s + 1 < nst; s += 2) {
                chunk(Rs + 8, T, T, F, 0, Rs + W4_RBUF, wso + 6144, P0, F);        // channels 0..7; prepares 8..15; writes the next stage's halo
                if constexpr (DIAG != 7) __syncthreads();
                chunk(Rs + W4_RBUF, T, F, T, (s + 2) * W4_KR * 4, nullptr, wso + 6144, P1, F);      // channels 8..15; prepares the next stage; loads two stages ahead
                if (s + 2 < nst) {
                    chunk(Rs + W4_RBUF + 8, T, T, F, 0, Rs, wso + 6144, P0, F);
                    if constexpr (DIAG != 7) __syncthreads();
                    chunk(Rs, T, F, T, (s + 3) * W4_KR * 4, nullptr, wso + 6144, P1, F);
                }
            }
            chunk(Rs + ((nst - 1) & 1) * W4_RBUF + 8, T, F, F, 0, nullptr, wso + 6144, P0, F);
            chunk(Rs, F, F, F, 0, nullptr, wso + 6144, P1, F);
            if constexpr (DIAG == 3) phase[2] = __builtin_amdgcn_s_memrealtime();
        } else {
            int ntile = 0;
            // DIAG 4: shader-clock stamps of the persistent form: 0 tile start, 1 main-loop end, 2 after the barrier in front of
            // the epilogue, 3 P values written + barrier, 4 P values read + barrier + next halo stored, 5 outputs stored,
            // 6 accumulators cleared + barrier, 7 next tile's first transform done
            auto tile_stamp = [&]() {
                if constexpr (DIAG == 4) {
                    tpx[7] = __builtin_amdgcn_s_memtime();
                    if (a.dbg && lane_now() == 0 && ntile <= 32) {
                        unsigned long long* d = a.dbg + (((size_t)blockIdx.x * 4 + WJ) * 32 + (ntile - 1)) * 8;
#pragma unroll
                        for (int q = 0; q < 8; ++q) d[q] = tpx[q];
                        if (ntile == 1 && WJ == 0) a.dbg[(size_t)gridDim.x * 1024 + blockIdx.x] =
                            ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
                    }
                }
            };
            for (;;) {
                if constexpr (DIAG == 4) tpx[0] = __builtin_amdgcn_s_memtime();
                // the tile after this one (or this one again when the list is exhausted: its loads are then never consumed)
                const int nitem = item + stride < hi ? item + stride : item;
                const int nseg = seg_of(nitem);
                const int nli = nitem - m.s[nseg].item0, ngtx = m.s[nseg].tiles_x, ntiles = ngtx * m.s[nseg].tiles_y;
                int nct_, npt, nimg;
                decode(nli, ntiles, m.s[nseg].n, nct_, npt, nimg);
                const int ntx0 = (npt % ngtx) * TILE_W, nty0 = (npt / ngtx) * TILE_H;
                const int wso_next = ((nct_ * 4 + wj) * nch) * 6144;
                if (prio_mode & 1) {
                    if ((ntile + (int)((blockIdx.x >> 3) >= (stride >> 1))) & 1) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
                } else if (prio_mode & 2) {
                    __builtin_amdgcn_s_setprio(0);
                }
                ++ntile;
                // The halo loads run two stages ahead, and the staging offsets (six vector registers, the scalar tile base, the source
                // descriptor) switch to the next tile at ONE straight-line point: behind the stage loop, in front of the tile's last two
                // chunks, whose first then requests the next tile's first stage (it still has that chunk's tail, a whole chunk and the
                // epilogue's exchange to arrive).  Round 2 switched inside the loop ("if (s == nst - 3)", two more call sites for
                // nst = 2 / 3): the registers then were loop-carried values redefined on one path, and the compiler kept two copies of
                // them, 12 v_mov per stage (3 % of the stage's vector instructions - which cost matrix-pipe time here) besides three
                // inlined copies of the offset code.  The price: the stage that has nothing left to request re-requests the tile's
                // last stage (six loads per thread and tile that hit in L1 / L2 and are never stored).
                auto next_halo = [&]() {
                    halo_offsets(ntx0, nty0, m.s[nseg].H, m.s[nseg].W, m.s[nseg].Ws, nseg);
                    src = src_of(m.s[nseg].in, m.s[nseg].Hs, m.s[nseg].Ws, nimg, nty0);
                };
                {   // stage 0, peeled: its first chunk starts the tile's accumulators (FIRST)
                    chunk(Rs + 8, T, T, F, 0, Rs + W4_RBUF, wso + 6144, P0, T);
                    __syncthreads();
                    chunk(Rs + W4_RBUF, T, F, T, min(2, nst - 1) * W4_KR * 4, nullptr, wso + 6144, P1, F);
                }
                // stages 1 .. nst-2 in pairs (odd stage: reads buffer 1, fills buffer 0; even stage: the reverse), so that every LDS
                // address of the loop is a register + an immediate: with the buffer a run-time value the loop spent 8 vector
                // instructions per stage on `base + offset` (of 152; every one costs matrix-pipe time).  cin a multiple of 32 gives
                // whole pairs; otherwise the last pair is its first half.
                for (int s = 1; s + 1 < nst; s += 2) {
                    chunk(Rs + W4_RBUF + 8, T, T, F, 0, Rs, wso + 6144, P0, F);
                    __syncthreads();
                    chunk(Rs, T, F, T, min(s + 2, nst - 1) * W4_KR * 4, nullptr, wso + 6144, P1, F);
                    if (s + 2 < nst) {
                        chunk(Rs + 8, T, T, F, 0, Rs + W4_RBUF, wso + 6144, P0, F);
                        __syncthreads();
                        chunk(Rs + W4_RBUF, T, F, T, min(s + 3, nst - 1) * W4_KR * 4, nullptr, wso + 6144, P1, F);
                    }
                }
                next_halo();
                chunk(Rs + ((nst - 1) & 1) * W4_RBUF + 8, T, F, T, 0, nullptr, wso + 6144, P0, F);      // + the next tile's first halo stage
                chunk(Rs, F, F, F, 0, nullptr, wso_next, P1, F);      // the ring's look-ahead continues in the next tile's weights
                wso = wso_next;
                if constexpr (DIAG == 4) tpx[1] = __builtin_amdgcn_s_memtime();
                if (prio_mode & 2) __builtin_amdgcn_s_setprio(3);      // experiment: the MFMA-free phase of a tile at top priority
                __syncthreads();
                if constexpr (DIAG == 4) tpx[2] = __builtin_amdgcn_s_memtime();
                epilogue();                                     // of (ct, img, tx0, ty0); also writes the next tile's first halo stage
                if constexpr (DIAG == 4) tpx[5] = __builtin_amdgcn_s_memtime();
                if (item + stride >= hi) { tpx[6] = tpx[5]; tile_stamp(); break; }
                item += stride;
                ct = nct_; img = nimg; tx0 = ntx0; ty0 = nty0; seg = nseg;
                __syncthreads();                                // (the accumulators are not cleared: see FIRST)
                if constexpr (DIAG == 4) tpx[6] = __builtin_amdgcn_s_memtime();
                raw_load(W4_KR * 4);
                xf_addr();
                xf_read3(Rs, 0); xf_cols(0);
                xf_read3(Rs, 3); xf_cols(3);
                rows_012(aq); rows_34(aq); rows_5();
                aq[4] = o4; aq[5] = o5;
                tile_stamp();
            }
        }
    };
    if (wj == 0) main_loop(std::integral_constant<int, 0>{});
    else if (wj == 1) main_loop(std::integral_constant<int, 1>{});
    else if (wj == 2) main_loop(std::integral_constant<int, 2>{});
    else main_loop(std::integral_constant<int, 3>{});
    if constexpr (PERSIST) return;
    __syncthreads();
    if constexpr (DIAG == 1 || DIAG == 2) {
        if (a.dbg && blockIdx.x < 32) {
            unsigned* d32 = (unsigned*)a.dbg + blockIdx.x * 384;
            for (int i = tid; i < 384; i += 256) d32[i] = steplog[i];
        }
        __syncthreads();
    }
    epilogue();
    if constexpr (DIAG == 3) {
        phase[3] = __builtin_amdgcn_s_memrealtime();
        if (a.dbg && lane_now() == 0) {     // [block][wave][4 stamps], then [block] hardware ids
            unsigned long long* d = a.dbg + ((size_t)blockIdx.x * 4 + wj) * 4;
            d[0] = phase[0]; d[1] = phase[1]; d[2] = phase[2]; d[3] = phase[3];
            if (wj == 0) a.dbg[(size_t)gridDim.x * 16 + blockIdx.x] =
                ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        }
    }
}

// Second half of a cin-split layer: out = [pool](relu(slab 0 + slab 1 + ... + bias)), the slabs added in that fixed order (bitwise
// reproducible; no atomics).  One thread per output pixel and channel quad; slabs and out are NHWC, the slabs at the conv's
// own size H x W, out at the pooled size when `pool` (MaxPool2d(2, 2, ceil_mode=True): windows clipped at the border).
__global__ __launch_bounds__(256) void splitk_combine_kernel(const float* __restrict__ slabs, size_t slab_stride, int S,
                                                             const float* __restrict__ bias, int relu, int pool, float* __restrict__ out,
                                                             int n, int H, int W, int cout) {
    const int Ho = pool ? (H + 1) >> 1 : H, Wo = pool ? (W + 1) >> 1 : W;
    const int cq = cout >> 2;
    const size_t total = (size_t)n * Ho * Wo * cq;
    const float lo = relu ? 0.f : -__builtin_inff();
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        size_t t = idx;
        const int c4 = (int)(t % cq); t /= cq;
        const int xo = (int)(t % Wo); t /= Wo;
        const int yo = (int)(t % Ho);
        const int img = (int)(t / Ho);
        const f32x4 b4 = *(const f32x4*)(bias + 4 * c4);
        auto at = [&](int y, int x) {
            const float* p = slabs + (((size_t)img * H + y) * W + x) * cout + 4 * c4;
            f32x4 v = *(const f32x4*)p;
            for (int k = 1; k < S; ++k) v += *(const f32x4*)(p + (size_t)k * slab_stride);
            v += b4;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], lo, __builtin_inff());
            return v;
        };
        f32x4 r;
        if (pool) {
            const int y0 = 2 * yo, x0 = 2 * xo;
            r = at(y0, x0);
            if (x0 + 1 < W) r = max4(r, at(y0, x0 + 1));
            if (y0 + 1 < H) {
                r = max4(r, at(y0 + 1, x0));
                if (x0 + 1 < W) r = max4(r, at(y0 + 1, x0 + 1));
            }
        } else {
            r = at(yo, xo);
        }
        *(f32x4*)(out + idx * 4) = r;
    }
}

int launch_pack_wino4(const float* w, float* p, int cin, int cout, hipStream_t s) {
    if (cin % 8 || cout % 32) { set_error("pack_wino4: cin %% 8 or cout %% 32 != 0 (%d, %d)", cin, cout); return -1; }
    const size_t total = (size_t)cin * cout * 24;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_wino4_kernel, dim3(blocks), dim3(256), 0, s, w, p, cin, cout);
    return check_launch("pack_wino4");
}

// a per-image source or output tensor of 2 GiB or more needs the per-tile descriptors (BIG instantiations)
static bool wino4_big(const ConvArgs& a) {
    return (size_t)a.Hs * a.Ws * a.cin * 4 >= 0x7ffffff0ULL || (size_t)a.H * a.W * a.cout * 4 >= 0x7ffffff0ULL;
}

static int check_wino4_shape(const ConvArgs& a, int src_mode) {
    if (a.cin % W4_KR || a.cin < W4_KR) { set_error("conv3x3_wino4: cin %d not a multiple of 16", a.cin); return -1; }
    if (a.cout % 32) { set_error("conv3x3_wino4: cout %d not a multiple of 32", a.cout); return -1; }
    if (a.H < 2 || a.W < 2 || a.n < 1) { set_error("conv3x3_wino4: H, W must be >= 2, got %dx%d", a.H, a.W); return -1; }
    // 32-bit offsets address a band of at most 18 source / 16 output rows (the taller tile geometry) from a per-tile descriptor: the
    // image itself may be of any size
    if ((size_t)a.Ws * a.cin * 4 * 18 >= 0x7ffffff0ULL || (size_t)a.W * a.cout * 4 * 16 >= 0x7ffffff0ULL) {
        set_error("conv3x3_wino4: a band of eighteen %d-channel source rows or sixteen %d-channel output rows of width %d reaches 2 GiB",
                  a.cin, a.cout, a.W);
        return -1;
    }
    if ((size_t)a.cin * a.cout * 96 >= 0xffffffffULL) { set_error("conv3x3_wino4: packed weights must stay below 4 GiB"); return -1; }
    if (src_mode == SRC_DIRECT) {
        if (a.Hs != a.H || a.Ws != a.W) { set_error("conv3x3_wino4: direct mode needs Hs==H, Ws==W"); return -1; }
    } else if (src_mode == SRC_UP2X) {
        if (a.H != 2 * a.Hs || a.W != 2 * a.Ws) { set_error("conv3x3_wino4: up2x mode needs H==2Hs, W==2Ws"); return -1; }
    } else {
        set_error("conv3x3_wino4: unsupported src_mode %d", src_mode);
        return -1;
    }
    return 0;
}

// Channel tiles per group of the persistent walk: the largest divisor of cout / 32 whose transformed weights (cin x 32 G x 24
// floats) fit W4_L2_WEIGHT_BYTES of an XCD's 4 MB L2 (the rest is left to the halos and outputs streaming through).
constexpr size_t W4_L2_WEIGHT_BYTES = 3u << 20;
constexpr int W4_STAGGER = 0;       // start delay of the second half of the persistent grid (x 1024 cycles); see the kernel
static int walk_group(int cin, int cout) {
    const int nct = cout / 32;
    static const int force = tune_env("ADAIN_W4_CTG", 0);       // diagnostic build: 0 = automatic, -1 = all channel tiles, > 0 = that many
    if (force < 0) return nct;
    int g = nct;
    if (force > 0) g = force < nct ? force : nct;
    else
        while (g > 1 && (size_t)cin * 32 * g * 96 > W4_L2_WEIGHT_BYTES) --g;
    while (nct % g) --g;
    return g;
}

static long long persistent_grid() {
    const int cus = device_cu_count();
    if (cus <= 0) return 0;
    long long pgrid = 2LL * cus;
#ifdef ADAIN_DIAG
    // timing experiment (tools/probes/one_wave_probe.py): ADAIN_W4_WGS=1 leaves one workgroup per CU = ONE wave per SIMD, nothing
    // beside a wave's stalls and a tile's epilogue - what a split with more accumulators per wave would have to live with
    static const int wgs = tune_env("ADAIN_W4_WGS", 2);
    pgrid = (long long)(wgs == 1 ? 1 : 2) * cus;
#endif
    return pgrid - pgrid % 8;
}

// Tile geometry of a launch: 0 = 8 x 32 output pixels per workgroup tile (the tuned default), 1 = 16 x 16.  Both cost the same per
// tile, so the one that covers the launch's feature maps with fewer tiles wins; the default keeps ties and near-ties (1 %).
static int tile_h(int geo) { return geo ? 16 : 8; }
static int tile_w(int geo) { return geo ? 16 : 32; }
static long long geo_tiles(int geo, int n, int H, int W) {
    return (long long)n * ((H + tile_h(geo) - 1) / tile_h(geo)) * ((W + tile_w(geo) - 1) / tile_w(geo));
}
static int pick_geo(const ConvSeg* segs, int count) {
    static const int force = tune_env("ADAIN_W4_GEO", -1);      // diagnostic build: -1 = automatic, 0 / 1 = that geometry
    if (force == 0 || force == 1) return force;
    long long t0 = 0, t1 = 0;
    for (int i = 0; i < count; ++i) {
        t0 += geo_tiles(0, segs[i].n, segs[i].H, segs[i].W);
        t1 += geo_tiles(1, segs[i].n, segs[i].H, segs[i].W);
    }
    return t1 * 101 < t0 * 100 ? 1 : 0;
}

// Rounds of the persistent grid that ONE image of a layer is worth: tiles of the geometry its launch would pick x channel tiles over
// the resident workgroups.  The encoder / decoder schedules (api.hip) use it to decide which layers of a batch of large frames run
// frame by frame and which over the whole batch.
double wino4_rounds_per_image(int H, int W, int cout) {
    const long long pgrid = persistent_grid();
    if (pgrid <= 0 || H < 1 || W < 1) return 0.0;
    ConvSeg s{};
    s.n = 1; s.H = H; s.W = W;
    const int geo = pick_geo(&s, 1);
    return (double)geo_tiles(geo, 1, H, W) * (cout / 32) / (double)pgrid;
}

template <int MODE, bool PERSIST, bool BIG>
static void w4_launch_geo(int geo, dim3 grid, hipStream_t s, const ConvArgs& a, const ConvSegs& m, int items, int prio) {
    if constexpr (!BIG) {       // (per-tile descriptors - tensors of 2 GiB and more - exist for the default geometry only: see pick_geo's callers)
        if (geo) { hipLaunchKernelGGL((conv3x3_wino4_kernel<MODE, 0, PERSIST, BIG, 1>), grid, dim3(256), 0, s, a, m, items, prio); return; }
    }
    hipLaunchKernelGGL((conv3x3_wino4_kernel<MODE, 0, PERSIST, BIG, 0>), grid, dim3(256), 0, s, a, m, items, prio);
}
static void w4_launch(int src_mode, bool persist, bool big, int geo, dim3 grid, hipStream_t s, const ConvArgs& a, const ConvSegs& m,
                      int items, int prio) {
    const bool up = src_mode == SRC_UP2X;
#ifdef ADAIN_DIAG
    // timing-only ablations of the PERSISTENT form inside whole passes (wrong results by construction; bench.py --diag-lib with
    // ADAIN_W4_PDIAG = 8: no halo loads or stores in the main loop, 10: no halo loads, 11: no halo stores): what the halo staging
    // costs the step, i.e. the most a direct-to-LDS staging (`buffer_load ... lds`) could win
    static const int pdiag = tune_env("ADAIN_W4_PDIAG", 0);
    if (persist && !big && geo == 0 && (pdiag == 8 || pdiag == 10 || pdiag == 11)) {
#define W4_PDIAG_CASE(D) if (pdiag == D) { if (up) hipLaunchKernelGGL((conv3x3_wino4_kernel<SRC_UP2X, D, true>), grid, dim3(256), 0, s, a, m, items, prio); \
                                           else hipLaunchKernelGGL((conv3x3_wino4_kernel<SRC_DIRECT, D, true>), grid, dim3(256), 0, s, a, m, items, prio); return; }
        W4_PDIAG_CASE(8) W4_PDIAG_CASE(10) W4_PDIAG_CASE(11)
#undef W4_PDIAG_CASE
    }
#endif
    if (persist) {
        if (big) up ? w4_launch_geo<SRC_UP2X, true, true>(geo, grid, s, a, m, items, prio) : w4_launch_geo<SRC_DIRECT, true, true>(geo, grid, s, a, m, items, prio);
        else up ? w4_launch_geo<SRC_UP2X, true, false>(geo, grid, s, a, m, items, prio) : w4_launch_geo<SRC_DIRECT, true, false>(geo, grid, s, a, m, items, prio);
    } else {
        if (big) up ? w4_launch_geo<SRC_UP2X, false, true>(geo, grid, s, a, m, items, prio) : w4_launch_geo<SRC_DIRECT, false, true>(geo, grid, s, a, m, items, prio);
        else up ? w4_launch_geo<SRC_UP2X, false, false>(geo, grid, s, a, m, items, prio) : w4_launch_geo<SRC_DIRECT, false, false>(geo, grid, s, a, m, items, prio);
    }
}

// Cin split of a small launch.  A one-tile workgroup walks its whole cin loop alone (about 1.5 us per 16 channels, 8 us of launch,
// prologue and epilogue around it: tools/probes/small_frame_trace.py), and a workgroup's vector and matrix instructions share the
// SIMD's issue, so a second workgroup on a compute unit buys nothing: what a split can win is the compute units a launch leaves
// EMPTY.  S = the largest power of two with items x S <= compute units and at least 64 channels (4 stages) per workgroup.
static int wino4_ksplit(long long items, int cin) {
    const int cus = device_cu_count();
    int S = 1;
    while (S < 8 && items * (2 * S) <= cus && cin % (2 * S * W4_KR) == 0 && cin / (2 * S) >= 64) S *= 2;
    return S;
}
static long long wino4_items(int n, int H, int W, int cout, int* geo_out) {
    ConvSeg sg{};
    sg.n = n; sg.H = H; sg.W = W;
    const int geo = pick_geo(&sg, 1);
    if (geo_out) *geo_out = geo;
    return geo_tiles(geo, n, H, W) * (cout / 32);
}
size_t wino4_split_floats(int n, int H, int W, int cin, int cout) {
    if (n < 1 || H < 2 || W < 2 || cin % W4_KR || cout % 32) return 0;
    const int S = wino4_ksplit(wino4_items(n, H, W, cout, nullptr), cin);
    return S > 1 ? (size_t)S * n * H * W * cout : 0;
}

int launch_conv3x3_wino4(const ConvArgs& a0, int src_mode, hipStream_t s, SplitWs split) {
    ConvArgs a = a0;
    a.ksplit = 1; a.cin_sub = a.cin; a.slab_stride = 0;
    if (check_wino4_shape(a, src_mode)) return -1;
    ConvSegs m{};
    m.count = 1;
    m.s[0] = ConvSeg{a.in, a.out, a.n, a.H, a.W, a.Hs, a.Ws, 0, 0, 0};
    const bool big = wino4_big(a);
    const int geo = (a.dbg || big) ? 0 : pick_geo(m.s, 1);      // the stamp / timing-only and the >= 2 GiB builds exist for the default geometry
    a.tiles_x = (a.W + tile_w(geo) - 1) / tile_w(geo);
    a.tiles_y = (a.H + tile_h(geo) - 1) / tile_h(geo);
    m.s[0].tiles_x = a.tiles_x; m.s[0].tiles_y = a.tiles_y;
    const long long blocks = (long long)a.tiles_x * a.tiles_y * (a.cout / 32) * a.n;
    if (blocks <= 0 || blocks > 0x7fffffffLL) { set_error("conv3x3_wino4: bad grid %lld", blocks); return -1; }
    static const int order_env = tune_env("ADAIN_W4_ORDER", 1);
    a.xcd_order = order_env;
    const dim3 g((unsigned)blocks);
    // persistent form (default) whenever the launch has at least two tiles per resident workgroup; ADAIN_W4_PERSIST = largest
    // cin it is used for (0 = never: one tile per workgroup).  +2-3 % on most layer shapes, +1.1 % on the config-2 step.
    static const int persist_env = tune_env("ADAIN_W4_PERSIST", 1 << 20);
    static const int prio_env = tune_env("ADAIN_W4_PRIO", 1);
    const long long pgrid = persistent_grid();
    if (pgrid <= 0) { set_error("conv3x3_wino4: device query failed"); return -1; }
    const bool persist_ok = a.cin <= persist_env && a.cin >= 2 * W4_KR && pgrid >= 8 && blocks >= 2 * pgrid;
    const bool persist = !a.dbg && persist_ok;
    const int items = (int)blocks;
    m.ctg = walk_group(a.cin, a.cout);
    m.stagger = tune_env("ADAIN_W4_STAGGER", W4_STAGGER);
    if (persist) {
        w4_launch(src_mode, true, big, geo, dim3((unsigned)pgrid), s, a, m, items, prio_env);
        return check_launch("conv3x3_wino4");
    }
#ifdef ADAIN_DIAG
    // timing / stamp builds of the one-tile form (tools/ only: libadain_hip_diag.so); selected by ADAIN_W4_DIAG when a stamp
    // buffer is set
    static const int diag_env = tune_env("ADAIN_W4_DIAG", 0);
    if (a.dbg && diag_env == 4 && persist_ok && src_mode == SRC_DIRECT) {      // per-tile phase stamps of the persistent form
        hipLaunchKernelGGL((conv3x3_wino4_kernel<SRC_DIRECT, 4, true>), dim3((unsigned)pgrid), dim3(256), 0, s, a, m, items, prio_env);
        return check_launch("conv3x3_wino4(diag)");
    }
    if (a.dbg && src_mode == SRC_DIRECT) {
        switch (diag_env) {
#define W4_DIAG_CASE(D) case D: hipLaunchKernelGGL((conv3x3_wino4_kernel<SRC_DIRECT, D>), g, dim3(256), 0, s, a, m, items, 0); break;
            W4_DIAG_CASE(2) W4_DIAG_CASE(3) W4_DIAG_CASE(5) W4_DIAG_CASE(6) W4_DIAG_CASE(7) W4_DIAG_CASE(8) W4_DIAG_CASE(9)
            W4_DIAG_CASE(10) W4_DIAG_CASE(11) W4_DIAG_CASE(12) W4_DIAG_CASE(13) W4_DIAG_CASE(14)
#undef W4_DIAG_CASE
            default: hipLaunchKernelGGL((conv3x3_wino4_kernel<SRC_DIRECT, 1>), g, dim3(256), 0, s, a, m, items, 0);
        }
        return check_launch("conv3x3_wino4(diag)");
    }
#endif
    const int S = (split.slab && !big && !a.dbg) ? wino4_ksplit(blocks, a.cin) : 1;
    if (S > 1) {
        const size_t slab = (size_t)a.n * a.H * a.W * a.cout;
        if (split.floats < (size_t)S * slab) { set_error("conv3x3_wino4: split workspace too small (%zu < %zu floats)", split.floats, (size_t)S * slab); return -1; }
        ConvArgs p = a;                  // first half: S workgroups per (tile, channel tile), partial sums into the slabs
        p.ksplit = S; p.cin_sub = a.cin / S; p.slab_stride = slab;
        p.out = split.slab; p.relu = 0; p.pool_out = 0; p.xcd_order = 1;
        w4_launch(src_mode, false, false, geo, dim3((unsigned)(blocks * S)), s, p, m, items * S, 0);
        if (int r = check_launch("conv3x3_wino4(split)")) return r;
        const int Ho = a.pool_out ? (a.H + 1) / 2 : a.H, Wo = a.pool_out ? (a.W + 1) / 2 : a.W;
        const size_t total = (size_t)a.n * Ho * Wo * (a.cout / 4);
        const unsigned cb = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(splitk_combine_kernel, dim3(cb), dim3(256), 0, s, (const float*)split.slab, slab, S, a.bias, a.relu, a.pool_out, a.out,
                           a.n, a.H, a.W, a.cout);
        return check_launch("conv3x3_wino4(combine)");
    }
    w4_launch(src_mode, false, big, geo, g, s, a, m, items, 0);
    return check_launch("conv3x3_wino4");
}

int launch_conv3x3_wino4_multi(const ConvArgs& layer, const ConvSeg* segs, int count, int src_mode, hipStream_t s, const SplitWs* split) {
    if (count < 1 || count > MAX_CONV_SEGS) { set_error("conv3x3_wino4_multi: 1..%d segments, got %d", MAX_CONV_SEGS, count); return -1; }
    ConvSegs m{};
    m.count = count;
    m.ctg = walk_group(layer.cin, layer.cout);
    m.stagger = tune_env("ADAIN_W4_STAGGER", W4_STAGGER);
    long long total = 0;
    bool big = false;
    ConvArgs a = layer;
    for (int i = 0; i < count; ++i) {
        a.n = segs[i].n; a.H = segs[i].H; a.W = segs[i].W; a.Hs = segs[i].Hs; a.Ws = segs[i].Ws;
        big = big || wino4_big(a);
    }
    const int geo = big ? 0 : pick_geo(segs, count);             // one geometry per launch: the one with fewer tiles over all segments
    for (int i = 0; i < count; ++i) {
        a.in = segs[i].in; a.out = segs[i].out; a.n = segs[i].n;
        a.H = segs[i].H; a.W = segs[i].W; a.Hs = segs[i].Hs; a.Ws = segs[i].Ws;
        if (!a.in || !a.out) { set_error("conv3x3_wino4_multi: null pointer in segment %d", i); return -1; }
        if (check_wino4_shape(a, src_mode)) return -1;
        big = big || wino4_big(a);
        m.s[i] = segs[i];
        m.s[i].tiles_x = (a.W + tile_w(geo) - 1) / tile_w(geo);
        m.s[i].tiles_y = (a.H + tile_h(geo) - 1) / tile_h(geo);
        m.s[i].item0 = (int)total;
        total += (long long)m.s[i].tiles_x * m.s[i].tiles_y * (a.cout / 32) * a.n;
        if (total > 0x7fffffffLL) { set_error("conv3x3_wino4_multi: too many tiles"); return -1; }
    }
    static const int merge_env = tune_env("ADAIN_W4_MERGE", 1);
    static const int prio_env = tune_env("ADAIN_W4_PRIO", 1);
    const long long pgrid = persistent_grid();
    if (pgrid <= 0) { set_error("conv3x3_wino4: device query failed"); return -1; }
    if (count == 1 || !merge_env || a.cin < 2 * W4_KR || pgrid < 8 || total < 2 * pgrid) {
        // not enough work for a shared persistent list (or a single segment): one launch per segment
        for (int i = 0; i < count; ++i) {
            a.in = segs[i].in; a.out = segs[i].out; a.n = segs[i].n;
            a.H = segs[i].H; a.W = segs[i].W; a.Hs = segs[i].Hs; a.Ws = segs[i].Ws;
            if (int r = launch_conv3x3_wino4(a, src_mode, s, split ? split[i] : SplitWs{nullptr, 0})) return r;
        }
        return 0;
    }
    // the kernel reads its geometry from the segments; the ConvArgs copy carries the layer (weights, bias, cin, cout, flags)
    a.in = m.s[0].in; a.out = m.s[0].out; a.n = m.s[0].n; a.H = m.s[0].H; a.W = m.s[0].W; a.Hs = m.s[0].Hs; a.Ws = m.s[0].Ws;
    a.tiles_x = m.s[0].tiles_x; a.tiles_y = m.s[0].tiles_y;
    a.xcd_order = 1;
    a.dbg = nullptr;
    w4_launch(src_mode, true, big, geo, dim3((unsigned)pgrid), s, a, m, (int)total, prio_env);
    return check_launch("conv3x3_wino4(multi)");
}

}  // namespace adain
