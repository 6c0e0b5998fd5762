// extern "C" boundary (include/adain_hip.h) and the encoder / decoder layer schedules.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/adain_hip.h"
#ifdef ADAIN_DIAG
#include "../../include/adain_hip_diag.h"
#endif
#include "common.h"

namespace adain {

static thread_local char g_err[512] = "";
// launch schedule of the calling thread (adain_set_schedule): ADAIN_SCHEDULE_BATCH or ADAIN_SCHEDULE_LATENCY
static thread_local int g_schedule = ADAIN_SCHEDULE_BATCH;

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Layer tables.  Encoder = net.vgg[:31] (reference net.py:38-69): conv0+conv1_1 folded ("first"), then
// 8 generic 3x3 convs; the ceil-mode max-pools in front of conv2_1, conv3_1, conv4_1 are fused into those
// convs' gathers.  Decoder = net.decoder (net.py:6-36): 8 generic convs (the nearest-2x upsamples in front
// of the 2nd, 6th and 8th are fused into their gathers) + the 64->3 "last" conv.
// `pool` = the layer's output feeds a max-pool: the pool is fused into this layer's EPILOGUE (only the pooled
// tensor is written), so the next conv reads it directly.  (The consumer-side form, SRC_POOL2, stays available
// through adain_conv3x3.)
struct Layer { int cin, cout, src, pool; };
static const Layer ENC[8] = {{64, 64, SRC_DIRECT, 1},   {64, 128, SRC_DIRECT, 0},  {128, 128, SRC_DIRECT, 1}, {128, 256, SRC_DIRECT, 0},
                             {256, 256, SRC_DIRECT, 0}, {256, 256, SRC_DIRECT, 0}, {256, 256, SRC_DIRECT, 1}, {256, 512, SRC_DIRECT, 0}};
static const Layer DEC[8] = {{512, 256, SRC_DIRECT, 0}, {256, 256, SRC_UP2X, 0},   {256, 256, SRC_DIRECT, 0}, {256, 256, SRC_DIRECT, 0},
                             {256, 128, SRC_DIRECT, 0}, {128, 128, SRC_UP2X, 0},   {128, 64, SRC_DIRECT, 0},  {64, 64, SRC_UP2X, 0}};

static size_t align64(size_t x) { return (x + 63) & ~(size_t)63; }
constexpr size_t FIRST_W = 2 * 14 * 64, FIRST_B = 64, LAST_W = 8 * 64 * 4, LAST_B = 3;

// The generic 3x3 layers run - and the library only contains - the Winograd F(4,3) x F(2,3) kernels (csrc/conv_wino4.hip).  The direct
// implicit GEMM and the F(2x2,3x3) families of rounds 1-2 were A/B baselines in the diagnostic build until round 6 and are retired
// (git history; docs/HISTORY.md has their numbers).
static size_t form_floats(int cin, int cout) { return (size_t)cin * cout * 24; }

// packed layout: [first w][first b] then per generic layer [w in the form the schedules launch][b], every block 256-B
// aligned: 75 MB for the two networks in the F(4,3) x F(2,3) form (only that form is packed and kept).
struct Offsets { size_t w[8], b[8], first_b, last_w, last_b, total; };
static Offsets enc_offsets() {
    Offsets f{};
    size_t o = 0;
    o += align64(FIRST_W);
    f.first_b = o;
    o += align64(FIRST_B);
    for (int i = 0; i < 8; ++i) {
        f.w[i] = o;
        o += align64(form_floats(ENC[i].cin, ENC[i].cout));
        f.b[i] = o;
        o += align64(ENC[i].cout);
    }
    f.total = o;
    return f;
}
static Offsets dec_offsets() {
    Offsets f{};
    size_t o = 0;
    for (int i = 0; i < 8; ++i) {
        f.w[i] = o;
        o += align64(form_floats(DEC[i].cin, DEC[i].cout));
        f.b[i] = o;
        o += align64(DEC[i].cout);
    }
    f.last_w = o;
    o += align64(LAST_W);
    f.last_b = o;
    o += align64(LAST_B);
    f.total = o;
    return f;
}

static int pack_layer(const float* w, float* dst, int cin, int cout, hipStream_t s) { return launch_pack_wino4(w, dst, cin, cout, s); }

static int launch_layer(ConvArgs& a, const float* packed, const Offsets& f, int i, int src, hipStream_t s, SplitWs split = SplitWs{nullptr, 0}) {
    a.bias = packed + f.b[i];
    a.wpk = packed + f.w[i];
    return launch_conv3x3_wino4(a, src, s, split);
}

static int copy_bias(const float* src, float* dst, int n, hipStream_t s) {
    if (hipMemcpyAsync(dst, src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) {
        set_error("bias copy failed: %s", hipGetErrorString(hipGetLastError()));
        return ADAIN_ELAUNCH;
    }
    return 0;
}

static void record(void* const* ev, int i, hipStream_t s) {
    if (ev && ev[i]) (void)hipEventRecord((hipEvent_t)ev[i], s);
}

}  // namespace adain

using namespace adain;

#define RET_IF(x) do { int _r = (x); if (_r) return _r; } while (0)

extern "C" {

int adain_abi_version(void) { return ADAIN_ABI_VERSION; }
const char* adain_last_error(void) { return g_err; }

int adain_set_schedule(int schedule) {
    if (schedule != ADAIN_SCHEDULE_BATCH && schedule != ADAIN_SCHEDULE_LATENCY) { set_error("set_schedule: unknown schedule %d", schedule); return ADAIN_EINVAL; }
    const int prev = g_schedule;
    g_schedule = schedule;
    return prev;
}
int adain_get_schedule(void) { return g_schedule; }

size_t adain_encoder_packed_floats(void) { return enc_offsets().total; }
size_t adain_decoder_packed_floats(void) { return dec_offsets().total; }

int adain_encoder_pack(const float* const* w, const float* const* b, float* packed, adain_stream_t stream) {
    if (!w || !b || !packed) { set_error("encoder_pack: null pointer"); return ADAIN_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    const Offsets f = enc_offsets();
    RET_IF(launch_pack_conv_first(w[0], b[0], w[1], b[1], packed, packed + f.first_b, s));
    for (int i = 0; i < 8; ++i) {
        RET_IF(pack_layer(w[i + 2], packed + f.w[i], ENC[i].cin, ENC[i].cout, s));
        RET_IF(copy_bias(b[i + 2], packed + f.b[i], ENC[i].cout, s));
    }
    return 0;
}

int adain_decoder_pack(const float* const* w, const float* const* b, float* packed, adain_stream_t stream) {
    if (!w || !b || !packed) { set_error("decoder_pack: null pointer"); return ADAIN_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    const Offsets f = dec_offsets();
    for (int i = 0; i < 8; ++i) {
        RET_IF(pack_layer(w[i], packed + f.w[i], DEC[i].cin, DEC[i].cout, s));
        RET_IF(copy_bias(b[i], packed + f.b[i], DEC[i].cout, s));
    }
    RET_IF(launch_pack_conv_last(w[8], packed + f.last_w, s));
    RET_IF(copy_bias(b[8], packed + f.last_b, 3, s));
    return 0;
}

void adain_encoded_size(int h, int w, int* hc, int* wc) {
    for (int i = 0; i < 3; ++i) { h = (h + 1) / 2; w = (w + 1) / 2; }
    if (hc) *hc = h;
    if (wc) *wc = w;
}

// Ping-pong activation buffers of the encoder: A takes conv1_1's output and then every second layer, B the
// others; sizes are the maxima over the schedule (ceil-mode pooling makes odd sizes non-monotonic).
static void enc_buf_sizes(int n, int h, int w, size_t* a_floats, size_t* b_floats) {
    size_t mx[2] = {(size_t)n * h * w * 64, 0};
    int cur = 0, ch = h, cw = w;
    for (int i = 0; i < 7; ++i) {   // layer 7 writes the caller's feature buffer
        const int oh = ENC[i].pool ? (ch + 1) / 2 : ch, ow = ENC[i].pool ? (cw + 1) / 2 : cw;
        const size_t sz = (size_t)n * oh * ow * ENC[i].cout;
        cur ^= 1;
        if (sz > mx[cur]) mx[cur] = sz;
        ch = oh; cw = ow;
    }
    *a_floats = align64(mx[0]);
    *b_floats = align64(mx[1]);
}
static size_t enc_buf_a(int n, int h, int w) { size_t a, b; enc_buf_sizes(n, h, w, &a, &b); return a; }
static size_t enc_buf_b(int n, int h, int w) { size_t a, b; enc_buf_sizes(n, h, w, &a, &b); return b; }
// Third block of an encoder / decoder workspace: the partial-sum slabs of the layers that ADAIN_SCHEDULE_LATENCY would split along
// cin (conv_wino4.hip: only launches with fewer tiles than compute units - single frames of the 256 class; at most 8 MB).  Always
// part of the workspace, whatever the calling thread's schedule is when it asks for the size.
static size_t enc_slab_floats(int n, int h, int w) {
    size_t mx = 0;
    int ch = h, cw = w;
    for (int l = 0; l < 8; ++l) {
        const size_t f = wino4_split_floats(n, ch, cw, ENC[l].cin, ENC[l].cout);
        if (f > mx) mx = f;
        if (ENC[l].pool) { ch = (ch + 1) / 2; cw = (cw + 1) / 2; }
    }
    return align64(mx);
}
static size_t dec_slab_floats(int n, int hc, int wc) {
    size_t mx = 0;
    int ch = hc, cw = wc;
    for (int l = 0; l < 8; ++l) {
        if (DEC[l].src == SRC_UP2X) { ch *= 2; cw *= 2; }
        const size_t f = wino4_split_floats(n, ch, cw, DEC[l].cin, DEC[l].cout);
        if (f > mx) mx = f;
    }
    return align64(mx);
}
static SplitWs split_ws(float* slab, size_t floats) {
    return (g_schedule == ADAIN_SCHEDULE_LATENCY && floats) ? SplitWs{slab, floats} : SplitWs{nullptr, 0};
}

size_t adain_encode_workspace_bytes(int n, int h, int w) {
    if (n < 1 || h < 1 || w < 1) return 0;
    return (enc_buf_a(n, h, w) + enc_buf_b(n, h, w) + enc_slab_floats(n, h, w)) * sizeof(float);
}

size_t adain_encode_multi_workspace_bytes(int count, const int* n, const int* h, const int* w) {
    if (count < 1 || count > MAX_CONV_SEGS || !n || !h || !w) return 0;
    size_t total = 0;
    for (int i = 0; i < count; ++i) total += adain_encode_workspace_bytes(n[i], h[i], w[i]);
    return total;
}

// ---- batches of WIDE frames: which layers run frame by frame -----------------------------------------------------------------------
// Measured (profiles/r04_batching_per_layer.md, tools/probes/frame_major_ab.sh): for 1080p and 1200 x 1600 frames a launch over a
// whole batch costs the mid-network layers a few per cent against one launch per frame in frame-major order, while the small
// relu4-level layers GAIN from the batch (one 1080p frame is 2.1 rounds of work for the resident workgroups in dec1, run in 3).
// Same-box A/B of the two schedules at batch 2 / 4: W = 1920 +0.5 / +1.7 %, W = 1600 +0.4 / +1.3 %, W = 1536 +0.3 %, but W = 1408
// -1.2 / -0.4 %, W = 1280 -1.0 / -1.6 %, W = 960 -1.8 %: the sign follows the frame's WIDTH - a tile row of the 128-channel half-size
// layers (10 halo rows x W/2 pixels x 512 B) outgrows an XCD's 4 MB L2 at W = 1638 - and below it a batch only helps (fewer launch
// ramps, fuller last rounds).  So a batch of frames at least BIG_FRAME_WIDTH wide runs its BIG layers - one frame alone is worth at
// least BIG_LAYER_ROUNDS rounds of the persistent grid - frame by frame (everything from the image down to the last big layer for
// frame 0, then for frame 1, ...) and only the layers behind them once over all frames; narrower frames run every layer over the
// batch, as they always did.  Results do not depend on the split.
constexpr double BIG_LAYER_ROUNDS = 6.0;
constexpr int BIG_FRAME_WIDTH = 1600;
static double big_layer_rounds() {           // (diagnostic build: ADAIN_BIG_ROUNDS_X10, e.g. 10000000 = never frame by frame)
    static const int v = tune_env("ADAIN_BIG_ROUNDS_X10", (int)(BIG_LAYER_ROUNDS * 10));
    return v / 10.0;
}
static int big_frame_width() {               // (diagnostic build: ADAIN_BIG_FRAME_WIDTH, 0 = any width)
    static const int v = tune_env("ADAIN_BIG_FRAME_WIDTH", BIG_FRAME_WIDTH);
    return v;
}

// encoder: number of leading generic layers (0..7) run frame by frame = index after the LAST big layer.  conv4_1 (layer 7) is never
// part of the prefix: it writes the caller's feature tensors, not the ping-pong buffers the frame-major pass works in, so it always
// runs over the whole batch in the layer-major loop behind (a 1440 x 2560 frame's conv4_1 is 7.2 rounds and would count as big).
static int enc_frame_major_layers(int n, int h, int w) {
    if (n < 2 || w < big_frame_width()) return 0;
    int k = 0, ch = h, cw = w;
    for (int l = 0; l < 7; ++l) {
        if (wino4_rounds_per_image(ch, cw, ENC[l].cout) >= big_layer_rounds()) k = l + 1;
        if (ENC[l].pool) { ch = (ch + 1) / 2; cw = (cw + 1) / 2; }
    }
    return k;
}
// decoder: number of leading layers (0..8) run over the whole batch = index of the FIRST big layer (8: none is big)
static int dec_batched_layers(int n, int hc, int wc) {
    if (n < 2 || 8 * wc < big_frame_width()) return 8;
    int ch = hc, cw = wc;
    for (int l = 0; l < 8; ++l) {
        if (DEC[l].src == SRC_UP2X) { ch *= 2; cw *= 2; }
        if (wino4_rounds_per_image(ch, cw, DEC[l].cout) >= big_layer_rounds()) return l;
    }
    return 8;
}

// The frame-major pass shares the layer-major schedule's two ping-pong buffers (image i's tensor of a layer sits at i x that layer's
// per-image size, where the batched layers behind expect it), so an image processed later must never write over the LAST
// frame-major tensor of an image processed earlier.  With VGG's sizes it never does (later tensors of the same buffer are
// larger and start further out); this replays the offsets and says so for the case at hand - if not, the schedule stays layer-major.
static bool enc_frame_major_is_safe(int n, int h, int w, int k) {
    if (k < 1) return false;
    size_t size[8];                    // per-image floats of layer l's output
    int ch = h, cw = w;
    for (int l = 0; l < k; ++l) {
        if (ENC[l].pool) { ch = (ch + 1) / 2; cw = (cw + 1) / 2; }
        size[l] = (size_t)ch * cw * ENC[l].cout;
    }
    const int last_buf = (k - 1) & 1 ? 0 : 1;          // layer l writes buffer B (1) for even l, A (0) for odd l
    const size_t keep = size[k - 1];
    // image j writes layer l's tensor at j * size[l]; the kept tensors of images 0 .. j-1 occupy [0, j * keep) of last_buf: safe iff
    // every earlier tensor of that buffer is at least as large per image (j cancels)
    if (n < 2) return true;
    if (last_buf == 0 && (size_t)h * w * 64 < keep) return false;                 // conv_first -> buffer A
    for (int l = 0; l + 1 < k; ++l)
        if (((l & 1) ? 0 : 1) == last_buf && size[l] < keep) return false;
    return true;
}
// decoder, images processed LAST to FIRST: image j's writes into the buffer that holds the batched layers' output must stay behind
// the tensors of the images still waiting, [0, j x that tensor's size)
static bool dec_frame_major_is_safe(int n, int hc, int wc, int batched) {
    if (batched < 1 || batched > 7) return true;       // 0: the input is the caller's feature tensor; 8: nothing runs frame by frame
    size_t size[8];
    int ch = hc, cw = wc;
    for (int l = 0; l < 8; ++l) {
        if (DEC[l].src == SRC_UP2X) { ch *= 2; cw *= 2; }
        size[l] = (size_t)ch * cw * DEC[l].cout;
    }
    const int in_buf = (batched - 1) & 1;               // layer l writes buffer B (1) for odd l, A (0) for even l
    for (int l = batched; l < 8; ++l)
        if ((l & 1) == in_buf && size[l] < size[batched - 1]) return false;
    (void)n;
    return true;
}

// one batch of n images [n][h][w]: the frame-major part of the encoder (conv_first + the first k generic layers, image by image),
// leaving layer k - 1's output for all images contiguous in *cur_out (dims *ch_out x *cw_out) exactly where the layer-major
// schedule would have put it
static int encode_frame_major(const void* images, int u8, int n, int h, int w, int k, const float* packed, const Offsets& f, float* bufA,
                              float* bufB, const float** cur_out, int* ch_out, int* cw_out, hipStream_t s) {
    const size_t img_stride = (size_t)h * w * 3 * (u8 ? 1 : 4);      // bytes per source image
    int ch = h, cw = w;
    const float* cur = bufA;
    for (int i = 0; i < n; ++i) {
        RET_IF(launch_conv_first((const char*)images + (size_t)i * img_stride, u8, bufA + (size_t)i * h * w * 64, packed, packed + f.first_b, 1, h, w, s));
        cur = bufA;
        ch = h; cw = w;
        for (int l = 0; l < k; ++l) {
            float* out = cur == bufA ? bufB : bufA;
            const int oh = ENC[l].pool ? (ch + 1) / 2 : ch, ow = ENC[l].pool ? (cw + 1) / 2 : cw;
            ConvArgs a{};
            a.cin = ENC[l].cin; a.cout = ENC[l].cout; a.relu = 1; a.pool_out = ENC[l].pool;
            a.bias = packed + f.b[l]; a.wpk = packed + f.w[l];
            a.in = cur + (size_t)i * ch * cw * ENC[l].cin;
            a.out = out + (size_t)i * oh * ow * ENC[l].cout;
            a.n = 1; a.H = a.Hs = ch; a.W = a.Ws = cw;
            RET_IF(launch_conv3x3_wino4(a, ENC[l].src, s));
            cur = out; ch = oh; cw = ow;
        }
    }
    *cur_out = cur; *ch_out = ch; *cw_out = cw;
    return 0;
}

// images[i]: NCHW float, or (u8 != 0) HWC uint8 converted as ToTensor does inside the first layer's kernel
static int encode_impl(int count, const void* const* images, int u8, float* const* feats, const int* n, const int* h, const int* w,
                       const float* packed, void* workspace, size_t ws_bytes, void* const* ev, adain_stream_t stream) {
    if (count < 1 || count > MAX_CONV_SEGS) { set_error("encode: 1..%d image batches per call, got %d", MAX_CONV_SEGS, count); return ADAIN_EINVAL; }
    if (!images || !feats || !n || !h || !w || !packed || !workspace) { set_error("encode: null pointer"); return ADAIN_EINVAL; }
    for (int i = 0; i < count; ++i) {
        if (!images[i] || !feats[i]) { set_error("encode: null pointer"); return ADAIN_EINVAL; }
        if (n[i] < 1 || h[i] < 9 || w[i] < 9) {
            // relu4_1 must be at least 2x2 for the reflection pad in front of conv4_1 (torch raises there too)
            set_error("encode: image %dx%d too small (needs h, w >= 9)", h[i], w[i]);
            return ADAIN_EINVAL;
        }
    }
    if (ws_bytes < adain_encode_multi_workspace_bytes(count, n, h, w)) { set_error("encode: workspace too small"); return ADAIN_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    const Offsets f = enc_offsets();
    float* bufA[MAX_CONV_SEGS];
    float* bufB[MAX_CONV_SEGS];
    SplitWs split[MAX_CONV_SEGS];
    const float* cur[MAX_CONV_SEGS];
    int ch[MAX_CONV_SEGS], cw[MAX_CONV_SEGS];
    float* base = (float*)workspace;
    record(ev, 0, s);
    // a batch of large frames (one tensor pair, no per-layer events wanted): its big layers frame by frame, see above
    int frame_major = (count == 1 && !ev) ? enc_frame_major_layers(n[0], h[0], w[0]) : 0;
    if (frame_major && !enc_frame_major_is_safe(n[0], h[0], w[0], frame_major)) frame_major = 0;
    for (int i = 0; i < count; ++i) {
        bufA[i] = base;
        bufB[i] = base + enc_buf_a(n[i], h[i], w[i]);
        base = bufB[i] + enc_buf_b(n[i], h[i], w[i]);
        split[i] = split_ws(base, enc_slab_floats(n[i], h[i], w[i]));
        base += enc_slab_floats(n[i], h[i], w[i]);
        if (frame_major) {
            RET_IF(encode_frame_major(images[i], u8, n[i], h[i], w[i], frame_major, packed, f, bufA[i], bufB[i], &cur[i], &ch[i], &cw[i], s));
            continue;
        }
        RET_IF(launch_conv_first(images[i], u8, bufA[i], packed, packed + f.first_b, n[i], h[i], w[i], s));
        cur[i] = bufA[i];
        ch[i] = h[i]; cw[i] = w[i];
    }
    record(ev, 1, s);
    for (int l = frame_major; l < 8; ++l) {
        ConvArgs a{};
        a.cin = ENC[l].cin; a.cout = ENC[l].cout;
        a.relu = 1;
        a.pool_out = ENC[l].pool;
        ConvSeg segs[MAX_CONV_SEGS];
        for (int i = 0; i < count; ++i) {
            float* out = (l == 7) ? feats[i] : (cur[i] == bufA[i] ? bufB[i] : bufA[i]);
            segs[i] = ConvSeg{cur[i], out, n[i], ch[i], cw[i], ch[i], cw[i], 0, 0, 0};
        }
        // one launch for every batch: the persistent kernel's tile list runs over all of them (csrc/conv_wino4.hip, SEGMENTS)
        a.bias = packed + f.b[l];
        a.wpk = packed + f.w[l];
        RET_IF(launch_conv3x3_wino4_multi(a, segs, count, ENC[l].src, s, split));
        record(ev, l + 2, s);
        for (int i = 0; i < count; ++i) {
            cur[i] = segs[i].out;
            if (ENC[l].pool) { ch[i] = (ch[i] + 1) / 2; cw[i] = (cw[i] + 1) / 2; }
        }
    }
    return 0;
}

int adain_encode_multi(int count, const float* const* images, float* const* feats, const int* n, const int* h, const int* w,
                       const float* packed, void* workspace, size_t ws_bytes, void* const* ev, adain_stream_t stream) {
    return encode_impl(count, (const void* const*)images, 0, feats, n, h, w, packed, workspace, ws_bytes, ev, stream);
}

int adain_encode(const float* image, float* feat, const float* packed, void* workspace, size_t ws_bytes, int n, int h,
                 int w, void* const* ev, adain_stream_t stream) {
    if (!image || !feat) { set_error("encode: null pointer"); return ADAIN_EINVAL; }
    const void* img = image;
    return encode_impl(1, &img, 0, &feat, &n, &h, &w, packed, workspace, ws_bytes, ev, stream);
}

int adain_encode_u8(const uint8_t* image, float* feat, const float* packed, void* workspace, size_t ws_bytes, int n, int h,
                    int w, void* const* ev, adain_stream_t stream) {
    if (!image || !feat) { set_error("encode: null pointer"); return ADAIN_EINVAL; }
    const void* img = image;
    return encode_impl(1, &img, 1, &feat, &n, &h, &w, packed, workspace, ws_bytes, ev, stream);
}

int adain_encode_relu1_1(const void* image, int is_u8, float* relu1_1, const float* packed, int n, int h, int w, adain_stream_t stream) {
    if (!image || !relu1_1 || !packed) { set_error("encode_relu1_1: null pointer"); return ADAIN_EINVAL; }
    if (n < 1 || h < 2 || w < 2) { set_error("encode_relu1_1: image %dx%d too small (the reflection pad needs h, w >= 2)", h, w); return ADAIN_EINVAL; }
    const Offsets f = enc_offsets();
    return launch_conv_first(image, is_u8 ? 1 : 0, relu1_1, packed, packed + f.first_b, n, h, w, (hipStream_t)stream);
}

size_t adain_decode_workspace_bytes(int n, int hc, int wc) {
    if (n < 1 || hc < 1 || wc < 1) return 0;
    return (align64((size_t)n * hc * wc * 1024) + align64((size_t)n * hc * wc * 4096) + dec_slab_floats(n, hc, wc)) * sizeof(float);
}

// image_u8 != nullptr: the last layer writes save_image's uint8 HWC frames itself (adain_stylize_u8 without a mask: the bytes of
// adain_decode + adain_quantize_u8, one launch and the float image's round trip through HBM less); image is then unused
static int decode_impl(const float* feat, float* image, uint8_t* image_u8, const float* packed, void* workspace, size_t ws_bytes, int n, int hc,
                       int wc, void* const* ev, adain_stream_t stream) {
    if (!feat || (!image && !image_u8) || !packed || !workspace) { set_error("decode: null pointer"); return ADAIN_EINVAL; }
    if (n < 1 || hc < 2 || wc < 2) { set_error("decode: feature map %dx%d too small (needs >= 2x2)", hc, wc); return ADAIN_EINVAL; }
    if (ws_bytes < adain_decode_workspace_bytes(n, hc, wc)) { set_error("decode: workspace too small"); return ADAIN_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    const Offsets f = dec_offsets();
    // buffer A (1024*hc*wc floats per image) takes the outputs of layers 0,2,4,6; buffer B (4096*hc*wc) of 1,3,5,7
    float* bufA = (float*)workspace;
    float* bufB = bufA + align64((size_t)n * hc * wc * 1024);
    const SplitWs split = split_ws(bufB + align64((size_t)n * hc * wc * 4096), dec_slab_floats(n, hc, wc));
    const float* cur = feat;
    int ch = hc, cw = wc;
    record(ev, 0, s);
    // a batch of large frames: the leading small layers over the whole batch, then everything from the first big layer to the
    // image frame by frame (see enc_frame_major_layers)
    int batched = !ev ? dec_batched_layers(n, hc, wc) : 8;
    if (!dec_frame_major_is_safe(n, hc, wc, batched)) batched = 8;
    for (int i = 0; i < batched; ++i) {
        ConvArgs a{};
        a.in = cur;
        a.out = (i & 1) ? bufB : bufA;
        a.n = n;
        a.Hs = ch; a.Ws = cw;
        if (DEC[i].src == SRC_UP2X) { ch *= 2; cw *= 2; }
        a.H = ch; a.W = cw;
        a.cin = DEC[i].cin; a.cout = DEC[i].cout;
        a.relu = 1;
        RET_IF(launch_layer(a, packed, f, i, DEC[i].src, s, split));
        record(ev, i + 1, s);
        cur = a.out;
    }
    if (batched == 8) {
        RET_IF(launch_conv_last(cur, image, packed + f.last_w, packed + f.last_b, n, ch, cw, s, image_u8));
        record(ev, 9, s);
        return 0;
    }
    const int ch0 = ch, cw0 = cw;                                   // size of layer `batched`'s source, per image
    for (int img = n - 1; img >= 0; --img) {                        // last to first: see dec_frame_major_is_safe
        const float* c = cur;
        ch = ch0; cw = cw0;
        for (int i = batched; i < 8; ++i) {
            ConvArgs a{};
            a.cin = DEC[i].cin; a.cout = DEC[i].cout; a.relu = 1;
            a.bias = packed + f.b[i]; a.wpk = packed + f.w[i];
            a.in = c + (size_t)img * ch * cw * DEC[i].cin;
            a.n = 1;
            a.Hs = ch; a.Ws = cw;
            if (DEC[i].src == SRC_UP2X) { ch *= 2; cw *= 2; }
            a.H = ch; a.W = cw;
            float* out = (i & 1) ? bufB : bufA;
            a.out = out + (size_t)img * ch * cw * DEC[i].cout;
            RET_IF(launch_conv3x3_wino4(a, DEC[i].src, s));
            c = out;
        }
        RET_IF(launch_conv_last(c + (size_t)img * ch * cw * 64, image ? image + (size_t)img * 3 * ch * cw : nullptr, packed + f.last_w, packed + f.last_b,
                                1, ch, cw, s, image_u8 ? image_u8 + (size_t)img * 3 * ch * cw : nullptr));
    }
    return 0;
}

int adain_decode(const float* feat, float* image, const float* packed, void* workspace, size_t ws_bytes, int n, int hc,
                 int wc, void* const* ev, adain_stream_t stream) {
    if (!image) { set_error("decode: null pointer"); return ADAIN_EINVAL; }
    return decode_impl(feat, image, nullptr, packed, workspace, ws_bytes, n, hc, wc, ev, stream);
}

size_t adain_mean_std_workspace_bytes(int nhwc, int n, int c, int hw) { return mean_std_workspace_bytes(nhwc, n, c, hw); }

int adain_mean_std(const float* feat, int nhwc, int n, int c, int hw, float eps, float* mean, float* std_out, void* workspace,
                   size_t ws_bytes, adain_stream_t stream) {
    if (!feat || !mean || !std_out) { set_error("mean_std: null pointer"); return ADAIN_EINVAL; }
    return launch_mean_std(feat, nhwc, n, c, hw, eps, mean, std_out, workspace, ws_bytes, (hipStream_t)stream);
}

int adain_blend_alpha(const float* x, int nhwc, int n, int c, int hw, const float* c_mean, const float* c_std,
                      const float* s_mean, const float* s_std, int style_n, float alpha, float one_minus_alpha, float* out,
                      adain_stream_t stream) {
    if (!x || !c_mean || !c_std || !s_mean || !s_std || !out) { set_error("blend_alpha: null pointer"); return ADAIN_EINVAL; }
    return launch_adain_blend_ex(x, nhwc, n, c, hw, c_mean, c_std, s_mean, s_std, style_n, alpha, one_minus_alpha, nullptr, 1, out,
                                 (hipStream_t)stream);
}

int adain_blend_pmap(const float* x, int nhwc, int n, int c, int hw, const float* c_mean, const float* c_std,
                     const float* s_mean, const float* s_std, int style_n, const float* pmap, int pmap_n, float* out,
                     adain_stream_t stream) {
    if (!x || !c_mean || !c_std || !s_mean || !s_std || !out || !pmap) { set_error("blend_pmap: null pointer"); return ADAIN_EINVAL; }
    return launch_adain_blend_ex(x, nhwc, n, c, hw, c_mean, c_std, s_mean, s_std, style_n, 0.f, 0.f, pmap, pmap_n, out,
                                 (hipStream_t)stream);
}

size_t adain_strength_map_workspace_bytes(int hc, int wc) { return strength_map_workspace_bytes(hc, wc); }

int adain_strength_map(const float* depth, int h0, int w0, int hc, int wc, float offset, float prominence, float* pmap,
                       void* workspace, size_t ws_bytes, adain_stream_t stream) {
    if (!depth || !pmap) { set_error("strength_map: null pointer"); return ADAIN_EINVAL; }
    return launch_strength_map(depth, h0, w0, hc, wc, offset, prominence, pmap, workspace, ws_bytes, (hipStream_t)stream);
}

int adain_resize_bilinear(const float* in, float* out, int planes, int hi, int wi, int ho, int wo, adain_stream_t stream) {
    if (!in || !out) { set_error("resize_bilinear: null pointer"); return ADAIN_EINVAL; }
    return launch_resize_bilinear(in, out, planes, hi, wi, ho, wo, (hipStream_t)stream);
}
int adain_resize_nearest(const float* in, float* out, int planes, int hi, int wi, int ho, int wo, adain_stream_t stream) {
    if (!in || !out) { set_error("resize_nearest: null pointer"); return ADAIN_EINVAL; }
    return launch_resize_nearest(in, out, planes, hi, wi, ho, wo, (hipStream_t)stream);
}
int adain_mask_composite(const float* content, const float* stylized, const float* mask, int mask_c, int mask_n, float* out, int n,
                         int c, int hw, adain_stream_t stream) {
    if (!content || !stylized || !mask || !out) { set_error("mask_composite: null pointer"); return ADAIN_EINVAL; }
    return launch_mask_composite(content, stylized, mask, mask_c, mask_n, out, n, c, hw, (hipStream_t)stream);
}
int adain_quantize_u8(const float* in, uint8_t* out, int n, int c, int h, int w, adain_stream_t stream) {
    if (!in || !out) { set_error("quantize_u8: null pointer"); return ADAIN_EINVAL; }
    return launch_quantize_u8(in, out, n, c, h, w, (hipStream_t)stream);
}
int adain_u8_to_f32(const uint8_t* in, float* out, int n, int c, int h, int w, adain_stream_t stream) {
    if (!in || !out) { set_error("u8_to_f32: null pointer"); return ADAIN_EINVAL; }
    return launch_u8_to_f32(in, out, n, c, h, w, (hipStream_t)stream);
}
int adain_warp_blend_u8(const uint8_t* cur, const uint8_t* prev, const float* flow, uint8_t* out, int h, int w, int c, float alpha,
                        float one_minus_alpha, adain_stream_t stream) {
    if (!cur || !prev || !flow || !out) { set_error("warp_blend_u8: null pointer"); return ADAIN_EINVAL; }
    return launch_warp_blend_u8(cur, prev, flow, out, h, w, c, alpha, one_minus_alpha, (hipStream_t)stream);
}
int adain_resize_area_u8(const uint8_t* in, uint8_t* out, int n, int hi, int wi, int c, int ho, int wo, adain_stream_t stream) {
    if (!in || !out) { set_error("resize_area_u8: null pointer"); return ADAIN_EINVAL; }
    return launch_resize_area_u8(in, out, n, hi, wi, c, ho, wo, (hipStream_t)stream);
}
size_t adain_resize_pil_bilinear_u8_workspace_bytes(int hi, int wi, int ho, int wo) { return resize_pil_workspace_bytes(hi, wi, ho, wo); }

int adain_resize_pil_bilinear_u8(const uint8_t* in, int pixel_bytes, int n, int hi, int wi, uint8_t* out, int ho, int wo, int crop_y0, int crop_x0,
                                 int crop_h, int crop_w, void* workspace, size_t ws_bytes, adain_stream_t stream) {
    if (!in || !out) { set_error("resize_pil_bilinear_u8: null pointer"); return ADAIN_EINVAL; }
    return launch_resize_pil_bilinear_u8(in, pixel_bytes, n, hi, wi, out, ho, wo, crop_y0, crop_x0, crop_h, crop_w, workspace, ws_bytes,
                                         (hipStream_t)stream);
}

int adain_nhwc_to_nchw(const float* in, float* out, int n, int c, int hw, adain_stream_t stream) {
    if (!in || !out) { set_error("nhwc_to_nchw: null pointer"); return ADAIN_EINVAL; }
    return launch_nhwc_to_nchw(in, out, n, c, hw, (hipStream_t)stream);
}
int adain_nchw_to_nhwc(const float* in, float* out, int n, int c, int hw, adain_stream_t stream) {
    if (!in || !out) { set_error("nchw_to_nhwc: null pointer"); return ADAIN_EINVAL; }
    return launch_nchw_to_nhwc(in, out, n, c, hw, (hipStream_t)stream);
}

// ---- one sub-batch of the batch callers in one call ---------------------------------------------------------------------------
// Workspace of adain_stylize_u8, carved in this order (every block 256-byte aligned):
struct StylizePlan {
    int hc, wc, H8, W8;               // relu4_1 map; decoder output size (8hc x 8wc)
    int identity;                     // mask, decoder output and frame share one size: the fused composite + quantise tail
    int mask_only;                    // decoder output and frame share one size, the mask has another: the same tail sampling the mask
    size_t conv, feat, stat, stats_ws, pmap, pmap_ws, img, content, mask_f, mask_r, sty_r, comp, total;   // floats (conv / *_ws: bytes / 4)
};
static StylizePlan stylize_plan(int n, int h, int w, int use_depth, int mask_n, int mask_c, int mask_h, int mask_w, int mask_is_float) {
    StylizePlan p{};
    adain_encoded_size(h, w, &p.hc, &p.wc);
    p.H8 = 8 * p.hc; p.W8 = 8 * p.wc;
    const size_t enc = adain_encode_workspace_bytes(n, h, w), dec = adain_decode_workspace_bytes(n, p.hc, p.wc);
    p.conv = align64(((enc > dec ? enc : dec) + 3) / 4);
    p.feat = align64((size_t)n * p.hc * p.wc * 512);
    p.stat = align64((size_t)n * 512);
    p.stats_ws = align64((mean_std_workspace_bytes(1, n, 512, p.hc * p.wc) + 3) / 4);
    if (use_depth) {
        p.pmap = align64((size_t)n * p.hc * p.wc);
        p.pmap_ws = align64((strength_map_workspace_bytes(p.hc, p.wc) + 3) / 4);
    }
    p.img = align64((size_t)n * 3 * p.H8 * p.W8);
    if (mask_n > 0) {
        p.identity = mask_h == h && mask_w == w && p.H8 == h && p.W8 == w;
        p.mask_only = !p.identity && p.H8 == h && p.W8 == w;
        if (!p.identity && !p.mask_only) {
            p.content = align64((size_t)n * 3 * h * w);
            p.mask_f = mask_is_float ? 0 : align64((size_t)mask_n * mask_c * mask_h * mask_w);
            p.mask_r = (mask_h == h && mask_w == w) ? 0 : align64((size_t)mask_n * mask_c * h * w);
            p.sty_r = (p.H8 == h && p.W8 == w) ? 0 : align64((size_t)n * 3 * h * w);
            p.comp = align64((size_t)n * 3 * h * w);
        }
    }
    p.total = p.conv + 2 * p.feat + 2 * p.stat + p.stats_ws + p.pmap + p.pmap_ws + p.img + p.content + p.mask_f + p.mask_r + p.sty_r + p.comp;
    return p;
}

size_t adain_stylize_u8_workspace_bytes(int n, int h, int w, int use_depth, int mask_n, int mask_c, int mask_h, int mask_w, int mask_is_float) {
    if (n < 1 || h < 9 || w < 9) return 0;
    return stylize_plan(n, h, w, use_depth, mask_n, mask_c, mask_h, mask_w, mask_is_float).total * sizeof(float);
}

void adain_stylize_u8_out_size(int h, int w, int has_mask, int* oh, int* ow) {
    int hc, wc;
    adain_encoded_size(h, w, &hc, &wc);
    if (oh) *oh = has_mask ? h : 8 * hc;
    if (ow) *ow = has_mask ? w : 8 * wc;
}

int adain_stylize_u8(const uint8_t* frames, int n, int h, int w, const float* enc_packed, const float* dec_packed, const float* s_mean,
                     const float* s_std, float alpha, float one_minus_alpha, const float* const* depth_maps, const int* depth_h, const int* depth_w,
                     float depth_offset, float depth_prominence, const void* mask, int mask_is_float, int mask_n, int mask_c, int mask_h,
                     int mask_w, uint8_t* out_u8, void* workspace, size_t ws_bytes, adain_stream_t stream) {
    if (!frames || !enc_packed || !dec_packed || !s_mean || !s_std || !out_u8 || !workspace) { set_error("stylize_u8: null pointer"); return ADAIN_EINVAL; }
    if (n < 1 || h < 9 || w < 9) { set_error("stylize_u8: frames %dx%d too small (needs h, w >= 9)", h, w); return ADAIN_EINVAL; }
    if (!depth_maps && !(alpha >= 0.f && alpha <= 1.f)) { set_error("stylize_u8: alpha %g outside [0, 1]", alpha); return ADAIN_EINVAL; }   // test.py:75
    if (depth_maps && (!depth_h || !depth_w)) { set_error("stylize_u8: depth maps without their sizes"); return ADAIN_EINVAL; }
    if (depth_maps && !(depth_offset >= 0.f && depth_offset <= 1.f)) { set_error("stylize_u8: offset %g outside [0, 1]", depth_offset); return ADAIN_EINVAL; }   // test.py:56
    if (depth_maps)         // every argument is checked before the first launch: an error return leaves nothing queued on the stream
        for (int i = 0; i < n; ++i) {
            if (!depth_maps[i]) { set_error("stylize_u8: depth map %d is null", i); return ADAIN_EINVAL; }
            if (depth_h[i] < 1 || depth_w[i] < 1) { set_error("stylize_u8: depth map %d is %dx%d", i, depth_h[i], depth_w[i]); return ADAIN_EINVAL; }
        }
    if (!mask) mask_n = 0;
    if (mask && ((mask_n != 1 && mask_n != n) || (mask_c != 1 && mask_c != 3) || mask_h < 1 || mask_w < 1)) {
        set_error("stylize_u8: mask [%d][%d][%d][%d] does not fit %d RGB frames", mask_n, mask_c, mask_h, mask_w, n);
        return ADAIN_EINVAL;
    }
    const StylizePlan p = stylize_plan(n, h, w, depth_maps != nullptr, mask_n, mask_c, mask_h, mask_w, mask_is_float);
    if (ws_bytes < p.total * sizeof(float)) { set_error("stylize_u8: workspace too small (%zu < %zu bytes)", ws_bytes, p.total * sizeof(float)); return ADAIN_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    float* at = (float*)workspace;
    auto take = [&at](size_t floats) { float* r = at; at += floats; return floats ? r : (float*)nullptr; };
    float* conv = take(p.conv);
    float* f = take(p.feat);
    float* g = take(p.feat);
    float* c_mean = take(p.stat);
    float* c_std = take(p.stat);
    float* stats_ws = take(p.stats_ws);
    float* pmap = take(p.pmap);
    float* pmap_ws = take(p.pmap_ws);
    float* img = take(p.img);
    float* content_f = take(p.content);
    float* mask_f = take(p.mask_f);
    float* mask_r = take(p.mask_r);
    float* sty_r = take(p.sty_r);
    float* comp = take(p.comp);
    const int hw_c = p.hc * p.wc;

    // vgg(content) with ToTensor inside the first layer (test.py:203-204, :57 / :76), calc_mean_std(content_f) (function.py:4-12)
    RET_IF(adain_encode_u8(frames, f, enc_packed, conv, p.conv * sizeof(float), n, h, w, nullptr, stream));
    RET_IF(launch_mean_std(f, 1, n, 512, hw_c, 1e-5f, c_mean, c_std, stats_ws, p.stats_ws * sizeof(float), s));
    if (depth_maps) {       // compute_stylization_strength_map per frame, then AdaIN * (1 - P) + content_f * P (test.py:66-70)
        for (int i = 0; i < n; ++i) {
            RET_IF(launch_strength_map(depth_maps[i], depth_h[i], depth_w[i], p.hc, p.wc, depth_offset, depth_prominence, pmap + (size_t)i * hw_c,
                                       pmap_ws, p.pmap_ws * sizeof(float), s));
        }
        RET_IF(launch_adain_blend_ex(f, 1, n, 512, hw_c, c_mean, c_std, s_mean, s_std, 1, 0.f, 0.f, pmap, n, g, s));
    } else {                // AdaIN * alpha + content_f * (1 - alpha) (test.py:79-80)
        RET_IF(launch_adain_blend_ex(f, 1, n, 512, hw_c, c_mean, c_std, s_mean, s_std, 1, alpha, one_minus_alpha, nullptr, 1, g, s));
    }
    if (mask_n == 0 && ((uintptr_t)out_u8 & 3) == 0)        // decoder with save_image's quantiser inside its last layer (test.py:71 / :81, :243-244): the finished uint8 frames
        return decode_impl(g, nullptr, out_u8, dec_packed, conv, p.conv * sizeof(float), n, p.hc, p.wc, nullptr, stream);
    RET_IF(adain_decode(g, img, dec_packed, conv, p.conv * sizeof(float), n, p.hc, p.wc, nullptr, stream));     // test.py:71 / :81
    if (mask_n == 0) return launch_quantize_u8(img, out_u8, n, 3, p.H8, p.W8, s);                                // test.py:243-244 (unaligned output)
    if (p.identity)         // both F.interpolate calls of test.py:227-234 are identities: composite + quantise in one pass
        return launch_composite_quantize_u8(frames, img, mask, mask_is_float, mask_c, mask_n, out_u8, n, h * w, s);
    if (p.mask_only)        // only the mask needs its nearest resize: an index map, sampled in place by the same fused tail
        return launch_composite_quantize_u8_nearest(frames, img, mask, mask_is_float, mask_c, mask_n, mask_h, mask_w, out_u8, n, h, w, s);
    // the general composite (test.py:222-236): mask.float() -> nearest to the frame size; output -> bilinear to the frame size
    RET_IF(launch_u8_to_f32(frames, content_f, n, 3, h, w, s));
    const float* m = (const float*)mask;
    if (!mask_is_float) {
        RET_IF(launch_mask_to_f32((const uint8_t*)mask, mask_f, (size_t)mask_n * mask_c * mask_h * mask_w, s));
        m = mask_f;
    }
    if (mask_r) {
        RET_IF(launch_resize_nearest(m, mask_r, mask_n * mask_c, mask_h, mask_w, h, w, s));
        m = mask_r;
    }
    const float* sty = img;
    if (sty_r) {
        RET_IF(launch_resize_bilinear(img, sty_r, n * 3, p.H8, p.W8, h, w, s));
        sty = sty_r;
    }
    RET_IF(launch_mask_composite(content_f, sty, m, mask_c, mask_n, comp, n, 3, h * w, s));
    return launch_quantize_u8(comp, out_u8, n, 3, h, w, s);
}

static unsigned long long* g_conv_dbg = nullptr;     // stamp buffer of the diagnostic kernels; always null in the product library

size_t adain_conv3x3_wino4_packed_floats(int cin, int cout) { return (size_t)cin * cout * 24; }

int adain_conv3x3_wino4_pack(const float* w, float* packed, int cin, int cout, adain_stream_t stream) {
    if (!w || !packed) { set_error("conv3x3_wino4_pack: null pointer"); return ADAIN_EINVAL; }
    return launch_pack_wino4(w, packed, cin, cout, (hipStream_t)stream);
}

int adain_conv3x3_wino(const float* in, float* out, const float* packed_w, const float* bias, int n, int h, int w, int hs, int ws,
                       int cin, int cout, int src_mode, int relu, int pool_out, int mh, adain_stream_t stream) {
    if (!in || !out || !packed_w || !bias) { set_error("conv3x3_wino: null pointer"); return ADAIN_EINVAL; }
    ConvArgs a{};
    a.in = in; a.out = out; a.wpk = packed_w; a.bias = bias;
    a.n = n; a.H = h; a.W = w; a.Hs = hs; a.Ws = ws; a.cin = cin; a.cout = cout; a.relu = relu; a.pool_out = pool_out ? 1 : 0;
    if (mh != 5) {      // F(4,3) x F(2,3), weights packed by adain_conv3x3_wino4_pack: the one form this library holds
        set_error("conv3x3_wino: form %d is retired (rounds 1-2: F(2x2,3x3) forms 1-4); this library runs form 5, F(4,3) x F(2,3)", mh);
        return ADAIN_EINVAL;
    }
    a.dbg = g_conv_dbg;     // diagnostic builds only (tools/wino4_probe.py sets it)
    return launch_conv3x3_wino4(a, src_mode, (hipStream_t)stream);
}

size_t adain_conv3x3_wino4_split_workspace_bytes(int n, int h, int w, int cin, int cout) {
    return wino4_split_floats(n, h, w, cin, cout) * sizeof(float);
}

int adain_conv3x3_wino4_split(const float* in, float* out, const float* packed_w, const float* bias, int n, int h, int w, int hs, int ws,
                              int cin, int cout, int src_mode, int relu, int pool_out, void* workspace, size_t ws_bytes, adain_stream_t stream) {
    if (!in || !out || !packed_w || !bias) { set_error("conv3x3_wino4_split: null pointer"); return ADAIN_EINVAL; }
    ConvArgs a{};
    a.in = in; a.out = out; a.wpk = packed_w; a.bias = bias;
    a.n = n; a.H = h; a.W = w; a.Hs = hs; a.Ws = ws; a.cin = cin; a.cout = cout; a.relu = relu; a.pool_out = pool_out ? 1 : 0;
    return launch_conv3x3_wino4(a, src_mode, (hipStream_t)stream, SplitWs{(float*)workspace, workspace ? ws_bytes / sizeof(float) : 0});
}

#ifdef ADAIN_DIAG
/* ---- diagnostic library only (include/adain_hip_diag.h) -------------------------------------------------------------------- */
int adain_debug_set_conv_stamp_buffer(void* p) { g_conv_dbg = (unsigned long long*)p; return 0; }

#endif

}  // extern "C"
