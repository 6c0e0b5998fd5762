/* Plain-C caller of the AdaIN C ABI (include/adain_hip.h): no Python, no torch.
 *
 *   cc -O2 -I include -I /opt/rocm/include examples/c_abi_smoke.c -o c_abi_smoke \
 *      -L applied-image-processing_amd -ladain_hip -L /opt/rocm/lib -lamdhip64 -lm
 *   LD_LIBRARY_PATH=applied-image-processing_amd ./c_abi_smoke [H W [Hs Ws]]
 *
 * Fills the reference architecture with an integer-hash weight pattern (mirrored bit for bit by
 * tests/test_gpu_configs.py::test_c_abi_from_plain_c), runs style_transfer_simple
 * (encode content, encode style, channel statistics, AdaIN + alpha blend, decode, uint8 quantise)
 * entirely through the C entry points and prints a checksum of the uint8 image.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "adain_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define AD(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "adain error %d: %s (%s:%d)\n", r_, adain_last_error(), __FILE__, __LINE__); exit(3); } } while (0)

/* u in [-0.5, 0.5): 24 bits of a multiplicative hash of (tag, i) */
static float hashf(uint32_t tag, uint32_t i) {
    uint32_t x = (tag * 0x9E3779B1u) ^ (i * 0x85EBCA77u);
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
    return (float)(x >> 8) * (1.0f / 16777216.0f) - 0.5f;
}

static float* dev_filled(uint32_t tag, size_t n, float scale, float offset) {
    float* h = (float*)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; ++i) h[i] = hashf(tag, (uint32_t)i) * scale + offset;
    float* d;
    CK(hipMalloc((void**)&d, n * sizeof(float)));
    CK(hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice));
    free(h);
    return d;
}

int main(int argc, char** argv) {
    const int H = argc > 2 ? atoi(argv[1]) : 72, W = argc > 2 ? atoi(argv[2]) : 104;
    const int Hs = argc > 4 ? atoi(argv[3]) : 64, Ws = argc > 4 ? atoi(argv[4]) : 80;
    /* reference architecture: encoder convs (cin, cout, k), decoder convs */
    static const int enc[10][3] = {{3, 3, 1}, {3, 64, 3}, {64, 64, 3}, {64, 128, 3}, {128, 128, 3}, {128, 256, 3},
                                   {256, 256, 3}, {256, 256, 3}, {256, 256, 3}, {256, 512, 3}};
    static const int dec[9][3] = {{512, 256, 3}, {256, 256, 3}, {256, 256, 3}, {256, 256, 3}, {256, 128, 3},
                                  {128, 128, 3}, {128, 64, 3}, {64, 64, 3}, {64, 3, 3}};
    const float *ew[10], *eb[10], *dw[9], *db[9];
    for (int i = 0; i < 10; ++i) {
        const size_t n = (size_t)enc[i][1] * enc[i][0] * enc[i][2] * enc[i][2];
        const float bound = sqrtf(6.0f / (float)(enc[i][0] * enc[i][2] * enc[i][2]));
        ew[i] = dev_filled(100 + i, n, 2.0f * bound, 0.f);
        eb[i] = dev_filled(200 + i, enc[i][1], 0.1f, 0.f);
    }
    for (int i = 0; i < 9; ++i) {
        const size_t n = (size_t)dec[i][1] * dec[i][0] * 9;
        const float bound = sqrtf(6.0f / (float)(dec[i][0] * 9));
        dw[i] = dev_filled(300 + i, n, 2.0f * bound, 0.f);
        db[i] = dev_filled(400 + i, dec[i][1], 0.1f, 0.f);
    }
    hipStream_t s;
    CK(hipStreamCreate(&s));
    float *pe, *pd;
    CK(hipMalloc((void**)&pe, adain_encoder_packed_floats() * sizeof(float)));
    CK(hipMalloc((void**)&pd, adain_decoder_packed_floats() * sizeof(float)));
    AD(adain_encoder_pack(ew, eb, pe, s));
    AD(adain_decoder_pack(dw, db, pd, s));

    float* content = dev_filled(1, (size_t)3 * H * W, 1.0f, 0.5f);      /* [1,3,H,W] in [0,1) */
    float* style = dev_filled(2, (size_t)3 * Hs * Ws, 1.0f, 0.5f);
    int hc, wc, hsc, wsc;
    adain_encoded_size(H, W, &hc, &wc);
    adain_encoded_size(Hs, Ws, &hsc, &wsc);
    size_t wsb = adain_encode_workspace_bytes(1, H, W), t;
    if ((t = adain_encode_workspace_bytes(1, Hs, Ws)) > wsb) wsb = t;
    if ((t = adain_decode_workspace_bytes(1, hc, wc)) > wsb) wsb = t;
    if ((t = adain_mean_std_workspace_bytes(1, 1, 512, hc * wc)) > wsb) wsb = t;
    if ((t = adain_mean_std_workspace_bytes(1, 1, 512, hsc * wsc)) > wsb) wsb = t;
    void* ws;
    CK(hipMalloc(&ws, wsb));
    float *cf, *sf, *feat, *stats, *out;
    uint8_t* u8;
    CK(hipMalloc((void**)&cf, (size_t)hc * wc * 512 * sizeof(float)));
    CK(hipMalloc((void**)&sf, (size_t)hsc * wsc * 512 * sizeof(float)));
    CK(hipMalloc((void**)&feat, (size_t)hc * wc * 512 * sizeof(float)));
    CK(hipMalloc((void**)&stats, 4 * 512 * sizeof(float)));
    CK(hipMalloc((void**)&out, (size_t)3 * 64 * hc * wc * sizeof(float)));
    CK(hipMalloc((void**)&u8, (size_t)3 * 64 * hc * wc));

    AD(adain_encode(content, cf, pe, ws, wsb, 1, H, W, NULL, s));
    AD(adain_mean_std(cf, 1, 1, 512, hc * wc, 1e-5f, stats, stats + 512, ws, wsb, s));
    AD(adain_encode(style, sf, pe, ws, wsb, 1, Hs, Ws, NULL, s));
    AD(adain_mean_std(sf, 1, 1, 512, hsc * wsc, 1e-5f, stats + 1024, stats + 1536, ws, wsb, s));
    AD(adain_blend_alpha(cf, 1, 1, 512, hc * wc, stats, stats + 512, stats + 1024, stats + 1536, 1, 0.5f, 0.5f, feat, s));
    AD(adain_decode(feat, out, pd, ws, wsb, 1, hc, wc, NULL, s));
    AD(adain_quantize_u8(out, u8, 1, 3, 8 * hc, 8 * wc, s));
    CK(hipStreamSynchronize(s));

    const size_t n8 = (size_t)3 * 64 * hc * wc;
    uint8_t* h8 = (uint8_t*)malloc(n8);
    CK(hipMemcpy(h8, u8, n8, hipMemcpyDeviceToHost));
    uint64_t sum = 0, fnv = 1469598103934665603ull;
    for (size_t i = 0; i < n8; ++i) { sum += h8[i]; fnv = (fnv ^ h8[i]) * 1099511628211ull; }
    printf("c_abi_smoke: abi %d, %dx%d content, %dx%d style -> %dx%d image, sum %llu, fnv %016llx\n", adain_abi_version(), H, W,
           Hs, Ws, 8 * hc, 8 * wc, (unsigned long long)sum, (unsigned long long)fnv);

    /* the same frame as a DECODED uint8 frame with a mask (byte > 96), through the one-call entry point: what a video / guide-view
     * loop does per frame once the style statistics exist (stats + 1024 / + 1536 above) */
    const size_t px = (size_t)H * W;
    uint8_t* hf = (uint8_t*)malloc(px * 3);
    uint8_t* hm = (uint8_t*)malloc(px * 3);
    for (size_t p = 0; p < px; ++p)
        for (int c = 0; c < 3; ++c) {
            hf[p * 3 + c] = (uint8_t)((hashf(1, (uint32_t)(c * px + p)) + 0.5f) * 255.0f);     /* HWC bytes of the float frame above */
            hm[c * px + p] = hf[p * 3 + c] > 96;                                              /* mask [1][3][H][W] */
        }
    uint8_t *df, *dm, *d8;
    CK(hipMalloc((void**)&df, px * 3));
    CK(hipMalloc((void**)&dm, px * 3));
    CK(hipMalloc((void**)&d8, px * 3));
    CK(hipMemcpy(df, hf, px * 3, hipMemcpyHostToDevice));
    CK(hipMemcpy(dm, hm, px * 3, hipMemcpyHostToDevice));
    int oh, ow;
    adain_stylize_u8_out_size(H, W, 1, &oh, &ow);
    const size_t sb = adain_stylize_u8_workspace_bytes(1, H, W, 0, 1, 3, H, W, 0);
    void* sws;
    CK(hipMalloc(&sws, sb));
    AD(adain_stylize_u8(df, 1, H, W, pe, pd, stats + 1024, stats + 1536, 0.5f, 0.5f, NULL, NULL, NULL, 0.f, 0.f, dm, 0, 1, 3, H, W, d8, sws, sb, s));
    CK(hipStreamSynchronize(s));
    uint8_t* o8 = (uint8_t*)malloc(px * 3);
    CK(hipMemcpy(o8, d8, px * 3, hipMemcpyDeviceToHost));
    sum = 0; fnv = 1469598103934665603ull;
    for (size_t i = 0; i < px * 3; ++i) { sum += o8[i]; fnv = (fnv ^ o8[i]) * 1099511628211ull; }
    printf("c_abi_smoke: adain_stylize_u8 masked frame -> %dx%d, sum %llu, fnv %016llx\n", oh, ow, (unsigned long long)sum, (unsigned long long)fnv);
    return 0;
}
