// Microbenchmark: what do LDS reads, LDS writes and streaming vector-memory loads cost the fp32 matrix pipe beside plain vector
// instructions?  Two waves per SIMD (two 256-thread workgroups per CU), every CU busy, v_mfma_f32_32x32x2_f32 on six accumulators
// round-robin; per group of 8 MFMAs: NV v_fma_f32, NR ds_read_b128, NW ds_write_b128, NG buffer_load_dwordx4 (8 MB buffer,
// L2 / Infinity-Cache resident), all independent of the MFMAs.  Loaded values are only "used" by an empty asm statement a
// ring's length later (4 LDS reads; a memory load's slot one pass of 24 MFMAs later), so the loads cost their issue and their wait, not arithmetic.  Prints
// chip-wide TFLOP/s by wall clock.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_mix_probe tools/probes/mfma_mix_probe.hip && /tmp/mfma_mix_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int;

template <int NV, int NR, int NW, int NG, int WPS = 2>
__global__ __launch_bounds__(256, WPS) void k(float* out, const float* wbuf, int iters) {
    constexpr int LDSF = WPS == 1 ? 30000 : 18000;                // 117 KB: one workgroup per CU; 70 KB: exactly two
    __shared__ __attribute__((aligned(16))) float lds[LDSF];
    for (int i = threadIdx.x; i < LDSF; i += 256) lds[i] = i * 0.001f;
    __syncthreads();
    f32x16 acc[6];
    for (int i = 0; i < 6; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
    f32x4 rd[4], gl[12];                                          // every index below is a compile-time constant
    for (int i = 0; i < 4; ++i) rd[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 12; ++i) gl[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float a = threadIdx.x * 0.001f + 1.f, b = 0.5f + threadIdx.x * 0.002f, c = 1.0001f;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wbuf), 0, 8u << 20, 0x00020000);
    const int lane16 = (threadIdx.x & 255) * 20;                  // 80-byte stride: conflict-free b128
    int goff = ((blockIdx.x * 256 + threadIdx.x) * 16) & ((8 << 20) - 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {                            // 3 groups of 8 MFMAs
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                acc[(g * 8 + h) % 6] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[(g * 8 + h) % 6], 0, 0, 0);
                // the group's other work spread over its 8 MFMAs
#pragma unroll
                for (int n = h; n < NV; n += 8) v[n & 7] = __builtin_fmaf(v[n & 7], c, b);
#pragma unroll
                for (int n = h; n < NR; n += 8) {
                    asm volatile("" ::"v"(rd[(g * NR + n) & 3]));   // the value read four reads ago
                    rd[(g * NR + n) & 3] = *(const f32x4*)(lds + lane16 + (n & 3) * 4);
                }
#pragma unroll
                for (int n = h; n < NW; n += 8) *(f32x4*)(lds + lane16 + 8 + (n & 1) * 4) = rd[n & 3];
#pragma unroll
                for (int n = h; n < NG; n += 8) {
                    asm volatile("" ::"v"(gl[g * NG + n]));         // the value this slot got one pass of 24 MFMAs ago
                    const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, goff, (g * 8 + n) * 4096, 0);
                    gl[g * NG + n] = __builtin_bit_cast(f32x4, w);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        goff = (goff + 98304) & ((8 << 20) - 1);
    }
    float s = 0.f;
    for (int i = 0; i < 6; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += rd[i][0] + rd[i][3];
    for (int i = 0; i < 12; ++i) s += gl[i][0] + gl[i][2];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Design (c) of DESIGN.md section 8: eight-wave workgroups, one per CU; wave w < 4 transforms (NV vector instructions per 8 MFMAs)
// and hands its fragments to wave w + 4 on the same SIMD through LDS (per 24 MFMAs 6 ds_write_b128 by the producer, 6
// ds_read_b128 by the consumer), one workgroup barrier per 48 MFMAs; both waves carry the patch reads / weight loads of the mix.
template <int NV>
__global__ __launch_bounds__(512, 1) void kc(float* out, const float* wbuf, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[30000];     // 117 KB: one workgroup per CU
    for (int i = threadIdx.x; i < 30000; i += 512) lds[i] = i * 0.001f;
    __syncthreads();
    const bool producer = threadIdx.x < 256;
    f32x16 acc[6];
    for (int i = 0; i < 6; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
    f32x4 rd[4], gl[9], frag[6];
    for (int i = 0; i < 4; ++i) rd[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 9; ++i) gl[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 6; ++i) frag[i] = f32x4{1.f, 2.f, 3.f, 4.f};
    const float a = threadIdx.x * 0.001f + 1.f, b = 0.5f + threadIdx.x * 0.002f, c = 1.0001f;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wbuf), 0, 8u << 20, 0x00020000);
    const int lane16 = (threadIdx.x & 255) * 20;
    float* hand = lds + 6000 + (threadIdx.x & 255) * 28;          // the pair's hand-off slots (6 fragments of 16 bytes, padded)
    int goff = ((blockIdx.x * 512 + threadIdx.x) * 16) & ((8 << 20) - 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {                   // 2 x 24 MFMAs, then the barrier
#pragma unroll
            for (int g = 0; g < 3; ++g) {
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    acc[(g * 8 + h) % 6] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, frag[(g * 8 + h) % 6][h & 3], acc[(g * 8 + h) % 6], 0, 0, 0);
                    if (producer) {
#pragma unroll
                        for (int n = h; n < NV; n += 8) v[n & 7] = __builtin_fmaf(v[n & 7], c, b);
                        if (h < 2) *(f32x4*)(hand + (g * 2 + h) * 4) = f32x4{v[0], v[1], v[2], v[3]};      // 6 per 24 MFMAs
                    } else if (h < 2) {
                        frag[g * 2 + h] = *(const f32x4*)(hand + (g * 2 + h) * 4);
                    }
#pragma unroll
                    for (int n = h; n < 4; n += 8) {
                        asm volatile("" ::"v"(rd[(g * 4 + n) & 3]));
                        rd[(g * 4 + n) & 3] = *(const f32x4*)(lds + lane16 + (n & 3) * 4);
                    }
#pragma unroll
                    for (int n = h; n < 3; n += 8) {
                        asm volatile("" ::"v"(gl[g * 3 + n]));
                        const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, goff, (g * 8 + n) * 4096, 0);
                        gl[g * 3 + n] = __builtin_bit_cast(f32x4, w);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __syncthreads();
        goff = (goff + 98304) & ((8 << 20) - 1);
    }
    float s = 0.f;
    for (int i = 0; i < 6; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += rd[i][0] + rd[i][3];
    for (int i = 0; i < 9; ++i) s += gl[i][0] + gl[i][2];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV>
static void run_pair(const char* what) {
    const int blocks = 256, iters = 200;
    float *out, *wbuf;
    hipMalloc(&out, blocks * 512 * sizeof(float));
    hipMalloc(&wbuf, 8 << 20);
    hipMemset(wbuf, 0, 8 << 20);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((kc<NV>), dim3(blocks), dim3(512), 0, 0, out, wbuf, iters);
    hipEventRecord(e0, 0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((kc<NV>), dim3(blocks), dim3(512), 0, 0, out, wbuf, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 5.0 * blocks * 8 * (double)iters * 48 * 4096;
    printf("producer / consumer pairs, %2d v_fma per 8 MFMAs on the producer only, barrier per 48 MFMAs : %6.1f TFLOP/s   %s\n", NV,
           flops / (ms * 1e-3) / 1e12, what);
    hipFree(out);
    hipFree(wbuf);
}

template <int NV, int NR, int NW, int NG, int WPS = 2>
static void run(const char* what) {
    const int blocks = 256 * WPS, iters = 400;
    float *out, *wbuf;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipMalloc(&wbuf, 8 << 20);
    hipMemset(wbuf, 0, 8 << 20);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<NV, NR, NW, NG, WPS>), dim3(blocks), dim3(256), 0, 0, out, wbuf, iters);
    hipEventRecord(e0, 0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((k<NV, NR, NW, NG, WPS>), dim3(blocks), dim3(256), 0, 0, out, wbuf, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 5.0 * blocks * 4 * (double)iters * 24 * 4096;
    printf("%d wave(s)/SIMD, per 8 MFMAs: %2d v_fma  %d ds_read_b128  %d ds_write_b128  %d buffer_load_dwordx4 : %6.1f TFLOP/s   %s\n", WPS, NV, NR,
           NW, NG, flops / (ms * 1e-3) / 1e12, what);
    hipFree(out);
    hipFree(wbuf);
}

int main() {
    run<0, 0, 0, 0>("(warm-up)");
    run<0, 0, 0, 0>("bare");
    run<0, 2, 0, 0>("");
    run<0, 4, 0, 0>("the kernel's patch reads: 12 per 24 MFMAs");
    run<0, 8, 0, 0>("");
    run<0, 0, 1, 0>("the kernel's staging writes: 6 per 48 MFMAs");
    run<0, 0, 2, 0>("");
    run<0, 0, 0, 2>("the weight stream: one b128 per 4 MFMAs");
    run<0, 0, 0, 3>("weights + halo loads: 18 per 48 MFMAs");
    run<0, 0, 0, 4>("");
    run<25, 0, 0, 0>("the kernel's vector instructions: 152 per 48 MFMAs");
    run<25, 4, 0, 0>("");
    run<25, 4, 1, 0>("");
    run<25, 4, 1, 3>("the whole main-loop mix");
    run<13, 4, 1, 3>("half the vector instructions");
    run<0, 0, 0, 0, 1>("bare, one wave per SIMD");
    run<25, 0, 0, 0, 1>("");
    run<25, 4, 1, 3, 1>("the whole mix on one wave per SIMD");
    run<13, 0, 0, 0, 1>("");
    run<13, 4, 1, 3, 1>("design (d): one wave per SIMD with two channel tiles = half the vector instructions per MFMA");
    run<13, 8, 1, 6, 1>("(d) with twice the LDS reads and memory loads");
    // round 3, DESIGN.md section 8 (e): F(4x4,3x3) with two 32-channel tiles per wave at ONE wave per SIMD: per 8 MFMAs 18 vector
    // instructions (2.25 per MFMA), 2.7 patch reads (24 per 72 MFMAs), 0.5 staging writes, 2 weight + 0.5 halo loads.  Multiply its
    // rate by 4 (2.25 multiplies per output) and today's mix <25, 4, 1, 3> at two waves by 3 to compare per OUTPUT.
    run<18, 3, 1, 3, 1>("design (e): F(4x4,3x3), two channel tiles per wave, one wave per SIMD");
    run<18, 3, 0, 2, 1>("(e) without staging traffic");
    run<18, 0, 0, 0, 1>("(e) vector instructions only");
    run<16, 3, 1, 3, 1>("(e) with a 2.0-per-MFMA transform");
    run<18, 3, 1, 3, 2>("(e)'s mix at two waves per SIMD (does not fit the registers: reference only)");
    run_pair<25>("design (c): one wave of a SIMD transforms for both");
    run_pair<0>("the same without vector instructions");
    return 0;
}
