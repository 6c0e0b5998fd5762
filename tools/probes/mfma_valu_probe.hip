// Microbenchmark: what do plain fp32 vector instructions cost the fp32 matrix pipe?  A wave issues MFMAs back to back
// (v_mfma_f32_32x32x2_f32, 6 accumulators round-robin, or the same flops as v_mfma_f32_16x16x4_f32 x 2) with NV independent
// v_fma_f32 behind each; one or two waves per SIMD, every CU busy.  Prints shader cycles per 4096-flop MFMA step per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu_probe tools/probes/mfma_valu_probe.hip && /tmp/mfma_valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int NV, int WPS, bool SMALL>
__global__ __launch_bounds__(256, WPS) void k(float* out, unsigned long long* cyc, int iters) {
    __shared__ float pad[WPS == 1 ? 30000 : 18000];     // 117 KB: one workgroup per CU; 70 KB: exactly two
    if (threadIdx.x == 0) pad[0] = 0.f;
    f32x16 acc[6];
    for (int i = 0; i < 6; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
    const float a = threadIdx.x * 0.001f + 1.f, b = 0.5f + threadIdx.x * 0.002f, c = 1.0001f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h = 0; h < 24; ++h) {
            const int r = h % 6;
            if constexpr (SMALL) {
                f32x4* q = (f32x4*)&acc[r];
                q[h & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, q[h & 3], 0, 0, 0);
                q[(h + 1) & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, q[(h + 1) & 3], 0, 0, 0);
            } else {
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[r], 0, 0, 0);
            }
#pragma unroll
            for (int n = 0; n < NV; ++n) v[(h + n) & 7] = __builtin_fmaf(v[(h + n) & 7], c, b);
            __builtin_amdgcn_sched_group_barrier(0x008, SMALL ? 2 : 1, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = pad[0];
    for (int i = 0; i < 6; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// the same instruction counts with the vector instructions BATCHED: G MFMAs back to back, then G * NV v_fma_f32
template <int NV, int G>
__global__ __launch_bounds__(256, 2) void kb(float* out, unsigned long long* cyc, int iters) {
    __shared__ float pad[18000];
    if (threadIdx.x == 0) pad[0] = 0.f;
    f32x16 acc[6];
    for (int i = 0; i < 6; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
    const float a = threadIdx.x * 0.001f + 1.f, b = 0.5f + threadIdx.x * 0.002f, c = 1.0001f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h0 = 0; h0 < 24; h0 += G) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                acc[(h0 + g) % 6] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[(h0 + g) % 6], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int n = 0; n < NV * G; ++n) v[n & 7] = __builtin_fmaf(v[n & 7], c, b);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = pad[0];
    for (int i = 0; i < 6; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    (void)cyc;
}

// packed fp32: NP v_pk_fma_f32 (two FMAs per lane each) per MFMA, against 2 NP plain v_fma_f32
template <int NP>
__global__ __launch_bounds__(256, 2) void kp(float* out, int iters) {
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    __shared__ float pad[18000];
    if (threadIdx.x == 0) pad[0] = 0.f;
    f32x16 acc[6];
    for (int i = 0; i < 6; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f32x2{threadIdx.x * 0.01f + i, threadIdx.x * 0.02f - i};
    const float a = threadIdx.x * 0.001f + 1.f, b = 0.5f + threadIdx.x * 0.002f;
    const f32x2 c = {1.0001f, 0.9999f}, d = {b, a};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h = 0; h < 24; ++h) {
            acc[h % 6] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[h % 6], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NP; ++n) v[(h + n) & 7] = __builtin_elementwise_fma(v[(h + n) & 7], c, d);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = pad[0];
    for (int i = 0; i < 6; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NP>
static void run_packed() {
    const int blocks = 512, iters = 400;
    float* out;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((kp<NP>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e0, 0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((kp<NP>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 5.0 * blocks * 4 * (double)iters * 24 * 4096;
    printf("packed: %d v_pk_fma_f32 per MFMA (= %d plain FMAs of work), 2 waves/SIMD: %.1f TFLOP/s\n", NP, 2 * NP, flops / (ms * 1e-3) / 1e12);
    hipFree(out);
}

template <int NV, int G>
static void run_batched() {
    const int blocks = 512, iters = 400;
    float* out;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((kb<NV, G>), dim3(blocks), dim3(256), 0, 0, out, nullptr, iters);
    hipEventRecord(e0, 0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((kb<NV, G>), dim3(blocks), dim3(256), 0, 0, out, nullptr, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 5.0 * blocks * 4 * (double)iters * 24 * 4096;
    printf("batched: %d MFMAs then %d v_fma_f32 (%d per MFMA), 2 waves/SIMD: %.1f TFLOP/s\n", G, NV * G, NV, flops / (ms * 1e-3) / 1e12);
    hipFree(out);
}

template <int NV, int WPS, bool SMALL>
static void run() {
    const int blocks = 256 * WPS, iters = 400;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipMalloc(&cyc, blocks * 4 * sizeof(unsigned long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<NV, WPS, SMALL>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e0, 0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((k<NV, WPS, SMALL>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 5.0 * blocks * 4 * (double)iters * 24 * 4096 * 64 / 64;     // 4096 flop per MFMA step per wave
    double mn = 1e30, mx = 0;
    static unsigned long long h[8192];
    hipMemcpy(h, cyc, blocks * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < blocks * 4; ++i) { sum += (double)h[i]; mn = h[i] < mn ? h[i] : mn; mx = h[i] > mx ? h[i] : mx; }
    const double per_wave = sum / (blocks * 4) / ((double)iters * 24);       // cycles per MFMA step as one wave sees it
    printf("%s  %d v_fma_f32 per step, %d wave(s)/SIMD: %7.1f cycles per step per wave = %6.1f per SIMD (ideal 64): matrix pipe %.3f\n",
           SMALL ? "16x16x4 x2" : "32x32x2   ", NV, WPS, per_wave, per_wave / WPS, 64.0 * WPS / per_wave);
    printf("            wall %.3f ms per launch = %.1f TFLOP/s; per-wave cycles min %.0f max %.0f per step\n", ms / 5, flops / (ms * 1e-3) / 1e12,
           mn / ((double)iters * 24), mx / ((double)iters * 24));
    hipFree(out);
    hipFree(cyc);
}

int main() {
    run<0, 1, false>(); run<2, 1, false>(); run<4, 1, false>(); run<5, 1, false>(); run<6, 1, false>(); run<8, 1, false>(); run<12, 1, false>();
    run<0, 2, false>(); run<2, 2, false>(); run<4, 2, false>(); run<5, 2, false>(); run<6, 2, false>(); run<8, 2, false>(); run<12, 2, false>();
    run<0, 2, true>(); run<5, 2, true>(); run<8, 2, true>();
    run<0, 1, true>(); run<5, 1, true>();
    run_batched<5, 1>(); run_batched<5, 2>(); run_batched<5, 4>(); run_batched<5, 6>(); run_batched<5, 12>(); run_batched<5, 24>();
    run_batched<3, 1>(); run_batched<3, 4>(); run_batched<3, 12>();
    run_packed<1>(); run_packed<2>(); run_packed<3>(); run_packed<4>();
    return 0;
}
