// Microbenchmark: cycles per v_mfma_f32_32x32x2_f32 when K accumulators are used round-robin (dependent chains of
// distance K), one wave per SIMD or two.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_chain_probe tools/probes/mfma_chain_probe.hip && /tmp/mfma_chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int K, int ROLE>     // ROLE 0: A varies per chain, 1: same operands
__global__ __launch_bounds__(256, 1) void chain(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc[K];
    for (int k = 0; k < K; ++k)
        for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
    float a = threadIdx.x * 0.001f + 1.f, b = 0.5f + threadIdx.x * 0.002f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
            for (int k = 0; k < K; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int k = 0; k < K; ++k)
        for (int e = 0; e < 16; ++e) s += acc[k][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// operands rotate through 8 + 8 different registers (as in a real kernel), K accumulators round-robin
template <int K>
__global__ __launch_bounds__(256, 1) void chain_ops(float* out, unsigned long long* cyc, int iters, const float* in) {
    f32x16 acc[K];
    for (int k = 0; k < K; ++k)
        for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[threadIdx.x + 64 * i]; b[i] = in[threadIdx.x + 64 * i + 512]; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
            for (int k = 0; k < K; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(rep + k) & 7], b[(rep + 3 * k) & 7], acc[k], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int k = 0; k < K; ++k)
        for (int e = 0; e < 16; ++e) s += acc[k][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int K>
static void run_ops() {
    float *out, *in;
    unsigned long long* cyc;
    const int blocks = 256;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipMalloc(&in, 2048 * sizeof(float));
    hipMemset(in, 0, 2048 * sizeof(float));
    hipMalloc(&cyc, blocks * sizeof(unsigned long long));
    const int iters = 2000;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((chain_ops<K>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, in);
    hipDeviceSynchronize();
    unsigned long long h[4096];
    hipMemcpy(h, cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < blocks; ++i) sum += (double)h[i];
    printf("rotating operands, K = %d accumulators, 1 wave per SIMD: %.1f cycles per MFMA\n", K, sum / blocks / ((double)iters * 8 * K));
    hipFree(out); hipFree(in); hipFree(cyc);
}

template <int K>
static void run(int waves_per_simd) {
    float* out;
    unsigned long long* cyc;
    const int blocks = 256 * waves_per_simd;        // 256-thread blocks: 1 wave per SIMD each
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipMalloc(&cyc, blocks * sizeof(unsigned long long));
    const int iters = 2000;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((chain<K, 0>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[4096];
    hipMemcpy(h, cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < blocks; ++i) sum += (double)h[i];
    const double per = sum / blocks / ((double)iters * 8 * K);
    printf("K = %d accumulators, %d wave(s) per SIMD: %.1f cycles per MFMA per wave  (pipe: %.1f cycles per MFMA)\n", K, waves_per_simd, per,
           per / waves_per_simd);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        run<1>(w); run<2>(w); run<3>(w); run<4>(w); run<6>(w);
    }
    run_ops<1>(); run_ops<2>(); run_ops<6>();
    return 0;
}
