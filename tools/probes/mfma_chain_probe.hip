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

// 8 MFMAs per step on 2 accumulators + NL weight-like loads (b128 per lane, 1 KB per wave, streamed from a `bytes`-sized
// buffer, consumed 3 steps later as the A operand), every CU active: what do the loads cost the matrix pipe?
using f32x4 = __attribute__((ext_vector_type(4))) float;
template <int NL>
__global__ __launch_bounds__(256, 2) void chain_loads(float* out, unsigned long long* cyc, int iters, const float* w, unsigned bytes) {
    f32x16 acc[2];
    for (int k = 0; k < 2; ++k)
        for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, bytes, 0x00020000);
    const int lane_off = (threadIdx.x & 63) * 16;
    unsigned soff = __builtin_amdgcn_readfirstlane((blockIdx.x * 4 + (threadIdx.x >> 6)) * 65536u % bytes);
    f32x4 ring[4][2];
    for (int p = 0; p < 4; ++p)
        for (int n = 0; n < 2; ++n) ring[p][n] = f32x4{1.f, 2.f, 3.f, 4.f};
    float b = threadIdx.x * 0.001f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int st = 0; st < 4; ++st) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[st][0][s], b, acc[0], 0, 0, 0);
                if (s < NL) {
                    auto v = __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, soff, 0);
                    ring[(st + 3) & 3][s & 1] = __builtin_bit_cast(f32x4, v);
                    soff = soff + 1024u < bytes ? soff + 1024u : 0u;
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[st][1][s], b, acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    for (int k = 0; k < 2; ++k)
        for (int e = 0; e < 16; ++e) sum += acc[k][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NL>
static void run_loads(int blocks_per_cu, unsigned bytes) {
    float *out, *w;
    unsigned long long* cyc;
    const int blocks = 256 * blocks_per_cu;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipMalloc(&w, bytes);
    hipMemset(w, 0, bytes);
    hipMalloc(&cyc, blocks * sizeof(unsigned long long));
    const int iters = 500;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((chain_loads<NL>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, w, bytes);
    hipDeviceSynchronize();
    unsigned long long h[4096];
    hipMemcpy(h, cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < blocks; ++i) sum += (double)h[i];
    printf("%d b128 loads per 8 MFMAs from a %.1f MB buffer, %d wave(s) per SIMD: %.1f cycles per 8 MFMAs per wave (ideal %d)\n", NL, bytes / 1048576.0,
           blocks_per_cu, sum / blocks / ((double)iters * 4), 512 * blocks_per_cu);
    hipFree(out); hipFree(w); hipFree(cyc);
}

// MFMA stream with NR ds_read_b128 per 24 MFMA-equivalents (4096 flop each), two waves per SIMD: 32x32x2 (16 acc registers in
// and out per instruction) against 16x16x4 (4 in / out, twice as many instructions): does LDS return traffic slow the matrix pipe?
using f32x4v = __attribute__((ext_vector_type(4))) float;
template <int NR, bool SMALL>
__global__ __launch_bounds__(256, 2) void mfma_lds(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = i * 1e-4f;
    __syncthreads();
    f32x16 acc[6];
    for (int k = 0; k < 6; ++k)
        for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
    f32x4v d[12];
    for (int i = 0; i < 12; ++i) d[i] = f32x4v{1.f, 2.f, 3.f, 4.f};
    const int lane_off = (threadIdx.x & 63) * 20 + (threadIdx.x >> 6) * 4000;
    float a = threadIdx.x * 0.001f + 1.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h = 0; h < 24; ++h) {
            const int r = h % 6;
            if constexpr (SMALL) {
                f32x4v* q = (f32x4v*)&acc[r];
#pragma unroll
                for (int t = 0; t < 4; ++t) q[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, d[h % 12][t & 3], q[t], 0, 0, 0);
            } else {
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, d[h % 12][h & 3], acc[r], 0, 0, 0);
            }
            if (h < NR) d[(h + 6) % 12] = *(const f32x4v*)(lds + lane_off + h * 320);
            __builtin_amdgcn_sched_group_barrier(0x008, SMALL ? 4 : 1, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    for (int k = 0; k < 6; ++k)
        for (int e = 0; e < 16; ++e) sum += acc[k][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NR, bool SMALL>
static void run_lds() {
    float* out;
    unsigned long long* cyc;
    const int blocks = 512;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipMalloc(&cyc, blocks * sizeof(unsigned long long));
    const int iters = 300;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((mfma_lds<NR, SMALL>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[4096];
    hipMemcpy(h, cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < blocks; ++i) sum += (double)h[i];
    printf("%s, %2d ds_read_b128 per 24 MFMA-equivalents, 2 waves per SIMD: %.0f cycles per 24 per wave (ideal 3072)\n", SMALL ? "16x16x4" : "32x32x2", NR,
           sum / blocks / iters);
    hipFree(out); hipFree(cyc);
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
    run_lds<0, false>(); run_lds<6, false>(); run_lds<12, false>(); run_lds<24, false>();
    run_lds<0, true>(); run_lds<6, true>(); run_lds<12, true>(); run_lds<24, true>();
    for (int bpc = 1; bpc <= 2; ++bpc) {
        run_loads<0>(bpc, 1u << 20); run_loads<1>(bpc, 1u << 20); run_loads<2>(bpc, 1u << 20); run_loads<4>(bpc, 1u << 20);
        run_loads<2>(bpc, 6u << 20); run_loads<4>(bpc, 6u << 20); run_loads<2>(bpc, 64u << 20);
    }
    return 0;
}
