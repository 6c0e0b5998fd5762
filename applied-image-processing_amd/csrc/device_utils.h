// Device-side helpers shared by the convolution kernels (conv_edge.hip, conv_wino4.hip).
#pragma once
#include "common.h"

namespace adain {

__device__ __forceinline__ int reflect1(int v, int n) {
    // ReflectionPad2d(1) index map, after clamping to [-1, n] (tiles may overhang the image).
    v = max(-1, min(v, n));
    v = v < 0 ? -v : v;
    return v >= n ? 2 * n - 2 - v : v;
}

// element-wise max as ONE v_med3_f32 per element: max(a, b) = med3(a, b, +inf).  fmaxf costs three instructions here (a
// canonicalising v_max of each operand in front of the real one), and beside MFMAs every vector instruction counts.
__device__ __forceinline__ f32x4 max4(f32x4 a, f32x4 b) {
    f32x4 r;
    r.x = __builtin_amdgcn_fmed3f(a.x, b.x, __builtin_inff()); r.y = __builtin_amdgcn_fmed3f(a.y, b.y, __builtin_inff());
    r.z = __builtin_amdgcn_fmed3f(a.z, b.z, __builtin_inff()); r.w = __builtin_amdgcn_fmed3f(a.w, b.w, __builtin_inff());
    return r;
}

// Buffer-descriptor loads: 32-bit per-lane byte offset + scalar byte offset, hardware range check
// (out-of-range reads return 0), no 64-bit address arithmetic in the loop.
using rsrc_t = __amdgpu_buffer_rsrc_t;
using u32x4 = __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int;

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(rsrc_t r, int voff, int soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return __builtin_bit_cast(f32x4, v);
}


}  // namespace adain
