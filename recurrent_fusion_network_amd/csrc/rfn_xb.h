// Cross-block ("XB") access forms for kernels whose blocks hand data to each other INSIDE one launch (the persistent
// recurrence kernels, rfn_chain.hip).  On MI355X a CU's vector L1 is never refreshed by another CU's stores and the per-XCD
// L2s are not coherent with each other; the forms below are the sc1 ones: loads that bypass L1 and are served coherently,
// stores that write through.  With XB = false every helper is the plain access, so a body templated on XB compiles to
// exactly the code it had before for the one-launch-per-product kernels.
#pragma once
#include "rfn_common.h"

typedef float xb_f32x4 __attribute__((ext_vector_type(4)));
// ---- cross-block access forms -----------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) float xb_gfloat;
typedef unsigned xb_u32x4 __attribute__((ext_vector_type(4)));
template <bool XB>
__device__ __forceinline__ float xb_ld1(const float* p) {
    if constexpr (XB) return __hip_atomic_load((xb_gfloat*)(uintptr_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // global_load_dword sc1
    else return *p;
}
template <bool XB>
__device__ __forceinline__ void xb_st1(float* p, float v) {
    if constexpr (XB) __hip_atomic_store((xb_gfloat*)(uintptr_t)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // global_store_dword sc1
    else *p = v;
}
// wave-uniform copies of values read from an LDS-resident descriptor
__device__ __forceinline__ int xb_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T* xb_uni_ptr(T* p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (T*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}
// 16-B sc1 access through a buffer descriptor on a wave-uniform base, 32-bit byte offset per lane
__device__ __forceinline__ __amdgpu_buffer_rsrc_t xb_rsrc(const float* base_uniform) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base_uniform), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ xb_f32x4 xb_buf_ld4_sc1(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    return __builtin_bit_cast(xb_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));
}
__device__ __forceinline__ void xb_buf_st4_sc1(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, xb_f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(xb_u32x4, v), r, byte_off, 0, 16);
}

