// Shared declarations for the gfx950 kernels of the recurrent-fusion decoder path.
// Written for MI355X (CDNA4) only: 64-lane waves, fp32 MFMA, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rfn.h"

#define RFN_WAVE 64

// Launch-error check used by every host launcher: kernels are asynchronous, the only error a
// launcher can see is a bad configuration.
#define RFN_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e_ = hipGetLastError();                   \
        if (e_ != hipSuccess) return RFN_ERR_LAUNCH;         \
    } while (0)

#define RFN_TRY(expr)                 \
    do {                              \
        int rc_ = (expr);             \
        if (rc_ != RFN_OK) return rc_; \
    } while (0)

static inline int rfn_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

static inline bool rfn_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

#ifdef __HIPCC__
__device__ __forceinline__ float rfn_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float rfn_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float rfn_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
// tanh for the attention score kernels, which evaluate B*L*A of them per call and were VALU-bound on
// ocml's tanhf: 1 - 2/(exp(2x)+1) on v_exp_f32 + v_rcp_f32.  Absolute error <= ~1.5e-7 over the whole
// range (the 1/(e+1) term is <= 1, each hardware op is 1 ulp), saturates cleanly to +-1.
__device__ __forceinline__ float rfn_tanh_fast(float x) {
    const float e = __expf(2.0f * x);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
// Philox4x32-10 keyed by `seed`; counter = (element index, call-site offset).  One 32-bit draw per element; forward and
// backward regenerate the same dropout mask from (seed, offset) instead of storing it.  THE one definition: the LSTM
// kernels (rfn_cell.hip), the gate GEMM's reduce (rfn_gemm.hip) and the cell GEMM's epilogues (rfn_cellgemm.hip) must
// draw identical bits for a forward mask and its backward twin to agree; rfn_dropout_mask (rfn.h) publishes them.
__device__ __forceinline__ float rfn_philox_uniform(uint64_t seed, uint64_t offset, uint64_t idx) {
    uint32_t c0 = (uint32_t)idx, c1 = (uint32_t)(idx >> 32), c2 = (uint32_t)offset, c3 = (uint32_t)(offset >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return (float)(c0 >> 8) * (1.0f / 16777216.0f);  // [0, 1)
}
// ---- f32 -> three bf16 planes (x = p0 + p1 + p2; csrc/rfn_gemm_x3.hip) ---------------------------------------------
__device__ __forceinline__ unsigned x3_bf16_rne(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void x3_split(float x, unsigned& p0, unsigned& p1, unsigned& p2) {
    const unsigned u = __float_as_uint(x);
    if ((u & 0x7F800000u) == 0x7F800000u) {   // infinity / NaN: the leading plane carries it, the others stay zero
        p0 = (u >> 16) | ((u & 0xFFFFu) ? 1u : 0u);   // a NaN stays a NaN even if its payload sat in the low half
        p1 = p2 = 0u;
        return;
    }
    p0 = x3_bf16_rne(x);
    if ((p0 & 0x7F80u) == 0x7F80u) p0 = u >> 16;   // rounding up would overflow: truncate, the residual takes the rest
    float r = x - __uint_as_float(p0 << 16);
    p1 = x3_bf16_rne(r);
    r -= __uint_as_float(p1 << 16);
    p2 = x3_bf16_rne(r);
}

#endif
