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
