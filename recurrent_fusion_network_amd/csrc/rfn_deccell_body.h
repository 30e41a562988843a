// Device body of the decoder's hoisted attention backward (rfn_deccell.hip has the story) as a function, so that two kernels can
// run it: dec_attn_bwd_fast_k (one launch per step) and cell_gemm_rows_k (rfn_cellgemm.hip), which runs the rows of a step
// beside the tiles of the step's K-split d gates . W_hh product in ONE launch -- the two do not depend on each other.
#pragma once
#include "rfn_attn_small_body.h"
#include "rfn_common.h"

#define DEC_LREG 8   /* thought vectors whose U values a thread keeps in registers */

struct DecAttnBwdArgs {
    const float* proj;     // (b, l, :) at proj + b * psb + l * psl
    const float* hproj;    // (B, A)
    const float* w_out;    // (A)
    const float* alpha;    // (B, L)
    const float* U;        // (b, l, :) at U + b * usb + l * usl, GD wide
    const float* dgates;   // (B, GD) row stride ldg
    float* dproj;          // same shape as proj, strides dpsb / dpsl
    float* dhproj;         // (B, A)
    float* dw_part;        // (B, A)
    long psb, psl, usb, usl, ldg, dpsb, dpsl;
    int L, A, GD, accumulate;
};

// The same backward for the shapes the path runs at (16-B accesses throughout, A <= 512, L <= 8): every operand -- the U rows and
// the gate gradients of the d alpha sums, the thread's four projection rows and their running gradients, hproj, w_out --
// requested before the first dependent instruction; all 256 threads in the tanh-backward (thread = 4 hidden units x 4 of the 8
// rows; the two row halves' column sums meet in LDS and are added lower half first).
__device__ __forceinline__ void dec_attn_bwd_fast_body(const DecAttnBwdArgs& a, const int b) {
    __shared__ float red[ATT_WAVES * DEC_LREG];
    __shared__ float da_s[DEC_LREG], al_s[DEC_LREG];
    __shared__ __attribute__((aligned(16))) float part_s[2 * 512];   // upper row half's [ah | aw]
    const int A = a.A, L = a.L, GD = a.GD;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i4 = tid & 127, half = tid >> 7, col = 4 * i4;
    const bool colok = col < A;
    const float* U = a.U + (long)b * a.usb;
    const float* dg = a.dgates + (long)b * a.ldg;
    const float* proj = a.proj + (long)b * a.psb;
    float* dproj = a.dproj + (long)b * a.dpsb;
    const bool acc = a.accumulate != 0;
    // ---- requests of the tanh-backward phase (they land under the d alpha sums) --------------------------------------------
    f32x4 pv[4], ov[4], hh = {0.f, 0.f, 0.f, 0.f}, ww = {0.f, 0.f, 0.f, 0.f};
    if (colok) {
        hh = *reinterpret_cast<const f32x4*>(a.hproj + (long)b * A + col);
        ww = *reinterpret_cast<const f32x4*>(a.w_out + col);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int l = half * 4 + j;
        const bool ok = colok && l < L;
        pv[j] = ok ? *reinterpret_cast<const f32x4*>(proj + l * a.psl + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        ov[j] = (ok && acc) ? *reinterpret_cast<const f32x4*>(dproj + l * a.dpsl + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (tid < L) al_s[tid] = a.alpha[(long)b * L + tid];
    // ---- d alpha_l = <d gates, U_l> --------------------------------------------------------------------------------------------
    float p[DEC_LREG];
#pragma unroll
    for (int j = 0; j < DEC_LREG; ++j) p[j] = 0.f;
    for (int c = 4 * tid; c < GD; c += 4 * ATT_THREADS) {
        const f32x4 gv = *reinterpret_cast<const f32x4*>(dg + c);
        f32x4 uv[DEC_LREG];
#pragma unroll
        for (int j = 0; j < DEC_LREG; ++j) uv[j] = *reinterpret_cast<const f32x4*>(U + ((j < L) ? j : L - 1) * a.usl + c);
#pragma unroll
        for (int j = 0; j < DEC_LREG; ++j) p[j] += (uv[j][0] * gv[0] + uv[j][1] * gv[1]) + (uv[j][2] * gv[2] + uv[j][3] * gv[3]);
    }
#pragma unroll
    for (int j = 0; j < DEC_LREG; ++j) {
        p[j] = rfn_wave_sum(p[j]);
        if (lane == 0) red[wave * DEC_LREG + j] = p[j];
    }
    __syncthreads();
    if (tid < L) {
        float sacc = red[tid];
#pragma unroll
        for (int w = 1; w < ATT_WAVES; ++w) sacc += red[w * DEC_LREG + tid];
        da_s[tid] = sacc;
    }
    __syncthreads();
    float dot = 0.f;
    for (int l = 0; l < L; ++l) dot += al_s[l] * da_s[l];
    // ---- softmax + tanh backward over this thread's rows -------------------------------------------------------------------
    f32x4 ah = {0.f, 0.f, 0.f, 0.f}, aw = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int l = half * 4 + j;
        if (l < L && colok) {
            const float ds = al_s[l] * (da_s[l] - dot);
            f32x4 dpre;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = rfn_tanh_fast(pv[j][e] + hh[e]);
                dpre[e] = ds * ww[e] * (1.0f - t * t);
                ah[e] += dpre[e];
                aw[e] += ds * t;
            }
            *reinterpret_cast<f32x4*>(dproj + l * a.dpsl + col) = acc ? ov[j] + dpre : dpre;
        }
    }
    if (half == 1 && colok) {
        *reinterpret_cast<f32x4*>(part_s + col) = ah;
        *reinterpret_cast<f32x4*>(part_s + 512 + col) = aw;
    }
    __syncthreads();
    if (half == 0 && colok) {
        ah += *reinterpret_cast<const f32x4*>(part_s + col);
        aw += *reinterpret_cast<const f32x4*>(part_s + 512 + col);
        *reinterpret_cast<f32x4*>(a.dhproj + (long)b * A + col) = ah;
        *reinterpret_cast<f32x4*>(a.dw_part + (long)b * A + col) = aw;
    }
}

