// Device bodies of the fused small-L attention (stage II and the decoder attend over L = T1 / T2 = 8 thought vectors; the
// story is in rfn_attn.hip) as functions, run by two kernels: attn_small_fwd_k / attn_small_bwd_k (one launch per cell step,
// rfn_attn.hip) and the persistent recurrence kernels (rfn_chain.hip).  Also holds what those bodies share with the stage-I
// kernels of rfn_attn.hip (load-policy switches, row_tanh_dot).
#pragma once
#include "rfn_common.h"
#include "rfn_xb.h"

// Stage-I attention streams every byte of its operands exactly once per launch (1.6 GB of features, 0.4 GB of
// projections per step and encoder group), so their loads / stores carry the nontemporal hint: they stop displacing each
// other and the small reused operands in L2 / Infinity Cache.  Measured at C3 (tools/bench_attn.py --contig, per
// encoder): context 70.2 -> 63.2 us (6.5 TB/s), fused backward 122.7 -> 100.2 us (6.15 TB/s), raw scores 26.7 -> 24.1 us.
// Same arithmetic, bit-identical results.  0 = plain accesses (A/B).
#ifndef ATT_NT_P
#define ATT_NT_P 1   /* the projection slabs of the stage-I score kernels */
#endif
#ifndef ATT_NT_X
#define ATT_NT_X 1   /* the feature stream */
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
// one 16-B load of the feature stream (each feature byte is read once per launch)
__device__ __forceinline__ f32x4 att_ldx(const float* p) {
#if ATT_NT_X
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
#else
    return *reinterpret_cast<const f32x4*>(p);
#endif
}

#define ATT_THREADS 256
#define ATT_WAVES 4

template <bool VEC, bool STREAM = false>
__device__ __forceinline__ float row_tanh_dot(const float* __restrict__ p, const float* __restrict__ hp_s,
                                              const float* __restrict__ w_s, int A, int lane) {
    float part = 0.f;
    if constexpr (VEC) {
        for (int a = lane * 4; a < A; a += 256) {
            const f32x4 x = (STREAM && ATT_NT_P) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + a))
                                                 : *reinterpret_cast<const f32x4*>(p + a);
            const f32x4 hh = *reinterpret_cast<const f32x4*>(hp_s + a);
            const f32x4 ww = *reinterpret_cast<const f32x4*>(w_s + a);
#pragma unroll
            for (int e = 0; e < 4; ++e) part += rfn_tanh_fast(x[e] + hh[e]) * ww[e];
        }
    } else {
        for (int a = lane; a < A; a += 64) part += rfn_tanh_fast(p[a] + hp_s[a]) * w_s[a];
    }
    return rfn_wave_sum(part);
}


struct AttnSmallArgs {
    const float* proj[RFN_MAX_ENC];    // (b, l, :) at proj + b*psb + l*psl
    const float* hproj[RFN_MAX_ENC];   // (B, A)
    const float* w_out[RFN_MAX_ENC];   // (A)
    const float* b_out[RFN_MAX_ENC];   // (1) or NULL
    const float* x[RFN_MAX_ENC];       // att_seq: (b, l, :) at x + b*xsb + l*xsl
    float* alpha[RFN_MAX_ENC];         // (B, L)
    float* z[RFN_MAX_ENC];             // (B, D) with row stride ldz          (forward out / backward: dz in)
    float* dproj[RFN_MAX_ENC];         // backward out (may alias proj)
    float* dhproj[RFN_MAX_ENC];        // (B, A)
    float* dw_part[RFN_MAX_ENC];       // (B, A)
    float* dx[RFN_MAX_ENC];            // accumulated: dx[b,l,:] += ...   (same strides as x)
    long psb, psl, xsb, xsl, ldz, dpsb, dpsl;
    int L, A, D, accumulate_dproj;
};
#define ATS_MAX_L 1024   /* every thread walks the L scores of its row: meant for L = T1 / T2, a handful */

// One (batch row b, encoder g) of the fused small-L attention forward.  XB: rfn_xb.h -- hproj and z are handed between blocks
// of one launch (the products before / after the attention), everything else is older than the launch or younger than it.
template <bool XB>
__device__ __forceinline__ void attn_small_fwd_body(const AttnSmallArgs& a, const int b, const int g, float* sm) {
    const int A = XB ? xb_uni(a.A) : a.A, L = XB ? xb_uni(a.L) : a.L, D = XB ? xb_uni(a.D) : a.D, Ap = (A + 3) & ~3;
    float* hp_s = sm;
    float* w_s = sm + Ap;
    float* s_s = sm + 2 * Ap;   // [L] scores, then alpha
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* proj = a.proj[g] + b * a.psb;
    for (int i = tid; i < A; i += ATT_THREADS) {
        hp_s[i] = xb_ld1<XB>(a.hproj[g] + (long)b * A + i);
        w_s[i] = a.w_out[g][i];
    }
    __syncthreads();
    const float bo = a.b_out[g] ? a.b_out[g][0] : 0.f;
    const bool vecA = (A % 4 == 0) && ((a.psb | a.psl) % 4 == 0) && ((((uintptr_t)a.proj[g]) & 15) == 0);
    for (int l = wave; l < L; l += ATT_WAVES) {
        const float s = (vecA ? row_tanh_dot<true>(proj + l * a.psl, hp_s, w_s, A, lane)
                              : row_tanh_dot<false>(proj + l * a.psl, hp_s, w_s, A, lane)) + bo;
        if (lane == 0) s_s[l] = s;
    }
    __syncthreads();
    float m = -INFINITY, sum = 0.f;
    for (int l = 0; l < L; ++l) m = fmaxf(m, s_s[l]);
    for (int l = 0; l < L; ++l) sum += expf(s_s[l] - m);
    const float inv = 1.0f / sum;
    __syncthreads();
    for (int l = tid; l < L; l += ATT_THREADS) {
        const float al = expf(s_s[l] - m) * inv;
        s_s[l] = al;
        a.alpha[g][(long)b * L + l] = al;
    }
    __syncthreads();
    const float* x = a.x[g] + b * a.xsb;
    float* z = a.z[g] + b * a.ldz;
    for (int d = tid; d < D; d += ATT_THREADS) {
        float acc = 0.f;
        for (int l = 0; l < L; ++l) acc += s_s[l] * x[l * a.xsl + d];
        xb_st1<XB>(z + d, acc);
    }
}

// The same row for the persistent recurrence kernels, software-pipelined across the grid barrier in front of it (rfn_chain.hip):
// everything that is older than the launch -- the projection rows, the attended vectors, w_out -- is loaded into registers
// BEFORE `hook()` (the wait half of the barrier); behind it only hproj, the one operand the previous phase produced, is
// fetched.  Same arithmetic in the same order as attn_small_fwd_body (row_tanh_dot<true>, the softmax, the l-ordered context
// sums), so the results are the same bits.  Shapes it is written for: A <= 512, L <= 8, D <= 512, 16-B aligned rows (the
// caller checks with attn_small_fwd_xp_ok and uses the plain body otherwise).
__device__ __forceinline__ bool attn_small_fwd_xp_ok(const AttnSmallArgs& a, int g) {
    return a.A % 4 == 0 && a.A <= 512 && a.L <= 8 && a.D <= 2 * ATT_THREADS && ((a.psb | a.psl) % 4 == 0) &&
           ((((uintptr_t)a.proj[g]) & 15) == 0);
}
template <typename Hook>
__device__ __forceinline__ void attn_small_fwd_body_xp(const AttnSmallArgs& a, const int b, const int g, float* sm, Hook hook) {
    const int A = xb_uni(a.A), L = xb_uni(a.L), D = xb_uni(a.D), Ap = (A + 3) & ~3;
    float* hp_s = sm;
    float* w_s = sm + Ap;
    float* s_s = sm + 2 * Ap;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* proj = a.proj[g] + b * a.psb;
    const float* x = a.x[g] + b * a.xsb;
    // ---- ahead of the barrier ----------------------------------------------------------------------------------------
    float wv[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) wv[q] = (tid + q * ATT_THREADS < A) ? a.w_out[g][tid + q * ATT_THREADS] : 0.f;
    f32x4 pv[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int l = wave + ATT_WAVES * j, c = lane * 4 + 256 * q;
            pv[j][q] = (l < L && c < A) ? *reinterpret_cast<const f32x4*>(proj + l * a.psl + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    float xv[8][2];
#pragma unroll
    for (int l = 0; l < 8; ++l)
#pragma unroll
        for (int q = 0; q < 2; ++q) xv[l][q] = (l < L && tid + q * ATT_THREADS < D) ? x[l * a.xsl + tid + q * ATT_THREADS] : 0.f;
    const float bo = a.b_out[g] ? a.b_out[g][0] : 0.f;
    hook();
    // ---- behind it: hproj is the previous phase's output -----------------------------------------------------------------
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = tid + q * ATT_THREADS;
        if (i < A) {
            hp_s[i] = xb_ld1<true>(a.hproj[g] + (long)b * A + i);
            w_s[i] = wv[q];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int l = wave + ATT_WAVES * j;
        if (l < L) {
            float part = 0.f;   // row_tanh_dot<true>: the same elements in the same order
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c = lane * 4 + 256 * q;
                if (c < A) {
                    const f32x4 hh = *reinterpret_cast<const f32x4*>(hp_s + c);
                    const f32x4 ww = *reinterpret_cast<const f32x4*>(w_s + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) part += rfn_tanh_fast(pv[j][q][e] + hh[e]) * ww[e];
                }
            }
            const float sc = rfn_wave_sum(part) + bo;
            if (lane == 0) s_s[l] = sc;
        }
    }
    __syncthreads();
    float m = -INFINITY, sum = 0.f;
    for (int l = 0; l < L; ++l) m = fmaxf(m, s_s[l]);
    for (int l = 0; l < L; ++l) sum += expf(s_s[l] - m);
    const float inv = 1.0f / sum;
    __syncthreads();
    for (int l = tid; l < L; l += ATT_THREADS) {
        const float al = expf(s_s[l] - m) * inv;
        s_s[l] = al;
        a.alpha[g][(long)b * L + l] = al;
    }
    __syncthreads();
    float* z = a.z[g] + b * a.ldz;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int d = tid + q * ATT_THREADS;
        if (d < D) {
            float acc = 0.f;
#pragma unroll
            for (int l = 0; l < 8; ++l)
                if (l < L) acc += s_s[l] * xv[l][q];
            xb_st1<true>(z + d, acc);
        }
    }
}

// One (batch row b, encoder g) of the fused small-L attention backward.  XB: dz comes from the product before it in the
// launch, dhproj feeds the product after it, dproj / dx are accumulated across the steps of the launch (by whichever block
// gets the row): all sc1; alpha, hproj, proj, x are the forward pass's (an earlier launch), dw_part is read by a later one.
template <bool VEC, bool XB>
__device__ __forceinline__ void attn_small_bwd_body(const AttnSmallArgs& a, const int b, const int g, float* sm) {
    // Contraction is off here: `acc ? old + dpre : dpre` and its neighbours fused into fmas or not depending on how the compiler
    // shaped the code around them, which differs between the launch form and the persistent form; the forms must agree bit for bit.
#pragma clang fp contract(off)
    const int A = XB ? xb_uni(a.A) : a.A, L = XB ? xb_uni(a.L) : a.L, D = XB ? xb_uni(a.D) : a.D, Ap = (A + 3) & ~3, Dp = (D + 3) & ~3;
    float* hp_s = sm;                 // [Ap]
    float* w_s = sm + Ap;             // [Ap]
    float* dz_s = sm + 2 * Ap;        // [Dp]
    const int Lp = (L + 3) & ~3;
    float* al_s = dz_s + Dp;          // [Lp]
    float* ds_s = al_s + Lp;          // [Lp]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* dz = a.z[g] + b * a.ldz;
    for (int i = tid; i < A; i += ATT_THREADS) {
        hp_s[i] = a.hproj[g][(long)b * A + i];
        w_s[i] = a.w_out[g][i];
    }
    for (int d = tid; d < D; d += ATT_THREADS) dz_s[d] = xb_ld1<XB>(dz + d);
    for (int l = tid; l < L; l += ATT_THREADS) al_s[l] = a.alpha[g][(long)b * L + l];
    __syncthreads();
    const float* x = a.x[g] + b * a.xsb;
    // dalpha[l] = <dz, x[l]> : each wave takes rows l, l + 4 together (independent load streams)
    for (int l = wave; l < L; l += 2 * ATT_WAVES) {
        const int l2 = l + ATT_WAVES;
        const bool two = l2 < L;
        const float* x0 = x + l * a.xsl;
        const float* x1 = x + (two ? l2 : l) * a.xsl;
        float p0 = 0.f, p1 = 0.f;
        if constexpr (VEC) {
            for (int d = lane * 4; d < D; d += 256) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(x0 + d);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(x1 + d);
                const f32x4 gv = *reinterpret_cast<const f32x4*>(dz_s + d);
                p0 += v0[0] * gv[0] + v0[1] * gv[1] + v0[2] * gv[2] + v0[3] * gv[3];
                p1 += v1[0] * gv[0] + v1[1] * gv[1] + v1[2] * gv[2] + v1[3] * gv[3];
            }
        } else {
            for (int d = lane; d < D; d += 64) {
                p0 += x0[d] * dz_s[d];
                p1 += x1[d] * dz_s[d];
            }
        }
        p0 = rfn_wave_sum(p0);
        p1 = rfn_wave_sum(p1);
        if (lane == 0) {
            ds_s[l] = p0;
            if (two) ds_s[l2] = p1;
        }
    }
    __syncthreads();
    float dot = 0.f;
    for (int l = 0; l < L; ++l) dot += al_s[l] * ds_s[l];
    __syncthreads();
    for (int l = tid; l < L; l += ATT_THREADS) ds_s[l] = al_s[l] * (ds_s[l] - dot);   // softmax backward
    __syncthreads();
    // d att_seq through the context: dx[l, :] += alpha[l] * dz
    if (a.dx[g]) {
        float* dx = a.dx[g] + b * a.xsb;
        if constexpr (VEC) {
            const int D4 = D >> 2, n4 = L * D4;
            for (int i0 = tid; i0 < n4; i0 += 4 * ATT_THREADS) {
                f32x4 v[4];
                float* o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = min(i0 + j * ATT_THREADS, n4 - 1);
                    const int l = i / D4;
                    o[j] = dx + l * a.xsl + 4 * (i - l * D4);
                    if constexpr (XB) v[j] = xb_buf_ld4_sc1(xb_rsrc(xb_uni_ptr(dx)), (uint32_t)((l * a.xsl + 4 * (i - l * D4)) * 4));
                    else v[j] = *reinterpret_cast<const f32x4*>(o[j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = i0 + j * ATT_THREADS;
                    if (i >= n4) break;
                    const int l = i / D4, d = 4 * (i - l * D4);
                    const f32x4 gv = *reinterpret_cast<const f32x4*>(dz_s + d);
                    if constexpr (XB) xb_buf_st4_sc1(xb_rsrc(xb_uni_ptr(dx)), (uint32_t)((l * a.xsl + d) * 4), v[j] + al_s[l] * gv);
                    else *reinterpret_cast<f32x4*>(o[j]) = v[j] + al_s[l] * gv;
                }
            }
        } else {
            for (int i = tid; i < L * D; i += ATT_THREADS) {
                const int l = i / D, d = i - l * D;
                xb_st1<XB>(dx + l * a.xsl + d, xb_ld1<XB>(dx + l * a.xsl + d) + al_s[l] * dz_s[d]);
            }
        }
    }
    // tanh backward over the (L, A) slice: one thread per (4) hidden unit(s), rows in order (deterministic sums),
    // four rows' loads issued together
    const float* proj = a.proj[g] + b * a.psb;
    float* dproj = a.dproj[g] + b * a.dpsb;
    const bool acc = a.accumulate_dproj != 0;
    if constexpr (VEC) {
        for (int i = 4 * tid; i < A; i += 4 * ATT_THREADS) {
            const f32x4 hh = *reinterpret_cast<const f32x4*>(hp_s + i);
            const f32x4 ww = *reinterpret_cast<const f32x4*>(w_s + i);
            f32x4 ah = {0.f, 0.f, 0.f, 0.f}, aw = {0.f, 0.f, 0.f, 0.f};
            for (int l0 = 0; l0 < L; l0 += 4) {
                f32x4 pv[4], ov[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int l = min(l0 + j, L - 1);
                    pv[j] = *reinterpret_cast<const f32x4*>(proj + l * a.psl + i);
                    if constexpr (XB) {
                        if (acc) ov[j] = xb_buf_ld4_sc1(xb_rsrc(xb_uni_ptr(dproj)), (uint32_t)((l * a.dpsl + i) * 4));
                    } else {
                        if (acc) ov[j] = *reinterpret_cast<const f32x4*>(dproj + l * a.dpsl + i);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int l = l0 + j;
                    if (l >= L) break;
                    f32x4 dpre;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = rfn_tanh_fast(pv[j][e] + hh[e]);
                        dpre[e] = ds_s[l] * ww[e] * (1.0f - t * t);
                        ah[e] += dpre[e];
                        aw[e] += ds_s[l] * t;
                    }
                    if constexpr (XB) xb_buf_st4_sc1(xb_rsrc(xb_uni_ptr(dproj)), (uint32_t)((l * a.dpsl + i) * 4), acc ? ov[j] + dpre : dpre);
                    else *reinterpret_cast<f32x4*>(dproj + l * a.dpsl + i) = acc ? ov[j] + dpre : dpre;
                }
            }
            if constexpr (XB) xb_buf_st4_sc1(xb_rsrc(xb_uni_ptr(a.dhproj[g] + (long)b * A)), (uint32_t)(i * 4), ah);
            else *reinterpret_cast<f32x4*>(a.dhproj[g] + (long)b * A + i) = ah;
            *reinterpret_cast<f32x4*>(a.dw_part[g] + (long)b * A + i) = aw;
        }
    } else {
        for (int i = tid; i < A; i += ATT_THREADS) {
            const float hh = hp_s[i], ww = w_s[i];
            float ah = 0.f, aw = 0.f;
            for (int l = 0; l < L; ++l) {
                const float t = rfn_tanh_fast(proj[l * a.psl + i] + hh);
                const float dpre = ds_s[l] * ww * (1.0f - t * t);
                float* o = dproj + l * a.dpsl + i;
                xb_st1<XB>(o, acc ? xb_ld1<XB>(o) + dpre : dpre);
                ah += dpre;
                aw += ds_s[l] * t;
            }
            xb_st1<XB>(a.dhproj[g] + (long)b * A + i, ah);
            a.dw_part[g][(long)b * A + i] = aw;
        }
    }
}

