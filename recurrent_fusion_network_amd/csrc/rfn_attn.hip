// Additive soft attention of the recurrent-fusion path, split at the GEMM boundary.
//
// Reference: AttentionModelCore.forward (misc/AttentionModelCore.py:31-48) and its inlined copy in
// the decoder cell (misc/LSTMSoftAttentionCore.py:64-79).  The two projections (att_2_att_h, hoisted
// over all steps; h_2_att_h) are MFMA GEMMs (rfn_gemm.hip); everything between them and the gate
// GEMM lives here and is HBM-bound streaming work:
//   scores  : s[b,l] = w . tanh(proj[b,l,:] + hproj[b,:]) + b_o ; alpha = softmax_l(s)
//             reads the (B,L,A) projection slice ONCE, the (B,L,A) tanh tensor never exists
//             (the reference materialises it three times).
//   context : z[b,:] = sum_l alpha[b,l] att_seq[b,l,:]   -- one coalesced pass over the features.
// Layout: lanes run along the contiguous feature/hidden index with 16-B loads, a wave owns whole
// (b,l) rows, softmax reductions are 64-lane shuffles (L <= a few hundred: SURVEY.md section 5).
#include <stdlib.h>
#include <string.h>

#include "rfn_attn_small_body.h"
#include "rfn_internal.h"

// Raw scores s[b,l] on a (L-chunk, batch) grid: 16 rows per block, 4 per wave, so B*ceil(L/16) blocks keep every
// CU full of independent row streams (one block per batch row left 4 waves per CU waiting on their own loads).
#define SC_ROWS 16
// Per-encoder pointers of one grouped launch (blockIdx.z = encoder); encoders share every stride and (L, A, D).
struct AttnEncPtrs {
    const float* proj[RFN_MAX_ENC];
    const float* hproj[RFN_MAX_ENC];
    const float* w_out[RFN_MAX_ENC];
    const float* b_out[RFN_MAX_ENC];
    const float* x[RFN_MAX_ENC];
    const float* alpha_in[RFN_MAX_ENC];   // raw scores (forward) / alpha (backward)
    const float* dalpha[RFN_MAX_ENC];
    const float* dz[RFN_MAX_ENC];
    float* scores[RFN_MAX_ENC];
    float* alpha_out[RFN_MAX_ENC];
    float* z[RFN_MAX_ENC];
    float* dproj[RFN_MAX_ENC];
    float* dhproj[RFN_MAX_ENC];
    float* dw_part[RFN_MAX_ENC];
    char* ks_img[RFN_MAX_ENC];   // attn_scores_bwd_k<.., EMIT>: k-slow bf16 plane image that receives dproj (rfn.h, rfn_x3_split_ks)
};
// Per-encoder extents and strides of one grouped launch.  Encoders of one launch share A (att_hid_size) only: their maps
// may differ in L and D (the reference ships 196 x 2048, 64 x 1536, 64 x 1280 and 49 x 2208 maps together,
// feat_array.py:240-244); the grid is sized for the largest and the blocks past an encoder's own extent leave at once.
struct AttnDims {
    int L[RFN_MAX_ENC], D[RFN_MAX_ENC];
    long psb[RFN_MAX_ENC], psl[RFN_MAX_ENC];   // proj:   (b, l, :) at b*psb + l*psl
    long xsb[RFN_MAX_ENC], xsl[RFN_MAX_ENC];   // att_seq
    long dsb[RFN_MAX_ENC], dsl[RFN_MAX_ENC];   // dproj
    long ldz[RFN_MAX_ENC];                     // row stride of z / dz
    int maxL, maxD;
};
static AttnDims attn_dims_uniform(int ng, int L, int D, long psb, long psl, long xsb, long xsl, long dsb, long dsl, long ldz) {
    AttnDims m;
    memset(&m, 0, sizeof(m));
    for (int g = 0; g < ng && g < RFN_MAX_ENC; ++g) {
        m.L[g] = L; m.D[g] = D; m.psb[g] = psb; m.psl[g] = psl; m.xsb[g] = xsb; m.xsl[g] = xsl;
        m.dsb[g] = dsb; m.dsl[g] = dsl; m.ldz[g] = ldz;
    }
    m.maxL = L; m.maxD = D;
    return m;
}
// contiguous maps: proj (B, L_g, A), att_seq (B, L_g, D_g), z / dz (B, D_g), dproj like proj
static int attn_dims_het(int ng, const int* L, const int* D, int A, AttnDims* out) {
    AttnDims m;
    memset(&m, 0, sizeof(m));
    if (!L || !D || ng < 1 || ng > RFN_MAX_ENC) return RFN_ERR_SHAPE;
    for (int g = 0; g < ng; ++g) {
        if (L[g] <= 0 || D[g] <= 0) return RFN_ERR_SHAPE;
        m.L[g] = L[g]; m.D[g] = D[g];
        m.psb[g] = (long)L[g] * A; m.psl[g] = A; m.xsb[g] = (long)L[g] * D[g]; m.xsl[g] = D[g];
        m.dsb[g] = m.psb[g]; m.dsl[g] = A; m.ldz[g] = D[g];
        m.maxL = L[g] > m.maxL ? L[g] : m.maxL;
        m.maxD = D[g] > m.maxD ? D[g] : m.maxD;
    }
    *out = m;
    return RFN_OK;
}

template <bool VEC>
__global__ __launch_bounds__(ATT_THREADS) void attn_scores_raw_k(const AttnEncPtrs E, const AttnDims Dm, int A) {
    const int L = Dm.L[blockIdx.z];
    if ((int)blockIdx.x * SC_ROWS >= L) return;     // a shorter map of a heterogeneous launch
    const long sb = Dm.psb[blockIdx.z], sl = Dm.psl[blockIdx.z];
    const float* __restrict__ proj = E.proj[blockIdx.z];
    const float* __restrict__ hproj = E.hproj[blockIdx.z];
    const float* __restrict__ w_out = E.w_out[blockIdx.z];
    const float* __restrict__ b_out = E.b_out[blockIdx.z];
    float* __restrict__ scores = E.scores[blockIdx.z];
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Ap = (A + 3) & ~3;
    float* hp_s = sm;
    float* w_s = sm + Ap;
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int a = tid; a < A; a += ATT_THREADS) {
        hp_s[a] = hproj[(long)b * A + a];
        w_s[a] = w_out[a];
    }
    __syncthreads();
    const float bo = b_out ? b_out[0] : 0.f;
    const int l0 = blockIdx.x * SC_ROWS;
    for (int r = wave; r < SC_ROWS; r += ATT_WAVES) {
        const int l = l0 + r;
        if (l >= L) break;
        const float sc = row_tanh_dot<VEC, true>(proj + b * sb + l * sl, hp_s, w_s, A, lane) + bo;
        if (lane == 0) scores[(long)b * L + l] = sc;
    }
}

// alpha = softmax_l(scores) in place, one block per batch row (L values: trivial)
__global__ __launch_bounds__(ATT_THREADS) void attn_softmax_k(float* __restrict__ alpha, int L) {
    __shared__ float red[ATT_WAVES];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* a = alpha + (long)b * L;
    float m = -INFINITY;
    for (int l = tid; l < L; l += ATT_THREADS) m = fmaxf(m, a[l]);
    m = rfn_wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int l = tid; l < L; l += ATT_THREADS) sum += expf(a[l] - m);
    sum = rfn_wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
    for (int l = tid; l < L; l += ATT_THREADS) a[l] = expf(a[l] - m) * inv;
}

static int launch_scores_raw(const float* proj, int64_t proj_sb, int64_t proj_sl, const float* hproj,
                             const float* w_out, const float* b_out, int B, int L, int A, float* scores,
                             hipStream_t st);

extern "C" int rfn_attn_scores_fwd(const float* proj, int64_t proj_sb, int64_t proj_sl, const float* hproj,
                                   const float* w_out, const float* b_out, int B, int L, int A, float* alpha,
                                   void* stream) {
    hipStream_t st = (hipStream_t)stream;
    RFN_TRY(launch_scores_raw(proj, proj_sb, proj_sl, hproj, w_out, b_out, B, L, A, alpha, st));
    hipLaunchKernelGGL(attn_softmax_k, dim3(B), dim3(ATT_THREADS), 0, st, alpha, L);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}

// ---- context: z[b,d] = sum_l alpha[b,l] x[b,l,d] ------------------------------------------------
// SOFTMAX: `alpha` holds RAW scores; every block of a batch row normalises them itself in LDS (L values: trivial,
// same reduction order as attn_softmax_k so the weights are bit-identical) and the first block publishes them
// to alpha_out -- the separate softmax launch of the split path disappears.
template <bool VEC, bool SOFTMAX>
__global__ __launch_bounds__(ATT_THREADS) void attn_context_fwd_k(const AttnEncPtrs E, const AttnDims Dm) {
    const int L = Dm.L[blockIdx.z], D = Dm.D[blockIdx.z];
    if ((int)blockIdx.x * ATT_THREADS * (VEC ? 4 : 1) >= D) return;     // a narrower map of a heterogeneous launch
    const long sb = Dm.xsb[blockIdx.z], sl = Dm.xsl[blockIdx.z], ldz = Dm.ldz[blockIdx.z];
    const float* __restrict__ x = E.x[blockIdx.z];
    const float* __restrict__ alpha = E.alpha_in[blockIdx.z];
    float* __restrict__ z = E.z[blockIdx.z];
    float* __restrict__ alpha_out = E.alpha_out[blockIdx.z];
    extern __shared__ __attribute__((aligned(16))) float al_s[];
    const int b = blockIdx.y, tid = threadIdx.x;
    for (int l = tid; l < L; l += ATT_THREADS) al_s[l] = alpha[(long)b * L + l];
    if constexpr (SOFTMAX) {
        __shared__ float red[ATT_WAVES];
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        float m = -INFINITY;   // each thread re-reads exactly the entries it wrote: no barrier needed yet
        for (int l = tid; l < L; l += ATT_THREADS) m = fmaxf(m, al_s[l]);
        m = rfn_wave_max(m);
        if (lane == 0) red[wave] = m;
        __syncthreads();
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        __syncthreads();
        float sum = 0.f;
        for (int l = tid; l < L; l += ATT_THREADS) sum += expf(al_s[l] - m);
        sum = rfn_wave_sum(sum);
        if (lane == 0) red[wave] = sum;
        __syncthreads();
        const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
        for (int l = tid; l < L; l += ATT_THREADS) {
            const float a = expf(al_s[l] - m) * inv;
            al_s[l] = a;
            if (blockIdx.x == 0) alpha_out[(long)b * L + l] = a;
        }
    }
    __syncthreads();
    const float* xb = x + b * sb;
    if constexpr (VEC) {
        const int d = (blockIdx.x * ATT_THREADS + tid) * 4;
        if (d >= D) return;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int l = 0;
        for (; l + 4 <= L; l += 4) {  // 4 independent 16-B loads in flight per lane
            const f32x4 x0 = att_ldx(xb + (l + 0) * sl + d);
            const f32x4 x1 = att_ldx(xb + (l + 1) * sl + d);
            const f32x4 x2 = att_ldx(xb + (l + 2) * sl + d);
            const f32x4 x3 = att_ldx(xb + (l + 3) * sl + d);
            acc += al_s[l] * x0;
            acc += al_s[l + 1] * x1;
            acc += al_s[l + 2] * x2;
            acc += al_s[l + 3] * x3;
        }
        for (; l < L; ++l) acc += al_s[l] * *reinterpret_cast<const f32x4*>(xb + l * sl + d);
        *reinterpret_cast<f32x4*>(z + b * ldz + d) = acc;
    } else {
        const int d = blockIdx.x * ATT_THREADS + tid;
        if (d >= D) return;
        float acc = 0.f;
        for (int l = 0; l < L; ++l) acc += al_s[l] * xb[l * sl + d];
        z[b * ldz + d] = acc;
    }
}

template <bool SOFTMAX>
static int launch_context_d(int ng, const AttnEncPtrs& E, const AttnDims& Dm, int B, hipStream_t st) {
    if (B <= 0 || ng < 1 || ng > RFN_MAX_ENC || Dm.maxL <= 0 || Dm.maxD <= 0) return RFN_ERR_SHAPE;
    if ((size_t)Dm.maxL * sizeof(float) > 64 * 1024) return RFN_ERR_SHAPE;
    bool vec = true;
    for (int g = 0; g < ng; ++g) {
        if (!E.x[g] || !E.alpha_in[g] || !E.z[g] || (SOFTMAX && !E.alpha_out[g])) return RFN_ERR_ARG;
        vec = vec && (Dm.D[g] % 4 == 0) && (Dm.xsb[g] % 4 == 0) && (Dm.xsl[g] % 4 == 0) && (Dm.ldz[g] % 4 == 0) &&
              rfn_aligned16(E.x[g]) && rfn_aligned16(E.z[g]);
    }
    if (vec)
        hipLaunchKernelGGL((attn_context_fwd_k<true, SOFTMAX>), dim3(rfn_cdiv(Dm.maxD, 4 * ATT_THREADS), B, ng),
                           dim3(ATT_THREADS), Dm.maxL * sizeof(float), st, E, Dm);
    else
        hipLaunchKernelGGL((attn_context_fwd_k<false, SOFTMAX>), dim3(rfn_cdiv(Dm.maxD, ATT_THREADS), B, ng),
                           dim3(ATT_THREADS), Dm.maxL * sizeof(float), st, E, Dm);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}
template <bool SOFTMAX>
static int launch_context(int ng, const AttnEncPtrs& E, int64_t sb, int64_t sl, int B, int L, int D, int64_t ldz,
                          hipStream_t st) {
    if (L <= 0 || D <= 0) return RFN_ERR_SHAPE;
    return launch_context_d<SOFTMAX>(ng, E, attn_dims_uniform(ng, L, D, 0, 0, sb, sl, 0, 0, ldz), B, st);
}

extern "C" int rfn_attn_context_fwd(const float* att_seq, int64_t sb, int64_t sl, const float* alpha, int B, int L,
                                    int D, float* z, int64_t ldz, void* stream) {
    AttnEncPtrs E;
    memset(&E, 0, sizeof(E));
    E.x[0] = att_seq;
    E.alpha_in[0] = alpha;
    E.z[0] = z;
    return launch_context<false>(1, E, sb, sl, B, L, D, ldz, (hipStream_t)stream);
}

static int launch_scores_raw_d(int ng, const AttnEncPtrs& E, const AttnDims& Dm, int B, int A, hipStream_t st) {
    if (B <= 0 || Dm.maxL <= 0 || A <= 0 || ng < 1 || ng > RFN_MAX_ENC) return RFN_ERR_SHAPE;
    const size_t lds = (2 * ((A + 3) & ~3)) * sizeof(float);
    if (lds > 64 * 1024) return RFN_ERR_SHAPE;
    bool vec = (A % 4 == 0);
    for (int g = 0; g < ng; ++g) {
        if (!E.proj[g] || !E.hproj[g] || !E.w_out[g] || !E.scores[g]) return RFN_ERR_ARG;
        vec = vec && (Dm.psb[g] % 4 == 0) && (Dm.psl[g] % 4 == 0) && rfn_aligned16(E.proj[g]);
    }
    dim3 grid(rfn_cdiv(Dm.maxL, SC_ROWS), B, ng);
    if (vec)
        hipLaunchKernelGGL(attn_scores_raw_k<true>, grid, dim3(ATT_THREADS), lds, st, E, Dm, A);
    else
        hipLaunchKernelGGL(attn_scores_raw_k<false>, grid, dim3(ATT_THREADS), lds, st, E, Dm, A);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}
static int launch_scores_raw_g(int ng, const AttnEncPtrs& E, int64_t proj_sb, int64_t proj_sl, int B, int L, int A,
                               hipStream_t st) {
    if (L <= 0) return RFN_ERR_SHAPE;
    return launch_scores_raw_d(ng, E, attn_dims_uniform(ng, L, 1, proj_sb, proj_sl, 0, 0, 0, 0, 0), B, A, st);
}
static int launch_scores_raw(const float* proj, int64_t proj_sb, int64_t proj_sl, const float* hproj,
                             const float* w_out, const float* b_out, int B, int L, int A, float* scores,
                             hipStream_t st) {
    AttnEncPtrs E;
    memset(&E, 0, sizeof(E));
    E.proj[0] = proj;
    E.hproj[0] = hproj;
    E.w_out[0] = w_out;
    E.b_out[0] = b_out;
    E.scores[0] = scores;
    return launch_scores_raw_g(1, E, proj_sb, proj_sl, B, L, A, st);
}

// AttentionModelCore.forward of `ngroups` encoders that share (L, A, D) and strides, in two launches: raw scores,
// then softmax + context (see attn_context_fwd_k).  Arrays: host arrays of device pointers, one entry per encoder.
extern "C" int rfn_attn_fwd_grouped(int ngroups, const float* const* proj, int64_t proj_sb, int64_t proj_sl,
                                    const float* const* hproj, const float* const* w_out, const float* const* b_out,
                                    const float* const* att_seq, int64_t sb, int64_t sl, int B, int L, int A, int D,
                                    float* const* scores_scratch, float* const* alpha, float* const* z, int64_t ldz,
                                    void* stream) {
    if (ngroups < 1 || ngroups > RFN_MAX_ENC) return RFN_ERR_SHAPE;
    if (!proj || !hproj || !w_out || !att_seq || !scores_scratch || !alpha || !z) return RFN_ERR_ARG;
    AttnEncPtrs E;
    memset(&E, 0, sizeof(E));
    for (int g = 0; g < ngroups; ++g) {
        if (!scores_scratch[g] || !alpha[g] || scores_scratch[g] == alpha[g]) return RFN_ERR_ARG;
        E.proj[g] = proj[g];
        E.hproj[g] = hproj[g];
        E.w_out[g] = w_out[g];
        E.b_out[g] = b_out ? b_out[g] : nullptr;
        E.scores[g] = scores_scratch[g];
        E.x[g] = att_seq[g];
        E.alpha_in[g] = scores_scratch[g];
        E.alpha_out[g] = alpha[g];
        E.z[g] = z[g];
    }
    RFN_TRY(launch_scores_raw_g(ngroups, E, proj_sb, proj_sl, B, L, A, (hipStream_t)stream));
    return launch_context<true>(ngroups, E, sb, sl, B, L, D, ldz, (hipStream_t)stream);
}
extern "C" int rfn_attn_fwd(const float* proj, int64_t proj_sb, int64_t proj_sl, const float* hproj,
                            const float* w_out, const float* b_out, const float* att_seq, int64_t sb, int64_t sl,
                            int B, int L, int A, int D, float* scores_scratch, float* alpha, float* z, int64_t ldz,
                            void* stream) {
    return rfn_attn_fwd_grouped(1, &proj, proj_sb, proj_sl, &hproj, &w_out, &b_out, &att_seq, sb, sl, B, L, A, D,
                                &scores_scratch, &alpha, &z, ldz, stream);
}

// ---- backward of the context: dalpha[b,l] = <dz[b,:], x[b,l,:]> ---------------------------------
// Each wave owns ROWS/4 rows in groups of RG that it walks TOGETHER along d, so RG (x2 with the unroll)
// independent 16-B loads are in flight per lane instead of one row's dependent stream.
template <bool VEC, int RG, int ROWS>
__global__ __launch_bounds__(ATT_THREADS) void attn_dalpha_k(const float* __restrict__ x, long sb, long sl,
                                                            const float* __restrict__ dz, long lddz, int L, int D,
                                                            float* __restrict__ dalpha) {
    extern __shared__ __attribute__((aligned(16))) float dz_s[];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int d = tid; d < D; d += ATT_THREADS) dz_s[d] = dz[b * lddz + d];
    __syncthreads();
    for (int grp = 0; grp < ROWS / (RG * ATT_WAVES); ++grp) {
        const int l0 = blockIdx.x * ROWS + (grp * ATT_WAVES + wave) * RG;   // rows l0 .. l0+RG-1 of this wave
        const int nr = min(RG, L - l0);
        if (nr <= 0) break;
        const float* p0 = x + b * sb + (long)l0 * sl;
        float part[RG];
#pragma unroll
        for (int r = 0; r < RG; ++r) part[r] = 0.f;
        if constexpr (VEC) {
#pragma unroll 2
            for (int d = lane * 4; d < D; d += 256) {
                const f32x4 gv = *reinterpret_cast<const f32x4*>(dz_s + d);
                f32x4 xv[RG];
#pragma unroll
                for (int r = 0; r < RG; ++r)   // rows past the end re-read the last valid row (result discarded):
                    xv[r] = att_ldx(p0 + min(r, nr - 1) * sl + d);  // no branch per load
#pragma unroll
                for (int r = 0; r < RG; ++r)
                    part[r] += xv[r][0] * gv[0] + xv[r][1] * gv[1] + xv[r][2] * gv[2] + xv[r][3] * gv[3];
            }
        } else {
            for (int d = lane; d < D; d += 64)
                for (int r = 0; r < nr; ++r) part[r] += p0[r * sl + d] * dz_s[d];
        }
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const float t = rfn_wave_sum(part[r]);
            if (lane == 0 && r < nr) dalpha[(long)b * L + l0 + r] = t;
        }
    }
}

template <bool VEC, int RG, int ROWS>
static void launch_dalpha(const float* att_seq, long sb, long sl, const float* dz, long lddz, int B, int L, int D,
                          float* dalpha, size_t lds, hipStream_t st) {
    hipLaunchKernelGGL((attn_dalpha_k<VEC, RG, ROWS>), dim3(rfn_cdiv(L, ROWS), B), dim3(ATT_THREADS), lds, st, att_seq,
                       sb, sl, dz, lddz, L, D, dalpha);
}

extern "C" int rfn_attn_context_bwd_dalpha(const float* att_seq, int64_t sb, int64_t sl, const float* dz,
                                           int64_t lddz, int B, int L, int D, float* dalpha, void* stream) {
    if (B <= 0 || L <= 0 || D <= 0) return RFN_ERR_SHAPE;
    if (!att_seq || !dz || !dalpha) return RFN_ERR_ARG;
    const size_t lds = (size_t)((D + 3) & ~3) * sizeof(float);
    if (lds > 64 * 1024) return RFN_ERR_SHAPE;
    const bool vec = (D % 4 == 0) && rfn_aligned16(att_seq) && (sb % 4 == 0) && (sl % 4 == 0);
    hipStream_t st = (hipStream_t)stream;
    if (vec) launch_dalpha<true, 4, 64>(att_seq, sb, sl, dz, lddz, B, L, D, dalpha, lds, st);
    else launch_dalpha<false, 4, 64>(att_seq, sb, sl, dz, lddz, B, L, D, dalpha, lds, st);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}

// ---- datt_seq[b,l,:] += alpha[b,l] * dz[b,:]  (stage II / decoder: att_seq is differentiable) ----
__global__ __launch_bounds__(ATT_THREADS) void attn_dseq_k(const float* __restrict__ alpha,
                                                          const float* __restrict__ dz, long lddz, int L, int D,
                                                          float* __restrict__ dx, long sb, long sl) {
    const int l = blockIdx.x, b = blockIdx.y;
    const float a = alpha[(long)b * L + l];
    float* o = dx + b * sb + l * sl;
    const float* g = dz + b * lddz;
    for (int d = threadIdx.x; d < D; d += ATT_THREADS) o[d] += a * g[d];
}

extern "C" int rfn_attn_context_bwd_dseq(const float* alpha, const float* dz, int64_t lddz, int B, int L, int D,
                                         float* datt_seq, int64_t sb, int64_t sl, void* stream) {
    if (B <= 0 || L <= 0 || D <= 0) return RFN_ERR_SHAPE;
    if (!alpha || !dz || !datt_seq) return RFN_ERR_ARG;
    hipLaunchKernelGGL(attn_dseq_k, dim3(L, B), dim3(ATT_THREADS), 0, (hipStream_t)stream, alpha, dz, (long)lddz, L,
                       D, datt_seq, (long)sb, (long)sl);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}

// ---- backward of scores: softmax bwd + tanh bwd, writes dproj (may alias proj) -------------------
// One block per batch row.  For each 64*W-wide chunk of A (W = 4 with 16-B accesses) every wave
// sweeps its rows l = wave, wave+4, ... keeping the chunk's column sums (dhproj, dw) in registers;
// the four waves' sums are combined through LDS in a fixed order (deterministic).
#ifndef SB_WAVES
#define SB_WAVES 16
#endif
#define SB_THREADS (64 * SB_WAVES)
#ifndef SB_UNROLL
#define SB_UNROLL 4
#endif
// FUSED: dalpha[l] = <dz, x[l]> is computed here first (into LDS, same per-row arithmetic as attn_dalpha_k), so the
// context backward and the score backward of one (step, encoder) are a single launch: x is streamed, then P.
// EMIT (with VEC, FUSED): dproj is not written as f32; its three bf16 planes go straight into the k-slow plane image the
// weight-gradient GEMM on the bf16 matrix cores reads (element (k = b * L + l, plane, column ks_col0 + a) at
// ((k * 3 + plane) * ks_mp + ks_col0 + a) * 2 bytes): 8 bytes per lane and plane, whole rows coalesced -- the separate split
// pass over dproj (a read of 4 and a write of 6 bytes per element) disappears.
template <bool VEC, bool FUSED, bool EMIT = false>
__global__ __launch_bounds__(SB_THREADS) void attn_scores_bwd_k(const AttnEncPtrs E, const AttnDims Dm, int A, int accumulate,
                                                               int vec_x, int ks_mp, int ks_col0) {
    const int L = Dm.L[blockIdx.y], D = Dm.D[blockIdx.y];
    const long sb = Dm.psb[blockIdx.y], sl = Dm.psl[blockIdx.y], dsb = Dm.dsb[blockIdx.y], dsl = Dm.dsl[blockIdx.y];
    const long xsb = Dm.xsb[blockIdx.y], xsl = Dm.xsl[blockIdx.y], lddz = Dm.ldz[blockIdx.y];
    const float* proj = E.proj[blockIdx.y];   // may alias dproj
    const float* __restrict__ hproj = E.hproj[blockIdx.y];
    const float* __restrict__ w_out = E.w_out[blockIdx.y];
    const float* __restrict__ alpha = E.alpha_in[blockIdx.y];
    const float* __restrict__ dalpha = E.dalpha[blockIdx.y];
    float* dproj = E.dproj[blockIdx.y];
    float* __restrict__ dhproj = E.dhproj[blockIdx.y];
    float* __restrict__ dw_part = E.dw_part[blockIdx.y];
    const float* __restrict__ x = E.x[blockIdx.y];
    const float* __restrict__ dz = E.dz[blockIdx.y];
    constexpr int W = VEC ? 4 : 1;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Ap = (A + 3) & ~3;
    float* hp_s = sm;             // [Ap]
    float* w_s = sm + Ap;         // [Ap]
    float* red_h = sm + 2 * Ap;                    // [SB_WAVES][Ap]
    float* red_w = sm + (2 + SB_WAVES) * Ap;       // [SB_WAVES][Ap]
    float* dot_s = sm + (2 + 2 * SB_WAVES) * Ap;   // [SB_WAVES]
    float* ds_s = dot_s + SB_WAVES;                // [L]
    // wave index as a scalar: the rows a wave walks (l = wave + ...) are then wave-uniform, so every row address below is a
    // scalar base + this lane's column offset instead of a 64-bit per-lane pointer (fewer VGPRs: the plane-emitting
    // instantiation spilled three of its 128)
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int a = tid; a < A; a += SB_THREADS) {
        hp_s[a] = hproj[(long)b * A + a];
        w_s[a] = w_out[a];
    }
    const float* dal = dalpha + (long)b * L;
    if constexpr (FUSED) {
        const int Lp = (L + 3) & ~3;
        float* dal_s = ds_s + Lp;                  // [Lp]
        float* dz_s = dal_s + Lp;                  // [D]
        for (int d = tid; d < D; d += SB_THREADS) dz_s[d] = dz[b * lddz + d];
        __syncthreads();
        for (int l0 = wave * 4; l0 < L; l0 += SB_WAVES * 4) {   // 4 rows per wave walked together, as attn_dalpha_k
            const int nr = min(4, L - l0);
            const float* p0 = x + b * xsb + (long)l0 * xsl;
            float pt[4] = {0.f, 0.f, 0.f, 0.f};
            if (vec_x) {
#pragma unroll 2
                for (int d = lane * 4; d < D; d += 256) {
                    const f32x4 gv = *reinterpret_cast<const f32x4*>(dz_s + d);
                    f32x4 xv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) xv[r] = att_ldx(p0 + min(r, nr - 1) * xsl + d);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        pt[r] += xv[r][0] * gv[0] + xv[r][1] * gv[1] + xv[r][2] * gv[2] + xv[r][3] * gv[3];
                }
            } else {
                for (int d = lane; d < D; d += 64)
                    for (int r = 0; r < nr; ++r) pt[r] += p0[r * xsl + d] * dz_s[d];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = rfn_wave_sum(pt[r]);
                if (lane == 0 && r < nr) dal_s[l0 + r] = t;
            }
        }
        __syncthreads();
        dal = dal_s;
    }
    // softmax backward: ds = alpha * (dalpha - <alpha, dalpha>)
    float part = 0.f;
    for (int l = tid; l < L; l += SB_THREADS) part += alpha[(long)b * L + l] * dal[l];
    part = rfn_wave_sum(part);
    if (lane == 0) dot_s[wave] = part;
    __syncthreads();
    float dot = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < SB_WAVES; ++w2) dot += dot_s[w2];
    for (int l = tid; l < L; l += SB_THREADS)
        ds_s[l] = alpha[(long)b * L + l] * (dal[l] - dot);
    __syncthreads();

    for (int a0 = 0; a0 < A; a0 += 64 * W) {
        const int a = a0 + lane * W;
        float ah[W], aw[W];
#pragma unroll
        for (int e = 0; e < W; ++e) ah[e] = aw[e] = 0.f;
        if (a < A) {
            float hh[W], ww[W];
#pragma unroll
            for (int e = 0; e < W; ++e) {
                hh[e] = hp_s[a + e];
                ww[e] = w_s[a + e];
            }
            // SB_UNROLL rows of this wave are in flight together: each lane issues its loads first, then does the
            // arithmetic and the stores (one row at a time left a single 1-KiB load in flight per wave and the sweep
            // latency-bound).  The per-column sums still add the rows in ascending order.
            for (int l0 = wave; l0 < L; l0 += SB_WAVES * SB_UNROLL) {
                float xv[SB_UNROLL][W];
#pragma unroll
                for (int u = 0; u < SB_UNROLL; ++u) {
                    const int l = l0 + u * SB_WAVES;
                    if (l < L) {
                        const float* p = proj + b * sb + l * sl + a;
                        if constexpr (VEC) {
                            const f32x4 t = ATT_NT_P ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p))
                                                     : *reinterpret_cast<const f32x4*>(p);
                            xv[u][0] = t[0]; xv[u][1] = t[1]; xv[u][2] = t[2]; xv[u][3] = t[3];
                        } else {
                            xv[u][0] = p[0];
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < SB_UNROLL; ++u) {
                    const int l = l0 + u * SB_WAVES;
                    if (l < L) {
                        const float dsl_v = ds_s[l];
                        float* o = dproj + b * dsb + l * dsl + a;
                        float ov[W];
#pragma unroll
                        for (int e = 0; e < W; ++e) {
                            const float t = rfn_tanh_fast(xv[u][e] + hh[e]);
                            const float dpre = dsl_v * ww[e] * (1.0f - t * t);
                            ov[e] = dpre;
                            ah[e] += dpre;
                            aw[e] += dsl_v * t;
                        }
                        if constexpr (EMIT) {
                            unsigned q[3][4];
#pragma unroll
                            for (int e = 0; e < W; ++e) x3_split(ov[e], q[0][e], q[1][e], q[2][e]);
                            char* img = E.ks_img[blockIdx.y] + (((long)b * L + l) * 3 * ks_mp + ks_col0 + a) * 2;
#pragma unroll
                            for (int p = 0; p < 3; ++p) {
                                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                                const u32x2 w2 = {q[p][0] | (q[p][1] << 16), q[p][2] | (q[p][3] << 16)};
                                *reinterpret_cast<u32x2*>(img + (long)p * ks_mp * 2) = w2;   // (a nontemporal store measured no better)
                            }
                        } else if constexpr (VEC) {
                            f32x4 t = {ov[0], ov[1], ov[2], ov[3]};
                            if (accumulate) t += *reinterpret_cast<const f32x4*>(o);
                            if (ATT_NT_P) __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(o));
                            else *reinterpret_cast<f32x4*>(o) = t;
                        } else {
                            o[0] = accumulate ? o[0] + ov[0] : ov[0];
                        }
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < W; ++e) {
                red_h[wave * Ap + a + e] = ah[e];
                red_w[wave * Ap + a + e] = aw[e];
            }
        }
    }
    __syncthreads();
    for (int a = tid; a < A; a += SB_THREADS) {
        float th = 0.f, tw = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < SB_WAVES; ++w2) {  // fixed order over the waves
            th += red_h[w2 * Ap + a];
            tw += red_w[w2 * Ap + a];
        }
        dhproj[(long)b * A + a] = th;
        dw_part[(long)b * A + a] = tw;
    }
}

template <bool FUSED>
static int launch_scores_bwd_d(int ng, const AttnEncPtrs& E, const AttnDims& Dm, int B, int A, int accumulate_dproj,
                               hipStream_t st, int ks_mp = 0, int ks_col0 = 0) {
    const bool emit = ks_mp > 0;   // dproj goes to E.ks_img as bf16 planes instead of f32
    const int L = Dm.maxL, D = Dm.maxD;     // LDS is sized for the largest map of the launch
    if (B <= 0 || L <= 0 || A <= 0 || ng < 1 || ng > RFN_MAX_ENC) return RFN_ERR_SHAPE;
    size_t fl = (size_t)(2 + 2 * SB_WAVES) * ((A + 3) & ~3) + SB_WAVES + L;
    if (FUSED) fl = (size_t)(2 + 2 * SB_WAVES) * ((A + 3) & ~3) + SB_WAVES + 2 * ((L + 3) & ~3) + ((D + 3) & ~3);
    const size_t lds = fl * sizeof(float);
    if (lds > 150 * 1024) return RFN_ERR_SHAPE;
    bool vec = (A % 4 == 0);
    bool vx = FUSED;
    for (int g = 0; g < ng; ++g) {
        if (!E.proj[g] || !E.hproj[g] || !E.w_out[g] || !E.alpha_in[g] || !E.dhproj[g] || !E.dw_part[g]) return RFN_ERR_ARG;
        if (emit ? !E.ks_img[g] : !E.dproj[g]) return RFN_ERR_ARG;
        if (FUSED ? (!E.x[g] || !E.dz[g]) : !E.dalpha[g]) return RFN_ERR_ARG;
        vec = vec && (Dm.psb[g] % 4 == 0) && (Dm.psl[g] % 4 == 0) && (Dm.dsb[g] % 4 == 0) && (Dm.dsl[g] % 4 == 0) &&
              rfn_aligned16(E.proj[g]) && (emit || rfn_aligned16(E.dproj[g]));
        vx = vx && (Dm.D[g] % 4 == 0) && (Dm.xsb[g] % 4 == 0) && (Dm.xsl[g] % 4 == 0) && rfn_aligned16(E.x[g]);
    }
    if (emit) {
        if (!FUSED || !vec || accumulate_dproj || (ks_mp & 3) || (ks_col0 & 3)) return RFN_ERR_SHAPE;
        if constexpr (FUSED) {
            auto k = attn_scores_bwd_k<true, true, true>;
            if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k, dim3(B, ng), dim3(SB_THREADS), lds, st, E, Dm, A, 0, (int)vx, ks_mp, ks_col0);
        }
    } else if (vec) {
        auto k = attn_scores_bwd_k<true, FUSED>;
        if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, dim3(B, ng), dim3(SB_THREADS), lds, st, E, Dm, A, accumulate_dproj, (int)vx, 0, 0);
    } else {
        auto k = attn_scores_bwd_k<false, FUSED>;
        if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, dim3(B, ng), dim3(SB_THREADS), lds, st, E, Dm, A, accumulate_dproj, (int)vx, 0, 0);
    }
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}
template <bool FUSED>
static int launch_scores_bwd(int ng, const AttnEncPtrs& E, int64_t proj_sb, int64_t proj_sl, int B, int L, int A,
                             int64_t dproj_sb, int64_t dproj_sl, int accumulate_dproj, int64_t xsb, int64_t xsl,
                             int64_t lddz, int D, hipStream_t st, int ks_mp = 0, int ks_col0 = 0) {
    if (L <= 0 || (FUSED && D <= 0)) return RFN_ERR_SHAPE;
    return launch_scores_bwd_d<FUSED>(ng, E, attn_dims_uniform(ng, L, D, proj_sb, proj_sl, xsb, xsl, dproj_sb, dproj_sl, lddz),
                                      B, A, accumulate_dproj, st, ks_mp, ks_col0);
}

extern "C" int rfn_attn_scores_bwd(const float* proj, int64_t proj_sb, int64_t proj_sl, const float* hproj,
                                   const float* w_out, const float* alpha, const float* dalpha, int B, int L, int A,
                                   float* dproj, int64_t dproj_sb, int64_t dproj_sl, int accumulate_dproj,
                                   float* dhproj, float* dw_part, void* stream) {
    AttnEncPtrs E;
    memset(&E, 0, sizeof(E));
    E.proj[0] = proj; E.hproj[0] = hproj; E.w_out[0] = w_out; E.alpha_in[0] = alpha; E.dalpha[0] = dalpha;
    E.dproj[0] = dproj; E.dhproj[0] = dhproj; E.dw_part[0] = dw_part;
    return launch_scores_bwd<false>(1, E, proj_sb, proj_sl, B, L, A, dproj_sb, dproj_sl, accumulate_dproj, 0, 0, 0, 0,
                                    (hipStream_t)stream);
}

// rfn_attn_context_bwd_dalpha + rfn_attn_scores_bwd of `ngroups` encoders (shared (L, A, D) and strides) in one
// launch: dalpha stays in LDS; bit-identical to the pairs.
extern "C" int rfn_attn_bwd_grouped(int ngroups, const float* const* proj, int64_t proj_sb, int64_t proj_sl,
                                    const float* const* hproj, const float* const* w_out, const float* const* alpha,
                                    const float* const* att_seq, int64_t sb, int64_t sl, const float* const* dz,
                                    int64_t lddz, int B, int L, int A, int D, float* const* dproj, int64_t dproj_sb,
                                    int64_t dproj_sl, int accumulate_dproj, float* const* dhproj,
                                    float* const* dw_part, void* stream) {
    if (ngroups < 1 || ngroups > RFN_MAX_ENC || D <= 0) return RFN_ERR_SHAPE;
    if (!proj || !hproj || !w_out || !alpha || !att_seq || !dz || !dproj || !dhproj || !dw_part) return RFN_ERR_ARG;
    AttnEncPtrs E;
    memset(&E, 0, sizeof(E));
    for (int g = 0; g < ngroups; ++g) {
        E.proj[g] = proj[g]; E.hproj[g] = hproj[g]; E.w_out[g] = w_out[g]; E.alpha_in[g] = alpha[g];
        E.x[g] = att_seq[g]; E.dz[g] = dz[g];
        E.dproj[g] = dproj[g]; E.dhproj[g] = dhproj[g]; E.dw_part[g] = dw_part[g];
    }
    return launch_scores_bwd<true>(ngroups, E, proj_sb, proj_sl, B, L, A, dproj_sb, dproj_sl, accumulate_dproj, sb, sl,
                                   lddz, D, (hipStream_t)stream);
}
// The same launch with d proj delivered as bf16 planes into k-slow plane images (one per encoder; rfn_x3_split_ks layout
// with row pitch ks_mp, this call's A columns starting at column ks_col0) instead of f32: what the weight gradient
// d att_2_att_h.weight = dproj^T . att_seq on the bf16 matrix cores (rfn_x3_gemm_ks) reads.  A % 4 == 0, 16-B aligned
// contiguous-row operands only (RFN_ERR_SHAPE otherwise).
extern "C" int rfn_attn_bwd_grouped_ks(int ngroups, const float* const* proj, int64_t proj_sb, int64_t proj_sl,
                                       const float* const* hproj, const float* const* w_out, const float* const* alpha,
                                       const float* const* att_seq, int64_t sb, int64_t sl, const float* const* dz,
                                       int64_t lddz, int B, int L, int A, int D, void* const* ks_images, int ks_mp,
                                       int ks_col0, float* const* dhproj, float* const* dw_part, void* stream) {
    if (ngroups < 1 || ngroups > RFN_MAX_ENC || D <= 0 || ks_mp < 1 || ks_col0 < 0 || ks_col0 + A > ks_mp) return RFN_ERR_SHAPE;
    if (!proj || !hproj || !w_out || !alpha || !att_seq || !dz || !ks_images || !dhproj || !dw_part) return RFN_ERR_ARG;
    AttnEncPtrs E;
    memset(&E, 0, sizeof(E));
    for (int g = 0; g < ngroups; ++g) {
        E.proj[g] = proj[g]; E.hproj[g] = hproj[g]; E.w_out[g] = w_out[g]; E.alpha_in[g] = alpha[g];
        E.x[g] = att_seq[g]; E.dz[g] = dz[g];
        E.ks_img[g] = (char*)ks_images[g]; E.dhproj[g] = dhproj[g]; E.dw_part[g] = dw_part[g];
    }
    return launch_scores_bwd<true>(ngroups, E, proj_sb, proj_sl, B, L, A, 0, 0, 0, sb, sl, lddz, D, (hipStream_t)stream,
                                   ks_mp, ks_col0);
}
extern "C" int rfn_attn_bwd(const float* proj, int64_t proj_sb, int64_t proj_sl, const float* hproj, const float* w_out,
                            const float* alpha, const float* att_seq, int64_t sb, int64_t sl, const float* dz,
                            int64_t lddz, int B, int L, int A, int D, float* dproj, int64_t dproj_sb, int64_t dproj_sl,
                            int accumulate_dproj, float* dhproj, float* dw_part, void* stream) {
    return rfn_attn_bwd_grouped(1, &proj, proj_sb, proj_sl, &hproj, &w_out, &alpha, &att_seq, sb, sl, &dz, lddz, B, L,
                                A, D, &dproj, dproj_sb, dproj_sl, accumulate_dproj, &dhproj, &dw_part, stream);
}

// The two calls above for encoders whose maps differ in (L, D) (they share A): contiguous layouts -- proj / dproj
// (B, L_g, A), att_seq (B, L_g, D_g), z / dz (B, D_g), alpha and the raw-score scratch (B, L_g).  Same kernels, same
// per-row arithmetic as the per-encoder calls (bit-identical); the grid covers the largest map.
extern "C" int rfn_attn_fwd_het(int ngroups, const float* const* proj, const float* const* hproj, const float* const* w_out,
                                const float* const* b_out, const float* const* att_seq, int B, const int* L, int A,
                                const int* D, float* const* scores_scratch, float* const* alpha, float* const* z,
                                void* stream) {
    if (ngroups < 1 || ngroups > RFN_MAX_ENC) return RFN_ERR_SHAPE;
    if (!proj || !hproj || !w_out || !att_seq || !scores_scratch || !alpha || !z) return RFN_ERR_ARG;
    AttnDims Dm;
    RFN_TRY(attn_dims_het(ngroups, L, D, A, &Dm));
    AttnEncPtrs E;
    memset(&E, 0, sizeof(E));
    for (int g = 0; g < ngroups; ++g) {
        if (!scores_scratch[g] || !alpha[g] || scores_scratch[g] == alpha[g]) return RFN_ERR_ARG;
        E.proj[g] = proj[g];
        E.hproj[g] = hproj[g];
        E.w_out[g] = w_out[g];
        E.b_out[g] = b_out ? b_out[g] : nullptr;
        E.scores[g] = scores_scratch[g];
        E.x[g] = att_seq[g];
        E.alpha_in[g] = scores_scratch[g];
        E.alpha_out[g] = alpha[g];
        E.z[g] = z[g];
    }
    RFN_TRY(launch_scores_raw_d(ngroups, E, Dm, B, A, (hipStream_t)stream));
    return launch_context_d<true>(ngroups, E, Dm, B, (hipStream_t)stream);
}
extern "C" int rfn_attn_bwd_het(int ngroups, const float* const* proj, const float* const* hproj, const float* const* w_out,
                                const float* const* alpha, const float* const* att_seq, const float* const* dz, int B,
                                const int* L, int A, const int* D, float* const* dproj, int accumulate_dproj,
                                float* const* dhproj, float* const* dw_part, void* stream) {
    if (ngroups < 1 || ngroups > RFN_MAX_ENC) return RFN_ERR_SHAPE;
    if (!proj || !hproj || !w_out || !alpha || !att_seq || !dz || !dproj || !dhproj || !dw_part) return RFN_ERR_ARG;
    AttnDims Dm;
    RFN_TRY(attn_dims_het(ngroups, L, D, A, &Dm));
    AttnEncPtrs E;
    memset(&E, 0, sizeof(E));
    for (int g = 0; g < ngroups; ++g) {
        E.proj[g] = proj[g]; E.hproj[g] = hproj[g]; E.w_out[g] = w_out[g]; E.alpha_in[g] = alpha[g];
        E.x[g] = att_seq[g]; E.dz[g] = dz[g];
        E.dproj[g] = dproj[g]; E.dhproj[g] = dhproj[g]; E.dw_part[g] = dw_part[g];
    }
    return launch_scores_bwd_d<true>(ngroups, E, Dm, B, A, accumulate_dproj, (hipStream_t)stream);
}

// =====================================================================================================
// Fused small-L attention (stage II and the decoder attend over L = T1 / T2 = 8 thought vectors): the whole
// AttentionModelCore.forward of up to RFN_MAX_ENC encoders in ONE launch, and its whole backward in one.
// One block per (batch row, encoder): everything of a row fits in LDS / registers, so scores, softmax and
// context (resp. dalpha, softmax/tanh backward, d att_seq) never leave the CU.  Replaces 3 + 3 launches per
// encoder per cell step of the split kernels above, which remain for the large-L stage-I maps.
// =====================================================================================================
__global__ __launch_bounds__(ATT_THREADS) void attn_small_fwd_k(const AttnSmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    attn_small_fwd_body<false>(a, blockIdx.x, blockIdx.y, sm);
}
template <bool VEC>
__global__ __launch_bounds__(ATT_THREADS) void attn_small_bwd_k(const AttnSmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    attn_small_bwd_body<VEC, false>(a, blockIdx.x, blockIdx.y, sm);
}

// The launch arguments without the launch (rfn_internal.h): rfn_attn_small_fwd / _bwd = prepare + launch; the persistent
// recurrence kernels (rfn_chain.hip) run the prepared form of every step inside one launch.
int rfn_attn_small_prepare_fwd(int ngroups, const float* const* proj, int64_t proj_sb, int64_t proj_sl, const float* const* hproj,
                               const float* const* w_out, const float* const* b_out, const float* const* att_seq, int64_t sb,
                               int64_t sl, int B, int L, int A, int D, float* const* alpha, float* const* z, int64_t ldz,
                               AttnSmallPrepared* pz) {
    if (ngroups < 1 || ngroups > RFN_MAX_ENC || B <= 0 || L <= 0 || L > ATS_MAX_L || A <= 0 || D <= 0)
        return RFN_ERR_SHAPE;
    if (!proj || !hproj || !w_out || !att_seq || !alpha || !z) return RFN_ERR_ARG;
    memset(pz, 0, sizeof(*pz));
    AttnSmallArgs& a = pz->a;
    for (int g = 0; g < ngroups; ++g) {
        if (!proj[g] || !hproj[g] || !w_out[g] || !att_seq[g] || !alpha[g] || !z[g]) return RFN_ERR_ARG;
        a.proj[g] = proj[g]; a.hproj[g] = hproj[g]; a.w_out[g] = w_out[g];
        a.b_out[g] = b_out ? b_out[g] : nullptr;
        a.x[g] = att_seq[g]; a.alpha[g] = alpha[g]; a.z[g] = z[g];
    }
    a.psb = proj_sb; a.psl = proj_sl; a.xsb = sb; a.xsl = sl; a.ldz = ldz;
    a.L = L; a.A = A; a.D = D;
    pz->lds = (size_t)(2 * ((A + 3) & ~3) + L) * sizeof(float);
    if (pz->lds > 64 * 1024) return RFN_ERR_SHAPE;
    pz->B = B; pz->ngroups = ngroups; pz->backward = 0; pz->vec = 0;
    return RFN_OK;
}

int rfn_attn_small_prepare_bwd(int ngroups, const float* const* proj, int64_t proj_sb, int64_t proj_sl, const float* const* hproj,
                               const float* const* w_out, const float* const* alpha, const float* const* att_seq, int64_t sb,
                               int64_t sl, const float* const* dz, int64_t lddz, int B, int L, int A, int D, float* const* dproj,
                               int64_t dproj_sb, int64_t dproj_sl, int accumulate_dproj, float* const* dhproj,
                               float* const* dw_part, float* const* datt_seq, AttnSmallPrepared* pz) {
    if (ngroups < 1 || ngroups > RFN_MAX_ENC || B <= 0 || L <= 0 || L > ATS_MAX_L || A <= 0 || D <= 0)
        return RFN_ERR_SHAPE;
    if (!proj || !hproj || !w_out || !alpha || !att_seq || !dz || !dproj || !dhproj || !dw_part) return RFN_ERR_ARG;
    memset(pz, 0, sizeof(*pz));
    AttnSmallArgs& a = pz->a;
    for (int g = 0; g < ngroups; ++g) {
        if (!proj[g] || !hproj[g] || !w_out[g] || !alpha[g] || !att_seq[g] || !dz[g] || !dproj[g] || !dhproj[g] ||
            !dw_part[g])
            return RFN_ERR_ARG;
        a.proj[g] = proj[g]; a.hproj[g] = hproj[g]; a.w_out[g] = w_out[g];
        a.x[g] = att_seq[g]; a.alpha[g] = const_cast<float*>(alpha[g]); a.z[g] = const_cast<float*>(dz[g]);
        a.dproj[g] = dproj[g]; a.dhproj[g] = dhproj[g]; a.dw_part[g] = dw_part[g];
        a.dx[g] = datt_seq ? datt_seq[g] : nullptr;
    }
    a.psb = proj_sb; a.psl = proj_sl; a.xsb = sb; a.xsl = sl; a.ldz = lddz; a.dpsb = dproj_sb; a.dpsl = dproj_sl;
    a.L = L; a.A = A; a.D = D; a.accumulate_dproj = accumulate_dproj;
    pz->lds = (size_t)(2 * ((A + 3) & ~3) + ((D + 3) & ~3) + 2 * ((L + 3) & ~3)) * sizeof(float);
    if (pz->lds > 64 * 1024) return RFN_ERR_SHAPE;
    bool vec = (A % 4 == 0) && (D % 4 == 0) && ((proj_sb | proj_sl | sb | sl | lddz | dproj_sb | dproj_sl) % 4 == 0);
    for (int g = 0; g < ngroups; ++g)
        vec = vec && rfn_aligned16(proj[g]) && rfn_aligned16(att_seq[g]) && rfn_aligned16(dproj[g]) &&
              rfn_aligned16(dhproj[g]) && rfn_aligned16(dw_part[g]) && (!datt_seq || rfn_aligned16(datt_seq[g]));
    pz->B = B; pz->ngroups = ngroups; pz->backward = 1; pz->vec = vec ? 1 : 0;
    return RFN_OK;
}

int rfn_attn_small_launch(const AttnSmallPrepared& pz, void* stream) {
    const dim3 grid(pz.B, pz.ngroups);
    if (!pz.backward)
        hipLaunchKernelGGL(attn_small_fwd_k, grid, dim3(ATT_THREADS), pz.lds, (hipStream_t)stream, pz.a);
    else if (pz.vec)
        hipLaunchKernelGGL(attn_small_bwd_k<true>, grid, dim3(ATT_THREADS), pz.lds, (hipStream_t)stream, pz.a);
    else
        hipLaunchKernelGGL(attn_small_bwd_k<false>, grid, dim3(ATT_THREADS), pz.lds, (hipStream_t)stream, pz.a);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}

extern "C" int rfn_attn_small_fwd(int ngroups, const float* const* proj, int64_t proj_sb, int64_t proj_sl,
                                  const float* const* hproj, const float* const* w_out, const float* const* b_out,
                                  const float* const* att_seq, int64_t sb, int64_t sl, int B, int L, int A, int D,
                                  float* const* alpha, float* const* z, int64_t ldz, void* stream) {
    AttnSmallPrepared pz;
    RFN_TRY(rfn_attn_small_prepare_fwd(ngroups, proj, proj_sb, proj_sl, hproj, w_out, b_out, att_seq, sb, sl, B, L, A, D, alpha, z,
                                       ldz, &pz));
    return rfn_attn_small_launch(pz, stream);
}

extern "C" int rfn_attn_small_bwd(int ngroups, const float* const* proj, int64_t proj_sb, int64_t proj_sl,
                                  const float* const* hproj, const float* const* w_out, const float* const* alpha,
                                  const float* const* att_seq, int64_t sb, int64_t sl, const float* const* dz,
                                  int64_t lddz, int B, int L, int A, int D, float* const* dproj, int64_t dproj_sb,
                                  int64_t dproj_sl, int accumulate_dproj, float* const* dhproj, float* const* dw_part,
                                  float* const* datt_seq, void* stream) {
    AttnSmallPrepared pz;
    RFN_TRY(rfn_attn_small_prepare_bwd(ngroups, proj, proj_sb, proj_sl, hproj, w_out, alpha, att_seq, sb, sl, dz, lddz, B, L, A, D,
                                       dproj, dproj_sb, dproj_sl, accumulate_dproj, dhproj, dw_part, datt_seq, &pz));
    return rfn_attn_small_launch(pz, stream);
}
