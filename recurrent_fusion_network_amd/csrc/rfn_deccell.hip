// Decoder cell with z2h hoisted through the attention (round 6).
//
// Reference: misc/LSTMSoftAttentionCore.py:60-102.  Per step the reference computes
//     alpha = softmax_l( att_h_2_out( tanh( att_2_att_h(v_l) + h_2_att_h(h) ) ) )          (:64-77)
//     z     = sum_l alpha_l v_l                                                             (:78-79)
//     gates = i2h(x) + h2h(h) + z2h(z)                                                      (:81)
// over the SAME T2 thought vectors v_l with the SAME z2h weights on every step, and nothing but a convex combination
// sits between v and z2h.  z2h is linear, so
//     z2h(z) = b_z + sum_l alpha_l (W_z v_l) = b_z + sum_l alpha_l U_l,      U = thought_vectors_comb . W_z^T   (once per call)
// and the per-step product K3 = z2h(z) -- one of the three dependent launches of a step -- disappears: a step is
//     K1  [h_2_att_h(h) | gates += h2h(h)]                       (rfn_cell_gemm, as before)
//     K2  scores, softmax, gates += b_z + sum_l alpha_l U_l, LSTM update          (dec_cell_fwd_k, this file)
// Backward: with z gone, the attention backward needs only d gates of its own step (d alpha_l = <d gates, U_l>), and
// d h = d gates . W_hh + d hproj . W_h splits into a long part that does NOT depend on the attention backward and a short one
// that does.  A step is two launches (rfn_path.hip, rfn_decoder_bwd):
//     X   the attention-backward rows (dec_attn_bwd_fast_body) BESIDE the tiles of d gates . W_hh cut 4 ways along K into
//         partial slabs, in one grid (cell_gemm_rows_k, rfn_cellgemm.hip)
//     Y   d hproj . W_h + the slabs, with the LSTM backward of the step below as epilogue (rfn_cell_gemm, acc_slabs)
// After the loop d U = sum_s alpha_s (x) d gates_s (dec_du_k), then d thoughts = d Pd . W_a + d U . W_z (one GEMM, two K
// segments) and d W_z = d U^T . thoughts over T2*B rows.
//
// Rows are independent and a row's arithmetic does not depend on the launch shape: every block of a row recomputes the row's
// scores with the same lane partition (row_tanh_dot) and the same serial softmax, the gate sums run over l in order.
#include "rfn_attn_small_body.h"
#include "rfn_common.h"
#include "rfn_deccell_body.h"

struct DecCellArgs {
    const float* proj;     // att_2_att_h(v): (b', l, :) at proj + b' * psb + l * psl, b' = b / row_div
    const float* hproj;    // h_2_att_h(h): (B, A)
    const float* w_out;    // (A)
    const float* b_out;    // (1) or NULL
    const float* U;        // W_z v: (b', l, :) at U + b' * usb + l * usl, NG * R wide, no bias
    const float* bz;       // (NG * R) z2h bias
    float* gates;          // (B, NG * R) row stride ldg: in i2h(x) + h2h(h), out the gate activations
    const float* c_prev;
    float* c_next;
    float* h_next;
    float* alpha;          // (B, L)
    long psb, psl, usb, usl, ldg, ldcp, ldcn, ldh;
    int B, L, A, R, row_div;
    float drop_p;
    uint64_t seed, drop_offset;
};


template <int NG>
__global__ __launch_bounds__(256) void dec_cell_fwd_k(const DecCellArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int A = a.A, L = a.L, R = a.R, Ap = (A + 3) & ~3;
    float* hp_s = sm;
    float* w_s = sm + Ap;
    float* s_s = sm + 2 * Ap;   // [L] scores
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nthr >> 6;
    const int b = blockIdx.y, pb = b / a.row_div;
    const int unit = blockIdx.x * nthr + tid;
    const bool live = unit < R;
    // ---- everything the gate phase reads is requested first: it lands under the score phase ------------------------------
    const float* Ub = a.U + (long)pb * a.usb;
    float uv[DEC_LREG][NG];
#pragma unroll
    for (int l = 0; l < DEC_LREG; ++l)
#pragma unroll
        for (int g = 0; g < NG; ++g) uv[l][g] = (live && l < L) ? Ub[l * a.usl + g * R + unit] : 0.f;
    float gin[NG], bzv[NG];
    float* G = a.gates + (long)b * a.ldg;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        gin[g] = live ? G[g * R + unit] : 0.f;
        bzv[g] = live ? a.bz[g * R + unit] : 0.f;
    }
    const float cprev = live ? a.c_prev[(long)b * a.ldcp + unit] : 0.f;
    // ---- scores (AttentionModelCore semantics, the arithmetic of attn_small_fwd_body) -------------------------------------
    for (int i = tid; i < A; i += nthr) {
        hp_s[i] = a.hproj[(long)b * A + i];
        w_s[i] = a.w_out[i];
    }
    __syncthreads();
    const float* proj = a.proj + (long)pb * a.psb;
    const float bo = a.b_out ? a.b_out[0] : 0.f;
    const bool vecA = (A % 4 == 0) && ((a.psb | a.psl) % 4 == 0) && ((((uintptr_t)a.proj) & 15) == 0);
    for (int l = wave; l < L; l += nw) {
        const float s = (vecA ? row_tanh_dot<true>(proj + l * a.psl, hp_s, w_s, A, lane)
                              : row_tanh_dot<false>(proj + l * a.psl, hp_s, w_s, A, lane)) + bo;
        if (lane == 0) s_s[l] = s;
    }
    __syncthreads();
    float m = -INFINITY, sum = 0.f;
    for (int l = 0; l < L; ++l) m = fmaxf(m, s_s[l]);
    for (int l = 0; l < L; ++l) sum += expf(s_s[l] - m);
    const float inv = 1.0f / sum;
    if (blockIdx.x == 0)
        for (int l = tid; l < L; l += nthr) a.alpha[(long)b * L + l] = expf(s_s[l] - m) * inv;
    if (!live) return;
    // ---- gates = (sum_l alpha_l U_l + b_z) + (i2h + h2h), l in order -----------------------------------------------------
    float acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = 0.f;
#pragma unroll
    for (int l = 0; l < DEC_LREG; ++l) {
        if (l < L) {
            const float al = expf(s_s[l] - m) * inv;
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g] += al * uv[l][g];
        }
    }
    for (int l = DEC_LREG; l < L; ++l) {
        const float al = expf(s_s[l] - m) * inv;
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[g] += al * Ub[l * a.usl + g * R + unit];
    }
    float pre[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) pre[g] = (acc[g] + bzv[g]) + gin[g];
    // ---- LSTM update (rfn_cell.hip lstm_fwd_k, same formulas; :83-101) ----------------------------------------------------
    const float ig = rfn_sigmoid(pre[0]), fg = rfn_sigmoid(pre[1]), og = rfn_sigmoid(pre[2]);
    float gg;
    if constexpr (NG == 5) {   // maxout: max of the two candidate chunks, no tanh; chunk 4 keeps the selector
        const float x = pre[3], y = pre[4];
        gg = fmaxf(x, y);
        G[4 * R + unit] = (x > y) ? 1.f : ((x == y) ? 0.5f : 0.f);
    } else {
        gg = tanhf(pre[3]);
    }
    G[unit] = ig;
    G[R + unit] = fg;
    G[2 * R + unit] = og;
    G[3 * R + unit] = gg;
    const float c = fg * cprev + ig * gg;
    a.c_next[(long)b * a.ldcn + unit] = c;
    float hv = og * tanhf(c);
    if (a.drop_p > 0.f) {
        const float u = rfn_philox_uniform(a.seed, a.drop_offset, (uint64_t)((long)b * R + unit));
        hv = (u >= a.drop_p) ? hv * (1.0f / (1.0f - a.drop_p)) : 0.f;
    }
    a.h_next[(long)b * a.ldh + unit] = hv;
}

// The same cell for the shapes the path runs at (A <= 512 in 16-B chunks, L <= 8, R a multiple of 256): 256 threads per block
// whatever the block's share of the row (UB = 256 / TPU units, TPU threads per unit taking the gates g = k, k + TPU, ...), and
// EVERYTHING the block reads -- its waves' projection rows, hproj and w_out in score-fragment order, the U values, the incoming
// gate sums -- requested before the first dependent instruction: two barriers, one global-memory latency.  Same arithmetic as
// dec_cell_fwd_k element for element (row_tanh_dot<true>'s lane partition and order, serial softmax, l-ordered gate sums), so a
// row's bits depend neither on the kernel nor on TPU.
template <int NG, int TPU>
__global__ __launch_bounds__(256) void dec_cell_fwd_fast_k(const DecCellArgs a) {
    constexpr int UB = 256 / TPU, NGT = (NG + TPU - 1) / TPU;
    __shared__ float s_s[DEC_LREG];
    __shared__ float pre_s[(TPU > 1) ? NG * UB : 1];
    const int A = a.A, L = a.L, R = a.R;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y, pb = b / a.row_div;
    const int u_loc = tid % UB, k = tid / UB;
    const int unit = blockIdx.x * UB + u_loc;
    const float* Ub = a.U + (long)pb * a.usb;
    float* G = a.gates + (long)b * a.ldg;
    // ---- requests ---------------------------------------------------------------------------------------------------------
    float uv[DEC_LREG][NGT], gin[NGT], bzv[NGT];
#pragma unroll
    for (int j = 0; j < NGT; ++j) {
        const int g = k + j * TPU;
        const bool ok = g < NG;
#pragma unroll
        for (int l = 0; l < DEC_LREG; ++l) uv[l][j] = (ok && l < L) ? Ub[l * a.usl + g * R + unit] : 0.f;
        gin[j] = ok ? G[g * R + unit] : 0.f;
        bzv[j] = ok ? a.bz[g * R + unit] : 0.f;
    }
    const float cprev = (k == 0) ? a.c_prev[(long)b * a.ldcp + unit] : 0.f;
    const float* proj = a.proj + (long)pb * a.psb;
    const float* hp = a.hproj + (long)b * A;
    f32x4 hh[2], ww[2], pv[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int c = lane * 4 + 256 * q;
        const bool ok = c < A;
        hh[q] = ok ? *reinterpret_cast<const f32x4*>(hp + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        ww[q] = ok ? *reinterpret_cast<const f32x4*>(a.w_out + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int l = wave + ATT_WAVES * j;
            pv[j][q] = (ok && l < L) ? *reinterpret_cast<const f32x4*>(proj + l * a.psl + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const float bo = a.b_out ? a.b_out[0] : 0.f;
    // ---- scores: row_tanh_dot<true>'s elements in its order ----------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int l = wave + ATT_WAVES * j;
        if (l < L) {
            float part = 0.f;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (lane * 4 + 256 * q < A) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) part += rfn_tanh_fast(pv[j][q][e] + hh[q][e]) * ww[q][e];
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
    if (blockIdx.x == 0 && tid < L) a.alpha[(long)b * L + tid] = expf(s_s[tid] - m) * inv;
    // ---- gates = (sum_l alpha_l U_l + b_z) + (i2h + h2h), l in order -----------------------------------------------------
    float pre[NGT];
#pragma unroll
    for (int j = 0; j < NGT; ++j) pre[j] = 0.f;
#pragma unroll
    for (int l = 0; l < DEC_LREG; ++l) {
        if (l < L) {
            const float al = expf(s_s[l] - m) * inv;
#pragma unroll
            for (int j = 0; j < NGT; ++j) pre[j] += al * uv[l][j];
        }
    }
#pragma unroll
    for (int j = 0; j < NGT; ++j) pre[j] = (pre[j] + bzv[j]) + gin[j];
    float pg[NG];
    if constexpr (TPU > 1) {
#pragma unroll
        for (int j = 0; j < NGT; ++j) {
            const int g = k + j * TPU;
            if (g < NG) pre_s[g * UB + u_loc] = pre[j];
        }
        __syncthreads();
        if (k != 0) return;
#pragma unroll
        for (int g = 0; g < NG; ++g) pg[g] = pre_s[g * UB + u_loc];
    } else {
#pragma unroll
        for (int g = 0; g < NG; ++g) pg[g] = pre[g];
    }
    // ---- LSTM update (rfn_cell.hip lstm_fwd_k, same formulas; :83-101) ----------------------------------------------------
    const float ig = rfn_sigmoid(pg[0]), fg = rfn_sigmoid(pg[1]), og = rfn_sigmoid(pg[2]);
    float gg;
    if constexpr (NG == 5) {
        const float x = pg[3], y = pg[4];
        gg = fmaxf(x, y);
        G[4 * R + unit] = (x > y) ? 1.f : ((x == y) ? 0.5f : 0.f);
    } else {
        gg = tanhf(pg[3]);
    }
    G[unit] = ig;
    G[R + unit] = fg;
    G[2 * R + unit] = og;
    G[3 * R + unit] = gg;
    const float c = fg * cprev + ig * gg;
    a.c_next[(long)b * a.ldcn + unit] = c;
    float hv = og * tanhf(c);
    if (a.drop_p > 0.f) {
        const float u = rfn_philox_uniform(a.seed, a.drop_offset, (uint64_t)((long)b * R + unit));
        hv = (u >= a.drop_p) ? hv * (1.0f / (1.0f - a.drop_p)) : 0.f;
    }
    a.h_next[(long)b * a.ldh + unit] = hv;
}
template <int NG>
static void dec_cell_fwd_fast_launch(const DecCellArgs& a, int tpu, hipStream_t st) {
    const dim3 grid(a.R / (256 / tpu), a.B);
    if (tpu == 4) hipLaunchKernelGGL((dec_cell_fwd_fast_k<NG, 4>), grid, dim3(256), 0, st, a);
    else if (tpu == 2) hipLaunchKernelGGL((dec_cell_fwd_fast_k<NG, 2>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dec_cell_fwd_fast_k<NG, 1>), grid, dim3(256), 0, st, a);
}

#ifndef DEC_TPU_MIN
#define DEC_TPU_MIN 1          /* A/B knobs (tools/build_variant.sh): least threads per unit, blocks wanted per CU */
#endif
#ifndef DEC_BLOCKS_PER_CU
#define DEC_BLOCKS_PER_CU 1
#endif
static int dec_device_cus() {
    static int cus[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int& c = cus[dev & 15];
    if (c == 0 && hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) c = 256;
    return c > 0 ? c : 256;
}

extern "C" int rfn_dec_cell_fwd(const float* proj, int64_t psb, int64_t psl, const float* hproj, const float* w_out,
                                const float* b_out, const float* U, int64_t usb, int64_t usl, const float* bz, float* gates,
                                int64_t ldg, const float* c_prev, int64_t ldcp, float* c_next, int64_t ldcn, float* h_next,
                                int64_t ldh, float* alpha, int B, int L, int A, int R, int maxout, int row_div, float drop_p,
                                uint64_t seed, uint64_t drop_offset, void* stream) {
    if (B <= 0 || L <= 0 || L > ATS_MAX_L || A <= 0 || R <= 0 || row_div < 1 || drop_p < 0.f || drop_p >= 1.f) return RFN_ERR_SHAPE;
    if (!proj || !hproj || !w_out || !U || !bz || !gates || !c_prev || !c_next || !h_next || !alpha) return RFN_ERR_ARG;
    DecCellArgs a;
    a.proj = proj; a.hproj = hproj; a.w_out = w_out; a.b_out = b_out; a.U = U; a.bz = bz; a.gates = gates;
    a.c_prev = c_prev; a.c_next = c_next; a.h_next = h_next; a.alpha = alpha;
    a.psb = psb; a.psl = psl; a.usb = usb; a.usl = usl; a.ldg = ldg; a.ldcp = ldcp; a.ldcn = ldcn; a.ldh = ldh;
    a.B = B; a.L = L; a.A = A; a.R = R; a.row_div = row_div; a.drop_p = drop_p; a.seed = seed; a.drop_offset = drop_offset;
    // units per block: the widest block that still gives every CU one (a row's result does not depend on the choice)
    const int cus = dec_device_cus();
    const bool fast = A % 4 == 0 && A <= 512 && L <= DEC_LREG && R % 256 == 0 && (psb | psl) % 4 == 0 && rfn_aligned16(proj) &&
                      rfn_aligned16(hproj) && rfn_aligned16(w_out);
    if (fast) {
        int tpu = DEC_TPU_MIN;
        while (tpu < 4 && (long)B * (R / (256 / tpu)) < DEC_BLOCKS_PER_CU * cus) tpu <<= 1;
        if (maxout) dec_cell_fwd_fast_launch<5>(a, tpu, (hipStream_t)stream);
        else dec_cell_fwd_fast_launch<4>(a, tpu, (hipStream_t)stream);
        RFN_CHECK_LAUNCH();
        return RFN_OK;
    }
    int ub = 256;
    while (ub > 64 && ((long)B * rfn_cdiv(R, ub) < cus || ub / 2 >= R)) ub >>= 1;
    const size_t lds = (size_t)(2 * ((A + 3) & ~3) + L) * sizeof(float);
    if (lds > 64 * 1024) return RFN_ERR_SHAPE;
    const dim3 grid(rfn_cdiv(R, ub), B);
    if (maxout) hipLaunchKernelGGL(dec_cell_fwd_k<5>, grid, dim3(ub), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(dec_cell_fwd_k<4>, grid, dim3(ub), lds, (hipStream_t)stream, a);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Attention backward of one step with the context hoisted: d alpha_l = <d gates, U_l>, softmax backward, tanh backward over
// the (L, A) slice -> d proj (accumulated across the steps: every step reads the same projection), d hproj, the per-row part
// of d att_h_2_out.weight.  One block per batch row.
// ---------------------------------------------------------------------------------------------------------------------------

template <bool VEC, bool VECU>   // VEC: 16-B accesses on the (L, A) side; VECU: on the GD-wide rows of U / d gates
__global__ __launch_bounds__(ATT_THREADS) void dec_attn_bwd_k(const DecAttnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int A = a.A, L = a.L, GD = a.GD, Ap = (A + 3) & ~3, Lp = (L + 3) & ~3;
    float* hp_s = sm;               // [Ap]
    float* w_s = sm + Ap;           // [Ap]
    float* al_s = sm + 2 * Ap;      // [Lp]
    float* ds_s = al_s + Lp;        // [Lp]
    float* red = ds_s + Lp;         // [ATT_WAVES][DEC_LREG]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const float* U = a.U + (long)b * a.usb;
    const float* dg = a.dgates + (long)b * a.ldg;
    for (int i = tid; i < A; i += ATT_THREADS) {
        hp_s[i] = a.hproj[(long)b * A + i];
        w_s[i] = a.w_out[i];
    }
    for (int l = tid; l < L; l += ATT_THREADS) al_s[l] = a.alpha[(long)b * L + l];
    // d alpha: every thread takes the same 16-B column chunks of all rows of a group of DEC_LREG thought vectors
    for (int l0 = 0; l0 < L; l0 += DEC_LREG) {
        float p[DEC_LREG];
#pragma unroll
        for (int j = 0; j < DEC_LREG; ++j) p[j] = 0.f;
        if constexpr (VECU) {
            for (int c = 4 * tid; c < GD; c += 4 * ATT_THREADS) {
                const f32x4 gv = *reinterpret_cast<const f32x4*>(dg + c);
                f32x4 uv[DEC_LREG];
#pragma unroll
                for (int j = 0; j < DEC_LREG; ++j) {
                    const int l = (l0 + j < L) ? l0 + j : L - 1;
                    uv[j] = *reinterpret_cast<const f32x4*>(U + l * a.usl + c);
                }
#pragma unroll
                for (int j = 0; j < DEC_LREG; ++j)
                    p[j] += (uv[j][0] * gv[0] + uv[j][1] * gv[1]) + (uv[j][2] * gv[2] + uv[j][3] * gv[3]);
            }
        } else {
            for (int c = tid; c < GD; c += ATT_THREADS) {
                const float gv = dg[c];
#pragma unroll
                for (int j = 0; j < DEC_LREG; ++j) {
                    const int l = (l0 + j < L) ? l0 + j : L - 1;
                    p[j] += U[l * a.usl + c] * gv;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < DEC_LREG; ++j) {
            p[j] = rfn_wave_sum(p[j]);
            if (lane == 0) red[wave * DEC_LREG + j] = p[j];
        }
        __syncthreads();
        if (tid < DEC_LREG && l0 + tid < L) {
            float s = red[tid];
#pragma unroll
            for (int w = 1; w < ATT_WAVES; ++w) s += red[w * DEC_LREG + tid];
            ds_s[l0 + tid] = s;
        }
        __syncthreads();
    }
    float dot = 0.f;
    for (int l = 0; l < L; ++l) dot += al_s[l] * ds_s[l];
    __syncthreads();
    for (int l = tid; l < L; l += ATT_THREADS) ds_s[l] = al_s[l] * (ds_s[l] - dot);   // softmax backward
    __syncthreads();
    // tanh backward over the (L, A) slice (the arithmetic of attn_small_bwd_body): rows in order
    const float* proj = a.proj + (long)b * a.psb;
    float* dproj = a.dproj + (long)b * a.dpsb;
    const bool acc = a.accumulate != 0;
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
                    if (acc) ov[j] = *reinterpret_cast<const f32x4*>(dproj + l * a.dpsl + i);
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
                    *reinterpret_cast<f32x4*>(dproj + l * a.dpsl + i) = acc ? ov[j] + dpre : dpre;
                }
            }
            *reinterpret_cast<f32x4*>(a.dhproj + (long)b * A + i) = ah;
            *reinterpret_cast<f32x4*>(a.dw_part + (long)b * A + i) = aw;
        }
    } else {
        for (int i = tid; i < A; i += ATT_THREADS) {
            const float hh = hp_s[i], ww = w_s[i];
            float ah = 0.f, aw = 0.f;
            for (int l = 0; l < L; ++l) {
                const float t = rfn_tanh_fast(proj[l * a.psl + i] + hh);
                const float dpre = ds_s[l] * ww * (1.0f - t * t);
                float* o = dproj + l * a.dpsl + i;
                *o = acc ? *o + dpre : dpre;
                ah += dpre;
                aw += ds_s[l] * t;
            }
            a.dhproj[(long)b * A + i] = ah;
            a.dw_part[(long)b * A + i] = aw;
        }
    }
}

__global__ __launch_bounds__(ATT_THREADS) void dec_attn_bwd_fast_k(const DecAttnBwdArgs a) { dec_attn_bwd_fast_body(a, blockIdx.x); }

// which kernel serves the call (the fast body's shapes): also asked by the fused launch of rfn_cellgemm.hip
bool rfn_dec_attn_bwd_fast_ok(const DecAttnBwdArgs& a) {
    return a.A % 4 == 0 && a.A <= 512 && a.L <= DEC_LREG && a.GD % 4 == 0 && (a.psb | a.psl | a.dpsb | a.dpsl | a.usb | a.usl | a.ldg) % 4 == 0 &&
           rfn_aligned16(a.proj) && rfn_aligned16(a.dproj) && rfn_aligned16(a.dhproj) && rfn_aligned16(a.dw_part) && rfn_aligned16(a.U) &&
           rfn_aligned16(a.dgates) && rfn_aligned16(a.hproj) && rfn_aligned16(a.w_out);
}
int rfn_dec_attn_bwd_args(const float* proj, int64_t psb, int64_t psl, const float* hproj, const float* w_out, const float* alpha,
                          const float* U, int64_t usb, int64_t usl, const float* dgates, int64_t ldg, int B, int L, int A, int GD,
                          float* dproj, int64_t dpsb, int64_t dpsl, int accumulate, float* dhproj, float* dw_part, DecAttnBwdArgs* out) {
    if (B <= 0 || L <= 0 || L > ATS_MAX_L || A <= 0 || GD <= 0) return RFN_ERR_SHAPE;
    if (!proj || !hproj || !w_out || !alpha || !U || !dgates || !dproj || !dhproj || !dw_part) return RFN_ERR_ARG;
    DecAttnBwdArgs& a = *out;
    a.proj = proj; a.hproj = hproj; a.w_out = w_out; a.alpha = alpha; a.U = U; a.dgates = dgates;
    a.dproj = dproj; a.dhproj = dhproj; a.dw_part = dw_part;
    a.psb = psb; a.psl = psl; a.usb = usb; a.usl = usl; a.ldg = ldg; a.dpsb = dpsb; a.dpsl = dpsl;
    a.L = L; a.A = A; a.GD = GD; a.accumulate = accumulate;
    return RFN_OK;
}

extern "C" int rfn_dec_attn_bwd(const float* proj, int64_t psb, int64_t psl, const float* hproj, const float* w_out,
                                const float* alpha, const float* U, int64_t usb, int64_t usl, const float* dgates, int64_t ldg,
                                int B, int L, int A, int GD, float* dproj, int64_t dpsb, int64_t dpsl, int accumulate,
                                float* dhproj, float* dw_part, void* stream) {
    DecAttnBwdArgs a;
    RFN_TRY(rfn_dec_attn_bwd_args(proj, psb, psl, hproj, w_out, alpha, U, usb, usl, dgates, ldg, B, L, A, GD, dproj, dpsb, dpsl,
                                  accumulate, dhproj, dw_part, &a));
    const bool vecu = GD % 4 == 0 && rfn_aligned16(U) && rfn_aligned16(dgates) && (usb | usl | ldg) % 4 == 0;
    const size_t lds = (size_t)(2 * ((A + 3) & ~3) + 2 * ((L + 3) & ~3) + ATT_WAVES * DEC_LREG) * sizeof(float);
    if (lds > 64 * 1024) return RFN_ERR_SHAPE;
    const bool vec = (A % 4 == 0) && ((psb | psl | dpsb | dpsl) % 4 == 0) && rfn_aligned16(proj) && rfn_aligned16(dproj) &&
                     rfn_aligned16(dhproj) && rfn_aligned16(dw_part);
    if (rfn_dec_attn_bwd_fast_ok(a))
        hipLaunchKernelGGL(dec_attn_bwd_fast_k, dim3(B), dim3(ATT_THREADS), 0, (hipStream_t)stream, a);
    else if (vec && vecu) hipLaunchKernelGGL((dec_attn_bwd_k<true, true>), dim3(B), dim3(ATT_THREADS), lds, (hipStream_t)stream, a);
    else if (vecu) hipLaunchKernelGGL((dec_attn_bwd_k<false, true>), dim3(B), dim3(ATT_THREADS), lds, (hipStream_t)stream, a);
    else if (vec) hipLaunchKernelGGL((dec_attn_bwd_k<true, false>), dim3(B), dim3(ATT_THREADS), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((dec_attn_bwd_k<false, false>), dim3(B), dim3(ATT_THREADS), lds, (hipStream_t)stream, a);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// d U after the loop: dU[b, l, :] = sum_s alpha[s, b, l] * dgates[s, b, :], s in order.  Reads the stored gate gradients
// once; one thread per 16-B column chunk of a row.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dec_du_k(const float* __restrict__ alpha, const float* __restrict__ dgates, int S, int B,
                                                int L, int GD, float* __restrict__ dU, long usb, long usl) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [S][L] alpha of this row
    const int b = blockIdx.y, tid = threadIdx.x;
    for (int i = tid; i < S * L; i += 256) {
        const int s = i / L, l = i - s * L;
        sm[i] = alpha[((long)s * B + b) * L + l];
    }
    __syncthreads();
    const int c = 4 * (blockIdx.x * 256 + tid);
    if (c >= GD) return;
    for (int l0 = 0; l0 < L; l0 += DEC_LREG) {
        f32x4 acc[DEC_LREG];
#pragma unroll
        for (int j = 0; j < DEC_LREG; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < S; ++s) {
            const f32x4 gv = *reinterpret_cast<const f32x4*>(dgates + ((long)s * B + b) * GD + c);
#pragma unroll
            for (int j = 0; j < DEC_LREG; ++j) {
                const float al = (l0 + j < L) ? sm[s * L + l0 + j] : 0.f;
                acc[j] += al * gv;
            }
        }
#pragma unroll
        for (int j = 0; j < DEC_LREG; ++j)
            if (l0 + j < L) *reinterpret_cast<f32x4*>(dU + (long)b * usb + (long)(l0 + j) * usl + c) = acc[j];
    }
}

__global__ __launch_bounds__(256) void dec_du_scalar_k(const float* __restrict__ alpha, const float* __restrict__ dgates, int S,
                                                       int B, int L, int GD, float* __restrict__ dU, long usb, long usl) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= GD) return;
    for (int l = 0; l < L; ++l) {
        float acc = 0.f;
        for (int s = 0; s < S; ++s) acc += alpha[((long)s * B + b) * L + l] * dgates[((long)s * B + b) * GD + c];
        dU[(long)b * usb + (long)l * usl + c] = acc;
    }
}

extern "C" int rfn_dec_du(const float* alpha, const float* dgates, int S, int B, int L, int GD, float* dU, int64_t usb,
                          int64_t usl, void* stream) {
    if (S <= 0 || B <= 0 || L <= 0 || GD <= 0) return RFN_ERR_SHAPE;
    if (!alpha || !dgates || !dU) return RFN_ERR_ARG;
    if (GD % 4 || !rfn_aligned16(dgates) || !rfn_aligned16(dU) || (usb | usl) % 4) {
        hipLaunchKernelGGL(dec_du_scalar_k, dim3(rfn_cdiv(GD, 256), B), dim3(256), 0, (hipStream_t)stream, alpha, dgates, S, B, L, GD,
                           dU, (long)usb, (long)usl);
        RFN_CHECK_LAUNCH();
        return RFN_OK;
    }
    const size_t lds = (size_t)S * L * sizeof(float);
    if (lds > 64 * 1024) return RFN_ERR_SHAPE;
    hipLaunchKernelGGL(dec_du_k, dim3(rfn_cdiv(GD, 1024), B), dim3(256), lds, (hipStream_t)stream, alpha, dgates, S, B, L, GD, dU,
                       (long)usb, (long)usl);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}
