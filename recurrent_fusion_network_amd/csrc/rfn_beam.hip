// Batched, device-resident beam search bookkeeping for sample_beam (misc/RecurrentFusionModel.py:451-531).
//
// The reference searches one image at a time with a full-vocabulary sort on the host and Python lists.  Here all
// images' beams share one decoder batch (NB*W rows) and one block per image does the step's bookkeeping on the
// device, reproducing the reference exactly:
//   * per beam row the top min(W, V+1) log-probs in descending order (only those columns are ever used, :466);
//   * candidates in (sorted column c outer, beam q inner) order, skipping beams that already emitted END (:470-478);
//     at t == 1 only row 0 is live (:468-469);
//   * STABLE descending sort by cumulative log-prob p (Python's sorted, :482), fp32 sums as the reference's tensors;
//   * fork bookkeeping of beam_seq / beam_seq_logprobs / beam_logprobs_sum and the recurrent-state gather (:491-506);
//   * done beams appended in construction order when END is emitted or t == seq_length (:508-514);
//   * an image with no candidate left stops (:480-481).
// Nothing is read back per step; the host sorts the (few) done beams once at the end.
#include "rfn_common.h"

#define BEAM_MAX_W 16

__global__ __launch_bounds__(64 * BEAM_MAX_W) void beam_step_k(
    const float* __restrict__ logp, long ldl, int V1, int W, int S, int t, int NB, int MAXD, int64_t* __restrict__ bs,
    float* __restrict__ bl, float* __restrict__ bsum, int32_t* __restrict__ order, int64_t* __restrict__ nxt,
    int64_t* __restrict__ done_seq, float* __restrict__ done_lp, float* __restrict__ done_p,
    int32_t* __restrict__ done_n, int32_t* __restrict__ active) {
    __shared__ float ys[BEAM_MAX_W][BEAM_MAX_W];
    __shared__ int ix[BEAM_MAX_W][BEAM_MAX_W];
    __shared__ int prev_seq[32][BEAM_MAX_W];
    __shared__ float prev_lp[32][BEAM_MAX_W];
    __shared__ float cand_p[BEAM_MAX_W * BEAM_MAX_W], cand_r[BEAM_MAX_W * BEAM_MAX_W];
    __shared__ int cand_c[BEAM_MAX_W * BEAM_MAX_W], cand_q[BEAM_MAX_W * BEAM_MAX_W], cand_ord[BEAM_MAX_W * BEAM_MAX_W];
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
    const int cols = min(W, V1);
    if (!active[k]) {  // this image's search has ended: keep its rows inert
        if (tid < W) {
            order[k * W + tid] = k * W + tid;
            nxt[k * W + tid] = 0;
        }
        return;
    }
    // ---- top `cols` of every live beam row (descending, lowest index first on ties) -------------------------
    const int live_rows = (t == 1) ? 1 : W;
    if (q < live_rows) {
        const float* row = logp + (long)(k * W + q) * ldl;
        int chosen[BEAM_MAX_W];
        for (int c = 0; c < cols; ++c) {
            float best = -INFINITY;
            int bi = 0x7fffffff;
            for (int v = lane; v < V1; v += 64) {
                bool taken = false;
#pragma unroll 4
                for (int p = 0; p < c; ++p) taken = taken || (chosen[p] == v);
                const float x = row[v];
                if (!taken && (x > best || (x == best && v < bi))) {
                    best = x;
                    bi = v;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ob = __shfl_xor(best, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ob > best || (ob == best && oi < bi)) {
                    best = ob;
                    bi = oi;
                }
            }
            chosen[c] = bi;  // identical in every lane after the butterfly
            if (lane == 0) {
                ys[q][c] = best;
                ix[q][c] = bi;
            }
        }
    }
    // ---- snapshot of the beams before this step (forks read the OLD columns, :488-489) ----------------------
    for (int i = tid; i < (t - 1) * W; i += blockDim.x) {
        const int s = i / W, w = i - s * W;
        prev_seq[s][w] = (int)bs[((long)s * NB + k) * W + w];
        prev_lp[s][w] = bl[((long)s * NB + k) * W + w];
    }
    __syncthreads();
    if (tid != 0) return;

    // ---- candidates in (c outer, q inner) order ------------------------------------------------------------------
    int nc = 0;
    for (int c = 0; c < cols; ++c)
        for (int qq = 0; qq < live_rows; ++qq) {
            if (t > 1 && prev_seq[t - 2][qq] == 0) continue;
            const float local = ys[qq][c];
            cand_c[nc] = ix[qq][c];
            cand_q[nc] = qq;
            cand_r[nc] = local;
            cand_p[nc] = bsum[k * W + qq] + local;  // fp32 add, as the reference's tensor arithmetic
            ++nc;
        }
    if (nc == 0) {  // :480-481
        active[k] = 0;
        for (int w = 0; w < W; ++w) {
            order[k * W + w] = k * W + w;
            nxt[k * W + w] = 0;
        }
        return;
    }
    // stable insertion sort of indices by descending p
    for (int i = 0; i < nc; ++i) {
        int j = i;
        const float pi = cand_p[i];
        while (j > 0 && cand_p[cand_ord[j - 1]] < pi) {
            cand_ord[j] = cand_ord[j - 1];
            --j;
        }
        cand_ord[j] = i;
    }
    // ---- new beams ---------------------------------------------------------------------------------------------
    const int nnew = min(W, nc);
    float new_sum[BEAM_MAX_W];
    for (int vix = 0; vix < W; ++vix) {
        if (vix >= nnew) {  // keeps its previous state and tokens (new_state = clone(state), :485)
            order[k * W + vix] = k * W + vix;
            new_sum[vix] = bsum[k * W + vix];
            nxt[k * W + vix] = bs[((long)(t - 1) * NB + k) * W + vix];
            continue;
        }
        const int ci = cand_ord[vix];
        const int qq = cand_q[ci];
        for (int s = 0; s < t - 1; ++s) {
            bs[((long)s * NB + k) * W + vix] = prev_seq[s][qq];
            bl[((long)s * NB + k) * W + vix] = prev_lp[s][qq];
        }
        order[k * W + vix] = k * W + qq;
        bs[((long)(t - 1) * NB + k) * W + vix] = cand_c[ci];
        bl[((long)(t - 1) * NB + k) * W + vix] = cand_r[ci];
        new_sum[vix] = cand_p[ci];
        nxt[k * W + vix] = cand_c[ci];
        if (cand_c[ci] == 0 || t == S) {  // :508-514
            const int n = done_n[k];
            if (n < MAXD) {
                for (int s = 0; s < S; ++s) {
                    done_seq[((long)k * MAXD + n) * S + s] = bs[((long)s * NB + k) * W + vix];
                    done_lp[((long)k * MAXD + n) * S + s] = bl[((long)s * NB + k) * W + vix];
                }
                done_p[(long)k * MAXD + n] = new_sum[vix];
                done_n[k] = n + 1;
            }
        }
    }
    for (int vix = 0; vix < W; ++vix) bsum[k * W + vix] = new_sum[vix];
}

extern "C" int rfn_beam_step(const float* logp, int64_t ldl, int V1, int W, int S, int t, int NB, int max_done,
                             int64_t* beam_seq, float* beam_lp, float* beam_sum, int32_t* order, int64_t* next_ids,
                             int64_t* done_seq, float* done_lp, float* done_p, int32_t* done_n, int32_t* active,
                             void* stream) {
    if (W < 1 || W > BEAM_MAX_W || S < 1 || S > 32 || t < 1 || t > S || NB < 1 || V1 < 1 || max_done < 1)
        return RFN_ERR_SHAPE;
    if (!logp || !beam_seq || !beam_lp || !beam_sum || !order || !next_ids || !done_seq || !done_lp || !done_p ||
        !done_n || !active)
        return RFN_ERR_ARG;
    hipLaunchKernelGGL(beam_step_k, dim3(NB), dim3(64 * W), 0, (hipStream_t)stream, logp, (long)ldl, V1, W, S, t, NB,
                       max_done, beam_seq, beam_lp, beam_sum, order, next_ids, done_seq, done_lp, done_p, done_n,
                       active);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}

// dst[r, :] = src[order[r], :]  -- recurrent-state re-gather of the forked beams (:499-501)
__global__ __launch_bounds__(256) void gather_rows_k(const float* __restrict__ src, float* __restrict__ dst,
                                                     const int32_t* __restrict__ order, int R) {
    const int r = blockIdx.x;
    const float* s = src + (long)order[r] * R;
    for (int j = threadIdx.x; j < R; j += 256) dst[(long)r * R + j] = s[j];
}
extern "C" int rfn_gather_rows(const float* src, float* dst, const int32_t* order, int rows, int R, void* stream) {
    if (rows < 1 || R < 1) return RFN_ERR_SHAPE;
    if (!src || !dst || !order || src == dst) return RFN_ERR_ARG;
    hipLaunchKernelGGL(gather_rows_k, dim3(rows), dim3(256), 0, (hipStream_t)stream, src, dst, order, R);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}
