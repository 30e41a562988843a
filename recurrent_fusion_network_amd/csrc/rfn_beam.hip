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

#define BEAM_MAX_W 32     /* beams per image; the full-row form (rfn_beam_step: phase 1 below) serves up to BEAM_WAVES of them */
#define BEAM_MAX_S 64     /* decode steps a block keeps the beams' histories for (seq_length) */
#define BEAM_THREADS 1024
#define BEAM_WAVES 16

// (value desc, index asc): the order in which the reference's descending sort lists the columns it uses
__device__ __forceinline__ bool beam_before(float x, int i, float y, int j) { return x > y || (x == y && i < j); }

// One block per image.  Phase 1: top `cols` columns of every live beam row -- each row is cut over 16 / rows waves,
// every lane keeps a sorted top-LW list of its strided share in registers (one pass over the row), the wave merges
// the lanes' lists with `cols` arg-max butterflies, one lane merges the waves' lists.  Phase 2: thread 0 builds and
// stably sorts the <= W*W candidates in LDS and decides forks / done slots; all threads then write the forked
// beam_seq / beam_logprobs columns and the done beams in parallel.
template <int LW>
__global__ __launch_bounds__(BEAM_THREADS) void beam_step_k(
    const float* __restrict__ logp, long ldl, int V1, int W, int S, int t, int NB, int MAXD, int64_t* __restrict__ bs,
    float* __restrict__ bl, float* __restrict__ bsum, int32_t* __restrict__ order, int64_t* __restrict__ nxt,
    int64_t* __restrict__ done_seq, float* __restrict__ done_lp, float* __restrict__ done_p,
    int32_t* __restrict__ done_n, int32_t* __restrict__ active, const float* __restrict__ topv,
    const int32_t* __restrict__ topi) {
    __shared__ float ys[BEAM_MAX_W][BEAM_MAX_W];
    __shared__ int ix[BEAM_MAX_W][BEAM_MAX_W];
    __shared__ float wys[BEAM_WAVES][BEAM_MAX_W];
    __shared__ int wix[BEAM_WAVES][BEAM_MAX_W];
    __shared__ int prev_seq[BEAM_MAX_S][BEAM_MAX_W];
    __shared__ float prev_lp[BEAM_MAX_S][BEAM_MAX_W];
    __shared__ float cand_p[BEAM_MAX_W * BEAM_MAX_W], cand_r[BEAM_MAX_W * BEAM_MAX_W];
    __shared__ int cand_c[BEAM_MAX_W * BEAM_MAX_W], cand_q[BEAM_MAX_W * BEAM_MAX_W], cand_ord[BEAM_MAX_W * BEAM_MAX_W];
    __shared__ int sel_ci[BEAM_MAX_W], slot_s[BEAM_MAX_W];
    __shared__ int nnew_s;
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cols = min(W, V1);
    if (!active[k]) {  // this image's search has ended: keep its rows inert
        if (tid < W) {
            order[k * W + tid] = k * W + tid;
            nxt[k * W + tid] = 0;
        }
        return;
    }
    // ---- phase 1: top `cols` of every live beam row (descending, lowest index first on ties) -----------------
    const int live_rows = (t == 1) ? 1 : W;
    const int wpr = topv ? 1 : BEAM_WAVES / live_rows;   // waves per row (full-row form: >= 1, the host checks W <= BEAM_WAVES)
    const int q = wave / wpr, part = wave - q * wpr;
    if (topv) {
        // the rows' top-W lists were produced with their log-softmax (rfn_log_softmax_topk): same values, same order
    } else if (q < live_rows) {
        const float* row = logp + (long)(k * W + q) * ldl;
        const int chunk = (V1 + wpr - 1) / wpr;
        const int v0 = part * chunk, v1 = min(V1, v0 + chunk);
        float tv[LW];
        int ti[LW];
#pragma unroll
        for (int j = 0; j < LW; ++j) {
            tv[j] = -INFINITY;
            ti[j] = 0x7fffffff;
        }
        for (int v = v0 + lane; v < v1; v += 64) {   // ascending v per lane: a later equal value stays behind
            float x = row[v];
            if (x > tv[LW - 1] || (ti[LW - 1] == 0x7fffffff)) {
                int xi = v;
#pragma unroll
                for (int j = 0; j < LW; ++j) {       // sorted insert by compare-exchange down the list
                    const bool fwd = beam_before(x, xi, tv[j], ti[j]);
                    const float ov = tv[j];
                    const int oi = ti[j];
                    tv[j] = fwd ? x : ov;
                    ti[j] = fwd ? xi : oi;
                    x = fwd ? ov : x;
                    xi = fwd ? oi : xi;
                }
            }
        }
        for (int c = 0; c < cols; ++c) {             // wave merge: pop the best head `cols` times
            float best = tv[0];
            int bi = ti[0];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ob = __shfl_xor(best, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (beam_before(ob, oi, best, bi)) {
                    best = ob;
                    bi = oi;
                }
            }
            if (ti[0] == bi && bi != 0x7fffffff) {   // this lane owned the winner: pop it
#pragma unroll
                for (int j = 0; j + 1 < LW; ++j) {
                    tv[j] = tv[j + 1];
                    ti[j] = ti[j + 1];
                }
                tv[LW - 1] = -INFINITY;
                ti[LW - 1] = 0x7fffffff;
            }
            if (lane == 0) {
                wys[wave][c] = best;
                wix[wave][c] = bi;
            }
        }
    }
    // ---- snapshot of the beams before this step (forks read the OLD columns, :488-489) ----------------------
    for (int i = tid; i < (t - 1) * W; i += (int)blockDim.x) {
        const int s = i / W, w = i - s * W;
        prev_seq[s][w] = (int)bs[((long)s * NB + k) * W + w];
        prev_lp[s][w] = bl[((long)s * NB + k) * W + w];
    }
    __syncthreads();
    if (topv) {
        for (int i = tid; i < live_rows * cols; i += (int)blockDim.x) {
            const int qq = i / cols, c = i - qq * cols;
            ys[qq][c] = topv[(long)(k * W + qq) * W + c];
            ix[qq][c] = topi[(long)(k * W + qq) * W + c];
        }
    } else if (tid < live_rows) {                    // merge the row's wave lists (each already sorted)
        int pos[BEAM_WAVES];
        for (int p = 0; p < wpr; ++p) pos[p] = 0;
        for (int c = 0; c < cols; ++c) {
            int bp = -1;
            for (int p = 0; p < wpr; ++p) {
                if (pos[p] >= cols) continue;
                const int w = tid * wpr + p;
                if (wix[w][pos[p]] == 0x7fffffff) continue;
                if (bp < 0 || beam_before(wys[w][pos[p]], wix[w][pos[p]], wys[tid * wpr + bp][pos[bp]],
                                          wix[tid * wpr + bp][pos[bp]]))
                    bp = p;
            }
            ys[tid][c] = wys[tid * wpr + bp][pos[bp]];
            ix[tid][c] = wix[tid * wpr + bp][pos[bp]];
            ++pos[bp];
        }
    }
    __syncthreads();

    // ---- phase 2a (thread 0): candidates in (c outer, q inner) order, stable sort, fork / done decisions --------
    if (tid == 0) {
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
        if (nc == 0) active[k] = 0;  // :480-481
        for (int i = 0; i < nc; ++i) {  // stable insertion sort of indices by descending p
            int j = i;
            const float pi = cand_p[i];
            while (j > 0 && cand_p[cand_ord[j - 1]] < pi) {
                cand_ord[j] = cand_ord[j - 1];
                --j;
            }
            cand_ord[j] = i;
        }
        const int nnew = min(W, nc);
        int n = done_n[k];
        for (int vix = 0; vix < W; ++vix) {
            slot_s[vix] = -1;
            if (vix >= nnew) continue;
            const int ci = cand_ord[vix];
            sel_ci[vix] = ci;
            if ((cand_c[ci] == 0 || t == S) && n < MAXD) {  // :508-514, appended in construction order
                slot_s[vix] = n;
                done_p[(long)k * MAXD + n] = cand_p[ci];
                ++n;
            }
        }
        done_n[k] = n;
        nnew_s = (nc == 0) ? -1 : nnew;
    }
    __syncthreads();
    const int nnew = nnew_s;
    if (nnew < 0) {  // no candidate left: the image stops, its rows stay inert
        if (tid < W) {
            order[k * W + tid] = k * W + tid;
            nxt[k * W + tid] = 0;
        }
        return;
    }
    // ---- phase 2b (all threads): forked columns and done beams ---------------------------------------------------
    for (int i = tid; i < nnew * S; i += (int)blockDim.x) {
        const int vix = i / S, s = i - vix * S;
        const int ci = sel_ci[vix], qq = cand_q[ci];
        int64_t tok;
        float lp;
        const long at = ((long)s * NB + k) * W + vix;
        if (s < t - 1) {
            tok = prev_seq[s][qq];
            lp = prev_lp[s][qq];
            bs[at] = tok;
            bl[at] = lp;
        } else if (s == t - 1) {
            tok = cand_c[ci];
            lp = cand_r[ci];
            bs[at] = tok;
            bl[at] = lp;
        } else {  // columns this search has not reached yet (still the initial zeros)
            tok = bs[at];
            lp = bl[at];
        }
        const int slot = slot_s[vix];
        if (slot >= 0) {
            done_seq[((long)k * MAXD + slot) * S + s] = tok;
            done_lp[((long)k * MAXD + slot) * S + s] = lp;
        }
    }
    if (tid < W) {
        const int vix = tid;
        if (vix < nnew) {
            const int ci = sel_ci[vix];
            order[k * W + vix] = k * W + cand_q[ci];
            nxt[k * W + vix] = cand_c[ci];
            bsum[k * W + vix] = cand_p[ci];          // every read of bsum happened before the barrier above
        } else {  // keeps its previous state and tokens (new_state = clone(state), :485)
            order[k * W + vix] = k * W + vix;
            nxt[k * W + vix] = bs[((long)(t - 1) * NB + k) * W + vix];
        }
    }
}

static int beam_step_launch(const float* logp, int64_t ldl, const float* topv, const int32_t* topi, int V1, int W, int S, int t,
                            int NB, int max_done, int64_t* beam_seq, float* beam_lp, float* beam_sum, int32_t* order,
                            int64_t* next_ids, int64_t* done_seq, float* done_lp, float* done_p, int32_t* done_n,
                            int32_t* active, void* stream) {
    if (W < 1 || W > BEAM_MAX_W || S < 1 || S > BEAM_MAX_S || t < 1 || t > S || NB < 1 || V1 < 1 || max_done < 1)
        return RFN_ERR_SHAPE;
    if (!topv && W > BEAM_WAVES) return RFN_ERR_SHAPE;   // the full-row form cuts a row over 16 / W waves
    if ((!logp && !(topv && topi)) || !beam_seq || !beam_lp || !beam_sum || !order || !next_ids || !done_seq || !done_lp ||
        !done_p || !done_n || !active)
        return RFN_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int threads = topv ? 256 : BEAM_THREADS;      // without the row scan the block only does the bookkeeping
#define BEAM_LAUNCH(LWV)                                                                                             \
    hipLaunchKernelGGL(beam_step_k<LWV>, dim3(NB), dim3(threads), 0, st, logp, (long)ldl, V1, W, S, t, NB,            \
                       max_done, beam_seq, beam_lp, beam_sum, order, next_ids, done_seq, done_lp, done_p, done_n,     \
                       active, topv, topi)
    if (W <= 2) BEAM_LAUNCH(2);
    else if (W <= 4) BEAM_LAUNCH(4);
    else if (W <= 8) BEAM_LAUNCH(8);
    else BEAM_LAUNCH(16);
#undef BEAM_LAUNCH
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}
extern "C" int rfn_beam_step(const float* logp, int64_t ldl, int V1, int W, int S, int t, int NB, int max_done,
                             int64_t* beam_seq, float* beam_lp, float* beam_sum, int32_t* order, int64_t* next_ids,
                             int64_t* done_seq, float* done_lp, float* done_p, int32_t* done_n, int32_t* active,
                             void* stream) {
    if (!logp) return RFN_ERR_ARG;
    return beam_step_launch(logp, ldl, nullptr, nullptr, V1, W, S, t, NB, max_done, beam_seq, beam_lp, beam_sum, order, next_ids,
                            done_seq, done_lp, done_p, done_n, active, stream);
}
// The same step fed with every beam row's W best log-probs (rfn_log_softmax_topk) instead of the full rows.
extern "C" int rfn_beam_step_topk(const float* topv, const int32_t* topi, int V1, int W, int S, int t, int NB, int max_done,
                                  int64_t* beam_seq, float* beam_lp, float* beam_sum, int32_t* order, int64_t* next_ids,
                                  int64_t* done_seq, float* done_lp, float* done_p, int32_t* done_n, int32_t* active,
                                  void* stream) {
    if (!topv || !topi) return RFN_ERR_ARG;
    return beam_step_launch(nullptr, 0, topv, topi, V1, W, S, t, NB, max_done, beam_seq, beam_lp, beam_sum, order, next_ids,
                            done_seq, done_lp, done_p, done_n, active, stream);
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
