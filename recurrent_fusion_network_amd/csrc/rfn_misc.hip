// Small streaming kernels of the recurrent-fusion path: bias-gradient column sums, embedding
// gather / deterministic scatter, vocabulary log-softmax, reason-head max over steps, the two
// criteria of misc/utils.py, the clamp+Adam update and the greedy pick of sample().
// All are HBM/L2-bound byte movers: coalesced 16-B accesses where alignment allows, wave shuffles for
// reductions, fixed summation orders (no float atomics) so results are bitwise reproducible.
#include <string.h>

#include "rfn_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float block_sum_256(float v, float* red /* [4] LDS */) {
    v = rfn_wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max_256(float v, float* red) {
    v = rfn_wave_max(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ---- out[n] (+)= sum_r X[r, n] ----------------------------------------------------------------
// block = CW columns x (1024 / CW) row lanes, 4 independent row streams per thread; the lanes' partial sums are
// combined through LDS in a fixed order.  Narrow matrices (the (S*B, A) attention partials) take CW = 16 so that
// 4x more blocks share the rows; wide ones keep 256-B row segments per wave.
template <int CW>
__global__ __launch_bounds__(1024) void colsum_k(const float* __restrict__ X, long ldx, int rows, int cols,
                                                 float* __restrict__ out, int accumulate) {
    constexpr int RL = 1024 / CW;
    __shared__ float red[RL][CW];
    const int cl = threadIdx.x % CW, rl = threadIdx.x / CW;
    const int c = blockIdx.x * CW + cl;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < cols) {
        const float* p = X + c;
        int r = rl;
        for (; r + 3 * RL < rows; r += 4 * RL) {
            a0 += p[(long)r * ldx];
            a1 += p[(long)(r + RL) * ldx];
            a2 += p[(long)(r + 2 * RL) * ldx];
            a3 += p[(long)(r + 3 * RL) * ldx];
        }
        for (; r < rows; r += RL) a0 += p[(long)r * ldx];
    }
    red[rl][cl] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (rl == 0 && c < cols) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < RL; ++k) s += red[k][cl];
        out[c] = accumulate ? out[c] + s : s;
    }
}
extern "C" int rfn_colsum_f32(const float* X, int64_t ldx, int rows, int cols, float* out, int accumulate,
                              void* stream) {
    if (rows < 0 || cols <= 0) return RFN_ERR_SHAPE;
    if (!X || !out) return RFN_ERR_ARG;
    if (cols >= 64 * 128)
        hipLaunchKernelGGL(colsum_k<64>, dim3(rfn_cdiv(cols, 64)), dim3(1024), 0, (hipStream_t)stream, X, (long)ldx,
                           rows, cols, out, accumulate);
    else
        hipLaunchKernelGGL(colsum_k<16>, dim3(rfn_cdiv(cols, 16)), dim3(1024), 0, (hipStream_t)stream, X, (long)ldx,
                           rows, cols, out, accumulate);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}

struct ColsumOuts { float* out[64]; };
// 64 columns x 16 row-lanes per block: a thread sums every 16th row of its column, the 16 partial sums are added in a
// fixed order through LDS (deterministic).
#define CSG_LANES 16
struct ColsumOuts2 { float* out[64]; float* out2[64]; };
__global__ __launch_bounds__(64 * CSG_LANES) void colsum_grouped_k(const float* __restrict__ X, long gstride, long ldx,
                                                                  int rows, int cols, const ColsumOuts2 outs) {
    __shared__ float red[CSG_LANES][64];
    const float* Xg = X + blockIdx.y * gstride;
    const int l = threadIdx.x & 63, c = blockIdx.x * 64 + l, rl = threadIdx.x >> 6;
    float acc = 0.f;
    if (c < cols)
        for (int r = rl; r < rows; r += CSG_LANES) acc += Xg[r * ldx + c];
    red[rl][l] = acc;
    __syncthreads();
    if (rl == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < CSG_LANES; ++k) t += red[k][l];
        outs.out[blockIdx.y][c] = t;
        if (outs.out2[blockIdx.y]) outs.out2[blockIdx.y][c] = t;
    }
}
extern "C" int rfn_colsum_grouped2_f32(const float* X, int64_t group_stride, int64_t ldx, int rows, int cols,
                                       float* const* outs, float* const* outs2, int ngroups, void* stream) {
    if (rows < 0 || cols <= 0 || ngroups < 1) return RFN_ERR_SHAPE;
    if (!X || !outs) return RFN_ERR_ARG;
    for (int g0 = 0; g0 < ngroups; g0 += 64) {
        ColsumOuts2 o;
        const int ng = ngroups - g0 < 64 ? ngroups - g0 : 64;
        for (int g = 0; g < ng; ++g) {
            if (!outs[g0 + g]) return RFN_ERR_ARG;
            o.out[g] = outs[g0 + g];
            o.out2[g] = outs2 ? outs2[g0 + g] : nullptr;
        }
        hipLaunchKernelGGL(colsum_grouped_k, dim3(rfn_cdiv(cols, 64), ng), dim3(64 * CSG_LANES), 0, (hipStream_t)stream,
                           X + (long)g0 * group_stride, (long)group_stride, (long)ldx, rows, cols, o);
        RFN_CHECK_LAUNCH();
    }
    return RFN_OK;
}
extern "C" int rfn_colsum_grouped_f32(const float* X, int64_t group_stride, int64_t ldx, int rows, int cols,
                                      float* const* outs, int ngroups, void* stream) {
    return rfn_colsum_grouped2_f32(X, group_stride, ldx, rows, cols, outs, nullptr, ngroups, stream);
}

// out[g][0..n) = value for up to 64 small buffers per launch (the exactly-zero att_h_2_out.bias gradients)
__global__ __launch_bounds__(64) void fill_small_k(const ColsumOuts outs, int n, float value) {
    float* o = outs.out[blockIdx.x];
    for (int i = threadIdx.x; i < n; i += 64) o[i] = value;
}
extern "C" int rfn_fill_small_f32(float* const* outs, int ngroups, int n, float value, void* stream) {
    if (ngroups < 1 || n < 1) return RFN_ERR_SHAPE;
    if (!outs) return RFN_ERR_ARG;
    for (int g0 = 0; g0 < ngroups; g0 += 64) {
        ColsumOuts o;
        const int ng = ngroups - g0 < 64 ? ngroups - g0 : 64;
        for (int g = 0; g < ng; ++g) {
            if (!outs[g0 + g]) return RFN_ERR_ARG;
            o.out[g] = outs[g0 + g];
        }
        hipLaunchKernelGGL(fill_small_k, dim3(ng), dim3(64), 0, (hipStream_t)stream, o, n, value);
        RFN_CHECK_LAUNCH();
    }
    return RFN_OK;
}

// dst[g][0..n) = src[g][0..n) for up to 64 small buffer pairs per launch (equal-by-construction bias gradients)
struct CopyPairs { float* dst[64]; const float* src[64]; };
__global__ __launch_bounds__(64) void copy_small_k(const CopyPairs p, int n) {
    float* o = p.dst[blockIdx.x];
    const float* s = p.src[blockIdx.x];
    for (int i = threadIdx.x; i < n; i += 64) o[i] = s[i];
}
extern "C" int rfn_copy_small_f32(float* const* dst, const float* const* src, int ngroups, int n, void* stream) {
    if (ngroups < 1 || n < 1) return RFN_ERR_SHAPE;
    if (!dst || !src) return RFN_ERR_ARG;
    for (int g0 = 0; g0 < ngroups; g0 += 64) {
        CopyPairs p;
        const int ng = ngroups - g0 < 64 ? ngroups - g0 : 64;
        for (int g = 0; g < ng; ++g) {
            if (!dst[g0 + g] || !src[g0 + g]) return RFN_ERR_ARG;
            p.dst[g] = dst[g0 + g];
            p.src[g] = src[g0 + g];
        }
        hipLaunchKernelGGL(copy_small_k, dim3(ng), dim3(64), 0, (hipStream_t)stream, p, n);
        RFN_CHECK_LAUNCH();
    }
    return RFN_OK;
}

// ---- embedding ----------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void embed_fwd_k(const float* __restrict__ W, int E, long V1,
                                                   const int64_t* __restrict__ ids, int inner, long si, long so,
                                                   float* __restrict__ out, long ldo) {
    const int r = blockIdx.x;
    long id = ids[(long)(r % inner) * si + (long)(r / inner) * so];
    if (id < 0 || id >= V1) id = 0;  // the reference would raise; never fault the GPU
    for (int e = threadIdx.x; e < E; e += 128) out[r * ldo + e] = W[id * E + e];
}
extern "C" int rfn_embed_fwd(const float* W, int E, int64_t V1, const int64_t* ids, int inner, int64_t ids_s_inner,
                             int64_t ids_s_outer, int rows, float* out, int64_t ldo, void* stream) {
    if (rows <= 0 || E <= 0 || V1 <= 0 || inner <= 0) return RFN_ERR_SHAPE;
    if (!W || !ids || !out) return RFN_ERR_ARG;
    hipLaunchKernelGGL(embed_fwd_k, dim3(rows), dim3(128), 0, (hipStream_t)stream, W, E, (long)V1, ids, inner,
                       (long)ids_s_inner, (long)ids_s_outer, out, (long)ldo);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}
// One wave per vocabulary row.  The token list is scanned 64 entries at a time: every lane tests one entry, a ballot gives
// the matches, the matching rows are queued in LDS in scan order, and the queue is drained EMB_DEPTH rows at a time -- their
// loads in flight together, added in queue order: deterministic, no float atomics.  A hot token (BOS feeds step 0 of every
// caption: B matches) used to be one serial chain of dependent loads, the long pole of the launch (147 us at B = 256).
// The scan walks the ids in MEMORY order when the rows form a whole (inner x rows/inner) rectangle (time-major rows
// r = t*B + b read ids[b][t]), otherwise in row order; either way the order is a function of the shape only.
#define EMB_UNROLL 4
#define EMB_CH 2      /* chunks of 64 x W floats per lane: E = 512 is covered by one wave */
#define EMB_QCAP 512  /* queued rows before a drain (>= 64 * EMB_UNROLL) */
#define EMB_DEPTH 8
template <bool VEC>
__global__ __launch_bounds__(64) void embed_bwd_k(const float* __restrict__ dout, long ldo,
                                                  const int64_t* __restrict__ ids, int inner, long si, long so,
                                                  int rows, int E, float* __restrict__ dW) {
    constexpr int W = VEC ? 4 : 1;
    typedef float vec_t __attribute__((ext_vector_type(W)));
    __shared__ int queue[EMB_QCAP];
    const long v = blockIdx.x;
    const int lane = threadIdx.x;
    // scan index q -> (hi, lo) = (q / mod, q % mod), carried along (q advances by 64: no division in the loop)
    const bool mem_order = rows % inner == 0;
    const int mod = mem_order ? rows / inner : inner;
    const long s_hi = mem_order ? si : so, s_lo = mem_order ? so : si;
    const int step_q = 64 / mod, step_r = 64 - step_q * mod;
    for (int e0 = 0; e0 < E; e0 += 64 * W * EMB_CH) {
        vec_t acc[EMB_CH];
#pragma unroll
        for (int c = 0; c < EMB_CH; ++c)
#pragma unroll
            for (int k = 0; k < W; ++k) acc[c][k] = 0.f;
        int nq = 0;                                   // queued rows (uniform)
        auto drain = [&]() {                          // adds queue[0 .. nq) in order
            __syncthreads();
            for (int i0 = 0; i0 < nq; i0 += EMB_DEPTH) {
                vec_t x[EMB_DEPTH][EMB_CH];
#pragma unroll
                for (int j = 0; j < EMB_DEPTH; ++j) {
                    const long rr = queue[i0 + j < nq ? i0 + j : nq - 1];
#pragma unroll
                    for (int c = 0; c < EMB_CH; ++c) {
                        const int e = e0 + (c * 64 + lane) * W;
#pragma unroll
                        for (int k = 0; k < W; ++k) x[j][c][k] = 0.f;
                        if (e < E) x[j][c] = *reinterpret_cast<const vec_t*>(dout + rr * ldo + e);
                    }
                }
#pragma unroll
                for (int j = 0; j < EMB_DEPTH; ++j)
                    if (i0 + j < nq) {
#pragma unroll
                        for (int c = 0; c < EMB_CH; ++c) acc[c] += x[j][c];
                    }
            }
            nq = 0;
            __syncthreads();
        };
        int hi = lane / mod, lo = lane - hi * mod;
        for (int base = 0; base < rows; base += 64 * EMB_UNROLL) {
            long id[EMB_UNROLL];
            int row[EMB_UNROLL];
#pragma unroll
            for (int u = 0; u < EMB_UNROLL; ++u) {
                const int q = base + 64 * u + lane;
                id[u] = -1;
                if (q < rows) id[u] = ids[hi * s_hi + lo * s_lo];
                row[u] = mem_order ? lo * inner + hi : q;      // the row of d out this entry belongs to
                lo += step_r;
                hi += step_q;
                if (lo >= mod) { lo -= mod; ++hi; }
            }
            if (nq + 64 * EMB_UNROLL > EMB_QCAP) drain();
#pragma unroll
            for (int u = 0; u < EMB_UNROLL; ++u) {
                const bool hit = id[u] == v;
                const unsigned long long m = __ballot(hit);
                if (hit) queue[nq + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = row[u];
                nq += __builtin_popcountll(m);
            }
        }
        drain();
#pragma unroll
        for (int c = 0; c < EMB_CH; ++c) {
            const int e = e0 + (c * 64 + lane) * W;
            if (e < E) *reinterpret_cast<vec_t*>(dW + v * E + e) = acc[c];
        }
    }
}
extern "C" int rfn_embed_bwd(const float* dout, int64_t ldo, const int64_t* ids, int inner, int64_t ids_s_inner,
                             int64_t ids_s_outer, int rows, int E, int64_t V1, float* dW, void* stream) {
    if (rows < 0 || E <= 0 || V1 <= 0 || inner <= 0) return RFN_ERR_SHAPE;
    if (!dout || !ids || !dW) return RFN_ERR_ARG;
    const bool vec = (E % 4 == 0) && (ldo % 4 == 0) && rfn_aligned16(dout) && rfn_aligned16(dW);
    if (vec)
        hipLaunchKernelGGL(embed_bwd_k<true>, dim3((unsigned)V1), dim3(64), 0, (hipStream_t)stream, dout, (long)ldo,
                           ids, inner, (long)ids_s_inner, (long)ids_s_outer, rows, E, dW);
    else
        hipLaunchKernelGGL(embed_bwd_k<false>, dim3((unsigned)V1), dim3(64), 0, (hipStream_t)stream, dout,
                           (long)ldo, ids, inner, (long)ids_s_inner, (long)ids_s_outer, rows, E, dW);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}

// ---- log-softmax over the vocabulary --------------------------------------------------------------
// One block per row.  Rows of up to 256 x 4 x LSM_R4 = 10240 logits (16-B aligned, V1 % 4 == 0) are read ONCE, 16 B per
// lane, and held in registers for the max, the sum and the output pass; other rows take the three-pass scalar form.
// log_softmax_row() is shared by the plain kernel and by the top-k kernel of the beam search, so both produce the same
// log-prob bits for a row.
typedef float lsm_f32x4 __attribute__((ext_vector_type(4)));
#define LSM_R4 10
template <bool VEC>
__device__ __forceinline__ float log_softmax_row(const float* __restrict__ x, int V1, lsm_f32x4 (&xr)[LSM_R4], float* red) {
    float m = -INFINITY, s = 0.f;
    if constexpr (VEC) {
        const int n4 = V1 >> 2;
#pragma unroll
        for (int j = 0; j < LSM_R4; ++j) {
            const int i = threadIdx.x + 256 * j;
            xr[j] = lsm_f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            if (i < n4) xr[j] = *reinterpret_cast<const lsm_f32x4*>(x + 4 * i);
        }
#pragma unroll
        for (int j = 0; j < LSM_R4; ++j) m = fmaxf(m, fmaxf(fmaxf(xr[j][0], xr[j][1]), fmaxf(xr[j][2], xr[j][3])));
        m = block_max_256(m, red);
#pragma unroll
        for (int j = 0; j < LSM_R4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) s += expf(xr[j][e] - m);      // exp(-inf) = 0 for the padding lanes
    } else {
        for (int v = threadIdx.x; v < V1; v += 256) m = fmaxf(m, x[v]);
        m = block_max_256(m, red);
        for (int v = threadIdx.x; v < V1; v += 256) s += expf(x[v] - m);
    }
    s = block_sum_256(s, red);
    return m + logf(s);
}
template <bool VEC>
__global__ __launch_bounds__(256) void log_softmax_fwd_k(const float* __restrict__ logits, long ldl, int V1,
                                                         int inner, long s_inner, long s_outer,
                                                         float* __restrict__ out) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    const float* x = logits + r * ldl;
    float* o = out + (long)(r % inner) * s_inner + (long)(r / inner) * s_outer;
    lsm_f32x4 xr[LSM_R4];
    const float lse = log_softmax_row<VEC>(x, V1, xr, red);
    if constexpr (VEC) {
        const int n4 = V1 >> 2;
#pragma unroll
        for (int j = 0; j < LSM_R4; ++j) {
            const int i = threadIdx.x + 256 * j;
            if (i < n4) *reinterpret_cast<lsm_f32x4*>(o + 4 * i) = xr[j] - lse;
        }
    } else {
        for (int v = threadIdx.x; v < V1; v += 256) o[v] = x[v] - lse;
    }
}
static bool lsm_vec_ok(const float* logits, int64_t ldl, int V1, const float* out, int64_t s_inner, int64_t s_outer) {
    return V1 % 4 == 0 && V1 <= 256 * 4 * LSM_R4 && ldl % 4 == 0 && rfn_aligned16(logits) &&
           (!out || (rfn_aligned16(out) && s_inner % 4 == 0 && s_outer % 4 == 0));
}
extern "C" int rfn_log_softmax_fwd(const float* logits, int64_t ldl, int rows, int V1, int inner,
                                   int64_t out_s_inner, int64_t out_s_outer, float* out, void* stream) {
    if (rows <= 0 || V1 <= 0 || inner <= 0) return RFN_ERR_SHAPE;
    if (!logits || !out) return RFN_ERR_ARG;
    if (lsm_vec_ok(logits, ldl, V1, out, out_s_inner, out_s_outer))
        hipLaunchKernelGGL(log_softmax_fwd_k<true>, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, (long)ldl, V1,
                           inner, (long)out_s_inner, (long)out_s_outer, out);
    else
        hipLaunchKernelGGL(log_softmax_fwd_k<false>, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, (long)ldl, V1,
                           inner, (long)out_s_inner, (long)out_s_outer, out);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}

// ---- log-softmax + top-W of every row, without materialising the log-probs (beam search) ---------------------------------
// sample_beam only ever looks at the W best log-probs of a beam row (misc/RecurrentFusionModel.py:463-466: a full sort, of
// which columns 0 .. beam_size-1 are read).  topv[r, c] / topi[r, c], c < W: the c-th largest log-prob of row r and its
// token, ordered (value descending, token ascending) -- the order the reference's descending sort lists them -- computed
// from the same log-prob bits rfn_log_softmax_fwd writes.  W <= 32.
#define LSM_TOPW 32
__device__ __forceinline__ bool lsm_before(float x, int i, float y, int j) { return x > y || (x == y && i < j); }
template <bool VEC, int LW>
__global__ __launch_bounds__(256) void log_softmax_topk_k(const float* __restrict__ logits, long ldl, int V1, int W,
                                                          float* __restrict__ topv, int* __restrict__ topi) {
    __shared__ float red[4];
    __shared__ float wv[4][LSM_TOPW];
    __shared__ int wi[4][LSM_TOPW];
    const int r = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* x = logits + r * ldl;
    lsm_f32x4 xr[LSM_R4];
    const float lse = log_softmax_row<VEC>(x, V1, xr, red);
    const int cols = min(W, V1);
    // Threshold first: the `cols`-th largest of the 256 threads' LOCAL maxima is a lower bound of the row's `cols`-th largest
    // log-prob (those maxima are distinct elements of the row), so only entries >= it can make the list -- a handful per
    // row instead of every entry going through a sorted insert.
    float lm = -INFINITY;
    if constexpr (VEC) {
#pragma unroll
        for (int j = 0; j < LSM_R4; ++j) lm = fmaxf(lm, fmaxf(fmaxf(xr[j][0], xr[j][1]), fmaxf(xr[j][2], xr[j][3])));
    } else {
        for (int v = threadIdx.x; v < V1; v += 256) lm = fmaxf(lm, x[v]);
    }
    lm -= lse;      // x - lse is monotone in x: the maximum of the log-probs is the log-prob of the maximum
    {
        float mine = lm;
        for (int c = 0; c < cols; ++c) {         // wave: pop the largest local maximum `cols` times
            float best = mine;
            int bl_ = lane;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ob = __shfl_xor(best, o, 64);
                const int ol = __shfl_xor(bl_, o, 64);
                if (ob > best || (ob == best && ol < bl_)) {
                    best = ob;
                    bl_ = ol;
                }
            }
            if (lane == bl_) mine = -INFINITY;
            if (lane == 0) wv[wave][c] = best;
        }
    }
    __syncthreads();
    float thr;
    {
        int pos[4] = {0, 0, 0, 0};
        thr = -INFINITY;
        for (int c = 0; c < cols; ++c) {         // every thread merges the four sorted lists redundantly (cols <= 32)
            int bp = 0;
            float bv = -INFINITY;
            for (int p = 0; p < 4; ++p)
                if (pos[p] < cols && wv[p][pos[p]] > bv) {
                    bv = wv[p][pos[p]];
                    bp = p;
                }
            thr = bv;
            ++pos[bp];
        }
    }
    __syncthreads();                             // wv is reused by the merge below
    float tv[LW];
    int ti[LW];
#pragma unroll
    for (int j = 0; j < LW; ++j) {
        tv[j] = -INFINITY;
        ti[j] = 0x7fffffff;
    }
    auto offer = [&](float lp, int v) {      // sorted insert by compare-exchange down the list (ascending v per thread)
        if (lp >= thr && lsm_before(lp, v, tv[LW - 1], ti[LW - 1])) {
#pragma unroll
            for (int j = 0; j < LW; ++j) {
                const bool fwd = lsm_before(lp, v, tv[j], ti[j]);
                const float ov = tv[j];
                const int oi = ti[j];
                tv[j] = fwd ? lp : ov;
                ti[j] = fwd ? v : oi;
                lp = fwd ? ov : lp;
                v = fwd ? oi : v;
            }
        }
    };
    if constexpr (VEC) {
        const int n4 = V1 >> 2;
#pragma unroll
        for (int j = 0; j < LSM_R4; ++j) {
            const int i = threadIdx.x + 256 * j;
            if (i < n4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) offer(xr[j][e] - lse, 4 * i + e);
            }
        }
    } else {
        for (int v = threadIdx.x; v < V1; v += 256) offer(x[v] - lse, v);
    }
    for (int c = 0; c < cols; ++c) {             // wave merge: pop the best head `cols` times
        float best = tv[0];
        int bi = ti[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (lsm_before(ob, oi, best, bi)) {
                best = ob;
                bi = oi;
            }
        }
        if (ti[0] == bi && bi != 0x7fffffff) {
#pragma unroll
            for (int j = 0; j + 1 < LW; ++j) {
                tv[j] = tv[j + 1];
                ti[j] = ti[j + 1];
            }
            tv[LW - 1] = -INFINITY;
            ti[LW - 1] = 0x7fffffff;
        }
        if (lane == 0) {
            wv[wave][c] = best;
            wi[wave][c] = bi;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {                      // merge the four waves' sorted lists
        int pos[4] = {0, 0, 0, 0};
        for (int c = 0; c < cols; ++c) {
            int bp = -1;
            for (int p = 0; p < 4; ++p) {
                if (pos[p] >= cols || wi[p][pos[p]] == 0x7fffffff) continue;
                if (bp < 0 || lsm_before(wv[p][pos[p]], wi[p][pos[p]], wv[bp][pos[bp]], wi[bp][pos[bp]])) bp = p;
            }
            if (bp < 0) {                        // a row with fewer than `cols` comparable entries (NaN logits): no
                topv[(long)r * W + c] = -INFINITY;   // candidate is left -- never index the lists with -1; the slot
                topi[(long)r * W + c] = 0;           // reads as "token 0 at -inf", which the beam step can never pick
                continue;                            // over a real candidate
            }
            topv[(long)r * W + c] = wv[bp][pos[bp]];
            topi[(long)r * W + c] = wi[bp][pos[bp]];
            ++pos[bp];
        }
    }
}
extern "C" int rfn_log_softmax_topk(const float* logits, int64_t ldl, int rows, int V1, int W, float* topv, int32_t* topi,
                                    void* stream) {
    if (rows <= 0 || V1 <= 0 || W < 1 || W > LSM_TOPW) return RFN_ERR_SHAPE;
    if (!logits || !topv || !topi) return RFN_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = lsm_vec_ok(logits, ldl, V1, nullptr, 0, 0);
#define LSM_LAUNCH(VECV, LWV)                                                                                              \
    hipLaunchKernelGGL((log_softmax_topk_k<VECV, LWV>), dim3(rows), dim3(256), 0, st, logits, (long)ldl, V1, W, topv, topi)
    if (W <= 4) { if (vec) LSM_LAUNCH(true, 4); else LSM_LAUNCH(false, 4); }
    else if (W <= 8) { if (vec) LSM_LAUNCH(true, 8); else LSM_LAUNCH(false, 8); }
    else if (W <= 16) { if (vec) LSM_LAUNCH(true, 16); else LSM_LAUNCH(false, 16); }
    else { if (vec) LSM_LAUNCH(true, 32); else LSM_LAUNCH(false, 32); }
#undef LSM_LAUNCH
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}
// d logits = g - softmax * sum(g), one block per row.  VEC: rows of up to 10240 logits (16-B aligned, V1 % 4 == 0) keep g in
// registers between the sum and the update: one read of g and of the log-probs, one write, 16 B per lane.
template <bool VEC>
__global__ __launch_bounds__(256) void log_softmax_bwd_k(const float* __restrict__ g, const float* __restrict__ logp,
                                                         int V1, int inner, long s_inner, long s_outer,
                                                         float* __restrict__ dlogits, long ldd) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    const long off = (long)(r % inner) * s_inner + (long)(r / inner) * s_outer;
    const float* gr = g + off;
    const float* lp = logp + off;
    float* d = dlogits + r * ldd;
    if constexpr (VEC) {
        const int n4 = V1 >> 2;
        lsm_f32x4 gv[LSM_R4], lv[LSM_R4];
#pragma unroll
        for (int j = 0; j < LSM_R4; ++j) {
            const int i = threadIdx.x + 256 * j;
            gv[j] = lsm_f32x4{0.f, 0.f, 0.f, 0.f};
            lv[j] = lsm_f32x4{0.f, 0.f, 0.f, 0.f};
            if (i < n4) {
                gv[j] = *reinterpret_cast<const lsm_f32x4*>(gr + 4 * i);
                lv[j] = __builtin_nontemporal_load(reinterpret_cast<const lsm_f32x4*>(lp + 4 * i));
            }
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < LSM_R4; ++j) s += (gv[j][0] + gv[j][1]) + (gv[j][2] + gv[j][3]);
        s = block_sum_256(s, red);
#pragma unroll
        for (int j = 0; j < LSM_R4; ++j) {
            const int i = threadIdx.x + 256 * j;
            if (i < n4) {
                lsm_f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = gv[j][e] - expf(lv[j][e]) * s;
                *reinterpret_cast<lsm_f32x4*>(d + 4 * i) = o;
            }
        }
    } else {
        float s = 0.f;
        for (int v = threadIdx.x; v < V1; v += 256) s += gr[v];
        s = block_sum_256(s, red);
        for (int v = threadIdx.x; v < V1; v += 256) d[v] = gr[v] - expf(lp[v]) * s;
    }
}
extern "C" int rfn_log_softmax_bwd(const float* g, const float* logp, int rows, int V1, int inner, int64_t s_inner,
                                   int64_t s_outer, float* dlogits, int64_t ldd, void* stream) {
    if (rows <= 0 || V1 <= 0 || inner <= 0) return RFN_ERR_SHAPE;
    if (!g || !logp || !dlogits) return RFN_ERR_ARG;
    const bool vec = V1 % 4 == 0 && V1 <= 256 * 4 * LSM_R4 && s_inner % 4 == 0 && s_outer % 4 == 0 && ldd % 4 == 0 &&
                     rfn_aligned16(g) && rfn_aligned16(logp) && rfn_aligned16(dlogits);
    if (vec)
        hipLaunchKernelGGL(log_softmax_bwd_k<true>, dim3(rows), dim3(256), 0, (hipStream_t)stream, g, logp, V1, inner,
                           (long)s_inner, (long)s_outer, dlogits, (long)ldd);
    else
        hipLaunchKernelGGL(log_softmax_bwd_k<false>, dim3(rows), dim3(256), 0, (hipStream_t)stream, g, logp, V1, inner,
                           (long)s_inner, (long)s_outer, dlogits, (long)ldd);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}

// ---- reason heads: max over steps ------------------------------------------------------------------
// blockIdx.y = head: X slabs (T, B*K) lie `T*BK` apart, out / arg / dout (B*K) `BK` apart.
__global__ __launch_bounds__(256) void max_steps_fwd_k(const float* __restrict__ X, int T, long BK,
                                                       float* __restrict__ out, int32_t* __restrict__ arg) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= BK) return;
    X += (long)blockIdx.y * T * BK;
    float m = X[i];
    int a = 0;
    for (int t = 1; t < T; ++t) {
        const float v = X[t * BK + i];
        if (v > m) {
            m = v;
            a = t;
        }
    }
    out[blockIdx.y * BK + i] = m;
    if (arg) arg[blockIdx.y * BK + i] = a;
}
extern "C" int rfn_max_over_steps_fwd_grouped(const float* X, int T, int B, int K, float* out, int32_t* arg,
                                              int ngroups, void* stream) {
    if (T <= 0 || B <= 0 || K <= 0 || ngroups < 1) return RFN_ERR_SHAPE;
    if (!X || !out) return RFN_ERR_ARG;
    const long BK = (long)B * K;
    hipLaunchKernelGGL(max_steps_fwd_k, dim3(rfn_cdiv(BK, 256), ngroups), dim3(256), 0, (hipStream_t)stream, X, T, BK,
                       out, arg);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}
extern "C" int rfn_max_over_steps_fwd(const float* X, int T, int B, int K, float* out, int32_t* arg, void* stream) {
    return rfn_max_over_steps_fwd_grouped(X, T, B, K, out, arg, 1, stream);
}
__global__ __launch_bounds__(256) void max_steps_bwd_k(const float* __restrict__ dout, const int32_t* __restrict__ arg,
                                                       int T, long BK, float* __restrict__ dX) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= BK) return;
    const float g = dout ? dout[blockIdx.y * BK + i] : 0.f;
    const int a = arg[blockIdx.y * BK + i];
    dX += (long)blockIdx.y * T * BK;
    for (int t = 0; t < T; ++t) dX[t * BK + i] = (t == a) ? g : 0.f;
}
extern "C" int rfn_max_over_steps_bwd_grouped(const float* dout, const int32_t* arg, int T, int B, int K, float* dX,
                                              int ngroups, void* stream) {
    if (T <= 0 || B <= 0 || K <= 0 || ngroups < 1) return RFN_ERR_SHAPE;
    if (!arg || !dX) return RFN_ERR_ARG;
    const long BK = (long)B * K;
    hipLaunchKernelGGL(max_steps_bwd_k, dim3(rfn_cdiv(BK, 256), ngroups), dim3(256), 0, (hipStream_t)stream, dout, arg,
                       T, BK, dX);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}
extern "C" int rfn_max_over_steps_bwd(const float* dout, const int32_t* arg, int T, int B, int K, float* dX,
                                      void* stream) {
    return rfn_max_over_steps_bwd_grouped(dout, arg, T, B, K, dX, 1, stream);
}

// ---- y = alpha*x + beta*y on a strided 2-D view ------------------------------------------------------
__global__ __launch_bounds__(256) void axpby_2d_k(float alpha, const float* __restrict__ x, long ldx, float beta,
                                                  float* __restrict__ y, long ldy, int rows, int cols) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i - (long)r * cols);
    const float xv = x ? alpha * x[r * ldx + c] : 0.f;
    float* p = y + r * ldy + c;
    *p = (beta == 0.f) ? xv : xv + beta * *p;
}
extern "C" int rfn_axpby_2d(float alpha, const float* x, int64_t ldx, float beta, float* y, int64_t ldy, int rows,
                            int cols, void* stream) {
    if (rows <= 0 || cols <= 0) return RFN_ERR_SHAPE;
    if (!y) return RFN_ERR_ARG;
    hipLaunchKernelGGL(axpby_2d_k, dim3(rfn_cdiv((long)rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, alpha,
                       x, (long)ldx, beta, y, (long)ldy, rows, cols);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}

// ---- y[r,c] /= divisor (true division: the state mean sum/M of misc/RecurrentFusionModel.py:234-235) ----
__global__ __launch_bounds__(256) void div_2d_k(float* __restrict__ y, long ldy, int rows, int cols, float divisor) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i - (long)r * cols);
    y[r * ldy + c] = y[r * ldy + c] / divisor;
}
extern "C" int rfn_div_2d(float* y, int64_t ldy, int rows, int cols, float divisor, void* stream) {
    if (rows <= 0 || cols <= 0 || divisor == 0.f) return RFN_ERR_SHAPE;
    if (!y) return RFN_ERR_ARG;
    hipLaunchKernelGGL(div_2d_k, dim3(rfn_cdiv((long)rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, y,
                       (long)ldy, rows, cols, divisor);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}

// ---- mean over the M encoder slices of a concatenated state and its backward (misc/RecurrentFusionModel.py:233-235)
// y[r, c] = (((x[r, c] + x[r, G1 + c]) + x[r, 2*G1 + c]) + ...) / G   -- summed first, then divided, like the
// reference; blockIdx.y selects one of up to two (x, y) pairs (h and c share the launch).
struct MeanArgs {
    const float* x[2];
    float* y[2];
    float beta[2];   // backward only: y = alpha*x + beta*y per pair
};
__global__ __launch_bounds__(256) void mean_groups_k(const MeanArgs a, long ldx, long gstride, int G, long ldy,
                                                     int rows, int cols) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i - (long)r * cols);
    const float* x = a.x[blockIdx.y] + r * ldx + c;
    float s = x[0];
    for (int g = 1; g < G; ++g) s += x[g * gstride];
    a.y[blockIdx.y][r * ldy + c] = s / (float)G;
}
extern "C" int rfn_mean_over_groups(int npairs, const float* const* x, int64_t ldx, int64_t gstride, int G,
                                    float* const* y, int64_t ldy, int rows, int cols, void* stream) {
    if (npairs < 1 || npairs > 2 || G < 1 || rows <= 0 || cols <= 0) return RFN_ERR_SHAPE;
    if (!x || !y) return RFN_ERR_ARG;
    MeanArgs a;
    memset(&a, 0, sizeof(a));
    for (int p = 0; p < npairs; ++p) {
        if (!x[p] || !y[p]) return RFN_ERR_ARG;
        a.x[p] = x[p];
        a.y[p] = y[p];
    }
    hipLaunchKernelGGL(mean_groups_k, dim3(rfn_cdiv((long)rows * cols, 256), npairs), dim3(256), 0,
                       (hipStream_t)stream, a, (long)ldx, (long)gstride, G, (long)ldy, rows, cols);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}
// y[r, g*gstride + c] = alpha * x[r, c] + beta_p * y[r, g*gstride + c]  for every group g (the mean's backward)
__global__ __launch_bounds__(256) void bcast_groups_k(const MeanArgs a, float alpha, long ldx, long gstride, int G,
                                                      long ldy, int rows, int cols) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i - (long)r * cols);
    const float xv = alpha * a.x[blockIdx.y][r * ldx + c];
    const float beta = a.beta[blockIdx.y];
    float* y = a.y[blockIdx.y] + r * ldy + c;
    for (int g = 0; g < G; ++g) {
        float* p = y + g * gstride;
        *p = (beta == 0.f) ? xv : xv + beta * *p;
    }
}
extern "C" int rfn_bcast_to_groups(int npairs, float alpha, const float* const* x, int64_t ldx, const float* beta,
                                   float* const* y, int64_t ldy, int64_t gstride, int G, int rows, int cols,
                                   void* stream) {
    if (npairs < 1 || npairs > 2 || G < 1 || rows <= 0 || cols <= 0) return RFN_ERR_SHAPE;
    if (!x || !y || !beta) return RFN_ERR_ARG;
    MeanArgs a;
    memset(&a, 0, sizeof(a));
    for (int p = 0; p < npairs; ++p) {
        if (!x[p] || !y[p]) return RFN_ERR_ARG;
        a.x[p] = x[p];
        a.y[p] = y[p];
        a.beta[p] = beta[p];
    }
    hipLaunchKernelGGL(bcast_groups_k, dim3(rfn_cdiv((long)rows * cols, 256), npairs), dim3(256), 0,
                       (hipStream_t)stream, a, alpha, (long)ldx, (long)gstride, G, (long)ldy, rows, cols);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}

// ---- fixed-order sum of n floats -> out[0] ---------------------------------------------------------
__global__ __launch_bounds__(256) void sum_k(const float* __restrict__ x, int n, float scale, float* __restrict__ out,
                                             int accumulate) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += x[i];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + s * scale : s * scale;
}

// ---- XE language loss (misc/utils.py:163-184) -------------------------------------------------------
__global__ __launch_bounds__(256) void xe_loss_k(const float* __restrict__ logp, int T, int V1,
                                                 const int64_t* __restrict__ target, long ld_t,
                                                 const float* __restrict__ mask, long ld_m, float eps, float gcoef,
                                                 const float* __restrict__ gdev, float* __restrict__ row_loss,
                                                 float* __restrict__ dlogp) {
    __shared__ float red[4];
    const int r = blockIdx.x, b = r / T, t = r - b * T;
    const float* lp = logp + (long)r * V1;
    long tg = target[b * ld_t + t];
    if (tg < 0 || tg >= V1) tg = 0;
    const float mk = mask[b * ld_m + t];
    const float uni = eps / (float)V1;
    float term = 0.f;
    if (row_loss) {
        if (eps > 0.f) {
            float s = 0.f;
            for (int v = threadIdx.x; v < V1; v += 256) s += lp[v];
            s = block_sum_256(s, red);
            term = (1.0f - eps) * lp[tg] + uni * s;
        } else {
            term = lp[tg];
        }
        if (threadIdx.x == 0) row_loss[r] = -mk * term;
    }
    if (dlogp) {
        float* d = dlogp + (long)r * V1;
        const float gc = gdev ? gcoef * gdev[0] : gcoef;  // upstream d loss, read on the device (no host sync)
        const float base = -mk * gc * uni;              // 0 when eps == 0
        const float hot = -mk * gc * (1.0f - eps);
        for (int v = threadIdx.x; v < V1; v += 256) d[v] = (v == tg) ? base + hot : base;
    }
}
extern "C" int rfn_xe_loss_ex(const float* logp, int B, int T, int V1, const int64_t* target, int64_t ld_target,
                              const float* mask, int64_t ld_mask, float eps, float gscale, const float* gscale_dev,
                              float* scratch, float* loss_out, int accumulate_loss, float* dlogp, void* stream) {
    if (B <= 0 || T <= 0 || V1 <= 0) return RFN_ERR_SHAPE;
    if (!logp || !target || !mask || (loss_out && !scratch)) return RFN_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(xe_loss_k, dim3(B * T), dim3(256), 0, st, logp, T, V1, target, (long)ld_target, mask,
                       (long)ld_mask, eps, gscale / (float)B, gscale_dev, loss_out ? scratch : nullptr, dlogp);
    RFN_CHECK_LAUNCH();
    if (loss_out) {
        hipLaunchKernelGGL(sum_k, dim3(1), dim3(256), 0, st, scratch, B * T, 1.0f / (float)B, loss_out,
                           accumulate_loss);
        RFN_CHECK_LAUNCH();
    }
    return RFN_OK;
}
extern "C" int rfn_xe_loss(const float* logp, int B, int T, int V1, const int64_t* target, int64_t ld_target,
                           const float* mask, int64_t ld_mask, float eps, float gscale, float* scratch,
                           float* loss_out, int accumulate_loss, float* dlogp, void* stream) {
    return rfn_xe_loss_ex(logp, B, T, V1, target, ld_target, mask, ld_mask, eps, gscale, nullptr, scratch, loss_out,
                          accumulate_loss, dlogp, stream);
}

// ---- XE language loss straight from the logits (misc/utils.py:163-184 + F.log_softmax of :276) --------------
// The unfused pass writes log_prob (B, T, V+1), the criterion reads it back for the loss, writes d log_prob (B, T, V+1), and the
// log-softmax backward reads both to write d logits: four passes over a 165 MB tensor at C3 that exist only because
// `forward` has to hand `log_prob` to a caller-owned criterion (train.py:154-159).  When the caller asks for the LOSS
// (RecurrentFusionModel.forward_loss) neither tensor is needed:
//   forward : one pass over the time-major logits rows r = t * B + b -- the row's logsumexp (log_softmax_row: the same bits
//             rfn_log_softmax_fwd would subtract) -> lse[r], and the row's loss term
//             -mask * ((1 - eps) * (x[tg] - lse) + eps / V1 * (sum_v x[v] - V1 * lse)) -> row_loss[b * T + t]
//   backward: d logits[v] = mask * g / B * (exp(x[v] - lse) - (1 - eps) * [v == tg] - eps / V1), written over the logits.
// (sum_v d logp = -mask * g / B whatever eps is, which is what the log-softmax backward multiplies the probabilities with.)
template <bool VEC>
__global__ __launch_bounds__(256) void xe_logits_fwd_k(const float* __restrict__ logits, long ldl, int B, int T, int V1,
                                                       const int64_t* __restrict__ target, long ld_t,
                                                       const float* __restrict__ mask, long ld_m, float eps,
                                                       float* __restrict__ lse_out, float* __restrict__ row_loss) {
    __shared__ float red[4];
    const int r = blockIdx.x, t = r / B, b = r - t * B;      // time-major rows
    const float* x = logits + r * ldl;
    lsm_f32x4 xr[LSM_R4];
    const float lse = log_softmax_row<VEC>(x, V1, xr, red);
    long tg = target[b * ld_t + t];
    if (tg < 0 || tg >= V1) tg = 0;
    const float mk = mask[b * ld_m + t];
    float term = x[tg] - lse;
    if (eps > 0.f) {
        float s = 0.f;
        if constexpr (VEC) {
            const int n4 = V1 >> 2;
#pragma unroll
            for (int j = 0; j < LSM_R4; ++j)
                if (threadIdx.x + 256 * j < n4) s += ((xr[j][0] - lse) + (xr[j][1] - lse)) + ((xr[j][2] - lse) + (xr[j][3] - lse));
        } else {
            for (int v = threadIdx.x; v < V1; v += 256) s += x[v] - lse;
        }
        s = block_sum_256(s, red);
        term = (1.0f - eps) * term + (eps / (float)V1) * s;
    }
    if (threadIdx.x == 0) {
        lse_out[r] = lse;
        row_loss[b * T + t] = -mk * term;
    }
}
template <bool VEC>
__global__ __launch_bounds__(256) void xe_logits_bwd_k(float* __restrict__ logits, long ldl, int B, int T, int V1,
                                                       const int64_t* __restrict__ target, long ld_t,
                                                       const float* __restrict__ mask, long ld_m, float eps, float gcoef,
                                                       const float* __restrict__ gdev, const float* __restrict__ lse_in) {
    const int r = blockIdx.x, t = r / B, b = r - t * B;
    float* x = logits + r * ldl;
    long tg = target[b * ld_t + t];
    if (tg < 0 || tg >= V1) tg = 0;
    const float c = mask[b * ld_m + t] * (gdev ? gcoef * gdev[0] : gcoef);     // mask * d loss / B
    const float lse = lse_in[r], uni = eps / (float)V1, hot = 1.0f - eps;
    if constexpr (VEC) {
        const int n4 = V1 >> 2;
        for (int i = threadIdx.x; i < n4; i += 256) {
            lsm_f32x4 v = *reinterpret_cast<const lsm_f32x4*>(x + 4 * i), o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = c * ((expf(v[e] - lse) - uni) - ((4 * i + e == tg) ? hot : 0.f));
            *reinterpret_cast<lsm_f32x4*>(x + 4 * i) = o;
        }
    } else {
        for (int v = threadIdx.x; v < V1; v += 256) x[v] = c * ((expf(x[v] - lse) - uni) - ((v == tg) ? hot : 0.f));
    }
}
extern "C" int rfn_xe_logits_fwd(const float* logits, int64_t ldl, int B, int T, int V1, const int64_t* target,
                                 int64_t ld_target, const float* mask, int64_t ld_mask, float eps, float* lse,
                                 float* scratch, float* loss_out, int accumulate_loss, void* stream) {
    if (B <= 0 || T <= 0 || V1 <= 0 || ldl < V1 || eps < 0.f || eps >= 1.f) return RFN_ERR_SHAPE;
    if (!logits || !target || !mask || !lse || !scratch || !loss_out) return RFN_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (lsm_vec_ok(logits, ldl, V1, nullptr, 0, 0))
        hipLaunchKernelGGL(xe_logits_fwd_k<true>, dim3(B * T), dim3(256), 0, st, logits, (long)ldl, B, T, V1, target,
                           (long)ld_target, mask, (long)ld_mask, eps, lse, scratch);
    else
        hipLaunchKernelGGL(xe_logits_fwd_k<false>, dim3(B * T), dim3(256), 0, st, logits, (long)ldl, B, T, V1, target,
                           (long)ld_target, mask, (long)ld_mask, eps, lse, scratch);
    RFN_CHECK_LAUNCH();
    hipLaunchKernelGGL(sum_k, dim3(1), dim3(256), 0, st, scratch, B * T, 1.0f / (float)B, loss_out, accumulate_loss);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}
extern "C" int rfn_xe_logits_bwd(float* logits, int64_t ldl, int B, int T, int V1, const int64_t* target, int64_t ld_target,
                                 const float* mask, int64_t ld_mask, float eps, const float* lse, float gscale,
                                 const float* gscale_dev, void* stream) {
    if (B <= 0 || T <= 0 || V1 <= 0 || ldl < V1 || eps < 0.f || eps >= 1.f) return RFN_ERR_SHAPE;
    if (!logits || !target || !mask || !lse) return RFN_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (lsm_vec_ok(logits, ldl, V1, nullptr, 0, 0))
        hipLaunchKernelGGL(xe_logits_bwd_k<true>, dim3(B * T), dim3(256), 0, st, logits, (long)ldl, B, T, V1, target,
                           (long)ld_target, mask, (long)ld_mask, eps, gscale / (float)B, gscale_dev, lse);
    else
        hipLaunchKernelGGL(xe_logits_bwd_k<false>, dim3(B * T), dim3(256), 0, st, logits, (long)ldl, B, T, V1, target,
                           (long)ld_target, mask, (long)ld_mask, eps, gscale / (float)B, gscale_dev, lse);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}

// ---- RL reward criterion, policy + entropy terms (misc/utils.py:50-72) ------------------------------------
// One block per (b, t) row:  term = -pol(b,t) * mask(b,t) + entropy_reg * mask0(b,t) * sum_v lp*exp(lp),
// mask0 = seq > 0, mask = [1, mask0[:, :-1]] (the step AFTER an END still counts), pol = input*reward or the
// PPO-clip surrogate min(surr1, clamp(surr1, 1-c, 1+c)*reward) with surr1 = exp(input)/(1e-5+exp(old))*reward
// (written as in the reference: it clamps surr1, not the ratio).  Gradients w.r.t. input and logprobs_all.
__global__ __launch_bounds__(256) void rl_loss_k(const float* __restrict__ inp, long ld_in, const int64_t* __restrict__ seq,
                                                 long ld_seq, const float* __restrict__ reward, long ld_rw,
                                                 const float* __restrict__ lp_all, long lp_sb, long lp_st, int T, int V1,
                                                 float entropy_reg, const float* __restrict__ old_lp, long ld_old,
                                                 int use_ppo, float ppo_clip, float inv_B_host, float* __restrict__ row_loss,
                                                 float* __restrict__ d_inp, long ld_din, float* __restrict__ d_lp,
                                                 long dlp_sb, long dlp_st, int T_all, const float* __restrict__ gdev) {
    __shared__ float red[4];
    // T_all >= T rows per caption are launched; rows t >= T do not enter the loss: their d_logprobs_all row is zero
    const int r = blockIdx.x, b = r / T_all, t = r - b * T_all;
    if (t >= T) {
        if (d_lp) {
            float* d = d_lp + b * dlp_sb + t * dlp_st;
            for (int v = threadIdx.x; v < V1; v += 256) d[v] = 0.f;
        }
        return;
    }
    const float inv_B = gdev ? inv_B_host * gdev[0] : inv_B_host;   // upstream d loss, read on the device (no host sync)
    const float m0 = (seq[b * ld_seq + t] > 0) ? 1.f : 0.f;
    const float mk = (t == 0) ? 1.f : ((seq[b * ld_seq + t - 1] > 0) ? 1.f : 0.f);
    const float* lp = lp_all + b * lp_sb + t * lp_st;
    if (row_loss) {
        float e = 0.f;
        if (m0 != 0.f && entropy_reg != 0.f)
            for (int v = threadIdx.x; v < V1; v += 256) e += lp[v] * expf(lp[v]);
        e = block_sum_256(e, red);
        if (threadIdx.x == 0) {
            const float x = inp[b * ld_in + t], rw = reward[b * ld_rw + t];
            float pol;
            if (use_ppo) {
                const float ratio = expf(x) / (1e-5f + expf(old_lp[b * ld_old + t]));
                const float s1 = ratio * rw;
                const float s2 = fminf(fmaxf(s1, 1.f - ppo_clip), 1.f + ppo_clip) * rw;
                pol = fminf(s1, s2);
            } else {
                pol = x * rw;
            }
            row_loss[b * T + t] = -pol * mk + entropy_reg * m0 * e;
        }
    }
    if (d_lp) {
        float* d = d_lp + b * dlp_sb + t * dlp_st;
        const float c = entropy_reg * m0 * inv_B;
        for (int v = threadIdx.x; v < V1; v += 256) d[v] = (c != 0.f) ? c * expf(lp[v]) * (1.f + lp[v]) : 0.f;
    }
    if (d_inp && threadIdx.x == 0) {
        const float x = inp[b * ld_in + t], rw = reward[b * ld_rw + t];
        float g;
        if (use_ppo) {
            const float ratio = expf(x) / (1e-5f + expf(old_lp[b * ld_old + t]));
            const float s1 = ratio * rw;
            const float cl = fminf(fmaxf(s1, 1.f - ppo_clip), 1.f + ppo_clip);
            const float s2 = cl * rw;
            const float ds1 = ratio * rw;                                        // d surr1 / d input
            const float ds2 = (s1 > 1.f - ppo_clip && s1 < 1.f + ppo_clip) ? ds1 * rw : 0.f;
            g = (s1 <= s2) ? ds1 : ds2;
        } else {
            g = rw;
        }
        d_inp[b * ld_din + t] = -g * mk * inv_B;
    }
}
extern "C" int rfn_rl_loss_ex(const float* input, int64_t ld_in, const int64_t* seq, int64_t ld_seq, const float* reward,
                              int64_t ld_rw, const float* logprobs_all, int64_t lp_sb, int64_t lp_st, int B, int T, int T_all,
                              int V1, float entropy_reg, const float* old_logprobs, int64_t ld_old, int use_ppo,
                              float ppo_clip, const float* gscale_dev, float* scratch, float* loss_out, int accumulate_loss,
                              float* d_input, int64_t ld_din, float* d_logprobs_all, int64_t dlp_sb, int64_t dlp_st,
                              void* stream) {
    if (B <= 0 || T <= 0 || V1 <= 0 || T_all < T) return RFN_ERR_SHAPE;
    if (!input || !seq || !reward || !logprobs_all || (loss_out && !scratch) || (use_ppo && !old_logprobs))
        return RFN_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int rows = d_logprobs_all ? T_all : T;      // the extra rows exist only to zero d_logprobs_all[:, T:]
    hipLaunchKernelGGL(rl_loss_k, dim3(B * rows), dim3(256), 0, st, input, (long)ld_in, seq, (long)ld_seq, reward,
                       (long)ld_rw, logprobs_all, (long)lp_sb, (long)lp_st, T, V1, entropy_reg, old_logprobs,
                       (long)ld_old, use_ppo, ppo_clip, 1.0f / (float)B, loss_out ? scratch : nullptr, d_input,
                       (long)ld_din, d_logprobs_all, (long)dlp_sb, (long)dlp_st, rows, gscale_dev);
    RFN_CHECK_LAUNCH();
    if (loss_out) {
        hipLaunchKernelGGL(sum_k, dim3(1), dim3(256), 0, st, scratch, B * T, 1.0f / (float)B, loss_out, accumulate_loss);
        RFN_CHECK_LAUNCH();
    }
    return RFN_OK;
}
extern "C" int rfn_rl_loss(const float* input, int64_t ld_in, const int64_t* seq, int64_t ld_seq, const float* reward,
                           int64_t ld_rw, const float* logprobs_all, int64_t lp_sb, int64_t lp_st, int B, int T, int V1,
                           float entropy_reg, const float* old_logprobs, int64_t ld_old, int use_ppo, float ppo_clip,
                           float* scratch, float* loss_out, int accumulate_loss, float* d_input, int64_t ld_din,
                           float* d_logprobs_all, int64_t dlp_sb, int64_t dlp_st, void* stream) {
    return rfn_rl_loss_ex(input, ld_in, seq, ld_seq, reward, ld_rw, logprobs_all, lp_sb, lp_st, B, T, T, V1, entropy_reg,
                          old_logprobs, ld_old, use_ppo, ppo_clip, nullptr, scratch, loss_out, accumulate_loss, d_input,
                          ld_din, d_logprobs_all, dlp_sb, dlp_st, stream);
}

// ---- nn.MultiLabelMarginLoss (mean) -----------------------------------------------------------------
// Row b: targets = ids before the first -1; loss_b = sum_{j in targets} sum_{i not target}
// max(0, 1 - x[j] + x[i]) / K.  One block per row; hit counts per target through LDS integer atomics.
struct MlmHeads {
    const float* pred[RFN_MAX_ENC + 1];
    float* dpred[RFN_MAX_ENC + 1];
};
__global__ __launch_bounds__(256) void mlm_k(const MlmHeads hd, int K, const int64_t* __restrict__ target, float gcoef,
                                             const float* __restrict__ gdev, float* __restrict__ row_loss) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* x = sm;                                   // [K]
    int* tg = reinterpret_cast<int*>(sm + K);        // [K] target list
    int* cnt = tg + K;                               // [K] hits per target slot
    unsigned char* is_t = reinterpret_cast<unsigned char*>(cnt + K);  // [K]
    __shared__ int nt_s;
    __shared__ float red[4];
    const int b = blockIdx.x, head = blockIdx.y;
    const float* pred = hd.pred[head];
    float* dpred = hd.dpred[head];
    if (gdev) gcoef *= gdev[0];
    for (int i = threadIdx.x; i < K; i += 256) {
        x[i] = pred[(long)b * K + i];
        is_t[i] = 0;
        cnt[i] = 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int n = 0;
        for (; n < K; ++n) {
            const long t = target[(long)b * K + n];
            if (t < 0 || t >= K) break;
            tg[n] = (int)t;
            is_t[t] = 1;
        }
        nt_s = n;
    }
    __syncthreads();
    const int nt = nt_s;
    const float invK = 1.0f / (float)K;
    float loss = 0.f;
    for (int i = threadIdx.x; i < K; i += 256) {
        float gi = 0.f;
        if (!is_t[i]) {
            for (int n = 0; n < nt; ++n) {
                const float z = 1.0f - x[tg[n]] + x[i];
                if (z > 0.f) {
                    loss += z;
                    gi += 1.0f;
                    atomicAdd(&cnt[n], 1);
                }
            }
        }
        if (dpred) dpred[(long)b * K + i] = gi * invK * gcoef;  // targets fixed up below
    }
    loss = block_sum_256(loss, red);  // also orders the dpred writes above before the fix-up
    if (row_loss && threadIdx.x == 0) row_loss[(long)head * gridDim.x + b] = loss * invK;
    if (dpred && threadIdx.x == 0) {
        // a duplicated target id receives the hits of every slot that names it
        for (int n = 0; n < nt; ++n) dpred[(long)b * K + tg[n]] = 0.f;
        for (int n = 0; n < nt; ++n) dpred[(long)b * K + tg[n]] -= (float)cnt[n] * invK * gcoef;
    }
}
// out[0] (+)= scale * sum(x[g*n .. g*n+n)) for g = 0..G-1 in order: the per-head sums are added one after the other,
// exactly as G calls of sum_k would
__global__ __launch_bounds__(256) void sum_groups_k(const float* __restrict__ x, int n, int G, float scale,
                                                    float* __restrict__ out, int accumulate) {
    __shared__ float red[4];
    float total = accumulate ? out[0] : 0.f;
    for (int g = 0; g < G; ++g) {
        float s = 0.f;
        for (int i = threadIdx.x; i < n; i += 256) s += x[(long)g * n + i];
        s = block_sum_256(s, red);
        total = (g == 0 && !accumulate) ? s * scale : total + s * scale;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = total;
}
extern "C" int rfn_multilabel_margin_grouped(int nheads, const float* const* preds, int B, int K, const int64_t* target,
                                             float scale, float gscale, const float* gscale_dev, float* scratch,
                                             float* loss_out, int accumulate_loss, float* const* dpreds,
                                             void* stream) {
    if (nheads < 1 || nheads > RFN_MAX_ENC + 1 || B <= 0 || K <= 0) return RFN_ERR_SHAPE;
    if (!preds || !target || (loss_out && !scratch)) return RFN_ERR_ARG;
    const size_t lds = (size_t)K * (sizeof(float) + 2 * sizeof(int) + 1) + 16;
    if (lds > 60 * 1024) return RFN_ERR_SHAPE;
    MlmHeads hd;
    memset(&hd, 0, sizeof(hd));
    for (int h = 0; h < nheads; ++h) {
        if (!preds[h]) return RFN_ERR_ARG;
        hd.pred[h] = preds[h];
        hd.dpred[h] = dpreds ? dpreds[h] : nullptr;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(mlm_k, dim3(B, nheads), dim3(256), lds, st, hd, K, target, scale * gscale / (float)B, gscale_dev,
                       loss_out ? scratch : nullptr);
    RFN_CHECK_LAUNCH();
    if (loss_out) {
        hipLaunchKernelGGL(sum_groups_k, dim3(1), dim3(256), 0, st, scratch, B, nheads, scale / (float)B, loss_out,
                           accumulate_loss);
        RFN_CHECK_LAUNCH();
    }
    return RFN_OK;
}
extern "C" int rfn_multilabel_margin(const float* pred, int B, int K, const int64_t* target, float scale, float gscale,
                                     float* scratch, float* loss_out, int accumulate_loss, float* dpred,
                                     void* stream) {
    if (!pred) return RFN_ERR_ARG;
    return rfn_multilabel_margin_grouped(1, &pred, B, K, target, scale, gscale, nullptr, scratch, loss_out,
                                         accumulate_loss, dpred ? &dpred : nullptr, stream);
}

// ---- clip_gradient + Adam (misc/utils.py:292-296, train.py:69-71) -----------------------------------
__device__ __forceinline__ void adam_elem(float& pv, float gv, float& mv, float& vv, float lr_over_bc1, float beta1,
                                          float beta2, float eps, float inv_sqrt_bc2, float wd, float clip,
                                          float gscale) {
    // The operation tree is pinned: no implicit contraction, fused multiply-adds only where written.  The per-bucket kernel,
    // the multi-bucket kernel and its graph-replayable form (scalars from memory instead of kernel arguments) must produce the
    // same bits, and left to itself the compiler fuses differently depending on where the scalars live.
#pragma clang fp contract(off)
    gv *= gscale;
    gv = fminf(fmaxf(gv, -clip), clip);
    gv = __builtin_fmaf(wd, pv, gv);
    mv = __builtin_fmaf(beta1, mv, (1.0f - beta1) * gv);
    vv = __builtin_fmaf(beta2, vv, ((1.0f - beta2) * gv) * gv);
    pv = pv - (lr_over_bc1 * mv) / __builtin_fmaf(sqrtf(vv), inv_sqrt_bc2, eps);
}
// VEC: 16 B per lane on all seven streams (flat buckets are 16-B aligned and a multiple of 4 long)
template <bool VEC>
__global__ __launch_bounds__(256) void adam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, long n, float lr_over_bc1, float beta1,
                                              float beta2, float eps, float inv_sqrt_bc2, float wd, float clip,
                                              float gscale) {
    const long stride = (long)gridDim.x * 256;
    if constexpr (VEC) {
        const long n4 = n >> 2;
        f32x4* p4 = reinterpret_cast<f32x4*>(p);
        const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
        f32x4* m4 = reinterpret_cast<f32x4*>(m);
        f32x4* v4 = reinterpret_cast<f32x4*>(v);
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            f32x4 pv = p4[i], mv = m4[i], vv = v4[i];
            const f32x4 gv = g4[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = pv[e], me = mv[e], ve = vv[e];
                adam_elem(pe, gv[e], me, ve, lr_over_bc1, beta1, beta2, eps, inv_sqrt_bc2, wd, clip, gscale);
                pv[e] = pe;
                mv[e] = me;
                vv[e] = ve;
            }
            m4[i] = mv;
            v4[i] = vv;
            p4[i] = pv;
        }
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
            float pv = p[i], mv = m[i], vv = v[i];
            adam_elem(pv, g[i], mv, vv, lr_over_bc1, beta1, beta2, eps, inv_sqrt_bc2, wd, clip, gscale);
            m[i] = mv;
            v[i] = vv;
            p[i] = pv;
        }
    }
}
extern "C" int rfn_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, float grad_clip, float grad_scale, int step,
                             void* stream) {
    if (n <= 0 || step < 1) return RFN_ERR_SHAPE;
    if (!p || !g || !m || !v) return RFN_ERR_ARG;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const bool vec = (n % 4 == 0) && rfn_aligned16(p) && rfn_aligned16(g) && rfn_aligned16(m) && rfn_aligned16(v);
    const long work = vec ? n / 4 : n;
    const int blocks = (int)(work / 256 + 1 < 4096 ? work / 256 + 1 : 4096);
    if (vec)
        hipLaunchKernelGGL(adam_k<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n,
                           (float)(lr / bc1), beta1, beta2, eps, (float)(1.0 / sqrt(bc2)), weight_decay, grad_clip,
                           grad_scale);
    else
        hipLaunchKernelGGL(adam_k<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n,
                           (float)(lr / bc1), beta1, beta2, eps, (float)(1.0 / sqrt(bc2)), weight_decay, grad_clip,
                           grad_scale);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}

// The same update for up to RFN_ADAM_MAXBUCKET flat buckets in ONE launch: every block walks the buckets in order with the
// grid-stride loop of adam_k.  Ten launches of 34-277 MB buckets each paid their own ramp and tail (4.7-5.1 TB/s over the
// step); one launch streams at the rate a single large bucket reaches (5.6 TB/s).  Element arithmetic is adam_elem's:
// bit-identical to the per-bucket calls.
struct AdamBuckets {
    float* p[RFN_ADAM_MAXBUCKET];
    const float* g[RFN_ADAM_MAXBUCKET];
    float* m[RFN_ADAM_MAXBUCKET];
    float* v[RFN_ADAM_MAXBUCKET];
    long n[RFN_ADAM_MAXBUCKET];
    int vec[RFN_ADAM_MAXBUCKET];
    int nb;
};
// coef != NULL: the two step-dependent scalars (lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t)) are read from device memory
// instead of the kernel arguments -- the form a captured HIP graph replays, where kernel arguments are frozen
// (rfn_adam_step_multi_coef).  Same values, same arithmetic.
__global__ __launch_bounds__(256) void adam_multi_k(const AdamBuckets B, float lr_over_bc1, float beta1, float beta2, float eps,
                                                    float inv_sqrt_bc2, float wd, float clip, float gscale,
                                                    const float* __restrict__ coef) {
    if (coef) {
        lr_over_bc1 = coef[0];
        inv_sqrt_bc2 = coef[1];
    }
    const long stride = (long)gridDim.x * 256, i0 = (long)blockIdx.x * 256 + threadIdx.x;
    for (int k = 0; k < B.nb; ++k) {
        if (B.vec[k]) {
            const long n4 = B.n[k] >> 2;
            f32x4* p4 = reinterpret_cast<f32x4*>(B.p[k]);
            const f32x4* g4 = reinterpret_cast<const f32x4*>(B.g[k]);
            f32x4* m4 = reinterpret_cast<f32x4*>(B.m[k]);
            f32x4* v4 = reinterpret_cast<f32x4*>(B.v[k]);
            for (long i = i0; i < n4; i += stride) {
                f32x4 pv = p4[i], mv = m4[i], vv = v4[i];
                const f32x4 gv = g4[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float pe = pv[e], me = mv[e], ve = vv[e];
                    adam_elem(pe, gv[e], me, ve, lr_over_bc1, beta1, beta2, eps, inv_sqrt_bc2, wd, clip, gscale);
                    pv[e] = pe;
                    mv[e] = me;
                    vv[e] = ve;
                }
                m4[i] = mv;
                v4[i] = vv;
                p4[i] = pv;
            }
        } else {
            float* p = B.p[k];
            const float* g = B.g[k];
            float* m = B.m[k];
            float* v = B.v[k];
            for (long i = i0; i < B.n[k]; i += stride) {
                float pv = p[i], mv = m[i], vv = v[i];
                adam_elem(pv, g[i], mv, vv, lr_over_bc1, beta1, beta2, eps, inv_sqrt_bc2, wd, clip, gscale);
                m[i] = mv;
                v[i] = vv;
                p[i] = pv;
            }
        }
    }
}
static int adam_multi_launch(int nbuckets, float* const* p, const float* const* g, float* const* m, float* const* v,
                             const int64_t* n, float lr, float beta1, float beta2, float eps, float weight_decay,
                             float grad_clip, float grad_scale, int step, const float* coef, void* stream) {
    if (nbuckets < 1 || nbuckets > RFN_ADAM_MAXBUCKET || (!coef && step < 1)) return RFN_ERR_SHAPE;
    if (!p || !g || !m || !v || !n) return RFN_ERR_ARG;
    AdamBuckets B;
    memset(&B, 0, sizeof(B));
    long most = 0;
    for (int k = 0; k < nbuckets; ++k) {
        if (n[k] <= 0) return RFN_ERR_SHAPE;
        if (!p[k] || !g[k] || !m[k] || !v[k]) return RFN_ERR_ARG;
        B.p[k] = p[k]; B.g[k] = g[k]; B.m[k] = m[k]; B.v[k] = v[k]; B.n[k] = (long)n[k];
        B.vec[k] = (n[k] % 4 == 0) && rfn_aligned16(p[k]) && rfn_aligned16(g[k]) && rfn_aligned16(m[k]) && rfn_aligned16(v[k]);
        const long work = B.vec[k] ? (long)n[k] / 4 : (long)n[k];
        most = work > most ? work : most;
    }
    B.nb = nbuckets;
    float c0 = 0.f, c1 = 0.f;
    if (!coef) {
        const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
        c0 = (float)(lr / bc1);
        c1 = (float)(1.0 / sqrt(bc2));
    }
    const int blocks = (int)(most / 256 + 1 < 4096 ? most / 256 + 1 : 4096);
    hipLaunchKernelGGL(adam_multi_k, dim3(blocks), dim3(256), 0, (hipStream_t)stream, B, c0, beta1, beta2, eps, c1, weight_decay,
                       grad_clip, grad_scale, coef);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}
extern "C" int rfn_adam_step_multi(int nbuckets, float* const* p, const float* const* g, float* const* m, float* const* v,
                                   const int64_t* n, float lr, float beta1, float beta2, float eps, float weight_decay,
                                   float grad_clip, float grad_scale, int step, void* stream) {
    return adam_multi_launch(nbuckets, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, grad_clip, grad_scale, step, nullptr,
                             stream);
}
extern "C" int rfn_adam_step_multi_coef(int nbuckets, float* const* p, const float* const* g, float* const* m, float* const* v,
                                        const int64_t* n, const float* coef_dev, float beta1, float beta2, float eps,
                                        float weight_decay, float grad_clip, float grad_scale, void* stream) {
    if (!coef_dev) return RFN_ERR_ARG;
    return adam_multi_launch(nbuckets, p, g, m, v, n, 0.f, beta1, beta2, eps, weight_decay, grad_clip, grad_scale, 0, coef_dev,
                             stream);
}

// ---- multinomial pick of sample() / scheduled sampling (misc/RecurrentFusionModel.py:623-631, 260-270) --------
// One block per row: inverse-CDF draw from p[v] ~ exp(logp[v] * inv_temperature) with the caller's uniform u[b].
// Thread t owns the contiguous index range [t*C, (t+1)*C); the 256 range sums are scanned in LDS in index order, the
// thread whose range contains u * total walks it.  The reference draws on the host with torch.multinomial; RNG streams
// are not portable anyway, the DISTRIBUTION is the same and the draw is a deterministic function of (logp, u).
__global__ __launch_bounds__(256) void multinomial_pick_k(const float* __restrict__ logp, long ldl, int V1,
                                                          float inv_temp, const float* __restrict__ u,
                                                          const float* __restrict__ coin, float keep_prob,
                                                          int64_t* __restrict__ ids, long ld_ids) {
    __shared__ float part[256];
    __shared__ int pick;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (coin && !(coin[b] < keep_prob)) return;   // block-uniform: this row keeps the token it already has
    const float* x = logp + b * ldl;
    const int C = (V1 + 255) / 256, v0 = tid * C, v1 = min(V1, v0 + C);
    float sum = 0.f;
    for (int v = v0; v < v1; ++v) sum += __expf(x[v] * inv_temp);
    part[tid] = sum;
    if (tid == 0) pick = -1;
    __syncthreads();
    float before = 0.f, total = 0.f;
    for (int k = 0; k < 256; ++k) {   // 256 LDS broadcasts per thread: same order for everyone
        const float pk = part[k];
        if (k < tid) before += pk;
        total += pk;
    }
    const float target = u[b] * total;
    // the owner is the first range with before <= target < before + sum; target == total (u -> 1) goes to the last
    // non-empty range
    const bool owner = sum > 0.f && target >= before && (target < before + sum);
    if (owner) atomicMax(&pick, tid);
    __syncthreads();
    if (pick < 0) {   // rounding put target at / past the total: last range with mass
        if (sum > 0.f) atomicMax(&pick, tid);
        __syncthreads();
    }
    if (tid == pick) {
        float acc = before;
        int chosen = v0;   // rounding corner (target at / past the range's end): the last index WITH mass
        for (int v = v0; v < v1; ++v) {
            const float e = __expf(x[v] * inv_temp);
            acc += e;
            if (e > 0.f) chosen = v;
            if (target < acc) break;
        }
        ids[b * ld_ids] = chosen;
    }
}
extern "C" int rfn_multinomial_pick(const float* logp, int64_t ldl, int B, int V1, float inv_temperature, const float* u,
                                    const float* coin, float keep_prob, int64_t* ids, int64_t ld_ids, void* stream) {
    if (B <= 0 || V1 <= 0 || ldl < V1 || !(inv_temperature > 0.f)) return RFN_ERR_SHAPE;
    if (!logp || !u || !ids) return RFN_ERR_ARG;
    hipLaunchKernelGGL(multinomial_pick_k, dim3(B), dim3(256), 0, (hipStream_t)stream, logp, (long)ldl, V1,
                       inv_temperature, u, coin, keep_prob, ids, (long)ld_ids);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}

// ---- greedy pick of sample() (misc/RecurrentFusionModel.py:619-649) -----------------------------------
__global__ __launch_bounds__(256) void greedy_pick_k(const float* __restrict__ logp, long ldl, int V1, int t,
                                                     int64_t* __restrict__ next_ids, int64_t* __restrict__ seq_out,
                                                     long ld_seq, float* __restrict__ lp_out, long ld_lp,
                                                     const int32_t* unf_prev, int32_t* unf_out,
                                                     const int64_t* given /* may alias next_ids */) {
    __shared__ float vs[4];
    __shared__ int is[4];
    const int b = blockIdx.x;
    const float* x = logp + b * ldl;
    if (given) {   // the token was drawn elsewhere (multinomial): record its log-prob and the finished flags only
        if (threadIdx.x == 0) {
            long it = given[b];
            if (it < 0 || it >= V1) it = 0;
            int unf = (t == 1) ? 1 : unf_prev[b];
            unf = unf && (it > 0);
            unf_out[b] = unf;
            next_ids[b] = it;
            seq_out[b * ld_seq] = unf ? it : 0;
            lp_out[b * ld_lp] = x[it];
        }
        return;
    }
    float m = -INFINITY;
    int mi = 0x7fffffff;
    for (int v = threadIdx.x; v < V1; v += 256) {
        const float xv = x[v];
        if (xv > m) {  // strided ascending scan keeps the first maximum per thread
            m = xv;
            mi = v;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(m, o, 64);
        const int oi = __shfl_xor(mi, o, 64);
        if (om > m || (om == m && oi < mi)) {
            m = om;
            mi = oi;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        vs[wave] = m;
        is[wave] = mi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (vs[w] > m || (vs[w] == m && is[w] < mi)) {
                m = vs[w];
                mi = is[w];
            }
        int unf = (t == 1) ? 1 : unf_prev[b];
        unf = unf && (mi > 0);
        unf_out[b] = unf;
        next_ids[b] = mi;
        seq_out[b * ld_seq] = unf ? mi : 0;
        lp_out[b * ld_lp] = m;
    }
}
extern "C" int rfn_pick_record(const float* logp, int64_t ldl, int B, int V1, int t, const int64_t* given,
                               int64_t* next_ids, int64_t* seq_out, int64_t ld_seq, float* lp_out, int64_t ld_lp,
                               const int32_t* unf_prev, int32_t* unf_out, void* stream) {
    if (B <= 0 || V1 <= 0 || t < 1) return RFN_ERR_SHAPE;
    if (!logp || !next_ids || !seq_out || !lp_out || !unf_out || (t > 1 && !unf_prev)) return RFN_ERR_ARG;
    hipLaunchKernelGGL(greedy_pick_k, dim3(B), dim3(given ? 64 : 256), 0, (hipStream_t)stream, logp, (long)ldl, V1, t, next_ids,
                       seq_out, (long)ld_seq, lp_out, (long)ld_lp, unf_prev, unf_out, given);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}
extern "C" int rfn_greedy_pick(const float* logp, int64_t ldl, int B, int V1, int t, int64_t* next_ids,
                               int64_t* seq_out, int64_t ld_seq, float* lp_out, int64_t ld_lp,
                               const int32_t* unf_prev, int32_t* unf_out, void* stream) {
    return rfn_pick_record(logp, ldl, B, V1, t, nullptr, next_ids, seq_out, ld_seq, lp_out, ld_lp, unf_prev, unf_out, stream);
}
