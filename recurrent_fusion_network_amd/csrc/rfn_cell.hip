// LSTM gate epilogue shared by the three cells of the recurrent-fusion path.
//
// Reference: misc/RecurrentFusionModel.py:55-73 (stage I), misc/LSTMSoftMultiAttentionFeatArrayNoInputCore.py
// :54-72 (stage II), misc/LSTMSoftAttentionCore.py:83-101 (decoder).  Gate chunk order in the 4R
// vector is [in | forget | out | g] (NOT cuDNN's i,f,g,o); the recurrent h is the POST-dropout
// value, c is never dropped.
// Element-wise and tiny (B*R elements): one thread per (b, j), coalesced along j.
#include "rfn_common.h"

// dropout masks: rfn_philox_uniform (rfn_common.h)

__global__ __launch_bounds__(256) void lstm_fwd_k(float* __restrict__ gates, long ldg, const float* c_prev /* may alias c_next */,
                                                  long ldcp, float* c_next, long ldcn,
                                                  float* __restrict__ h_next, long ldh, int B, int R, int maxout,
                                                  float drop_p, uint64_t seed, uint64_t offset, long gs_g, long gs_cp,
                                                  long gs_cn, long gs_h) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * R) return;
    const int grp = blockIdx.y;   // independent cells of one step (the M encoders of stage I)
    gates += grp * gs_g; c_prev += grp * gs_cp; c_next += grp * gs_cn; h_next += grp * gs_h;
    offset += (uint64_t)grp;
    const int b = (int)(idx / R), j = (int)(idx - (long)b * R);
    float* g = gates + b * ldg;
    const float ig = rfn_sigmoid(g[j]);
    const float fg = rfn_sigmoid(g[R + j]);
    const float og = rfn_sigmoid(g[2 * R + j]);
    float gg;
    if (maxout) {  // in_transform = max of the two candidate chunks, no tanh; chunk 4 keeps the selector
        const float a = g[3 * R + j], b2 = g[4 * R + j];
        gg = fmaxf(a, b2);
        g[4 * R + j] = (a > b2) ? 1.f : ((a == b2) ? 0.5f : 0.f);   // a tie splits the gradient, as torch.max(a, b) does
    } else {
        gg = tanhf(g[3 * R + j]);
    }
    g[j] = ig;
    g[R + j] = fg;
    g[2 * R + j] = og;
    g[3 * R + j] = gg;
    const float c = fg * c_prev[b * ldcp + j] + ig * gg;
    c_next[b * ldcn + j] = c;
    float hv = og * tanhf(c);
    if (drop_p > 0.f) {
        const float u = rfn_philox_uniform(seed, offset, (uint64_t)idx);
        hv = (u >= drop_p) ? hv * (1.0f / (1.0f - drop_p)) : 0.f;
    }
    h_next[b * ldh + j] = hv;
}

extern "C" int rfn_lstm_fwd_grouped(float* gates, int64_t ldg, const float* c_prev, int64_t ldcp, float* c_next,
                                    int64_t ldcn, float* h_next, int64_t ldh, int B, int R, int maxout, float drop_p,
                                    uint64_t seed, uint64_t offset, int G, int64_t gs_gates, int64_t gs_cprev,
                                    int64_t gs_cnext, int64_t gs_h, void* stream) {
    if (B <= 0 || R <= 0 || G < 1 || drop_p < 0.f || drop_p >= 1.f) return RFN_ERR_SHAPE;
    if (!gates || !c_prev || !c_next || !h_next) return RFN_ERR_ARG;
    hipLaunchKernelGGL(lstm_fwd_k, dim3(rfn_cdiv((long)B * R, 256), G), dim3(256), 0, (hipStream_t)stream, gates,
                       (long)ldg, c_prev, (long)ldcp, c_next, (long)ldcn, h_next, (long)ldh, B, R, maxout ? 1 : 0, drop_p,
                       seed, offset, (long)gs_gates, (long)gs_cprev, (long)gs_cnext, (long)gs_h);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}
extern "C" int rfn_lstm_fwd(float* gates, int64_t ldg, const float* c_prev, int64_t ldcp, float* c_next,
                            int64_t ldcn, float* h_next, int64_t ldh, int B, int R, int maxout, float drop_p,
                            uint64_t seed, uint64_t offset, void* stream) {
    return rfn_lstm_fwd_grouped(gates, ldg, c_prev, ldcp, c_next, ldcn, h_next, ldh, B, R, maxout, drop_p, seed, offset,
                                1, 0, 0, 0, 0, stream);
}

__global__ __launch_bounds__(256) void lstm_bwd_k(float* __restrict__ gates, long ldg, const float* __restrict__ c_prev,
                                                  long ldcp, const float* __restrict__ c_next, long ldcn,
                                                  const float* __restrict__ dh, long lddh,
                                                  const float* dc_next /* may alias dc_prev */, long lddcn,
                                                  float* dc_prev, long lddcp, int B, int R, int maxout, float drop_p,
                                                  uint64_t seed, uint64_t offset, long gs_g, long gs_c, long gs_dh,
                                                  long gs_dc) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * R) return;
    const int grp = blockIdx.y;
    gates += grp * gs_g; c_prev += grp * gs_c; c_next += grp * gs_c; dh += grp * gs_dh;
    if (dc_next) dc_next += grp * gs_dc;
    dc_prev += grp * gs_dc;
    offset += (uint64_t)grp;
    const int b = (int)(idx / R), j = (int)(idx - (long)b * R);
    float* g = gates + b * ldg;
    const float ig = g[j], fg = g[R + j], og = g[2 * R + j], gg = g[3 * R + j];
    float dhv = dh[b * lddh + j];
    if (drop_p > 0.f) {
        const float u = rfn_philox_uniform(seed, offset, (uint64_t)idx);
        dhv = (u >= drop_p) ? dhv * (1.0f / (1.0f - drop_p)) : 0.f;
    }
    const float tc = tanhf(c_next[b * ldcn + j]);
    float dc = dhv * og * (1.0f - tc * tc);
    if (dc_next) dc += dc_next[b * lddcn + j];
    const float d_o = dhv * tc;
    const float d_i = dc * gg;
    const float d_f = dc * c_prev[b * ldcp + j];
    const float d_g = dc * ig;
    g[j] = d_i * ig * (1.0f - ig);
    g[R + j] = d_f * fg * (1.0f - fg);
    g[2 * R + j] = d_o * og * (1.0f - og);
    if (maxout) {  // the gradient goes to the chunk that won the max (half each on a tie)
        const float sel = g[4 * R + j];
        g[3 * R + j] = d_g * sel;
        g[4 * R + j] = d_g * (1.0f - sel);
    } else {
        g[3 * R + j] = d_g * (1.0f - gg * gg);
    }
    dc_prev[b * lddcp + j] = dc * fg;
}

extern "C" int rfn_lstm_bwd_grouped(float* gates, int64_t ldg, const float* c_prev, int64_t ldcp,
                                    const float* c_next, int64_t ldcn, const float* dh, int64_t lddh,
                                    const float* dc_next, int64_t lddcn, float* dc_prev, int64_t lddcp, int B, int R,
                                    int maxout, float drop_p, uint64_t seed, uint64_t offset, int G, int64_t gs_gates,
                                    int64_t gs_c, int64_t gs_dh, int64_t gs_dc, void* stream) {
    if (B <= 0 || R <= 0 || G < 1 || drop_p < 0.f || drop_p >= 1.f) return RFN_ERR_SHAPE;
    if (!gates || !c_prev || !c_next || !dh || !dc_prev) return RFN_ERR_ARG;
    hipLaunchKernelGGL(lstm_bwd_k, dim3(rfn_cdiv((long)B * R, 256), G), dim3(256), 0, (hipStream_t)stream, gates,
                       (long)ldg, c_prev, (long)ldcp, c_next, (long)ldcn, dh, (long)lddh, dc_next, (long)lddcn,
                       dc_prev, (long)lddcp, B, R, maxout ? 1 : 0, drop_p, seed, offset, (long)gs_gates, (long)gs_c, (long)gs_dh,
                       (long)gs_dc);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}
extern "C" int rfn_lstm_bwd(float* gates, int64_t ldg, const float* c_prev, int64_t ldcp, const float* c_next,
                            int64_t ldcn, const float* dh, int64_t lddh, const float* dc_next, int64_t lddcn,
                            float* dc_prev, int64_t lddcp, int B, int R, int maxout, float drop_p, uint64_t seed,
                            uint64_t offset, void* stream) {
    return rfn_lstm_bwd_grouped(gates, ldg, c_prev, ldcp, c_next, ldcn, dh, lddh, dc_next, lddcn, dc_prev, lddcp, B, R,
                                maxout, drop_p, seed, offset, 1, 0, 0, 0, 0, stream);
}

// keep mask of one dropout call site, as lstm_fwd_k / lstm_bwd_k regenerate it (rfn.h: rfn_dropout_mask)
__global__ __launch_bounds__(256) void dropout_mask_k(uint64_t seed, uint64_t offset, long n, float drop_p,
                                                      float* __restrict__ keep) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    keep[idx] = (drop_p > 0.f && rfn_philox_uniform(seed, offset, (uint64_t)idx) < drop_p) ? 0.f : 1.f;
}
extern "C" int rfn_dropout_mask(uint64_t seed, uint64_t offset, int64_t n, float drop_p, float* keep_out, void* stream) {
    if (n <= 0 || drop_p < 0.f || drop_p >= 1.f) return RFN_ERR_SHAPE;
    if (!keep_out) return RFN_ERR_ARG;
    hipLaunchKernelGGL(dropout_mask_k, dim3(rfn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, seed, offset, (long)n,
                       drop_p, keep_out);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}
