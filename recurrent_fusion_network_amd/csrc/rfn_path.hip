// Host-side sequencing of the recurrent-fusion caption-decoder path on one MI355X.
//
// Restates the control flow of RecurrentFusionModel.forward / sample
// (misc/RecurrentFusionModel.py:198-281, 545-658) as a fixed schedule of HIP launches on one stream:
//   phase 1 (rfn_prefix_*): fc2h -> T1 x M fusion-stage-I cells -> reason heads -> state mean ->
//                           T2 fusion-stage-II cells                       (:199-255, 283-343)
//   phase 2 (rfn_decoder_*): embed -> attention-LSTM decoder -> logit + log_softmax (:257-281)
// What is different from the reference's per-step Python loop of ~1500 ATen calls:
//   * every attention feature projection att_2_att_h does not depend on h, so all T1 (stage I),
//     T2 (stage II) step weights of one encoder are applied in ONE grouped MFMA GEMM before the
//     recurrence, and the decoder's is applied once instead of 17 times;
//   * all activations are kept time-major ((step, batch, feature)), so "stack + transpose +
//     contiguous" (:223-227, 246-251) disappears, the stage-II / decoder attention reads the thought
//     vectors through strides, and every weight gradient that is shared across steps is ONE GEMM
//     over (steps*batch) rows;
//   * the teacher-forced i2h(x_t) and logit(h_t) of all steps are batched GEMMs;
//   * backward never materialises tanh outputs: the projection slice is overwritten in place by its
//     gradient, which then feeds the grouped weight-gradient GEMM (the 77 % CPU hotspot).
// No allocation, no synchronisation, no global state: buffers come from the caller's workspace.
#include <stdio.h>
#include <string.h>
#include <initializer_list>

#include <vector>

#include "rfn_internal.h"

namespace {

// ---------------------------------------------------------------------------------------------
// canonical parameter order (names: SURVEY.md 8b)
// ---------------------------------------------------------------------------------------------
struct PIdx {
    int M, T1, T2;
    explicit PIdx(const rfn_dims* d) : M(d->M), T1(d->T1), T2(d->T2) {}
    int fc_w(int i) const { return 2 * i; }
    int fc_b(int i) const { return 2 * i + 1; }
    int embed() const { return 2 * M; }
    int logit_w() const { return 2 * M + 1; }
    int logit_b() const { return 2 * M + 2; }
    // stage I cell (t, i): k = 0 att_2_att_h.w, 1 .b, 2 h_2_att_h.w, 3 .b, 4 att_h_2_out.w, 5 .b,
    //                          6 H2h.w, 7 H2h.b, 8 z2h.w, 9 z2h.b
    int s1(int t, int i, int k) const { return 2 * M + 3 + (t * M + i) * 10 + k; }
    int rind_w(int i) const { return 2 * M + 3 + T1 * M * 10 + 2 * i; }
    int rind_b(int i) const { return rind_w(i) + 1; }
    int s2base(int t) const { return 2 * M + 3 + T1 * M * 10 + 2 * M + t * (2 + 8 * M); }
    int s2_hh_w(int t) const { return s2base(t); }
    int s2_hh_b(int t) const { return s2base(t) + 1; }
    // stage II cell t, encoder i: k = 0 z_2_h.w, 1 .b, 2 att_2_att_h.w, 3 .b, 4 h_2_att_h.w, 5 .b,
    //                                 6 att_h_2_out.w, 7 .b
    int s2(int t, int i, int k) const { return s2base(t) + 2 + 8 * i + k; }
    int r_w() const { return s2base(T2); }
    int r_b() const { return r_w() + 1; }
    // decoder: 0 i2h.w, 1 .b, 2 h2h.w, 3 .b, 4 z2h.w, 5 .b, 6 att_2_att_h.w, 7 .b, 8 h_2_att_h.w,
    //          9 .b, 10 att_h_2_out.w, 11 .b
    int dec(int k) const { return r_w() + 2 + k; }
    int count() const { return r_w() + 2 + 12; }
};

int check_dims(const rfn_dims* d) {
    if (!d) return RFN_ERR_ARG;
    if (d->M < 1 || d->M > RFN_MAX_ENC || d->R < 1 || d->A < 1 || d->E < 1 || d->T1 < 1 || d->T2 < 1 || d->K < 1 ||
        d->V1 < 2)
        return RFN_ERR_SHAPE;
    for (int i = 0; i < d->M; ++i)
        if (d->L[i] < 1 || d->D[i] < 1 || d->F[i] < 1) return RFN_ERR_SHAPE;
    // the per-phase grouped launches carry at most 64 (step, encoder) pointer slots (rfn.h, rfn_dims)
    if (d->T1 * d->M > 64 || d->T2 * d->M > 64) return RFN_ERR_SHAPE;
    if (d->drop_fusion < 0 || d->drop_fusion >= 1 || d->drop_reason < 0 || d->drop_reason >= 1 || d->drop_lm < 0 ||
        d->drop_lm >= 1)
        return RFN_ERR_SHAPE;
    return RFN_OK;
}

// ---------------------------------------------------------------------------------------------
// GEMM helpers.  lin(): Y = X W^T + b.  dx(): dX = dY W.  dw(): dW = dY^T X.
// ---------------------------------------------------------------------------------------------
rfn_gemm_seg seg_lin(const float* X, long ldx, const float* W, long ldw, int K, const float* bias) {
    rfn_gemm_seg s;
    memset(&s, 0, sizeof(s));
    s.A = X; s.lda = ldx; s.a_kfast = 1;
    s.B = W; s.ldb = ldw; s.b_kfast = 1;
    s.K = K; s.bias = bias;
    return s;
}
// dX[m, k] = sum_n dY[m, n] W[n, k]   (W is (N, Kout) row-major, ld ldw)
rfn_gemm_seg seg_dx(const float* dY, long lddy, const float* W, long ldw, int N) {
    rfn_gemm_seg s;
    memset(&s, 0, sizeof(s));
    s.A = dY; s.lda = lddy; s.a_kfast = 1;
    s.B = W; s.ldb = ldw; s.b_kfast = 0;
    s.K = N;
    return s;
}
// dW[n, k] = sum_m dY[m, n] X[m, k]
rfn_gemm_seg seg_dw(const float* dY, long lddy, const float* X, long ldx, int rows) {
    rfn_gemm_seg s;
    memset(&s, 0, sizeof(s));
    s.A = dY; s.lda = lddy; s.a_kfast = 0;
    s.B = X; s.ldb = ldx; s.b_kfast = 0;
    s.K = rows;
    return s;
}
// stream + split-K scratch handed to every GEMM of a phase
struct GemmCtx {
    void* st;
    float* ws;
    size_t ws_bytes;
    unsigned flags;   // RFN_GEMM_OPT_* bits of the phase (rfn_dims.gemm_flags)
    int32_t* tickets = nullptr;   // zeroed split-K tile counters of the phase's workspace (rfn_gemm_f32_tk), or NULL
    int n_tickets = 0;
};
const int GEMM_TICKETS = 16384;   // 64 KB of counters per workspace: every split launch of the path has fewer output tiles
rfn_gemm_problem prob1(float* C, long ldc, const rfn_gemm_seg& s) {
    rfn_gemm_problem p;
    memset(&p, 0, sizeof(p));
    p.C = C; p.ldc = ldc; p.nseg = 1; p.seg[0] = s;
    return p;
}
// weight gradient + its bias gradient (column sums of dY) in one problem
rfn_gemm_problem prob_dw(float* dW, long ldw, float* db, const float* dY, long lddy, const float* X, long ldx,
                         int rows) {
    rfn_gemm_problem p = prob1(dW, ldw, seg_dw(dY, lddy, X, ldx, rows));
    p.a_colsum = db;
    return p;
}
int gemm1(int M, int N, const rfn_gemm_seg& s, float* C, long ldc, int acc, const GemmCtx& gx) {
    rfn_gemm_problem p = prob1(C, ldc, s);
    return rfn_gemm_f32_tk(M, N, 1, &p, acc, gx.ws, gx.ws_bytes, gx.flags, gx.tickets, gx.n_tickets, gx.st);
}
int gemm_dw(int N, int K, float* dW, long ldw, float* db, const float* dY, long lddy, const float* X, long ldx,
            int rows, const GemmCtx& gx) {
    rfn_gemm_problem p = prob_dw(dW, ldw, db, dY, lddy, X, ldx, rows);
    return rfn_gemm_f32_tk(N, K, 1, &p, 0, gx.ws, gx.ws_bytes, gx.flags, gx.tickets, gx.n_tickets, gx.st);
}
// The vocabulary (V+1 = 9488 at the headline size) is not a multiple of the 128-wide tile: the three logit-layer GEMMs
// are issued as an aligned main part that takes the unchecked fast path plus a thin remainder (< 128 columns / rows /
// reduction elements) on the bounds-checked kernel.  Same sums, same order per output element except the dX split,
// which adds the remainder's partial product last.
static inline int aligned_part(int n) { return (n / 128) * 128; }

// rfn_dims.probe_events (rfn.h): the caller's timing events around the dominant launches, recorded on the launch stream
static inline int probe_mark(const rfn_dims* d, int idx, void* st) {
    if (!d->probe_events || !d->probe_events[idx]) return RFN_OK;
    return hipEventRecord((hipEvent_t)d->probe_events[idx], (hipStream_t)st) == hipSuccess ? RFN_OK : RFN_ERR_LAUNCH;
}
int gemm_logits(int rows, int V1, const float* h, int R, const float* Wl, const float* bl, float* C, const GemmCtx& gx) {
    // both operands are [row][k]: the LDS-DMA kernel takes the ragged vocabulary in one launch (edge tiles clamp their
    // source rows); only a hidden size that is not a whole K step keeps the main + remainder split
    const int Va = aligned_part(V1);
    if (Va == V1 || Va == 0 || R % 32 == 0) return gemm1(rows, V1, seg_lin(h, R, Wl, R, R, bl), C, V1, 0, gx);
    RFN_TRY(gemm1(rows, Va, seg_lin(h, R, Wl, R, R, bl), C, V1, 0, gx));
    return gemm1(rows, V1 - Va, seg_lin(h, R, Wl + (long)Va * R, R, R, bl + Va), C + Va, V1, 0, gx);
}
int gemm_logits_dw(int V1, int R, float* dW, float* db, const float* dlg, const float* h, int rows, const GemmCtx& gx) {
    // The bias gradient (column sums of the 165 MB dlogits) comes from its own streaming pass (~35 us) instead of riding
    // on the GEMM: without the rider the 42-GFLOP weight gradient takes the LDS-DMA kernel (0.55 -> 0.33 ms at C3).
    const bool big = (double)rows * V1 * R >= 2e9 && db;
    if (big) {
        RFN_TRY(rfn_colsum_f32(dlg, V1, rows, V1, db, 0, gx.st));
        db = nullptr;
    }
    const int Va = aligned_part(V1);
    if (Va == V1 || Va == 0) return gemm_dw(V1, R, dW, R, db, dlg, V1, h, R, rows, gx);
    RFN_TRY(gemm_dw(Va, R, dW, R, db, dlg, V1, h, R, rows, gx));
    return gemm_dw(V1 - Va, R, dW + (long)Va * R, R, db ? db + Va : nullptr, dlg + Va, V1, h, R, rows, gx);
}
int gemm_logits_dx(int rows, int R, int V1, const float* dlg, const float* Wl, float* dh, const GemmCtx& gx) {
    const int Va = aligned_part(V1);
    if (Va == V1 || Va == 0) return gemm1(rows, R, seg_dx(dlg, V1, Wl, R, V1), dh, R, 0, gx);
    RFN_TRY(gemm1(rows, R, seg_dx(dlg, V1, Wl, R, Va), dh, R, 0, gx));
    return gemm1(rows, R, seg_dx(dlg + Va, V1, Wl + (long)Va * R, R, V1 - Va), dh, R, 1, gx);
}
// any number of K segments into one C (chunks of RFN_GEMM_MAXSEG, later chunks accumulate)
int gemm_segs(int M, int N, int nseg, const rfn_gemm_seg* segs, float* C, long ldc, int acc, const GemmCtx& gx) {
    for (int s0 = 0; s0 < nseg; s0 += RFN_GEMM_MAXSEG) {
        rfn_gemm_problem p;
        memset(&p, 0, sizeof(p));
        p.C = C; p.ldc = ldc;
        p.nseg = (nseg - s0 < RFN_GEMM_MAXSEG) ? nseg - s0 : RFN_GEMM_MAXSEG;
        for (int s = 0; s < p.nseg; ++s) p.seg[s] = segs[s0 + s];
        RFN_TRY(rfn_gemm_f32_tk(M, N, 1, &p, (s0 > 0) ? 1 : acc, gx.ws, gx.ws_bytes, gx.flags, gx.tickets, gx.n_tickets, gx.st));
    }
    return RFN_OK;
}
// any number of same-shape problems (chunks of RFN_GEMM_MAXGROUP)
int gemm_groups(int M, int N, int n, const rfn_gemm_problem* p, int acc, const GemmCtx& gx) {
    for (int g0 = 0; g0 < n; g0 += RFN_GEMM_MAXGROUP) {
        const int ng = (n - g0 < RFN_GEMM_MAXGROUP) ? n - g0 : RFN_GEMM_MAXGROUP;
        RFN_TRY(rfn_gemm_f32_tk(M, N, ng, p + g0, acc, gx.ws, gx.ws_bytes, gx.flags, gx.tickets, gx.n_tickets, gx.st));
    }
    return RFN_OK;
}
// The same for big problems whose column count is not a multiple of the 128-wide tile (a 2208-wide DenseNet feature
// map, feat_array.py:147-150): the aligned main part takes the interior fast path (LDS-DMA kernel), a thin remainder
// (< 128 columns) the bounds-checked one -- as the logit layer does for the vocabulary.  Same sums, same k order per
// output element.  A bias-gradient rider (row sums of the A operand) rides on the main part only.
int gemm_groups_split_cols(int M, int N, int n, const rfn_gemm_problem* p, int acc, const GemmCtx& gx) {
    const int Na = aligned_part(N);
    if (Na == N || Na == 0 || M % 128 != 0) return gemm_groups(M, N, n, p, acc, gx);
    RFN_TRY(gemm_groups(M, Na, n, p, acc, gx));
    rfn_gemm_problem rest[64];
    if (n > 64) return RFN_ERR_SHAPE;
    for (int g = 0; g < n; ++g) {
        rest[g] = p[g];
        rest[g].C = p[g].C + Na;
        rest[g].a_colsum = nullptr;
        for (int s = 0; s < p[g].nseg; ++s) {
            rfn_gemm_seg& sg = rest[g].seg[s];
            sg.B = sg.b_kfast ? sg.B + (long)Na * sg.ldb : sg.B + Na;
            if (sg.bias) sg.bias += Na;
        }
    }
    return gemm_groups(M, N - Na, n, rest, acc, gx);
}

// ---------------------------------------------------------------------------------------------
// fused per-step products of the recurrences (csrc/rfn_cellgemm.hip): several outputs per launch, K cut across the
// waves of a block, LSTM forward / backward epilogues.  cell_lin(): Y = X W^T + b;  cell_dx(): dX = dY W.
// ---------------------------------------------------------------------------------------------
rfn_cell_out cell_out(float* C, long ldc, int N, int accumulate) {
    rfn_cell_out o;
    memset(&o, 0, sizeof(o));
    o.C = C; o.ldc = ldc; o.N = N; o.accumulate = accumulate; o.epilogue = RFN_CELL_EPI_STORE;
    return o;
}
void cell_lin(rfn_cell_out& o, const float* X, long ldx, const float* W, long ldw, int K, const float* bias) {
    if (o.nseg < RFN_CELL_MAXSEG) o.seg[o.nseg] = seg_lin(X, ldx, W, ldw, K, bias);
    ++o.nseg;     // an overflow makes rfn_cell_gemm_supported() say no
}
void cell_dx(rfn_cell_out& o, const float* dY, long lddy, const float* W, long ldw, int N) {
    if (o.nseg < RFN_CELL_MAXSEG) o.seg[o.nseg] = seg_dx(dY, lddy, W, ldw, N);
    ++o.nseg;
}
void cell_lstm(rfn_cell_out& o, const float* c_prev, long ldcp, float* c_next, long ldcn, float* h_next, long ldh,
               uint64_t drop_offset) {
    o.epilogue = RFN_CELL_EPI_LSTM;
    o.c_prev = c_prev; o.ldcp = ldcp; o.c_next = c_next; o.ldcn = ldcn; o.h_next = h_next; o.ldh = ldh;
    o.drop_offset = drop_offset;
}
void cell_lstm_bwd(rfn_cell_out& o, float* gates, long ldg, const float* c_prev, long ldcp, const float* c_next, long ldcn,
                   const float* dh_ext, long lddh, const float* dc_next, long lddcn, float* dc_prev, long lddcp,
                   uint64_t drop_offset) {
    o.epilogue = RFN_CELL_EPI_LSTM_BWD;
    o.gates = gates; o.ldg = ldg; o.c_prev = c_prev; o.ldcp = ldcp; o.c_next = const_cast<float*>(c_next); o.ldcn = ldcn;
    o.dh_ext = dh_ext; o.lddh = lddh; o.dc_next = dc_next; o.lddcn = lddcn; o.dc_prev = dc_prev; o.lddcp = lddcp;
    o.drop_offset = drop_offset;
}
bool cell_ok(int B, int n, const rfn_cell_out* outs, int R) {
    for (int i = 0; i < n; ++i)
        if (outs[i].nseg > RFN_CELL_MAXSEG) return false;
    return n <= RFN_CELL_MAXOUT && rfn_cell_gemm_supported(B, n, outs, R) != 0;
}
// variant: 0, or RFN_CELL_VARIANT_DEEP from cell_variant(d) (A/B hook: few-tile launches on the deep-ring kernel)
int cell_run(int B, int n, const rfn_cell_out* outs, int R, float drop_p, uint64_t seed, void* st, int variant) {
    return rfn_cell_gemm(B, n, outs, R, drop_p, seed, variant, st);
}
inline int cell_variant(const rfn_dims* d) {
    const int tiles = (d->path_flags & RFN_PATH_OPT_NO_SMALL_TILES) ? 9 : (d->path_flags & RFN_PATH_OPT_SHARED_SMALL_TILES) ? 7 : 0;
    return ((d->path_flags & RFN_PATH_OPT_DEEP_CELLS) ? RFN_CELL_VARIANT_DEEP : 0) | tiles;
}
// the same launch prepared instead of launched: one phase of a recurrence-chain step (rfn_chain.hip)
int cell_prepare(int B, int n, const rfn_cell_out* outs, int R, float drop_p, uint64_t seed, CgPrepared* pz, int variant) {
    return rfn_cg_prepare(B, n, outs, R, drop_p, seed, variant, pz);
}
inline int chain_persist(const rfn_dims* d, uint32_t which) { return (d->path_flags & which) ? 1 : 0; }

// Up to MEM_BATCH copies / zero fills of f32 buffers in ONE launch (src == NULL: zero).  The path's bookkeeping moves (initial
// states into the workspace, gradient slabs zeroed before they are accumulated into) come in twos and threes; each was a
// hipMemcpyAsync / hipMemsetAsync of its own, i.e. a ~5 us launch in the dependent chain.
const int MEM_BATCH = 4;
struct MemOp { float* dst; const float* src; long n; };
struct MemOps { MemOp op[MEM_BATCH]; };
__global__ __launch_bounds__(256) void mem_batch_k(const MemOps o) {
    const MemOp m = o.op[blockIdx.y];
    const long stride = (long)gridDim.x * 256, i0 = (long)blockIdx.x * 256 + threadIdx.x;
    const bool v4 = (((uintptr_t)m.dst | (uintptr_t)m.src) & 15) == 0;
    const long n4 = v4 ? m.n >> 2 : 0;
    typedef float f4 __attribute__((ext_vector_type(4)));
    if (m.src) {
        for (long i = i0; i < n4; i += stride) reinterpret_cast<f4*>(m.dst)[i] = reinterpret_cast<const f4*>(m.src)[i];
        for (long i = 4 * n4 + i0; i < m.n; i += stride) m.dst[i] = m.src[i];
    } else {
        for (long i = i0; i < n4; i += stride) reinterpret_cast<f4*>(m.dst)[i] = f4{0.f, 0.f, 0.f, 0.f};
        for (long i = 4 * n4 + i0; i < m.n; i += stride) m.dst[i] = 0.f;
    }
}
int mem_batch(std::initializer_list<MemOp> ops, void* st) {
    MemOps o;
    int n = 0;
    long big = 0;
    for (const MemOp& m : ops) {
        if (!m.dst || m.n <= 0) continue;
        if (n == MEM_BATCH) return RFN_ERR_SHAPE;
        o.op[n++] = m;
        big = m.n > big ? m.n : big;
    }
    if (!n) return RFN_OK;
    for (int i = n; i < MEM_BATCH; ++i) o.op[i] = MemOp{nullptr, nullptr, 0};
    long blocks = (big / 4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(mem_batch_k, dim3((unsigned)blocks, n), dim3(256), 0, (hipStream_t)st, o);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}
int copy_f32(float* dst, const float* src, size_t n, void* st) {
    if (hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)st) != hipSuccess)
        return RFN_ERR_LAUNCH;
    return RFN_OK;
}
int zero_f32(float* dst, size_t n, void* st) {
    if (hipMemsetAsync(dst, 0, n * sizeof(float), (hipStream_t)st) != hipSuccess) return RFN_ERR_LAUNCH;
    return RFN_OK;
}

// ---------------------------------------------------------------------------------------------
// workspace layouts (offsets in floats, 256-B aligned)
// ---------------------------------------------------------------------------------------------
// LSTM gate width: [in | forget | out | g] or, with maxout, [in | forget | out | g1 | g2] (max of the last two,
// misc/LSTMSoftMultiAttentionFeatArrayNoInputCore.py:25-27,60-62; misc/LSTMSoftAttentionCore.py:25-28,89-91)
inline int gate_width(int maxout, int R) { return (maxout ? 5 : 4) * R; }

struct Bump {
    size_t off = 0;
    size_t take(size_t n) {
        const size_t o = off;
        off += (n + 63) & ~(size_t)63;
        return o;
    }
};

const size_t GEMM_WS_FLOATS = (size_t)64 << 20;  // 256 MiB of split-K partial tiles (a 4096 x 2208 weight gradient cut 8 ways)
const int FUSED_ATTN_BWD_MIN_B = 96;   // below this the (L/64, B) grid of the split dalpha kernel fills the chip better
const long FUSED_ATTN_BWD_SMALL_MAP = 32768;   // ... unless the map (L x D) is so small that launches, not bytes, are the cost
const size_t STEP_GEMM_WS_FLOATS = GEMM_WS_FLOATS;   // the free-running step makes the same split-K choices as the teacher-forced pass

struct PrefixLayout {
    size_t P1[RFN_MAX_ENC], al1[RFN_MAX_ENC], z1[RFN_MAX_ENC], P2[RFN_MAX_ENC], dz1[RFN_MAX_ENC];
    size_t Hs, Cs, hp1, g1, rmat, rarg, h2, c2, hp2, al2, z2, g2;
    size_t dHs, dHpart, dC, dal, dwp, dhp1, dh2e, dhrec, dc2, dz2, dhp2;
    size_t gws;
    size_t bar;     // grid-barrier counters of the persistent recurrence kernels (RFN_CHAIN_BAR_WORDS uint32, rfn_chain.hip)
    size_t tk;      // split-K tile counters (GEMM_TICKETS int32), zeroed at the head of every entry point that splits
    size_t x3;      // plane images + split-K partials of the bf16-plane GEMMs (RFN_GEMM_OPT_BF16X3), 0 floats otherwise
    size_t x3p[RFN_MAX_ENC];   // train: encoder i's dP1 as a k-slow plane image (all T1 steps), kept from the backward
                               // recurrence to its weight-gradient GEMM
    size_t total;
};
// Does encoder i's hoisted projection (and its weight gradient) take the bf16-plane GEMM?  Only when asked for
// (RFN_GEMM_OPT_BF16X3), when the per-step output groups are whole 256-wide tiles and when the product is long enough to
// pay for the two split passes; everything else stays on the exact-f32 kernels.
static bool x3_takes(const rfn_dims* d, int B, int i) {
    if (!(d->gemm_flags & RFN_GEMM_OPT_BF16X3)) return false;
    if (d->A % 256 || d->D[i] % 4 || d->T1 > 64) return false;
    if (d->gemm_flags & RFN_GEMM_OPT_BF16X3_ANY_SIZE) return true;
    return 2.0 * B * d->L[i] * d->D[i] * d->A * d->T1 >= 2e10;
}
static int x3_row_pad(int rows) { return (rows + 255) / 256 * 256; }   // row pitch of a k-slow image
// Stage-I attention backward, which form runs (decided here, once, because the weight-gradient pass has to know whether the
// attention launches already wrote dP1 as bf16 planes).  The fused kernel (d alpha kept in LDS, one block per batch row and
// encoder) needs enough blocks to stream at the chip's rate: the GROUPED launch of all M encoders counts M * B of them, a
// per-encoder launch B.  Below that the split d-alpha kernel's (L/64, B) grid fills the chip better -- unless the map is so
// small that launches, not bytes, are the cost.
// ONE definition, used by the backward sweep (which launches it) and by the weight-gradient pass (which has to know whether
// encoder i's dP1 already sits in its plane image) -- ADVICE r04: the two used to restate each other's conditions.
//   GROUPED(_KS)  all M encoders of a step in one launch (they share (L, D)); _KS: dP1 written as bf16 planes
//   HET           all M encoders in one launch, maps of different (L, D), exact f32 only
//   FUSED(_KS)    one launch per encoder, d alpha kept in LDS
//   SPLIT         d alpha kernel + score-backward kernel per encoder
// A launch needs enough blocks to stream at the chip's rate: it has (encoders in the launch) x B of them.
enum AttnBwdForm { AB_GROUPED_KS, AB_GROUPED, AB_HET, AB_FUSED_KS, AB_FUSED, AB_SPLIT };
static bool attn_bwd_small_map(const rfn_dims* d, int B, int i) {
    return !x3_takes(d, B, i) && (long)d->L[i] * d->D[i] <= FUSED_ATTN_BWD_SMALL_MAP;
}
static AttnBwdForm attn_bwd_form(const rfn_dims* d, int B, int i, bool dz_in_one_launch) {
    const int M = d->M;
    bool same_ld = M > 1;
    for (int j = 1; j < M; ++j) same_ld = same_ld && d->D[j] == d->D[0] && d->L[j] == d->L[0];
    if (same_ld && ((long)B * M >= FUSED_ATTN_BWD_MIN_B || attn_bwd_small_map(d, B, 0)))
        return x3_takes(d, B, 0) ? AB_GROUPED_KS : AB_GROUPED;
    if (!same_ld && M > 1 && dz_in_one_launch) {   // every encoder must qualify for the exact-f32 fused form in the shared launch
        bool het = true;
        for (int j = 0; j < M && het; ++j)
            het = !x3_takes(d, B, j) && ((long)B * M >= FUSED_ATTN_BWD_MIN_B || attn_bwd_small_map(d, B, j));
        if (het) return AB_HET;
    }
    if (B >= FUSED_ATTN_BWD_MIN_B && x3_takes(d, B, i)) return AB_FUSED_KS;
    if (B >= FUSED_ATTN_BWD_MIN_B || attn_bwd_small_map(d, B, i)) return AB_FUSED;
    return AB_SPLIT;
}
static bool x3_dp_emitted(const rfn_dims* d, int B, int i) {   // dP1 of encoder i reaches its k-slow plane image from the attention launches
    const AttnBwdForm f = attn_bwd_form(d, B, i, false);       // (the HET form never takes plane products)
    return f == AB_GROUPED_KS || f == AB_FUSED_KS;
}
static size_t x3_scratch_floats(const rfn_dims* d, int B, int train) {
    size_t most = 0;
    for (int i = 0; i < d->M; ++i) {
        if (!x3_takes(d, B, i)) continue;
        const int BL = B * d->L[i], TA = d->T1 * d->A, Di = d->D[i];
        size_t fwd = rfn_x3_image_bytes(BL, Di) + rfn_x3_image_bytes(TA, Di);
        size_t bwd = 0;
        if (train) bwd = rfn_x3_image_bytes(Di, BL) + 4 * rfn_x3_part_floats(TA, Di, rfn_x3_splitk_for(TA, Di, BL));
        const size_t need = (fwd > bwd ? fwd : bwd) / 4 + 256;
        if (need > most) most = need;
    }
    return most;
}
PrefixLayout prefix_layout(const rfn_dims* d, int B, int train) {
    const size_t G2 = (size_t)gate_width(d->review_maxout, d->R);
    PrefixLayout L;
    memset(&L, 0, sizeof(L));
    Bump b;
    const size_t M = d->M, R = d->R, A = d->A, T1 = d->T1, T2 = d->T2, K = d->K, Bz = B;
    size_t maxL = T1;
    for (int i = 0; i < d->M; ++i) {
        L.P1[i] = b.take(Bz * d->L[i] * T1 * A);
        L.al1[i] = b.take(T1 * Bz * d->L[i]);
        L.z1[i] = b.take(T1 * Bz * d->D[i]);
        L.P2[i] = b.take(T1 * Bz * T2 * A);
        if ((size_t)d->L[i] > maxL) maxL = d->L[i];
    }
    L.Hs = b.take((T1 + 1) * Bz * M * R);
    L.Cs = b.take((T1 + 1) * Bz * M * R);
    L.hp1 = b.take(T1 * M * Bz * A);
    L.g1 = b.take(T1 * M * Bz * 4 * R);
    L.rmat = b.take((T1 * M > T2 ? T1 * M : T2) * Bz * K);   // M slabs (T1,B,K) of the stage-I heads / one (T2,B,K)
    L.rarg = b.take((M + 1) * Bz * K);  // int32, same width
    L.h2 = b.take((T2 + 1) * Bz * R);
    L.c2 = b.take((T2 + 1) * Bz * R);
    L.hp2 = b.take(T2 * M * Bz * A);
    L.al2 = b.take(T2 * M * Bz * T1);
    L.z2 = b.take(T2 * M * Bz * R);
    L.g2 = b.take(T2 * Bz * G2);
    L.gws = b.take(GEMM_WS_FLOATS);
    L.tk = b.take(GEMM_TICKETS);
    L.bar = b.take(RFN_CHAIN_BAR_WORDS);
    L.x3 = b.take(x3_scratch_floats(d, B, train));
    if (train)
        for (int i = 0; i < d->M; ++i)
            if (x3_takes(d, B, i)) L.x3p[i] = b.take(rfn_x3_image_bytes(d->T1 * d->A, B * d->L[i]) / 4 + 64);
    if (train) {
        for (int i = 0; i < d->M; ++i) L.dz1[i] = b.take(Bz * d->D[i]);
        L.dHs = b.take((T1 + 1) * Bz * M * R);
        L.dHpart = b.take(M * Bz * M * R);       // small batches: the M partial products d gates_j . W_H_j of a stage-I step
        L.dC = b.take(Bz * M * R);
        L.dal = b.take(M * Bz * maxL);
        L.dwp = b.take((T1 > T2 ? T1 : T2) * M * Bz * A);
        L.dhp1 = b.take(T1 * M * Bz * A);
        L.dh2e = b.take(T2 * Bz * R);
        L.dhrec = b.take(Bz * R);
        L.dc2 = b.take(Bz * R);
        L.dz2 = b.take(M * Bz * R);
        L.dhp2 = b.take(T2 * M * Bz * A);
    }
    L.total = b.off;
    return L;
}

const int DEC_KSPLIT = 4;   // the decoder's d gates . W_hh (K = 4R) is computed as this many K-split partial products (rfn_decoder_bwd)
struct DecoderLayout {
    size_t Pd, Ud, xs, gd, hd, cd, hpd, ald, zd, logits;
    size_t dhe, dhrec, dc, dz, dal, dwp, dhpd, dPd, dUd, dxs;
    size_t gws, tk, bar;
    size_t total;
};
DecoderLayout decoder_layout(const rfn_dims* d, int B, int S, int train) {
    const size_t GD = (size_t)gate_width(d->decoder_maxout, d->R);
    DecoderLayout L;
    memset(&L, 0, sizeof(L));
    Bump b;
    const size_t R = d->R, A = d->A, E = d->E, T2 = d->T2, V1 = d->V1, Bz = B, Sz = S;
    L.Pd = b.take(T2 * Bz * A);
    L.Ud = b.take(T2 * Bz * GD);      // U = thought vectors . W_z^T (z2h hoisted through the attention, rfn_deccell.hip)
    L.xs = b.take(Sz * Bz * E);
    L.gd = b.take(Sz * Bz * GD);
    L.hd = b.take((Sz + 1) * Bz * R);
    L.cd = b.take((Sz + 1) * Bz * R);
    L.hpd = b.take(Sz * Bz * A);
    L.ald = b.take(Sz * Bz * T2);
    L.zd = b.take(Sz * Bz * R);
    L.logits = b.take(Sz * Bz * V1);  // logits in forward, dlogits in backward
    L.gws = b.take(GEMM_WS_FLOATS);
    L.tk = b.take(GEMM_TICKETS);
    L.bar = b.take(RFN_CHAIN_BAR_WORDS);
    if (train) {
        L.dhe = b.take(Sz * Bz * R);
        L.dhrec = b.take(DEC_KSPLIT * Bz * R);   // the recurrent d h; hoisted form: + the K-split partial slabs of d gates . W_hh
        L.dc = b.take(Bz * R);
        L.dz = b.take(Bz * R);
        L.dal = b.take(Bz * T2);
        L.dwp = b.take(Sz * Bz * A);
        L.dhpd = b.take(Sz * Bz * A);
        L.dPd = b.take(T2 * Bz * A);
        L.dUd = b.take(T2 * Bz * GD);
        L.dxs = b.take(Sz * Bz * E);
    }
    L.total = b.off;
    return L;
}

inline int stage1_cell_cus() {   // CU count of the current device (cached); 256 when it cannot be read
    static int cus[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int& c = cus[dev & 15];
    if (c == 0 && hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) c = 256;
    return c > 0 ? c : 256;
}

const uint64_t OFF_STAGE2 = RFN_DROP_OFFSET_STAGE2, OFF_DECODER = RFN_DROP_OFFSET_DECODER;   // rfn.h: rfn_dropout_mask

// The decoder cell's form: z2h hoisted through the attention (rfn_deccell.hip: two dependent launches per step each way)
// unless the caller asks for the three-launch form of rounds 3-5 (A/B hook; the persistent decoder chains are built from it).
inline bool dec_hoisted(const rfn_dims* d) {
    return !(d->path_flags & (RFN_PATH_OPT_DEC_UNHOISTED | RFN_PATH_OPT_PERSIST_DEC_FWD | RFN_PATH_OPT_PERSIST_DEC_BWD));
}
// rfn_decoder_prepare's buffer: [Pd = att_2_att_h(comb) (T2 * Bc, A) | U = comb . W_z^T (T2 * Bc, GD)], U 256-B aligned
inline size_t cproj_u_off(const rfn_dims* d, int Bc) { return ((size_t)d->T2 * Bc * d->A + 63) & ~(size_t)63; }

}  // namespace

// =============================================================================================
// parameter table
// =============================================================================================
extern "C" int rfn_abi_version(void) { return RFN_ABI_VERSION; }

extern "C" const char* rfn_error_string(int code) {
    switch (code) {
        case RFN_OK: return "ok";
        case RFN_ERR_SHAPE: return "unsupported or inconsistent dimensions";
        case RFN_ERR_UNSUPPORTED: return "configuration not implemented by the HIP path (maxout)";
        case RFN_ERR_LAUNCH: return "HIP launch / runtime failure";
        case RFN_ERR_WORKSPACE: return "workspace too small";
        case RFN_ERR_ARG: return "null or misaligned pointer";
        default: return "unknown error";
    }
}

extern "C" int rfn_param_count(const rfn_dims* d) {
    const int rc = check_dims(d);
    if (rc != RFN_OK) return rc;
    return PIdx(d).count();
}

static const char* const kAtt[6] = {"att_2_att_h.weight", "att_2_att_h.bias", "h_2_att_h.weight",
                                    "h_2_att_h.bias",     "att_h_2_out.weight", "att_h_2_out.bias"};

extern "C" int rfn_param_name(const rfn_dims* d, int idx, char* buf, size_t n) {
    const int rc = check_dims(d);
    if (rc != RFN_OK) return rc;
    if (!buf || n == 0) return RFN_ERR_ARG;
    const PIdx P(d);
    const int M = d->M;
    if (idx < 0 || idx >= P.count()) return RFN_ERR_SHAPE;
    if (idx < 2 * M) {
        snprintf(buf, n, "fc2h.%d.%s", idx / 2, (idx & 1) ? "bias" : "weight");
    } else if (idx == P.embed()) {
        snprintf(buf, n, "embed.weight");
    } else if (idx == P.logit_w()) {
        snprintf(buf, n, "logit.weight");
    } else if (idx == P.logit_b()) {
        snprintf(buf, n, "logit.bias");
    } else if (idx < P.rind_w(0)) {
        const int r = idx - P.s1(0, 0, 0), cell = r / 10, k = r % 10, t = cell / M, i = cell % M;
        if (k < 6)
            snprintf(buf, n, "review_steps_individual.%d.lstm.%d.att_model.%s", t, i, kAtt[k]);
        else
            snprintf(buf, n, "review_steps_individual.%d.lstm.%d.%s.%s", t, i, (k < 8) ? "H2h" : "z2h",
                     (k & 1) ? "bias" : "weight");
    } else if (idx < P.s2base(0)) {
        const int r = idx - P.rind_w(0);
        snprintf(buf, n, "reason_linear_individual.%d.%s", r / 2, (r & 1) ? "bias" : "weight");
    } else if (idx < P.r_w()) {
        const int per = 2 + 8 * M, r = idx - P.s2base(0), t = r / per, k = r % per;
        if (k < 2) {
            snprintf(buf, n, "review_steps.%d.h2h.%s", t, k ? "bias" : "weight");
        } else {
            const int i = (k - 2) / 8, kk = (k - 2) % 8;
            if (kk < 2)
                snprintf(buf, n, "review_steps.%d.z_2_h.%d.%s", t, i, kk ? "bias" : "weight");
            else
                snprintf(buf, n, "review_steps.%d.att_model.%d.%s", t, i, kAtt[kk - 2]);
        }
    } else if (idx == P.r_w()) {
        snprintf(buf, n, "reason_linear.weight");
    } else if (idx == P.r_b()) {
        snprintf(buf, n, "reason_linear.bias");
    } else {
        const int k = idx - P.dec(0);
        static const char* const names[3] = {"i2h", "h2h", "z2h"};
        if (k < 6)
            snprintf(buf, n, "decoder.%s.%s", names[k / 2], (k & 1) ? "bias" : "weight");
        else
            snprintf(buf, n, "decoder.%s", kAtt[k - 6]);
    }
    return RFN_OK;
}

extern "C" int rfn_param_shape(const rfn_dims* d, int idx, int64_t* rows, int64_t* cols) {
    const int rc = check_dims(d);
    if (rc != RFN_OK) return rc;
    if (!rows || !cols) return RFN_ERR_ARG;
    const PIdx P(d);
    const int M = d->M, R = d->R, A = d->A;
    const int G2 = gate_width(d->review_maxout, R), GD = gate_width(d->decoder_maxout, R);
    if (idx < 0 || idx >= P.count()) return RFN_ERR_SHAPE;
    auto att_shape = [&](int k, int feat, int64_t* r, int64_t* c) {
        const int64_t rr[6] = {A, A, A, A, 1, 1};
        const int64_t cc[6] = {feat, 1, R, 1, A, 1};
        *r = rr[k];
        *c = cc[k];
    };
    if (idx < 2 * M) {
        *rows = R;
        *cols = (idx & 1) ? 1 : d->F[idx / 2];
    } else if (idx == P.embed()) {
        *rows = d->V1; *cols = d->E;
    } else if (idx == P.logit_w()) {
        *rows = d->V1; *cols = R;
    } else if (idx == P.logit_b()) {
        *rows = d->V1; *cols = 1;
    } else if (idx < P.rind_w(0)) {
        const int r = idx - P.s1(0, 0, 0), cell = r / 10, k = r % 10, i = cell % M;
        if (k < 6) att_shape(k, d->D[i], rows, cols);
        else { *rows = 4 * R; *cols = (k & 1) ? 1 : (k < 8 ? M * R : d->D[i]); }
    } else if (idx < P.s2base(0)) {
        *rows = d->K; *cols = ((idx - P.rind_w(0)) & 1) ? 1 : R;
    } else if (idx < P.r_w()) {
        const int per = 2 + 8 * M, k = (idx - P.s2base(0)) % per;
        if (k < 2) { *rows = G2; *cols = k ? 1 : R; }
        else {
            const int kk = (k - 2) % 8;
            if (kk < 2) { *rows = G2; *cols = kk ? 1 : R; }
            else att_shape(kk - 2, R, rows, cols);
        }
    } else if (idx == P.r_w()) {
        *rows = d->K; *cols = R;
    } else if (idx == P.r_b()) {
        *rows = d->K; *cols = 1;
    } else {
        const int k = idx - P.dec(0);
        if (k < 6) { *rows = GD; *cols = (k & 1) ? 1 : (k < 2 ? d->E : R); }
        else att_shape(k - 6, R, rows, cols);
    }
    return RFN_OK;
}

// =============================================================================================
// phase 1: init state + fusion stages I and II
// =============================================================================================
extern "C" size_t rfn_prefix_ws_bytes(const rfn_dims* d, int B, int train) {
    if (check_dims(d) != RFN_OK || B < 1) return 0;
    return prefix_layout(d, B, train).total * sizeof(float);
}

static int prefix_fwd_impl(const rfn_dims* d, int B, const float* const* prm, const float* const* fc,
                           const float* const* init_h, const float* const* init_c, const float* const* att,
                           float* comb, float* h_out, float* c_out, float* reason_pred, void* ws, size_t ws_bytes,
                           int train, uint64_t seed, void* st) {
    RFN_TRY(check_dims(d));
    if (B < 1) return RFN_ERR_SHAPE;
    if (!prm || (!fc && !(init_h && init_c)) || !att || !ws) return RFN_ERR_ARG;
    const PrefixLayout Lo = prefix_layout(d, B, train);
    if (ws_bytes < Lo.total * sizeof(float)) return RFN_ERR_WORKSPACE;
    const PIdx P(d);
    const int M = d->M, R = d->R, A = d->A, T1 = d->T1, T2 = d->T2, K = d->K;
    const int G2 = gate_width(d->review_maxout, R);
    const long MR = (long)M * R, BMR = (long)B * MR, BR = (long)B * R;
    float* W = (float*)ws;
    const bool tk_on = (d->gemm_flags & RFN_GEMM_OPT_SPLITK_IN_KERNEL) != 0;   // measured slower: off unless asked for (rfn.h)
    const GemmCtx gx{st, W + Lo.gws, GEMM_WS_FLOATS * sizeof(float), d->gemm_flags, tk_on ? (int32_t*)(W + Lo.tk) : nullptr,
                     tk_on ? GEMM_TICKETS : 0};
    if (tk_on) RFN_TRY(zero_f32(W + Lo.tk, GEMM_TICKETS, st));   // tile counters: zero on entry, every launch leaves them zero
    float* Hs = W + Lo.Hs;
    float* Cs = W + Lo.Cs;
    int32_t* rarg = (int32_t*)(W + Lo.rarg);

    // K0: h0_i = fc2h_i(fc_i) written straight into the concatenated H of step 0; c0 = h0 (:202-208)
    if (init_h) {   // caller-provided stage-I state (get_thought_vectors(fc, att, state_list), :283)
        for (int i = 0; i < M; ++i) {
            if (!init_h[i] || !init_c[i]) return RFN_ERR_ARG;
            RFN_TRY(rfn_axpby_2d(1.f, init_h[i], R, 0.f, Hs + i * R, MR, B, R, st));
            RFN_TRY(rfn_axpby_2d(1.f, init_c[i], R, 0.f, Cs + i * R, MR, B, R, st));
        }
    } else {
        rfn_cell_out k0[RFN_MAX_ENC];
        for (int i = 0; i < M; ++i) {
            if (!fc[i]) return RFN_ERR_ARG;
            k0[i] = cell_out(Hs + i * R, MR, R, 0);
            cell_lin(k0[i], fc[i], d->F[i], prm[P.fc_w(i)], d->F[i], d->F[i], prm[P.fc_b(i)]);
        }
        if (cell_ok(B, M, k0, R)) {   // the M fc2h products in one launch
            RFN_TRY(cell_run(B, M, k0, R, 0.f, 0, st, cell_variant(d)));
        } else {
            for (int i = 0; i < M; ++i)
                RFN_TRY(gemm1(B, R, seg_lin(fc[i], d->F[i], prm[P.fc_w(i)], d->F[i], d->F[i], prm[P.fc_b(i)]), Hs + i * R,
                              MR, 0, gx));
        }
        RFN_TRY(copy_f32(Cs, Hs, BMR, st));
    }

    // hoisted feature projections of stage I, all T1 step weights grouped: step-major slabs P1_i[t][(b,l)][a], so that
    // every consumer streams contiguous memory (the attention kernels of step t, the weight-gradient GEMM's k-rows)
    rfn_gemm_problem pr[64];
    for (int i = 0; i < M; ++i) {
        if (T1 > 64) return RFN_ERR_SHAPE;
        if (x3_takes(d, B, i)) {
            // bf16-plane GEMM: the features and the T1 stacked step weights as plane images, one launch for all steps
            const int BL = B * d->L[i], Di = d->D[i];
            char* imgX = (char*)(W + Lo.x3);
            char* imgW = imgX + rfn_x3_image_bytes(BL, Di);
            const float* srcs[64];
            float* outs[64];
            const float* bias[64];
            srcs[0] = att[i];
            RFN_TRY(rfn_x3_split(srcs, 1, Di, BL, Di, 1, imgX, st));
            for (int t = 0; t < T1; ++t) {
                srcs[t] = prm[P.s1(t, i, 0)];
                outs[t] = W + Lo.P1[i] + (long)t * BL * A;
                bias[t] = prm[P.s1(t, i, 1)];
            }
            RFN_TRY(rfn_x3_split(srcs, T1, Di, A, Di, 1, imgW, st));
            RFN_TRY(probe_mark(d, 2 * i, st));
            RFN_TRY(rfn_x3_gemm(BL, T1 * A, Di, imgX, imgW, BL, A, outs, bias, A, 0, 1, nullptr, st));
            RFN_TRY(probe_mark(d, 2 * i + 1, st));
            continue;
        }
        for (int t = 0; t < T1; ++t)
            pr[t] = prob1(W + Lo.P1[i] + (long)t * B * d->L[i] * A, A,
                          seg_lin(att[i], d->D[i], prm[P.s1(t, i, 0)], d->D[i], d->D[i], prm[P.s1(t, i, 1)]));
        RFN_TRY(probe_mark(d, 2 * i, st));
        RFN_TRY(gemm_groups(B * d->L[i], A, T1, pr, 0, gx));
        RFN_TRY(probe_mark(d, 2 * i + 1, st));
    }

    // ---- stage I: T1 steps x M cells (:213-217, :101-114, :47-74) ---------------------------
    for (int t = 0; t < T1; ++t) {
        float* Hc = Hs + t * BMR;
        float* Hn = Hs + (t + 1) * BMR;
        float* Cc = Cs + t * BMR;
        float* Cn = Cs + (t + 1) * BMR;
        float* hp = W + Lo.hp1 + (long)t * M * B * A;
        float* g = W + Lo.g1 + (long)t * M * B * 4 * R;
        {   // h_2_att_h of the M cells: one launch, encoder i's own h = column block i of H
            rfn_cell_out k1[RFN_MAX_ENC];
            for (int i = 0; i < M; ++i) {
                k1[i] = cell_out(hp + (long)i * B * A, A, A, 0);
                cell_lin(k1[i], Hc + i * R, MR, prm[P.s1(t, i, 2)], R, R, prm[P.s1(t, i, 3)]);
            }
            if (cell_ok(B, M, k1, R)) {
                RFN_TRY(cell_run(B, M, k1, R, 0.f, 0, st, cell_variant(d)));
            } else {
                for (int i = 0; i < M; ++i)
                    pr[i] = prob1(hp + (long)i * B * A, A,
                                  seg_lin(Hc + i * R, MR, prm[P.s1(t, i, 2)], R, R, prm[P.s1(t, i, 3)]));
                RFN_TRY(gemm_groups(B, A, M, pr, 0, gx));
            }
        }
        // attention of the M encoders: one grouped pair of launches when they share (L, D), else one pair each.
        // Raw scores land in the (idle) split-K scratch; the context kernel normalises them on the fly.
        bool same_ld = M > 1;
        for (int i = 1; i < M; ++i) same_ld = same_ld && d->L[i] == d->L[0] && d->D[i] == d->D[0];
        if (same_ld) {
            const long L0 = d->L[0], D0 = d->D[0];
            if ((size_t)M * B * L0 > GEMM_WS_FLOATS) return RFN_ERR_SHAPE;
            const float *a_p[RFN_MAX_ENC], *a_hp[RFN_MAX_ENC], *a_w[RFN_MAX_ENC], *a_b[RFN_MAX_ENC];
            float *a_sc[RFN_MAX_ENC], *a_al[RFN_MAX_ENC], *a_z[RFN_MAX_ENC];
            for (int i = 0; i < M; ++i) {
                a_p[i] = W + Lo.P1[i] + (long)t * B * L0 * A;
                a_hp[i] = hp + (long)i * B * A;
                a_w[i] = prm[P.s1(t, i, 4)];
                a_b[i] = prm[P.s1(t, i, 5)];
                a_sc[i] = W + Lo.gws + (long)i * B * L0;
                a_al[i] = W + Lo.al1[i] + (long)t * B * L0;
                a_z[i] = W + Lo.z1[i] + (long)t * B * D0;
            }
            RFN_TRY(rfn_attn_fwd_grouped(M, a_p, L0 * A, (long)A, a_hp, a_w, a_b, att, L0 * D0, D0, B, (int)L0,
                                         A, (int)D0, a_sc, a_al, a_z, D0, st));
        }
        bool het_done = false;
        if (!same_ld && M > 1) {   // maps of different (L, D): still one pair of launches (rfn_attn_fwd_het)
            const float *a_p[RFN_MAX_ENC], *a_hp[RFN_MAX_ENC], *a_w[RFN_MAX_ENC], *a_b[RFN_MAX_ENC];
            float *a_sc[RFN_MAX_ENC], *a_al[RFN_MAX_ENC], *a_z[RFN_MAX_ENC];
            int a_L[RFN_MAX_ENC], a_D[RFN_MAX_ENC];
            size_t sc_off = 0;
            for (int i = 0; i < M; ++i) {
                const long Li = d->L[i], Di = d->D[i];
                a_p[i] = W + Lo.P1[i] + (long)t * B * Li * A;
                a_hp[i] = hp + (long)i * B * A;
                a_w[i] = prm[P.s1(t, i, 4)];
                a_b[i] = prm[P.s1(t, i, 5)];
                a_sc[i] = W + Lo.gws + sc_off;
                sc_off += ((size_t)B * Li + 3) & ~(size_t)3;
                a_al[i] = W + Lo.al1[i] + (long)t * B * Li;
                a_z[i] = W + Lo.z1[i] + (long)t * B * Di;
                a_L[i] = (int)Li;
                a_D[i] = (int)Di;
            }
            if (sc_off > GEMM_WS_FLOATS) return RFN_ERR_SHAPE;
            RFN_TRY(rfn_attn_fwd_het(M, a_p, a_hp, a_w, a_b, att, B, a_L, A, a_D, a_sc, a_al, a_z, st));
            het_done = true;
        }
        for (int i = 0; i < M; ++i) {
            const long Li = d->L[i], Di = d->D[i];
            float* al = W + Lo.al1[i] + (long)t * B * Li;
            float* z = W + Lo.z1[i] + (long)t * B * Di;
            if (!same_ld && !het_done) {
                if ((size_t)B * Li > GEMM_WS_FLOATS) return RFN_ERR_SHAPE;
                RFN_TRY(rfn_attn_fwd(W + Lo.P1[i] + (long)t * B * Li * A, Li * A, (long)A, hp + (long)i * B * A,
                                     prm[P.s1(t, i, 4)], prm[P.s1(t, i, 5)], att[i], Li * Di, Di, B, (int)Li, A,
                                     (int)Di, W + Lo.gws, al, z, Di, st));
            }
            rfn_gemm_problem& p = pr[i];
            memset(&p, 0, sizeof(p));
            p.C = g + (long)i * B * 4 * R;
            p.ldc = 4 * R;
            p.nseg = 2;
            p.seg[0] = seg_lin(Hc, MR, prm[P.s1(t, i, 6)], MR, (int)MR, prm[P.s1(t, i, 7)]);
            p.seg[1] = seg_lin(z, Di, prm[P.s1(t, i, 8)], Di, (int)Di, prm[P.s1(t, i, 9)]);
        }
        // Small batches (BASELINE config 2, the shards of a strong-scaled batch): the M gate products in ONE cell-GEMM launch
        // with the LSTM update as its epilogue (no split-K partials, no reduce launch) -- when the launch's 32-row tiles do not
        // outnumber the CUs two to one; beyond that the 128 x 128 split-K kernel below is the faster one (17 GFLOP per step at C3).
        if ((long)rfn_cdiv(B, 32) * M * (4 * R / 32) <= 2L * stage1_cell_cus()) {
            rfn_cell_out kg[RFN_MAX_ENC];
            for (int i = 0; i < M; ++i) {
                kg[i] = cell_out(g + (long)i * B * 4 * R, 4 * R, 4 * R, 0);
                cell_lin(kg[i], Hc, MR, prm[P.s1(t, i, 6)], MR, (int)MR, prm[P.s1(t, i, 7)]);
                cell_lin(kg[i], W + Lo.z1[i] + (long)t * B * d->D[i], d->D[i], prm[P.s1(t, i, 8)], d->D[i], d->D[i], prm[P.s1(t, i, 9)]);
                cell_lstm(kg[i], Cc + i * R, MR, Cn + i * R, MR, Hn + i * R, MR, (uint64_t)(t * M + i));
            }
            if (cell_ok(B, M, kg, R)) {
                RFN_TRY(cell_run(B, M, kg, R, d->drop_fusion, seed, st, cell_variant(d)));
                continue;
            }
        }
        // gate GEMM of the M cells (grouped) with the LSTM update riding on its split-K reduce: encoder i's state is column
        // block i of the (B, M*R) rows
        rfn_gemm_lstm lu;
        memset(&lu, 0, sizeof(lu));
        lu.c_prev = Cc; lu.c_next = Cn; lu.h_next = Hn;
        lu.ldcp = lu.ldcn = lu.ldh = MR;
        lu.gs_cprev = lu.gs_cnext = lu.gs_h = R;
        lu.drop_p = d->drop_fusion; lu.seed = seed; lu.offset = (uint64_t)(t * M);
        RFN_TRY(rfn_gemm_f32_lstm(B, R, M, pr, gx.ws, gx.ws_bytes, gx.flags, &lu, st));
    }

    // reason heads of stage I: max over steps of reason_linear_individual (:217, :229)
    float* rmat = W + Lo.rmat;
    for (int i = 0; i < M; ++i)   // all encoders' heads: one grouped GEMM into M slabs, one max-over-steps launch
        pr[i] = prob1(rmat + (long)i * T1 * B * K, K,
                      seg_lin(Hs + BMR + i * R, MR, prm[P.rind_w(i)], R, R, prm[P.rind_b(i)]));
    RFN_TRY(gemm_groups(T1 * B, K, M, pr, 0, gx));
    RFN_TRY(rfn_max_over_steps_fwd_grouped(rmat, T1, B, K, reason_pred, rarg, M, st));

    // state mean over encoders (:233-235): sum first, then divide, as the reference does
    float* h2 = W + Lo.h2;
    float* c2 = W + Lo.c2;
    {
        const float* mx[2] = {Hs + T1 * BMR, Cs + T1 * BMR};
        float* my[2] = {h2, c2};
        RFN_TRY(rfn_mean_over_groups(2, mx, MR, R, M, my, R, B, R, st));
    }

    // hoisted thought projections of stage II: rows (t', b) of encoder i's thoughts = Hs[1:]
    for (int i = 0; i < M; ++i) {
        if (T2 > 64) return RFN_ERR_SHAPE;
        for (int t = 0; t < T2; ++t)
            pr[t] = prob1(W + Lo.P2[i] + (long)t * A, (long)T2 * A,
                          seg_lin(Hs + BMR + i * R, MR, prm[P.s2(t, i, 2)], R, R, prm[P.s2(t, i, 3)]));
        RFN_TRY(gemm_groups(T1 * B, A, T2, pr, 0, gx));
    }

    // ---- stage II: T2 steps (:241-244, LSTMSoftMultiAttentionFeatArrayNoInputCore.py:41-73) ---
    rfn_gemm_seg segs[RFN_MAX_ENC + 1];
    // Fused form of a step (3 launches): K1 = every h_2_att_h_i(h) and h2h(h) in one launch (they share h); the M attentions;
    // K3 = sum_i z_2_h_i(z_i) accumulated onto the gates with the LSTM update as its epilogue.  When every step takes it the
    // T2 steps run inside one persistent launch (rfn_chain.hip).
    auto s2_step = [&](int t, ChainStep* cs) -> bool {
        float* hc = h2 + t * BR;
        float* hn = h2 + (t + 1) * BR;
        float* hp = W + Lo.hp2 + (long)t * M * B * A;
        float* al = W + Lo.al2 + (long)t * M * B * T1;
        float* z = W + Lo.z2 + (long)t * M * BR;
        float* g = W + Lo.g2 + (long)t * B * G2;
        if (d->review_maxout) return false;
        rfn_cell_out k1[RFN_MAX_ENC + 1], k3;
        for (int i = 0; i < M; ++i) {
            k1[i] = cell_out(hp + (long)i * B * A, A, A, 0);
            cell_lin(k1[i], hc, R, prm[P.s2(t, i, 4)], R, R, prm[P.s2(t, i, 5)]);
        }
        k1[M] = cell_out(g, G2, G2, 0);
        cell_lin(k1[M], hc, R, prm[P.s2_hh_w(t)], R, R, prm[P.s2_hh_b(t)]);
        k3 = cell_out(g, G2, G2, 1);
        for (int i = 0; i < M; ++i) cell_lin(k3, z + i * BR, R, prm[P.s2(t, i, 0)], R, R, prm[P.s2(t, i, 1)]);
        cell_lstm(k3, c2 + t * BR, R, c2 + (t + 1) * BR, R, hn, R, OFF_STAGE2 + (uint64_t)t);
        if (!cell_ok(B, M + 1, k1, R) || !cell_ok(B, 1, &k3, R)) return false;
        const float *a_p[RFN_MAX_ENC], *a_hp[RFN_MAX_ENC], *a_w[RFN_MAX_ENC], *a_b[RFN_MAX_ENC], *a_x[RFN_MAX_ENC];
        float *a_al[RFN_MAX_ENC], *a_z[RFN_MAX_ENC];
        for (int i = 0; i < M; ++i) {
            a_p[i] = W + Lo.P2[i] + (long)t * A;
            a_hp[i] = hp + (long)i * B * A;
            a_w[i] = prm[P.s2(t, i, 6)];
            a_b[i] = prm[P.s2(t, i, 7)];
            a_x[i] = Hs + BMR + i * R;
            a_al[i] = al + (long)i * B * T1;
            a_z[i] = z + i * BR;
        }
        return cell_prepare(B, M + 1, k1, R, 0.f, 0, &cs->g0, cell_variant(d)) == RFN_OK &&
               rfn_attn_small_prepare_fwd(M, a_p, (long)T2 * A, (long)B * T2 * A, a_hp, a_w, a_b, a_x, MR, BMR, B, T1, A, R, a_al, a_z,
                                          R, &cs->at) == RFN_OK &&
               cell_prepare(B, 1, &k3, R, d->drop_reason, seed, &cs->g2, cell_variant(d)) == RFN_OK;
    };
    bool s2_chained = false;
    {
        std::vector<ChainStep> steps((size_t)T2);
        bool ok = true;
        for (int t = 0; t < T2 && ok; ++t) ok = s2_step(t, &steps[t]);
        if (ok) {
            RFN_TRY(rfn_chain_run(steps.data(), T2, chain_persist(d, RFN_PATH_OPT_PERSIST_S2_FWD), (uint32_t*)(W + Lo.bar), st));
            s2_chained = true;
        }
    }
    for (int t = 0; t < T2 && !s2_chained; ++t) {
        float* hc = h2 + t * BR;
        float* hn = h2 + (t + 1) * BR;
        float* cc = c2 + t * BR;
        float* cn = c2 + (t + 1) * BR;
        float* hp = W + Lo.hp2 + (long)t * M * B * A;
        float* al = W + Lo.al2 + (long)t * M * B * T1;
        float* z = W + Lo.z2 + (long)t * M * BR;
        float* g = W + Lo.g2 + (long)t * B * G2;
        // Fused form (3 launches): K1 = every h_2_att_h_i(h) and h2h(h) in one launch (they share h); the M attentions;
        // K3 = sum_i z_2_h_i(z_i) accumulated onto the gates with the LSTM update as its epilogue.
        rfn_cell_out k1[RFN_MAX_ENC + 1], k3;
        for (int i = 0; i < M; ++i) {
            k1[i] = cell_out(hp + (long)i * B * A, A, A, 0);
            cell_lin(k1[i], hc, R, prm[P.s2(t, i, 4)], R, R, prm[P.s2(t, i, 5)]);
        }
        k1[M] = cell_out(g, G2, G2, 0);
        cell_lin(k1[M], hc, R, prm[P.s2_hh_w(t)], R, R, prm[P.s2_hh_b(t)]);
        k3 = cell_out(g, G2, G2, 1);
        for (int i = 0; i < M; ++i) cell_lin(k3, z + i * BR, R, prm[P.s2(t, i, 0)], R, R, prm[P.s2(t, i, 1)]);
        cell_lstm(k3, cc, R, cn, R, hn, R, OFF_STAGE2 + (uint64_t)t);
        const bool fused = !d->review_maxout && cell_ok(B, M + 1, k1, R) && cell_ok(B, 1, &k3, R);
        if (fused) {
            RFN_TRY(cell_run(B, M + 1, k1, R, 0.f, 0, st, cell_variant(d)));
        } else {
            for (int i = 0; i < M; ++i)
                pr[i] = prob1(hp + (long)i * B * A, A, seg_lin(hc, R, prm[P.s2(t, i, 4)], R, R, prm[P.s2(t, i, 5)]));
            RFN_TRY(gemm_groups(B, A, M, pr, 0, gx));
        }
        segs[0] = seg_lin(hc, R, prm[P.s2_hh_w(t)], R, R, prm[P.s2_hh_b(t)]);
        {   // attention of all M encoders over the T1 thoughts: one fused launch
            const float *a_p[RFN_MAX_ENC], *a_hp[RFN_MAX_ENC], *a_w[RFN_MAX_ENC], *a_b[RFN_MAX_ENC], *a_x[RFN_MAX_ENC];
            float *a_al[RFN_MAX_ENC], *a_z[RFN_MAX_ENC];
            for (int i = 0; i < M; ++i) {
                a_p[i] = W + Lo.P2[i] + (long)t * A;
                a_hp[i] = hp + (long)i * B * A;
                a_w[i] = prm[P.s2(t, i, 6)];
                a_b[i] = prm[P.s2(t, i, 7)];
                a_x[i] = Hs + BMR + i * R;
                a_al[i] = al + (long)i * B * T1;
                a_z[i] = z + i * BR;
                segs[1 + i] = seg_lin(z + i * BR, R, prm[P.s2(t, i, 0)], R, R, prm[P.s2(t, i, 1)]);
            }
            RFN_TRY(rfn_attn_small_fwd(M, a_p, (long)T2 * A, (long)B * T2 * A, a_hp, a_w, a_b, a_x, MR, BMR, B, T1, A,
                                       R, a_al, a_z, R, st));
        }
        if (fused) {
            RFN_TRY(cell_run(B, 1, &k3, R, d->drop_reason, seed, st, cell_variant(d)));
        } else {
            RFN_TRY(gemm_segs(B, G2, M + 1, segs, g, G2, 0, gx));
            RFN_TRY(rfn_lstm_fwd(g, G2, cc, R, cn, R, hn, R, B, R, d->review_maxout, d->drop_reason, seed,
                                 OFF_STAGE2 + (uint64_t)t, st));
        }
    }
    RFN_TRY(gemm1(T2 * B, K, seg_lin(h2 + BR, R, prm[P.r_w()], R, R, prm[P.r_b()]), rmat, K, 0, gx));
    RFN_TRY(rfn_max_over_steps_fwd(rmat, T2, B, K, reason_pred + (long)M * B * K, rarg + (long)M * B * K, st));

    return mem_batch({{comb, h2 + BR, (long)T2 * BR}, {h_out, h2 + (long)T2 * BR, BR}, {c_out, c2 + (long)T2 * BR, BR}}, st);
}

extern "C" int rfn_prefix_fwd(const rfn_dims* d, int B, const float* const* prm, const float* const* fc,
                              const float* const* att, float* comb, float* h_out, float* c_out, float* reason_pred,
                              void* ws, size_t ws_bytes, int train, uint64_t seed, void* st) {
    if (!fc) return RFN_ERR_ARG;
    return prefix_fwd_impl(d, B, prm, fc, nullptr, nullptr, att, comb, h_out, c_out, reason_pred, ws, ws_bytes, train,
                           seed, st);
}
// get_thought_vectors with a caller-provided state_list (inference only: no backward through the given state)
extern "C" int rfn_prefix_fwd_from_state(const rfn_dims* d, int B, const float* const* prm, const float* const* init_h,
                                         const float* const* init_c, const float* const* att, float* comb,
                                         float* h_out, float* c_out, float* reason_pred, void* ws, size_t ws_bytes,
                                         void* st) {
    if (!init_h || !init_c) return RFN_ERR_ARG;
    return prefix_fwd_impl(d, B, prm, nullptr, init_h, init_c, att, comb, h_out, c_out, reason_pred, ws, ws_bytes, 0, 0,
                           st);
}

extern "C" int rfn_prefix_bwd(const rfn_dims* d, int B, const float* const* prm, const float* const* fc,
                              const float* const* att, const float* d_comb, const float* d_h, const float* d_c,
                              const float* d_reason, float* const* grd, void* ws, size_t ws_bytes, uint64_t seed,
                              int defer_wgrad, void* st) {
    RFN_TRY(check_dims(d));
    if (B < 1) return RFN_ERR_SHAPE;
    if (!prm || !fc || !att || !grd || !ws) return RFN_ERR_ARG;
    const PrefixLayout Lo = prefix_layout(d, B, 1);
    if (ws_bytes < Lo.total * sizeof(float)) return RFN_ERR_WORKSPACE;
    const PIdx P(d);
    const int M = d->M, R = d->R, A = d->A, T1 = d->T1, T2 = d->T2, K = d->K;
    const int G2 = gate_width(d->review_maxout, R);
    const long MR = (long)M * R, BMR = (long)B * MR, BR = (long)B * R, BA = (long)B * A;
    float* W = (float*)ws;
    const bool tk_on = (d->gemm_flags & RFN_GEMM_OPT_SPLITK_IN_KERNEL) != 0;   // measured slower: off unless asked for (rfn.h)
    const GemmCtx gx{st, W + Lo.gws, GEMM_WS_FLOATS * sizeof(float), d->gemm_flags, tk_on ? (int32_t*)(W + Lo.tk) : nullptr,
                     tk_on ? GEMM_TICKETS : 0};
    if (tk_on) RFN_TRY(zero_f32(W + Lo.tk, GEMM_TICKETS, st));   // tile counters: zero on entry, every launch leaves them zero
    float* Hs = W + Lo.Hs;
    float* Cs = W + Lo.Cs;
    float* h2 = W + Lo.h2;
    float* c2 = W + Lo.c2;
    float* rmat = W + Lo.rmat;
    const int32_t* rarg = (const int32_t*)(W + Lo.rarg);
    float* dHs = W + Lo.dHs;
    float* dC = W + Lo.dC;
    float* dh2e = W + Lo.dh2e;
    float* dhrec = W + Lo.dhrec;
    float* dc2 = W + Lo.dc2;
    float* dz2 = W + Lo.dz2;
    float* dal = W + Lo.dal;
    float* dwp = W + Lo.dwp;
    rfn_gemm_problem pr[64];
    rfn_gemm_seg segs[64];
    if (T1 > 64 || T2 > 64) return RFN_ERR_SHAPE;

    // ---- reason head of stage II (:244, :253) -------------------------------------------------
    RFN_TRY(rfn_max_over_steps_bwd(d_reason ? d_reason + (long)M * B * K : nullptr, rarg + (long)M * B * K, T2, B, K,
                                   rmat, st));
    // d thoughts from the decoder (or zero) into the slab the reason head accumulates onto; d H of stage I zeroed
    RFN_TRY(mem_batch({{dh2e, d_comb, (long)T2 * BR}, {dHs, nullptr, (long)(T1 + 1) * BMR}}, st));
    RFN_TRY(gemm1(T2 * B, R, seg_dx(rmat, K, prm[P.r_w()], R, K), dh2e, R, 1, gx));
    RFN_TRY(gemm_dw(K, R, grd[P.r_w()], R, grd[P.r_b()], rmat, K, h2 + BR, R, T2 * B, gx));

    // gradient w.r.t. the stage-I hidden states (zeroed above): thoughts (through stage II) + reason heads + mean

    // ---- stage II backward ------------------------------------------------------------------------
    // Fused form of a step (3 launches): Kb1 = dh_rec and every dz_i = dgates . [W_hh | W_z_i] in one launch; the M
    // attention backwards; Kb2 = dh_rec += sum_i dhp_i . W_h_i whose epilogue completes d h of step t-1 (+ its external
    // share dh2e[t-1]) and runs that step's LSTM backward.
    // The operands of EVERY step are validated before the fused form is chosen (each step has its own weights and slabs;
    // a step the cell GEMM cannot take must not be discovered mid-sweep, when the gates are already gate gradients).
    auto s2_kb1 = [&](int t, rfn_cell_out* kb1) {
        float* g = W + Lo.g2 + (long)t * B * G2;
        kb1[0] = cell_out(dhrec, R, R, 0);
        cell_dx(kb1[0], g, G2, prm[P.s2_hh_w(t)], R, G2);
        for (int i = 0; i < M; ++i) {
            kb1[1 + i] = cell_out(dz2 + i * BR, R, R, 0);
            cell_dx(kb1[1 + i], g, G2, prm[P.s2(t, i, 0)], R, G2);
        }
    };
    auto s2_kb2 = [&](int t, rfn_cell_out& kb2) {
        float* dhp = W + Lo.dhp2 + (long)t * M * BA;
        kb2 = cell_out(dhrec, R, R, 1);
        for (int i = 0; i < M; ++i) cell_dx(kb2, dhp + i * BA, A, prm[P.s2(t, i, 4)], R, A);
        if (t > 0)
            cell_lstm_bwd(kb2, W + Lo.g2 + (long)(t - 1) * B * G2, G2, c2 + (t - 1) * BR, R, c2 + t * BR, R, dh2e + (t - 1) * BR, R,
                          dc2, R, dc2, R, OFF_STAGE2 + (uint64_t)(t - 1));
    };
    bool fused2 = !d->review_maxout;
    for (int t = 0; t < T2 && fused2; ++t) {
        rfn_cell_out t1[RFN_MAX_ENC + 1], t2;
        s2_kb1(t, t1);
        s2_kb2(t, t2);
        fused2 = cell_ok(B, M + 1, t1, R) && cell_ok(B, 1, &t2, R);
    }
    if (fused2) {   // LSTM backward of the last step: only the external gradients (thoughts, reason head, decoder state)
        float* dht = dh2e + (T2 - 1) * BR;
        if (d_h) RFN_TRY(rfn_axpby_2d(1.f, d_h, R, 1.f, dht, R, B, R, st));
        RFN_TRY(rfn_lstm_bwd(W + Lo.g2 + (long)(T2 - 1) * B * G2, G2, c2 + (T2 - 1) * BR, R, c2 + T2 * BR, R, dht, R, d_c, R, dc2, R,
                             B, R, 0, d->drop_reason, seed, OFF_STAGE2 + (uint64_t)(T2 - 1), st));
    }
    int t_hi = T2 - 1;
    if (fused2 && T2 >= 3) {
        // steps T2-1 ... 1 in one persistent launch (rfn_chain.hip); step 0, whose Kb2 is a plain accumulate, follows as launches
        std::vector<ChainStep> steps((size_t)(T2 - 1));
        bool ok = true;
        for (int t = T2 - 1; t >= 1 && ok; --t) {
            ChainStep& cs = steps[(size_t)(T2 - 1 - t)];
            rfn_cell_out kb1[RFN_MAX_ENC + 1], kb2;
            s2_kb1(t, kb1);
            s2_kb2(t, kb2);
            float* hp = W + Lo.hp2 + (long)t * M * BA;
            float* dhp = W + Lo.dhp2 + (long)t * M * BA;
            float* al = W + Lo.al2 + (long)t * M * B * T1;
            const float *a_p[RFN_MAX_ENC], *a_hp[RFN_MAX_ENC], *a_w[RFN_MAX_ENC], *a_al[RFN_MAX_ENC], *a_x[RFN_MAX_ENC],
                *a_dz[RFN_MAX_ENC];
            float *a_dp[RFN_MAX_ENC], *a_dhp[RFN_MAX_ENC], *a_dw[RFN_MAX_ENC], *a_dx[RFN_MAX_ENC];
            for (int i = 0; i < M; ++i) {
                a_dp[i] = W + Lo.P2[i] + (long)t * A;
                a_p[i] = a_dp[i];
                a_hp[i] = hp + i * BA;
                a_w[i] = prm[P.s2(t, i, 6)];
                a_al[i] = al + (long)i * B * T1;
                a_x[i] = Hs + BMR + i * R;
                a_dz[i] = dz2 + i * BR;
                a_dhp[i] = dhp + i * BA;
                a_dw[i] = dwp + ((long)t * M + i) * BA;
                a_dx[i] = dHs + BMR + i * R;
            }
            ok = cell_prepare(B, M + 1, kb1, R, 0.f, 0, &cs.g0, cell_variant(d)) == RFN_OK &&
                 rfn_attn_small_prepare_bwd(M, a_p, (long)T2 * A, (long)B * T2 * A, a_hp, a_w, a_al, a_x, MR, BMR, a_dz, R, B, T1, A, R,
                                            a_dp, (long)T2 * A, (long)B * T2 * A, 0, a_dhp, a_dw, a_dx, &cs.at) == RFN_OK &&
                 cell_prepare(B, 1, &kb2, R, d->drop_reason, seed, &cs.g2, cell_variant(d)) == RFN_OK;
        }
        if (ok) {
            RFN_TRY(rfn_chain_run(steps.data(), T2 - 1, chain_persist(d, RFN_PATH_OPT_PERSIST_S2_BWD), (uint32_t*)(W + Lo.bar), st));
            t_hi = 0;
        }
    }
    for (int t = t_hi; t >= 0; --t) {
        float* hp = W + Lo.hp2 + (long)t * M * BA;
        float* dhp = W + Lo.dhp2 + (long)t * M * BA;
        float* al = W + Lo.al2 + (long)t * M * B * T1;
        float* g = W + Lo.g2 + (long)t * B * G2;
        float* dht = dh2e + t * BR;  // total dh of h2[t+1]
        if (!fused2) {
            if (t == T2 - 1) {
                if (d_h) RFN_TRY(rfn_axpby_2d(1.f, d_h, R, 1.f, dht, R, B, R, st));
            } else {
                RFN_TRY(rfn_axpby_2d(1.f, dhrec, R, 1.f, dht, R, B, R, st));
            }
            const float* dcn = (t == T2 - 1) ? d_c : dc2;
            RFN_TRY(rfn_lstm_bwd(g, G2, c2 + t * BR, R, c2 + (t + 1) * BR, R, dht, R, dcn, R, dc2, R, B, R, d->review_maxout,
                                 d->drop_reason, seed, OFF_STAGE2 + (uint64_t)t, st));
            // dh_rec = dgates . W_hh ; dz_i = dgates . W_z_i   (same shape: one grouped launch)
            pr[0] = prob1(dhrec, R, seg_dx(g, G2, prm[P.s2_hh_w(t)], R, G2));
            for (int i = 0; i < M; ++i) pr[1 + i] = prob1(dz2 + i * BR, R, seg_dx(g, G2, prm[P.s2(t, i, 0)], R, G2));
            RFN_TRY(gemm_groups(B, R, M + 1, pr, 0, gx));
        } else {
            rfn_cell_out kb1[RFN_MAX_ENC + 1];
            s2_kb1(t, kb1);
            RFN_TRY(cell_run(B, M + 1, kb1, R, 0.f, 0, st, cell_variant(d)));
        }
        {   // whole attention backward of the M encoders in one fused launch (dP overwrites P in place)
            const float *a_p[RFN_MAX_ENC], *a_hp[RFN_MAX_ENC], *a_w[RFN_MAX_ENC], *a_al[RFN_MAX_ENC], *a_x[RFN_MAX_ENC],
                *a_dz[RFN_MAX_ENC];
            float *a_dp[RFN_MAX_ENC], *a_dhp[RFN_MAX_ENC], *a_dw[RFN_MAX_ENC], *a_dx[RFN_MAX_ENC];
            for (int i = 0; i < M; ++i) {
                a_dp[i] = W + Lo.P2[i] + (long)t * A;
                a_p[i] = a_dp[i];
                a_hp[i] = hp + i * BA;
                a_w[i] = prm[P.s2(t, i, 6)];
                a_al[i] = al + (long)i * B * T1;
                a_x[i] = Hs + BMR + i * R;  // thoughts_i[b, l] = Hs[1 + l][b, iR:]
                a_dz[i] = dz2 + i * BR;
                a_dhp[i] = dhp + i * BA;
                a_dw[i] = dwp + ((long)t * M + i) * BA;
                a_dx[i] = dHs + BMR + i * R;
                segs[i] = seg_dx(dhp + i * BA, A, prm[P.s2(t, i, 4)], R, A);
            }
            RFN_TRY(rfn_attn_small_bwd(M, a_p, (long)T2 * A, (long)B * T2 * A, a_hp, a_w, a_al, a_x, MR, BMR, a_dz, R,
                                       B, T1, A, R, a_dp, (long)T2 * A, (long)B * T2 * A, 0, a_dhp, a_dw, a_dx, st));
        }
        if (!fused2) {
            RFN_TRY(gemm_segs(B, R, M, segs, dhrec, R, 1, gx));
        } else {
            rfn_cell_out kb2;
            s2_kb2(t, kb2);
            RFN_TRY(cell_run(B, 1, &kb2, R, d->drop_reason, seed, st, cell_variant(d)));
        }
    }
    // weight gradients of stage II, grouped over steps; every bias gradient rides on the GEMM that streams
    // the same dY (h2h.b = z_2_h[i].b = colsum(dgates); h_2_att_h.b = att_2_att_h.b = colsum over (b) resp. (l,b))
    {
        float* outs[64];
        for (int t = 0; t < T2; ++t)
            for (int i = 0; i < M; ++i) outs[t * M + i] = grd[P.s2(t, i, 6)];
        if (T2 * M > 64) return RFN_ERR_SHAPE;
        RFN_TRY(rfn_colsum_grouped_f32(dwp, BA, A, B, A, outs, T2 * M, st));
        // att_h_2_out.bias shifts all scores of a softmax equally: its gradient is exactly 0
        for (int t = 0; t < T2; ++t)
            for (int i = 0; i < M; ++i) outs[t * M + i] = grd[P.s2(t, i, 7)];
        RFN_TRY(rfn_fill_small_f32(outs, T2 * M, 1, 0.f, st));
    }
    for (int t = 0; t < T2; ++t)
        pr[t] = prob_dw(grd[P.s2_hh_w(t)], R, grd[P.s2_hh_b(t)], W + Lo.g2 + (long)t * B * G2, G2, h2 + t * BR, R, B);
    RFN_TRY(gemm_groups(G2, R, T2, pr, 0, gx));
    for (int i = 0; i < M; ++i) {
        for (int t = 0; t < T2; ++t)
            pr[t] = prob_dw(grd[P.s2(t, i, 0)], R, grd[P.s2(t, i, 1)], W + Lo.g2 + (long)t * B * G2, G2,
                            W + Lo.z2 + ((long)t * M + i) * BR, R, B);
        RFN_TRY(gemm_groups(G2, R, T2, pr, 0, gx));
        for (int t = 0; t < T2; ++t)
            pr[t] = prob_dw(grd[P.s2(t, i, 4)], R, grd[P.s2(t, i, 5)], W + Lo.dhp2 + ((long)t * M + i) * BA, A,
                            h2 + t * BR, R, B);
        RFN_TRY(gemm_groups(A, R, T2, pr, 0, gx));
        // d att_2_att_h.weight[t] = dP2_i[:, t]^T . thoughts_i   (K = T1*B rows)
        for (int t = 0; t < T2; ++t)
            pr[t] = prob_dw(grd[P.s2(t, i, 2)], R, grd[P.s2(t, i, 3)], W + Lo.P2[i] + (long)t * A, (long)T2 * A,
                            Hs + BMR + i * R, MR, T1 * B);
        RFN_TRY(gemm_groups(A, R, T2, pr, 0, gx));
        // d thoughts_i += sum_t dP2_i[:, t] . W_a[t]
        for (int t = 0; t < T2; ++t) segs[t] = seg_dx(W + Lo.P2[i] + (long)t * A, (long)T2 * A, prm[P.s2(t, i, 2)], R, A);
        RFN_TRY(gemm_segs(T1 * B, R, T2, segs, dHs + BMR + i * R, MR, 1, gx));
    }

    // ---- state mean backward (:233-235): every encoder's final (h, c) gets d / M --------------------
    const float invM = 1.0f / (float)M;
    {
        const float* bx[2] = {dhrec, dc2};
        float* by[2] = {dHs + T1 * BMR, dC};
        const float bb[2] = {1.f, 0.f};
        RFN_TRY(rfn_bcast_to_groups(2, invM, bx, R, bb, by, MR, R, M, B, R, st));
    }
    // ---- reason heads of stage I ----------------------------------------------------------------------
    RFN_TRY(rfn_max_over_steps_bwd_grouped(d_reason, rarg, T1, B, K, rmat, M, st));
    for (int i = 0; i < M; ++i)
        pr[i] = prob1(dHs + BMR + i * R, MR, seg_dx(rmat + (long)i * T1 * B * K, K, prm[P.rind_w(i)], R, K));
    RFN_TRY(gemm_groups(T1 * B, R, M, pr, 1, gx));
    for (int i = 0; i < M; ++i)
        pr[i] = prob_dw(grd[P.rind_w(i)], R, grd[P.rind_b(i)], rmat + (long)i * T1 * B * K, K, Hs + BMR + i * R, MR,
                        T1 * B);
    RFN_TRY(gemm_groups(K, R, M, pr, 0, gx));

    // ---- stage I backward --------------------------------------------------------------------------------
    // Small batches (BASELINE config 2, the shards of a strong-scaled batch) take three launches per step instead of seven:
    //   X  every product of the step's gate gradients in ONE cell-GEMM launch: the M partial slabs d gates_j . W_H[t,j] of
    //      d H_t (each cell reads the whole concatenated H, :53) and the M d z_i = d gates_i . W_z[t,i];
    //   the attention backward of the M encoders (unchanged);
    //   Y  d H_t[:, i] = external + sum_j slab_j[:, i] + d hproj_i . W_h[t,i], whose epilogue runs the LSTM backward of cell
    //      (t-1, i) -- the next thing the sweep needs (no split-K partials, no reduce / axpby / lstm launches).
    // Taken while the X launch's 32-row tiles do not outnumber the CUs two to one and every product fits the cell GEMM.
    auto s1x_of = [&](int t, rfn_cell_out* kx) {
        float* g = W + Lo.g1 + (long)t * M * B * 4 * R;
        for (int j = 0; j < M; ++j) {
            kx[j] = cell_out(W + Lo.dHpart + (long)j * BMR, MR, (int)MR, 0);
            cell_dx(kx[j], g + (long)j * B * 4 * R, 4 * R, prm[P.s1(t, j, 6)], MR, 4 * R);
        }
        for (int i = 0; i < M; ++i) {
            kx[M + i] = cell_out(W + Lo.dz1[i], d->D[i], d->D[i], 0);
            cell_dx(kx[M + i], g + (long)i * B * 4 * R, 4 * R, prm[P.s1(t, i, 8)], d->D[i], 4 * R);
        }
    };
    auto s1y_of = [&](int t, rfn_cell_out* ky) {
        float* dHc = dHs + t * BMR;
        float* dhp = W + Lo.dhp1 + (long)t * M * BA;
        for (int i = 0; i < M; ++i) {
            ky[i] = cell_out(dHc + i * R, MR, R, 1);
            ky[i].acc_slabs = W + Lo.dHpart + i * R;
            ky[i].acc_parts = M;
            ky[i].acc_stride = BMR;
            cell_dx(ky[i], dhp + i * BA, A, prm[P.s1(t, i, 2)], R, A);
            if (t > 0)
                cell_lstm_bwd(ky[i], W + Lo.g1 + ((long)(t - 1) * M + i) * B * 4 * R, 4 * R, Cs + (t - 1) * BMR + i * R, MR,
                              Cs + t * BMR + i * R, MR, nullptr, 0, dC + i * R, MR, dC + i * R, MR, (uint64_t)((t - 1) * M + i));
        }
    };
    bool s1_small = 2 * M <= RFN_CELL_MAXOUT && M <= 8;
    {
        long cols = (long)M * (MR / 32);
        for (int i = 0; i < M; ++i) cols += d->D[i] / 32;
        s1_small = s1_small && (long)rfn_cdiv(B, 32) * cols <= 2L * stage1_cell_cus();
        for (int t = 0; t < T1 && s1_small; ++t) {
            rfn_cell_out tx[2 * RFN_MAX_ENC], ty[RFN_MAX_ENC];
            s1x_of(t, tx);
            s1y_of(t, ty);
            // step 0's Y is a plain accumulate, the others carry the gate-gradient epilogue: one epilogue per launch
            s1_small = cell_ok(B, 2 * M, tx, R) && cell_ok(B, M, ty, R);
        }
    }
    for (int t = T1 - 1; t >= 0; --t) {
        float* dHn = dHs + (t + 1) * BMR;  // total gradient of Hs[t+1]
        float* dHc = dHs + t * BMR;        // external gradient of Hs[t]; the recurrent part is added here
        float* g = W + Lo.g1 + (long)t * M * B * 4 * R;
        float* hp = W + Lo.hp1 + (long)t * M * BA;
        float* dhp = W + Lo.dhp1 + (long)t * M * BA;
        if (!s1_small || t == T1 - 1)
            RFN_TRY(rfn_lstm_bwd_grouped(g, 4 * R, Cs + t * BMR, MR, Cs + (t + 1) * BMR, MR, dHn, MR, dC, MR, dC, MR, B, R, 0,
                                         d->drop_fusion, seed, (uint64_t)(t * M), M, (long)B * 4 * R, R, R, R, st));
        if (s1_small) {
            rfn_cell_out kx[2 * RFN_MAX_ENC];
            s1x_of(t, kx);
            RFN_TRY(cell_run(B, 2 * M, kx, R, 0.f, 0, st, cell_variant(d)));
        } else {
        // dH_t += sum_i dgates_i . W_H[t,i]   (every cell reads the whole concatenated H, :53)
        for (int i = 0; i < M; ++i) segs[i] = seg_dx(g + (long)i * B * 4 * R, 4 * R, prm[P.s1(t, i, 6)], MR, 4 * R);
        RFN_TRY(gemm_segs(B, (int)MR, M, segs, dHc, MR, 1, gx));
        }
        // dz_i = dgates_i . W_z[t,i]: one grouped launch when the encoders share a feature width
        bool same_d = true;
        for (int i = 1; i < M; ++i) same_d = same_d && d->D[i] == d->D[0];
        if (s1_small) {
            // done by X
        } else if (same_d && M > 1) {
            for (int i = 0; i < M; ++i)
                pr[i] = prob1(W + Lo.dz1[i], d->D[0],
                              seg_dx(g + (long)i * B * 4 * R, 4 * R, prm[P.s1(t, i, 8)], d->D[0], 4 * R));
            RFN_TRY(gemm_groups(B, d->D[0], M, pr, 0, gx));
        }
        bool dz_done = (same_d && M > 1) || s1_small;
        if (!dz_done && M > 1) {   // heterogeneous feature widths: the M products still share one launch (an output each)
            rfn_cell_out kz[RFN_MAX_ENC];
            for (int i = 0; i < M; ++i) {
                kz[i] = cell_out(W + Lo.dz1[i], d->D[i], d->D[i], 0);
                cell_dx(kz[i], g + (long)i * B * 4 * R, 4 * R, prm[P.s1(t, i, 8)], d->D[i], 4 * R);
            }
            if (cell_ok(B, M, kz, R)) {
                RFN_TRY(cell_run(B, M, kz, R, 0.f, 0, st, cell_variant(d)));
                dz_done = true;
            }
        }
        const AttnBwdForm form0 = attn_bwd_form(d, B, 0, dz_done);
        const bool grouped_bwd = form0 == AB_GROUPED || form0 == AB_GROUPED_KS;
        if (grouped_bwd) {   // all encoders' attention backward of this step: one launch
            const long L0 = d->L[0], D0 = d->D[0];
            const float *a_p[RFN_MAX_ENC], *a_hp[RFN_MAX_ENC], *a_w[RFN_MAX_ENC], *a_al[RFN_MAX_ENC], *a_dz[RFN_MAX_ENC];
            float *a_dp[RFN_MAX_ENC], *a_dhp[RFN_MAX_ENC], *a_dw[RFN_MAX_ENC];
            for (int i = 0; i < M; ++i) {
                a_dp[i] = W + Lo.P1[i] + (long)t * B * L0 * A;
                a_p[i] = a_dp[i];
                a_hp[i] = hp + i * BA;
                a_w[i] = prm[P.s1(t, i, 4)];
                a_al[i] = W + Lo.al1[i] + (long)t * B * L0;
                a_dz[i] = W + Lo.dz1[i];
                a_dhp[i] = dhp + i * BA;
                a_dw[i] = dwp + ((long)t * M + i) * BA;
            }
            if (form0 == AB_GROUPED_KS) {   // dP1 of this step straight into the encoders' k-slow plane images
                void* a_img[RFN_MAX_ENC];
                for (int i = 0; i < M; ++i) a_img[i] = W + Lo.x3p[i];
                RFN_TRY(rfn_attn_bwd_grouped_ks(M, a_p, L0 * A, (long)A, a_hp, a_w, a_al, att, L0 * D0, D0, a_dz, D0, B,
                                                (int)L0, A, (int)D0, a_img, x3_row_pad(T1 * A), t * A, a_dhp, a_dw, st));
            } else {
                RFN_TRY(rfn_attn_bwd_grouped(M, a_p, L0 * A, (long)A, a_hp, a_w, a_al, att, L0 * D0, D0, a_dz, D0,
                                             B, (int)L0, A, (int)D0, a_dp, L0 * A, (long)A, 0, a_dhp, a_dw, st));
            }
        }
        const bool het_bwd = form0 == AB_HET;
        if (het_bwd) {   // maps of different (L, D): the M attention backwards of this step in one launch
            const float *a_p[RFN_MAX_ENC], *a_hp[RFN_MAX_ENC], *a_w[RFN_MAX_ENC], *a_al[RFN_MAX_ENC], *a_dz[RFN_MAX_ENC];
            float *a_dp[RFN_MAX_ENC], *a_dhp[RFN_MAX_ENC], *a_dw[RFN_MAX_ENC];
            int a_L[RFN_MAX_ENC], a_D[RFN_MAX_ENC];
            for (int i = 0; i < M; ++i) {
                a_dp[i] = W + Lo.P1[i] + (long)t * B * d->L[i] * A;
                a_p[i] = a_dp[i];
                a_hp[i] = hp + i * BA;
                a_w[i] = prm[P.s1(t, i, 4)];
                a_al[i] = W + Lo.al1[i] + (long)t * B * d->L[i];
                a_dz[i] = W + Lo.dz1[i];
                a_dhp[i] = dhp + i * BA;
                a_dw[i] = dwp + ((long)t * M + i) * BA;
                a_L[i] = d->L[i];
                a_D[i] = d->D[i];
            }
            RFN_TRY(rfn_attn_bwd_het(M, a_p, a_hp, a_w, a_al, att, a_dz, B, a_L, A, a_D, a_dp, 0, a_dhp, a_dw, st));
        }
        for (int i = 0; i < M; ++i) {
            const long Li = d->L[i], Di = d->D[i];
            float* dz = W + Lo.dz1[i];
            if (!dz_done)
                RFN_TRY(gemm1(B, (int)Di, seg_dx(g + (long)i * B * 4 * R, 4 * R, prm[P.s1(t, i, 8)], Di, 4 * R), dz, Di, 0, gx));
            float* dali = dal + (long)i * B * Li;
            float* p1 = W + Lo.P1[i] + (long)t * B * Li * A;
            const AttnBwdForm form = attn_bwd_form(d, B, i, dz_done);
            if (grouped_bwd || het_bwd) {
                // done above
            } else if (form == AB_FUSED_KS) {   // ... with dP1 as bf16 planes
                const float* p1c = p1;
                const float* hpc = hp + i * BA;
                const float* wc = prm[P.s1(t, i, 4)];
                const float* alc = W + Lo.al1[i] + (long)t * B * Li;
                const float* dzc = dz;
                void* img = W + Lo.x3p[i];
                float* dhpo = dhp + i * BA;
                float* dwo = dwp + ((long)t * M + i) * BA;
                RFN_TRY(rfn_attn_bwd_grouped_ks(1, &p1c, Li * A, (long)A, &hpc, &wc, &alc, &att[i], Li * Di, Di, &dzc, Di, B,
                                                (int)Li, A, (int)Di, &img, x3_row_pad(T1 * A), t * A, &dhpo, &dwo, st));
            } else if (form == AB_FUSED) {   // dalpha stays in LDS, one launch
                RFN_TRY(rfn_attn_bwd(p1, Li * A, (long)A, hp + i * BA, prm[P.s1(t, i, 4)],
                                     W + Lo.al1[i] + (long)t * B * Li, att[i], Li * Di, Di, dz, Di, B, (int)Li, A,
                                     (int)Di, p1, Li * A, (long)A, 0, dhp + i * BA,
                                     dwp + ((long)t * M + i) * BA, st));
            } else {
                RFN_TRY(rfn_attn_context_bwd_dalpha(att[i], Li * Di, Di, dz, Di, B, (int)Li, (int)Di, dali, st));
                RFN_TRY(rfn_attn_scores_bwd(p1, Li * A, (long)A, hp + i * BA, prm[P.s1(t, i, 4)],
                                            W + Lo.al1[i] + (long)t * B * Li, dali, B, (int)Li, A, p1, Li * A,
                                            (long)A, 0, dhp + i * BA, dwp + ((long)t * M + i) * BA, st));
            }
            pr[i] = prob1(dHc + i * R, MR, seg_dx(dhp + i * BA, A, prm[P.s1(t, i, 2)], R, A));
        }
        if (s1_small) {
            rfn_cell_out ky[RFN_MAX_ENC];
            s1y_of(t, ky);
            RFN_TRY(cell_run(B, M, ky, R, d->drop_fusion, seed, st, cell_variant(d)));
        } else {
            rfn_cell_out kb[RFN_MAX_ENC];
            for (int i = 0; i < M; ++i) {
                kb[i] = cell_out(dHc + i * R, MR, R, 1);
                cell_dx(kb[i], dhp + i * BA, A, prm[P.s1(t, i, 2)], R, A);
            }
            if (cell_ok(B, M, kb, R)) RFN_TRY(cell_run(B, M, kb, R, 0.f, 0, st, cell_variant(d)));
            else RFN_TRY(gemm_groups(B, R, M, pr, 1, gx));
        }
    }
    // c0 = h0.clone() (:206): dh0 += dc0 ; fc2h gradients
    RFN_TRY(rfn_axpby_2d(1.f, dC, MR, 1.f, dHs, MR, B, (int)MR, st));
    for (int i = 0; i < M; ++i)
        RFN_TRY(gemm_dw(R, d->F[i], grd[P.fc_w(i)], d->F[i], grd[P.fc_b(i)], dHs + i * R, MR, fc[i], d->F[i], B, gx));
    {
        float* outs[64];
        if (T1 * M > 64) return RFN_ERR_SHAPE;
        for (int t = 0; t < T1; ++t)
            for (int i = 0; i < M; ++i) outs[t * M + i] = grd[P.s1(t, i, 4)];
        RFN_TRY(rfn_colsum_grouped_f32(dwp, BA, A, B, A, outs, T1 * M, st));
        for (int t = 0; t < T1; ++t)
            for (int i = 0; i < M; ++i) outs[t * M + i] = grd[P.s1(t, i, 5)];
        RFN_TRY(rfn_fill_small_f32(outs, T1 * M, 1, 0.f, st));
    }
    // weight gradients of stage I (per encoder; see rfn_prefix_bwd_wgrad) unless the caller defers them
    if (!defer_wgrad)   // the short part-A products of every encoder first, then the long att_2_att_h products back to back
        for (int part = 1; part <= 2; ++part)
            for (int i = 0; i < M; ++i) RFN_TRY(rfn_prefix_bwd_wgrad(d, B, att, grd, ws, ws_bytes, i, part, st));
    return RFN_OK;
}

// Stage-I weight gradients of ONE encoder, grouped over the T1 steps (bias gradients ride along):
// H2h, z2h, h_2_att_h and the dominant d att_2_att_h.weight[t,i] = dP1_i[:, t]^T . att_i (K = B*L_i).
// Reads only the workspace rfn_prefix_bwd left behind, so a data-parallel host can all-reduce encoder i's
// gradient bucket while encoder i+1's GEMMs run.
extern "C" int rfn_prefix_bwd_wgrad(const rfn_dims* d, int B, const float* const* att, float* const* grd, void* ws,
                                    size_t ws_bytes, int enc, int parts, void* st) {
    RFN_TRY(check_dims(d));
    if (B < 1 || enc < 0 || enc >= d->M || (parts & ~3) || !parts) return RFN_ERR_SHAPE;
    if (!att || !grd || !ws) return RFN_ERR_ARG;
    const PrefixLayout Lo = prefix_layout(d, B, 1);
    if (ws_bytes < Lo.total * sizeof(float)) return RFN_ERR_WORKSPACE;
    const PIdx P(d);
    const int M = d->M, R = d->R, A = d->A, T1 = d->T1, i = enc;
    if (T1 > 64) return RFN_ERR_SHAPE;
    const long MR = (long)M * R, BMR = (long)B * MR, BA = (long)B * A;
    float* W = (float*)ws;
    const bool tk_on = (d->gemm_flags & RFN_GEMM_OPT_SPLITK_IN_KERNEL) != 0;
    const GemmCtx gx{st, W + Lo.gws, GEMM_WS_FLOATS * sizeof(float), d->gemm_flags, tk_on ? (int32_t*)(W + Lo.tk) : nullptr,
                     tk_on ? GEMM_TICKETS : 0};
    const float* Hs = W + Lo.Hs;
    rfn_gemm_problem pr[64];
    const long Li = d->L[i], Di = d->D[i];
    // att_2_att_h.bias and h_2_att_h.bias enter the same pre-activation (AttentionModelCore.py:36-38), so their
    // gradients are the same vector: the column-sum launch of part A writes it to both; the long att_2_att_h GEMM
    // carries no bias-gradient rider.
    auto part_b = [&]() -> int {   // the dominant att_2_att_h gradient (small bucket, long GEMM)
        if (x3_takes(d, B, i)) {
            // bf16-plane GEMM: dW[t] = dP1[t]^T . att.  Both operands are reduction-index-major in memory ((b,l) rows), so
            // they are kept that way as k-slow plane images and rfn_x3_gemm_ks transposes while reading LDS: no
            // transposing pass.  dP1's image was written by the attention backward itself when B >= 96 (fused kernel);
            // for small batches it is split from the f32 slabs here.
            const int BL = (int)(B * Li), TA = T1 * A;
            char* ksX = (char*)(W + Lo.x3);
            float* part = (float*)(ksX + rfn_x3_image_bytes((int)Di, BL));
            char* ksP = (char*)(W + Lo.x3p[i]);
            const int sk = rfn_x3_splitk_for(TA, (int)Di, BL);
            const float* srcs[64];
            float* outs[64];
            srcs[0] = att[i];
            RFN_TRY(rfn_x3_split_ks(srcs, 1, Di, BL, (int)Di, ksX, st));
            for (int t = 0; t < T1; ++t) {
                srcs[t] = W + Lo.P1[i] + (long)t * BL * A;
                outs[t] = grd[P.s1(t, i, 0)];
            }
            if (!x3_dp_emitted(d, B, i)) {
                RFN_TRY(rfn_x3_split_ks(srcs, T1, A, BL, A, ksP, st));
            } else if (BL % 32) {   // the attention kernels wrote the B*L real rows; the GEMM also reads the pad rows
                const long row_bytes = 3L * x3_row_pad(TA) * 2;
                if (hipMemsetAsync(ksP + BL * row_bytes, 0, (size_t)((BL + 31) / 32 * 32 - BL) * row_bytes, (hipStream_t)st) !=
                    hipSuccess)
                    return RFN_ERR_LAUNCH;
            }
            RFN_TRY(probe_mark(d, 2 * M + 2 * i, st));
            RFN_TRY(rfn_x3_gemm_ks(TA, (int)Di, BL, ksP, ksX, A, (int)Di, outs, nullptr, Di, 0, sk, part, st));
            return probe_mark(d, 2 * M + 2 * i + 1, st);
        }
        for (int t = 0; t < T1; ++t)
            pr[t] = prob_dw(grd[P.s1(t, i, 0)], Di, nullptr, W + Lo.P1[i] + (long)t * B * Li * A, A, att[i], Di,
                            (int)(B * Li));
        RFN_TRY(probe_mark(d, 2 * M + 2 * i, st));
        RFN_TRY(gemm_groups_split_cols(A, (int)Di, T1, pr, 0, gx));
        return probe_mark(d, 2 * M + 2 * i + 1, st);
    };
    if ((parts & 2) && !(parts & 1)) return part_b();
    // part A: H2h, z2h, h_2_att_h (large bucket, short GEMMs: K = B rows per step).  Their bias gradients are column
    // sums of tensors that are tiny next to the weight gradients (dgates: T1*B*4R floats per encoder), so they come from
    // one grouped column-sum launch each instead of riding on the GEMMs -- which keeps the two big products
    // (2 x T1 x 4R x {M*R, D}) on the LDS-DMA kernel.  H2h.bias and z2h.bias enter the same pre-activation and share
    // one gradient, like h_2_att_h.bias and att_2_att_h.bias.
    float* outs[64];
    const float* g1i = W + Lo.g1 + (long)i * B * 4 * R;           // (t, i) slab = g1i + t * M*B*4R
    float* outs2[64];
    for (int t = 0; t < T1; ++t) {
        outs[t] = grd[P.s1(t, i, 7)];
        outs2[t] = grd[P.s1(t, i, 9)];     // z2h.bias = H2h.bias gradient
    }
    RFN_TRY(rfn_colsum_grouped2_f32(g1i, (long)M * B * 4 * R, 4 * R, B, 4 * R, outs, outs2, T1, st));
    for (int t = 0; t < T1; ++t) {
        outs[t] = grd[P.s1(t, i, 3)];
        outs2[t] = grd[P.s1(t, i, 1)];     // att_2_att_h.bias = h_2_att_h.bias gradient
    }
    RFN_TRY(rfn_colsum_grouped2_f32(W + Lo.dhp1 + (long)i * BA, (long)M * BA, A, B, A, outs, outs2, T1, st));
    for (int t = 0; t < T1; ++t)
        pr[t] = prob_dw(grd[P.s1(t, i, 6)], MR, nullptr, g1i + (long)t * M * B * 4 * R, 4 * R, Hs + t * BMR, MR, B);
    RFN_TRY(gemm_groups(4 * R, (int)MR, T1, pr, 0, gx));
    for (int t = 0; t < T1; ++t)
        pr[t] = prob_dw(grd[P.s1(t, i, 8)], Di, nullptr, g1i + (long)t * M * B * 4 * R, 4 * R,
                        W + Lo.z1[i] + (long)t * B * Di, Di, B);
    RFN_TRY(gemm_groups_split_cols(4 * R, (int)Di, T1, pr, 0, gx));
    for (int t = 0; t < T1; ++t)
        pr[t] = prob_dw(grd[P.s1(t, i, 2)], R, nullptr, W + Lo.dhp1 + ((long)t * M + i) * BA, A, Hs + t * BMR + i * R, MR, B);
    RFN_TRY(gemm_groups(A, R, T1, pr, 0, gx));
    if (parts & 2) RFN_TRY(part_b());
    return RFN_OK;
}

// =============================================================================================
// phase 2: teacher-forced decoder
// =============================================================================================
// single-encoder forms of the fused small-L attention (decoder: one attention over the T2 fused thoughts)
static int attn1_fwd(const float* proj, long psb, long psl, const float* hp, const float* w, const float* bo,
                     const float* x, long sb, long sl, int B, int L, int A, int D, float* al, float* z, long ldz,
                     void* st) {
    return rfn_attn_small_fwd(1, &proj, psb, psl, &hp, &w, &bo, &x, sb, sl, B, L, A, D, &al, &z, ldz, st);
}
static int attn1_bwd(const float* proj, long psb, long psl, const float* hp, const float* w, const float* al,
                     const float* x, long sb, long sl, const float* dz, long lddz, int B, int L, int A, int D,
                     float* dproj, long dpsb, long dpsl, int acc, float* dhp, float* dwp, float* dx, void* st) {
    return rfn_attn_small_bwd(1, &proj, psb, psl, &hp, &w, &al, &x, sb, sl, &dz, lddz, B, L, A, D, &dproj, dpsb, dpsl,
                              acc, &dhp, &dwp, &dx, st);
}

extern "C" size_t rfn_decoder_ws_bytes(const rfn_dims* d, int B, int S, int train) {
    if (check_dims(d) != RFN_OK || B < 1 || S < 1) return 0;
    return decoder_layout(d, B, S, train).total * sizeof(float);
}

// One decoder cell call on given buffers (g already holds i2h(x)): h_2_att_h(h), attention over the fused thoughts,
// h2h(h) + z2h(z) accumulated onto the gates, LSTM update with the dropout mask of (seed, step).  Shared by the batched,
// the step-wise and the free-running pass, so the three are bit-identical by construction.
// Fused form (3 launches): K1 = h_2_att_h(h) and g += h2h(h) in one launch (they share h); the attention; K3 =
// g += z2h(z) with the LSTM update as its epilogue.  h_next / c_next may alias h / c (free-running step).
// The fused form of that step (3 launches), prepared: K1 = h_2_att_h(h) and g += h2h(h) in one launch (they share h); the
// attention; K3 = g += z2h(z) with the LSTM update as its epilogue.  false: the cell GEMM cannot take the step (maxout, widths).
static bool decoder_cell_prepare(const rfn_dims* d, int B, const float* const* prm, const float* comb, const float* cproj,
                                 const float* h, const float* c, float* h_next, float* c_next, float* hp, float* al, float* z,
                                 float* g, uint64_t seed, int step, ChainStep* cs) {
    const PIdx P(d);
    const int R = d->R, A = d->A, T2 = d->T2;
    const int GD = gate_width(d->decoder_maxout, R);
    const long BR = (long)B * R, BA = (long)B * A;
    if (d->decoder_maxout) return false;
    rfn_cell_out k1[2], k3;
    k1[0] = cell_out(hp, A, A, 0);
    cell_lin(k1[0], h, R, prm[P.dec(8)], R, R, prm[P.dec(9)]);
    k1[1] = cell_out(g, GD, GD, 1);
    cell_lin(k1[1], h, R, prm[P.dec(2)], R, R, prm[P.dec(3)]);
    k3 = cell_out(g, GD, GD, 1);
    cell_lin(k3, z, R, prm[P.dec(4)], R, R, prm[P.dec(5)]);
    cell_lstm(k3, c, R, c_next, R, h_next, R, OFF_DECODER + (uint64_t)step);
    if (!cell_ok(B, 2, k1, R) || !cell_ok(B, 1, &k3, R)) return false;
    const float *w = prm[P.dec(10)], *bo = prm[P.dec(11)];
    return cell_prepare(B, 2, k1, R, 0.f, 0, &cs->g0, cell_variant(d)) == RFN_OK &&
           rfn_attn_small_prepare_fwd(1, &cproj, A, BA, &hp, &w, &bo, &comb, R, BR, B, T2, A, R, &al, &z, R, &cs->at) == RFN_OK &&
           cell_prepare(B, 1, &k3, R, d->drop_lm, seed, &cs->g2, cell_variant(d)) == RFN_OK;
}

// Hoisted form (default, rfn_deccell.hip; 2 launches): K1 as above, then ONE per-row launch = scores, softmax,
// gates += b_z + sum_l alpha_l U_l, LSTM update.  `U` = comb . W_z^T (Bc = B / row_div rows per thought vector; the rows
// b of one beam-search image share row b / row_div of cproj / U).  z is not computed.
static int decoder_cell_core(const rfn_dims* d, int B, const float* const* prm, const float* comb, const float* cproj,
                             const float* U, int row_div, const float* h, const float* c, float* h_next, float* c_next,
                             float* hp, float* al, float* z, float* g, const GemmCtx& gx, uint64_t seed, int step, void* st) {
    const PIdx P(d);
    const int R = d->R, A = d->A, T2 = d->T2;
    const int GD = gate_width(d->decoder_maxout, R);
    const long BR = (long)B * R, BA = (long)B * A;
    if (dec_hoisted(d)) {
        const int Bc = B / row_div;
        rfn_cell_out k1[2];
        k1[0] = cell_out(hp, A, A, 0);
        cell_lin(k1[0], h, R, prm[P.dec(8)], R, R, prm[P.dec(9)]);
        k1[1] = cell_out(g, GD, GD, 1);
        cell_lin(k1[1], h, R, prm[P.dec(2)], R, R, prm[P.dec(3)]);
        if (cell_ok(B, 2, k1, R)) {
            RFN_TRY(cell_run(B, 2, k1, R, 0.f, 0, st, cell_variant(d)));
        } else {
            RFN_TRY(gemm1(B, A, seg_lin(h, R, prm[P.dec(8)], R, R, prm[P.dec(9)]), hp, A, 0, gx));
            RFN_TRY(gemm1(B, GD, seg_lin(h, R, prm[P.dec(2)], R, R, prm[P.dec(3)]), g, GD, 1, gx));
        }
        return rfn_dec_cell_fwd(cproj, A, (long)Bc * A, hp, prm[P.dec(10)], prm[P.dec(11)], U, GD, (long)Bc * GD, prm[P.dec(5)], g,
                                GD, c, R, c_next, R, h_next, R, al, B, T2, A, R, d->decoder_maxout, row_div, d->drop_lm, seed,
                                OFF_DECODER + (uint64_t)step, st);
    }
    if (row_div != 1) return RFN_ERR_UNSUPPORTED;   // the three-launch form reads comb per row
    {
        ChainStep cs;
        if (decoder_cell_prepare(d, B, prm, comb, cproj, h, c, h_next, c_next, hp, al, z, g, seed, step, &cs))
            return rfn_chain_run(&cs, 1, 0, nullptr, st);   // one step: its three launches
    }
    RFN_TRY(gemm1(B, A, seg_lin(h, R, prm[P.dec(8)], R, R, prm[P.dec(9)]), hp, A, 0, gx));
    RFN_TRY(attn1_fwd(cproj, A, BA, hp, prm[P.dec(10)], prm[P.dec(11)], comb, R, BR, B, T2, A, R, al, z, R, st));
    rfn_gemm_seg segs[2];
    segs[0] = seg_lin(h, R, prm[P.dec(2)], R, R, prm[P.dec(3)]);
    segs[1] = seg_lin(z, R, prm[P.dec(4)], R, R, prm[P.dec(5)]);
    RFN_TRY(gemm_segs(B, GD, 2, segs, g, GD, 1, gx));
    return rfn_lstm_fwd(g, GD, c, R, c_next, R, h_next, R, B, R, d->decoder_maxout, d->drop_lm, seed,
                        OFF_DECODER + (uint64_t)step, st);
}

// The decoder cell of step s on the training workspace (gd[s] already holds i2h(x_s)): h_2_att_h, attention over the
// fused thoughts, h2h + z2h accumulated onto the gates, LSTM epilogue with the dropout mask of (seed, s).
static int decoder_fwd_cell(const rfn_dims* d, int B, int s, const float* const* prm, const float* comb, float* W,
                            const DecoderLayout& Lo, const GemmCtx& gx, uint64_t seed, void* st) {
    const PIdx P(d);
    const int R = d->R, A = d->A, T2 = d->T2;
    const int GD = gate_width(d->decoder_maxout, R);
    const long BR = (long)B * R, BA = (long)B * A;
    float* hd = W + Lo.hd;
    float* cd = W + Lo.cd;
    float* hc = hd + s * BR;
    float* hp = W + Lo.hpd + s * BA;
    float* al = W + Lo.ald + (long)s * B * T2;
    float* z = W + Lo.zd + s * BR;
    float* g = W + Lo.gd + (long)s * B * GD;
    return decoder_cell_core(d, B, prm, comb, W + Lo.Pd, W + Lo.Ud, 1, hc, cd + s * BR, hd + (s + 1) * BR, cd + (s + 1) * BR, hp,
                             al, z, g, gx, seed, s, st);
}

static bool decoder_fwd_cell_prepare(const rfn_dims* d, int B, int s, const float* const* prm, const float* comb, float* W,
                                     const DecoderLayout& Lo, uint64_t seed, ChainStep* cs) {
    const int R = d->R, A = d->A, T2 = d->T2;
    const int GD = gate_width(d->decoder_maxout, R);
    const long BR = (long)B * R, BA = (long)B * A;
    float* hd = W + Lo.hd;
    float* cd = W + Lo.cd;
    return decoder_cell_prepare(d, B, prm, comb, W + Lo.Pd, hd + s * BR, cd + s * BR, hd + (s + 1) * BR, cd + (s + 1) * BR,
                                W + Lo.hpd + s * BA, W + Lo.ald + (long)s * B * T2, W + Lo.zd + s * BR,
                                W + Lo.gd + (long)s * B * GD, seed, s, cs);
}

// Loop-invariant part of phase 2: projection of the fused thoughts (applied once instead of every step) and the
// initial state.  The batched products of the pass (this projection, i2h, logits) are never split along K: the
// free-running step (rfn_decoder_prepare / rfn_decoder_step) and the step-wise training pass (rfn_decoder_fwd_step)
// compute the same products for one step's rows with the same unsplit k order, so log-probs of the same tokens are
// bit-identical across all three whatever the batch size (split-K choices depend on the row count).  The per-step
// products (h_2_att_h, h2h + z2h) have the same shape everywhere and take the same split.
static int decoder_fwd_begin(const rfn_dims* d, int B, const float* const* prm, const float* comb, const float* h0,
                             const float* c0, float* W, const DecoderLayout& Lo, void* st) {
    const PIdx P(d);
    const int R = d->R, A = d->A, T2 = d->T2;
    const GemmCtx gx_whole{st, nullptr, 0, d->gemm_flags};
    RFN_TRY(gemm1(T2 * B, A, seg_lin(comb, R, prm[P.dec(6)], R, R, prm[P.dec(7)]), W + Lo.Pd, A, 0, gx_whole));
    if (dec_hoisted(d)) {   // U = comb . W_z^T, no bias: z2h of every thought vector, once (misc/LSTMSoftAttentionCore.py:78-81)
        const int GD = gate_width(d->decoder_maxout, R);
        RFN_TRY(gemm1(T2 * B, GD, seg_lin(comb, R, prm[P.dec(4)], R, R, nullptr), W + Lo.Ud, GD, 0, gx_whole));
    }
    return mem_batch({{W + Lo.hd, h0, (long)B * R}, {W + Lo.cd, c0, (long)B * R}}, st);
}

extern "C" int rfn_decoder_fwd(const rfn_dims* d, int B, int S, const float* const* prm, const float* comb,
                               const float* h0, const float* c0, const int64_t* ids, int64_t ld_ids, float* log_prob,
                               void* ws, size_t ws_bytes, int train, uint64_t seed, void* st) {
    RFN_TRY(check_dims(d));
    if (B < 1 || S < 1) return RFN_ERR_SHAPE;
    if (!prm || !comb || !h0 || !c0 || !ids || !ws) return RFN_ERR_ARG;
    const DecoderLayout Lo = decoder_layout(d, B, S, train);
    if (ws_bytes < Lo.total * sizeof(float)) return RFN_ERR_WORKSPACE;
    const PIdx P(d);
    const int R = d->R, E = d->E, V1 = d->V1;
    const int GD = gate_width(d->decoder_maxout, R);
    float* W = (float*)ws;
    const GemmCtx gx{st, W + Lo.gws, GEMM_WS_FLOATS * sizeof(float), d->gemm_flags};
    const GemmCtx gx_whole{st, nullptr, 0, d->gemm_flags};
    RFN_TRY(decoder_fwd_begin(d, B, prm, comb, h0, c0, W, Lo, st));
    // all token embeddings and their i2h projections in one go (teacher forcing: ids are known)
    RFN_TRY(rfn_embed_fwd(prm[P.embed()], E, V1, ids, B, ld_ids, 1, S * B, W + Lo.xs, E, st));
    RFN_TRY(gemm1(S * B, GD, seg_lin(W + Lo.xs, E, prm[P.dec(0)], E, E, prm[P.dec(1)]), W + Lo.gd, GD, 0, gx_whole));
    if (dec_hoisted(d)) {   // the S cell steps, two launches each (rfn_deccell.hip)
        for (int s = 0; s < S; ++s) RFN_TRY(decoder_fwd_cell(d, B, s, prm, comb, W, Lo, gx, seed, st));
    } else {   // three-launch form: one persistent launch (rfn_chain.hip) when every step takes the fused form, else step by step
        std::vector<ChainStep> steps((size_t)S);
        bool fused = true;
        for (int s = 0; s < S && fused; ++s) fused = decoder_fwd_cell_prepare(d, B, s, prm, comb, W, Lo, seed, &steps[s]);
        if (fused) {
            RFN_TRY(rfn_chain_run(steps.data(), S, chain_persist(d, RFN_PATH_OPT_PERSIST_DEC_FWD), (uint32_t*)(W + Lo.bar), st));
        } else {
            for (int s = 0; s < S; ++s) RFN_TRY(decoder_fwd_cell(d, B, s, prm, comb, W, Lo, gx, seed, st));
        }
    }
    // logits of all steps, then log-softmax written in the reference's (B, S, V+1) layout
    RFN_TRY(gemm_logits(S * B, V1, W + Lo.hd + (long)B * R, R, prm[P.logit_w()], prm[P.logit_b()], W + Lo.logits, gx_whole));
    if (log_prob) RFN_TRY(rfn_log_softmax_fwd(W + Lo.logits, V1, S * B, V1, B, (long)S * V1, V1, log_prob, st));
    return RFN_OK;      // log_prob == NULL: the logits stay in the workspace for rfn_xe_logits_fwd (rfn_decoder_logits)
}

extern "C" float* rfn_decoder_logits(const rfn_dims* d, int B, int S, int train, void* ws) {
    if (check_dims(d) != RFN_OK || B < 1 || S < 1 || !ws) return nullptr;
    return (float*)ws + decoder_layout(d, B, S, train).logits;
}

// Step-wise form of the same pass for scheduled sampling (misc/RecurrentFusionModel.py:260-270): the token fed at
// step s may be drawn from the distribution of step s-1, so the host interleaves its draws with the steps.  begin +
// S steps leave the workspace and log_prob exactly as rfn_decoder_fwd on the final ids does (bit for bit), so
// rfn_decoder_bwd runs on it unchanged -- the sampled pass IS the differentiated pass, nothing is computed twice.
extern "C" int rfn_decoder_fwd_begin(const rfn_dims* d, int B, int S, const float* const* prm, const float* comb,
                                     const float* h0, const float* c0, void* ws, size_t ws_bytes, int train, void* st) {
    RFN_TRY(check_dims(d));
    if (B < 1 || S < 1) return RFN_ERR_SHAPE;
    if (!prm || !comb || !h0 || !c0 || !ws) return RFN_ERR_ARG;
    const DecoderLayout Lo = decoder_layout(d, B, S, train);
    if (ws_bytes < Lo.total * sizeof(float)) return RFN_ERR_WORKSPACE;
    return decoder_fwd_begin(d, B, prm, comb, h0, c0, (float*)ws, Lo, st);
}

extern "C" int rfn_decoder_fwd_step(const rfn_dims* d, int B, int S, int s, const float* const* prm, const float* comb,
                                    const int64_t* ids_s, int64_t ld_ids, float* log_prob, void* ws, size_t ws_bytes,
                                    int train, uint64_t seed, void* st) {
    RFN_TRY(check_dims(d));
    if (B < 1 || S < 1 || s < 0 || s >= S) return RFN_ERR_SHAPE;
    if (!prm || !comb || !ids_s || !log_prob || !ws) return RFN_ERR_ARG;
    const DecoderLayout Lo = decoder_layout(d, B, S, train);
    if (ws_bytes < Lo.total * sizeof(float)) return RFN_ERR_WORKSPACE;
    const PIdx P(d);
    const int R = d->R, E = d->E, V1 = d->V1;
    const int GD = gate_width(d->decoder_maxout, R);
    float* W = (float*)ws;
    const GemmCtx gx{st, W + Lo.gws, GEMM_WS_FLOATS * sizeof(float), d->gemm_flags};
    const GemmCtx gx_whole{st, nullptr, 0, d->gemm_flags};
    float* xs = W + Lo.xs + (long)s * B * E;
    float* lg = W + Lo.logits + (long)s * B * V1;
    RFN_TRY(rfn_embed_fwd(prm[P.embed()], E, V1, ids_s, B, ld_ids, 1, B, xs, E, st));
    RFN_TRY(gemm1(B, GD, seg_lin(xs, E, prm[P.dec(0)], E, E, prm[P.dec(1)]), W + Lo.gd + (long)s * B * GD, GD, 0, gx_whole));
    RFN_TRY(decoder_fwd_cell(d, B, s, prm, comb, W, Lo, gx, seed, st));
    RFN_TRY(gemm_logits(B, V1, W + Lo.hd + (long)(s + 1) * B * R, R, prm[P.logit_w()], prm[P.logit_b()], lg, gx_whole));
    return rfn_log_softmax_fwd(lg, V1, B, V1, B, (long)S * V1, V1, log_prob + (long)s * V1, st);
}

extern "C" int rfn_decoder_bwd(const rfn_dims* d, int B, int S, const float* const* prm, const float* comb,
                               const float* h0, const float* c0, const int64_t* ids, int64_t ld_ids,
                               const float* log_prob, const float* d_log_prob, float* d_comb, float* d_h0,
                               float* d_c0, float* const* grd, void* ws, size_t ws_bytes, uint64_t seed, void* st) {
    RFN_TRY(check_dims(d));
    if (B < 1 || S < 1) return RFN_ERR_SHAPE;
    if (!prm || !comb || !ids || (!log_prob != !d_log_prob) || !d_comb || !d_h0 || !d_c0 || !grd || !ws)
        return RFN_ERR_ARG;
    (void)h0; (void)c0;
    const DecoderLayout Lo = decoder_layout(d, B, S, 1);
    if (ws_bytes < Lo.total * sizeof(float)) return RFN_ERR_WORKSPACE;
    const PIdx P(d);
    const int R = d->R, A = d->A, E = d->E, T2 = d->T2, V1 = d->V1;
    const int GD = gate_width(d->decoder_maxout, R);
    const long BR = (long)B * R, BA = (long)B * A;
    float* W = (float*)ws;
    const bool tk_on = (d->gemm_flags & RFN_GEMM_OPT_SPLITK_IN_KERNEL) != 0;   // measured slower: off unless asked for (rfn.h)
    const GemmCtx gx{st, W + Lo.gws, GEMM_WS_FLOATS * sizeof(float), d->gemm_flags, tk_on ? (int32_t*)(W + Lo.tk) : nullptr,
                     tk_on ? GEMM_TICKETS : 0};
    if (tk_on) RFN_TRY(zero_f32(W + Lo.tk, GEMM_TICKETS, st));   // tile counters: zero on entry, every launch leaves them zero
    float* hd = W + Lo.hd;
    float* cd = W + Lo.cd;
    float* gd = W + Lo.gd;
    float* dlg = W + Lo.logits;
    float* dhe = W + Lo.dhe;
    float* dhrec = W + Lo.dhrec;
    float* dc = W + Lo.dc;
    float* dz = W + Lo.dz;
    float* dPd = W + Lo.dPd;
    // log-softmax backward into time-major rows, then the batched logit layer
    // (no log_prob: rfn_xe_logits_bwd already turned the logits rows into d logits)
    if (d_log_prob) RFN_TRY(rfn_log_softmax_bwd(d_log_prob, log_prob, S * B, V1, B, (long)S * V1, V1, dlg, V1, st));
    RFN_TRY(gemm_logits_dw(V1, R, grd[P.logit_w()], grd[P.logit_b()], dlg, hd + BR, S * B, gx));
    RFN_TRY(gemm_logits_dx(S * B, R, V1, dlg, prm[P.logit_w()], dhe, gx));
    if (dec_hoisted(d)) {
        // ---- z2h hoisted (rfn_deccell.hip): per step the attention backward from that step's gate gradients, then ONE product
        // d h = [d gates_s | d hproj_s] . [W_hh ; W_h] whose epilogue finishes d h of step s-1 (+ the logit layer's share) and
        // runs that step's LSTM backward.  d thoughts / d W_z / d att_2_att_h follow after the loop from dU and dPd.
        const float* Ud = W + Lo.Ud;
        float* dUd = W + Lo.dUd;
        RFN_TRY(mem_batch({{dPd, nullptr, (long)T2 * BA}}, st));
        // The fused form of a step, two launches: X = the step's attention-backward rows BESIDE the tiles of d gates_s . W_hh cut
        // DEC_KSPLIT ways along K into partial slabs (neither depends on the other; one grid, rfn_cg_launch_with_rows), then
        // Y = d hproj_s . W_h + the slabs, whose epilogue finishes d h of step s-1 and runs that step's LSTM backward.
        auto kx_of = [&](int s, rfn_cell_out* kx) {
            const int Kp = GD / DEC_KSPLIT;
            for (int j = 0; j < DEC_KSPLIT; ++j) {
                kx[j] = cell_out(dhrec + (long)j * BR, R, R, 0);
                cell_dx(kx[j], gd + (long)s * B * GD + (long)j * Kp, GD, prm[P.dec(2)] + (long)j * Kp * R, R, Kp);
            }
        };
        auto ky_of = [&](int s, rfn_cell_out& ky) {
            ky = cell_out(dhrec, R, R, 1);        // C = slab 0 (+ the other DEC_KSPLIT - 1 slabs behind it)
            ky.acc_slabs = dhrec + BR;
            ky.acc_parts = DEC_KSPLIT - 1;
            ky.acc_stride = BR;
            cell_dx(ky, W + Lo.dhpd + s * BA, A, prm[P.dec(8)], R, A);
            if (s > 0)
                cell_lstm_bwd(ky, gd + (long)(s - 1) * B * GD, GD, cd + (s - 1) * BR, R, cd + s * BR, R, dhe + (s - 1) * BR, R,
                              dc, R, dc, R, OFF_DECODER + (uint64_t)(s - 1));
        };
        bool fusedh = !d->decoder_maxout && GD % (DEC_KSPLIT * 32) == 0;
        for (int s = 0; s < S && fusedh; ++s) {
            rfn_cell_out tx[DEC_KSPLIT], ty;
            kx_of(s, tx);
            ky_of(s, ty);
            fusedh = cell_ok(B, DEC_KSPLIT, tx, R) && cell_ok(B, 1, &ty, R);
        }
        if (fusedh)   // LSTM backward of the last step: nothing recurrent flows into it
            RFN_TRY(rfn_lstm_bwd(gd + (long)(S - 1) * B * GD, GD, cd + (S - 1) * BR, R, cd + S * BR, R, dhe + (S - 1) * BR, R, nullptr,
                                 R, dc, R, B, R, 0, d->drop_lm, seed, OFF_DECODER + (uint64_t)(S - 1), st));
        for (int s = S - 1; s >= 0; --s) {
            float* g = gd + (long)s * B * GD;
            float* dhp = W + Lo.dhpd + s * BA;
            if (!fusedh) {
                float* dht = dhe + s * BR;
                if (s < S - 1) RFN_TRY(rfn_axpby_2d(1.f, dhrec, R, 1.f, dht, R, B, R, st));
                RFN_TRY(rfn_lstm_bwd(g, GD, cd + s * BR, R, cd + (s + 1) * BR, R, dht, R, (s < S - 1) ? dc : nullptr, R, dc, R, B, R,
                                     d->decoder_maxout, d->drop_lm, seed, OFF_DECODER + (uint64_t)s, st));
                RFN_TRY(rfn_dec_attn_bwd(W + Lo.Pd, A, BA, W + Lo.hpd + s * BA, prm[P.dec(10)], W + Lo.ald + (long)s * B * T2, Ud, GD,
                                         (long)B * GD, g, GD, B, T2, A, GD, dPd, A, BA, 1, dhp, W + Lo.dwp + s * BA, st));
                rfn_gemm_seg sg[2] = {seg_dx(g, GD, prm[P.dec(2)], R, GD), seg_dx(dhp, A, prm[P.dec(8)], R, A)};
                RFN_TRY(gemm_segs(B, R, 2, sg, dhrec, R, 0, gx));
                continue;
            }
            rfn_cell_out kx[DEC_KSPLIT], ky;
            kx_of(s, kx);
            CgPrepared px;
            DecAttnBwdArgs da;
            RFN_TRY(cell_prepare(B, DEC_KSPLIT, kx, R, 0.f, 0, &px, cell_variant(d)));
            RFN_TRY(rfn_dec_attn_bwd_args(W + Lo.Pd, A, BA, W + Lo.hpd + s * BA, prm[P.dec(10)], W + Lo.ald + (long)s * B * T2, Ud, GD,
                                          (long)B * GD, g, GD, B, T2, A, GD, dPd, A, BA, 1, dhp, W + Lo.dwp + s * BA, &da));
            const int rc = rfn_cg_launch_with_rows(px, da, B, st);
            if (rc == RFN_ERR_UNSUPPORTED) {   // shapes the fused grid does not take: the same two bodies as two launches
                RFN_TRY(rfn_dec_attn_bwd(W + Lo.Pd, A, BA, W + Lo.hpd + s * BA, prm[P.dec(10)], W + Lo.ald + (long)s * B * T2, Ud, GD,
                                         (long)B * GD, g, GD, B, T2, A, GD, dPd, A, BA, 1, dhp, W + Lo.dwp + s * BA, st));
                RFN_TRY(rfn_cg_launch(px, st));
            } else {
                RFN_TRY(rc);
            }
            ky_of(s, ky);
            RFN_TRY(cell_run(B, 1, &ky, R, d->drop_lm, seed, st, cell_variant(d)));
        }
        RFN_TRY(mem_batch({{d_h0, dhrec, BR}, {d_c0, dc, BR}, {grd[P.dec(11)], nullptr, 1}}, st));
        // d U = sum_s alpha_s (x) d gates_s; d thoughts = dPd . W_att + dU . W_z (one product, two K segments)
        RFN_TRY(rfn_dec_du(W + Lo.ald, gd, S, B, T2, GD, dUd, GD, (long)B * GD, st));
        {
            rfn_gemm_seg sg[2] = {seg_dx(dPd, A, prm[P.dec(6)], R, A), seg_dx(dUd, GD, prm[P.dec(4)], R, GD)};
            RFN_TRY(gemm_segs(T2 * B, R, 2, sg, d_comb, R, 0, gx));
        }
        RFN_TRY(gemm_dw(A, R, grd[P.dec(6)], R, grd[P.dec(7)], dPd, A, comb, R, T2 * B, gx));
        RFN_TRY(gemm_dw(GD, R, grd[P.dec(4)], R, nullptr, dUd, GD, comb, R, T2 * B, gx));   // d z2h.weight = dU^T . thoughts
        // weights shared across steps: one GEMM over (S*B) time-major rows each, bias gradients ride along
        RFN_TRY(rfn_colsum_f32(W + Lo.dwp, A, S * B, A, grd[P.dec(10)], 0, st));
        RFN_TRY(gemm_dw(A, R, grd[P.dec(8)], R, grd[P.dec(9)], W + Lo.dhpd, A, hd, R, S * B, gx));
        RFN_TRY(gemm_dw(GD, R, grd[P.dec(2)], R, grd[P.dec(3)], gd, GD, hd, R, S * B, gx));
        RFN_TRY(gemm_dw(GD, E, grd[P.dec(0)], E, grd[P.dec(1)], gd, GD, W + Lo.xs, E, S * B, gx));
        // d z2h.bias = column sums of the gate gradients = d h2h.bias (b_z enters every step's gates as b_h2h does)
        RFN_TRY(mem_batch({{grd[P.dec(5)], grd[P.dec(3)], (long)GD}}, st));
        RFN_TRY(gemm1(S * B, E, seg_dx(gd, GD, prm[P.dec(0)], E, GD), W + Lo.dxs, E, 0, gx));
        RFN_TRY(rfn_embed_bwd(W + Lo.dxs, E, ids, B, ld_ids, 1, S * B, E, V1, grd[P.embed()], st));
        return RFN_OK;
    }
    RFN_TRY(mem_batch({{d_comb, nullptr, (long)T2 * BR}, {dPd, nullptr, (long)T2 * BA}}, st));
    rfn_gemm_problem pr[2];
    // Fused form of a backward step (3 launches): Kb1 = [dh_rec | dz] = dgates . [W_hh | W_z] in one launch (they share the
    // gate gradients); the attention backward; Kb2 = dh_rec += dhp . W_h whose epilogue completes d h of step s-1
    // (+ the logit layer's share dhe[s-1]) and runs that step's LSTM backward -- the next thing the sweep needs.
    // every step's operands are validated before the fused form is chosen (see the stage-II sweep in rfn_prefix_bwd)
    auto dec_kb1 = [&](int s, rfn_cell_out* kb1) {
        float* g = gd + (long)s * B * GD;
        kb1[0] = cell_out(dhrec, R, R, 0);
        cell_dx(kb1[0], g, GD, prm[P.dec(2)], R, GD);
        kb1[1] = cell_out(dz, R, R, 0);
        cell_dx(kb1[1], g, GD, prm[P.dec(4)], R, GD);
    };
    auto dec_kb2 = [&](int s, rfn_cell_out& kb2) {
        kb2 = cell_out(dhrec, R, R, 1);
        cell_dx(kb2, W + Lo.dhpd + s * BA, A, prm[P.dec(8)], R, A);
        if (s > 0)
            cell_lstm_bwd(kb2, gd + (long)(s - 1) * B * GD, GD, cd + (s - 1) * BR, R, cd + s * BR, R, dhe + (s - 1) * BR, R,
                          dc, R, dc, R, OFF_DECODER + (uint64_t)(s - 1));
    };
    bool fused = !d->decoder_maxout;
    for (int s = 0; s < S && fused; ++s) {
        rfn_cell_out t1[2], t2;
        dec_kb1(s, t1);
        dec_kb2(s, t2);
        fused = cell_ok(B, 2, t1, R) && cell_ok(B, 1, &t2, R);
    }
    if (fused)   // LSTM backward of the last step: nothing recurrent flows into it
        RFN_TRY(rfn_lstm_bwd(gd + (long)(S - 1) * B * GD, GD, cd + (S - 1) * BR, R, cd + S * BR, R, dhe + (S - 1) * BR, R, nullptr,
                             R, dc, R, B, R, 0, d->drop_lm, seed, OFF_DECODER + (uint64_t)(S - 1), st));
    int s_hi = S - 1;
    if (fused && S >= 3) {
        // steps S-1 ... 1 share one form (Kb2 carries the LSTM backward of the step below): one persistent launch
        // (rfn_chain.hip); step 0, whose Kb2 is a plain accumulate, follows as its three launches
        std::vector<ChainStep> steps((size_t)(S - 1));
        bool ok = true;
        for (int s = S - 1; s >= 1 && ok; --s) {
            ChainStep& cs = steps[(size_t)(S - 1 - s)];
            rfn_cell_out kb1[2], kb2;
            dec_kb1(s, kb1);
            dec_kb2(s, kb2);
            const float *proj = W + Lo.Pd, *hp = W + Lo.hpd + s * BA, *w = prm[P.dec(10)], *al = W + Lo.ald + (long)s * B * T2;
            const float* dzc = dz;
            float *dpr = dPd, *dhp = W + Lo.dhpd + s * BA, *dwp = W + Lo.dwp + s * BA, *dxc = d_comb;
            ok = cell_prepare(B, 2, kb1, R, 0.f, 0, &cs.g0, cell_variant(d)) == RFN_OK &&
                 rfn_attn_small_prepare_bwd(1, &proj, A, BA, &hp, &w, &al, &comb, R, BR, &dzc, R, B, T2, A, R, &dpr, A, BA, 1, &dhp,
                                            &dwp, &dxc, &cs.at) == RFN_OK &&
                 cell_prepare(B, 1, &kb2, R, d->drop_lm, seed, &cs.g2, cell_variant(d)) == RFN_OK;
        }
        if (ok) {
            RFN_TRY(rfn_chain_run(steps.data(), S - 1, chain_persist(d, RFN_PATH_OPT_PERSIST_DEC_BWD), (uint32_t*)(W + Lo.bar), st));
            s_hi = 0;
        }
    }
    for (int s = s_hi; s >= 0; --s) {
        float* g = gd + (long)s * B * GD;
        float* dht = dhe + s * BR;
        float* al = W + Lo.ald + (long)s * B * T2;
        float* dhp = W + Lo.dhpd + s * BA;
        if (fused) {
            rfn_cell_out kb1[2], kb2;
            dec_kb1(s, kb1);
            RFN_TRY(cell_run(B, 2, kb1, R, 0.f, 0, st, cell_variant(d)));
            RFN_TRY(attn1_bwd(W + Lo.Pd, A, BA, W + Lo.hpd + s * BA, prm[P.dec(10)], al, comb, R, BR, dz, R, B, T2, A, R,
                              dPd, A, BA, 1, dhp, W + Lo.dwp + s * BA, d_comb, st));
            dec_kb2(s, kb2);
            RFN_TRY(cell_run(B, 1, &kb2, R, d->drop_lm, seed, st, cell_variant(d)));
            continue;
        }
        if (s < S - 1) RFN_TRY(rfn_axpby_2d(1.f, dhrec, R, 1.f, dht, R, B, R, st));
        RFN_TRY(rfn_lstm_bwd(g, GD, cd + s * BR, R, cd + (s + 1) * BR, R, dht, R, (s < S - 1) ? dc : nullptr, R, dc,
                             R, B, R, d->decoder_maxout, d->drop_lm, seed, OFF_DECODER + (uint64_t)s, st));
        pr[0] = prob1(dhrec, R, seg_dx(g, GD, prm[P.dec(2)], R, GD));
        pr[1] = prob1(dz, R, seg_dx(g, GD, prm[P.dec(4)], R, GD));
        RFN_TRY(gemm_groups(B, R, 2, pr, 0, gx));
        RFN_TRY(attn1_bwd(W + Lo.Pd, A, BA, W + Lo.hpd + s * BA, prm[P.dec(10)], al, comb, R, BR, dz, R, B, T2, A, R,
                          dPd, A, BA, 1, dhp, W + Lo.dwp + s * BA, d_comb, st));
        RFN_TRY(gemm1(B, R, seg_dx(dhp, A, prm[P.dec(8)], R, A), dhrec, R, 1, gx));
    }
    RFN_TRY(mem_batch({{d_h0, dhrec, BR}, {d_c0, dc, BR}, {grd[P.dec(11)], nullptr, 1}}, st));
    // attention projection of the fused thoughts (shared by all steps)
    RFN_TRY(gemm1(T2 * B, R, seg_dx(dPd, A, prm[P.dec(6)], R, A), d_comb, R, 1, gx));
    RFN_TRY(gemm_dw(A, R, grd[P.dec(6)], R, grd[P.dec(7)], dPd, A, comb, R, T2 * B, gx));
    // weights shared across steps: one GEMM over (S*B) time-major rows each, bias gradients ride along
    RFN_TRY(rfn_colsum_f32(W + Lo.dwp, A, S * B, A, grd[P.dec(10)], 0, st));
    RFN_TRY(gemm_dw(A, R, grd[P.dec(8)], R, grd[P.dec(9)], W + Lo.dhpd, A, hd, R, S * B, gx));
    RFN_TRY(gemm_dw(GD, R, grd[P.dec(2)], R, grd[P.dec(3)], gd, GD, hd, R, S * B, gx));
    RFN_TRY(gemm_dw(GD, R, grd[P.dec(4)], R, grd[P.dec(5)], gd, GD, W + Lo.zd, R, S * B, gx));
    RFN_TRY(gemm_dw(GD, E, grd[P.dec(0)], E, grd[P.dec(1)], gd, GD, W + Lo.xs, E, S * B, gx));
    // embedding: dx = dgates . W_i2h, then the fixed-order scatter
    RFN_TRY(gemm1(S * B, E, seg_dx(gd, GD, prm[P.dec(0)], E, GD), W + Lo.dxs, E, 0, gx));
    RFN_TRY(rfn_embed_bwd(W + Lo.dxs, E, ids, B, ld_ids, 1, S * B, E, V1, grd[P.embed()], st));
    return RFN_OK;
}

// =============================================================================================
// free-running decoder step (sample / beam / one_time_step)
// =============================================================================================
extern "C" size_t rfn_decoder_step_ws_bytes(const rfn_dims* d, int B) {
    if (check_dims(d) != RFN_OK || B < 1) return 0;
    Bump b;
    b.take((size_t)B * d->E);
    b.take((size_t)B * d->A);
    b.take((size_t)B * d->T2);
    b.take((size_t)B * d->R);
    b.take((size_t)B * gate_width(d->decoder_maxout, d->R));
    b.take((size_t)B * d->V1);
    b.take(STEP_GEMM_WS_FLOATS);
    return b.off * sizeof(float);
}

extern "C" size_t rfn_decoder_cproj_floats(const rfn_dims* d, int B) {
    if (check_dims(d) != RFN_OK || B < 1) return 0;
    return cproj_u_off(d, B) + (size_t)d->T2 * B * gate_width(d->decoder_maxout, d->R);
}

extern "C" int rfn_decoder_prepare(const rfn_dims* d, int B, const float* const* prm, const float* comb, float* cproj,
                                   void* st) {
    RFN_TRY(check_dims(d));
    if (B < 1) return RFN_ERR_SHAPE;
    if (!prm || !comb || !cproj) return RFN_ERR_ARG;
    const PIdx P(d);
    const GemmCtx gx{st, nullptr, 0, d->gemm_flags};
    RFN_TRY(gemm1(d->T2 * B, d->A, seg_lin(comb, d->R, prm[P.dec(6)], d->R, d->R, prm[P.dec(7)]), cproj, d->A, 0, gx));
    if (!dec_hoisted(d)) return RFN_OK;
    const int GD = gate_width(d->decoder_maxout, d->R);   // the same unsplit product as decoder_fwd_begin's
    return gemm1(d->T2 * B, GD, seg_lin(comb, d->R, prm[P.dec(4)], d->R, d->R, nullptr), cproj + cproj_u_off(d, B), GD, 0, gx);
}

// One decoder step computed with exactly the operation sequence of one step of rfn_decoder_fwd (unsplit i2h, then
// h2h + z2h accumulated onto it, same split-K scratch size, same dropout stream (seed, OFF_DECODER + step)), so the
// distribution a host samples from here IS the one the teacher-forced gradient pass differentiates
// (misc/RecurrentFusionModel.py:260-270, 623-631: the reference samples from the dropout-affected outputs themselves).
static int decoder_step_impl(const rfn_dims* d, int B, const float* const* prm, const float* comb, const float* cproj,
                             const int64_t* ids, const float* xt, int64_t ld_xt, float* h, float* c, float* logits,
                             float* logp, int64_t ld_logp, void* ws, size_t ws_bytes, uint64_t seed, int step,
                             void* st, float* topv = nullptr, int32_t* topi = nullptr, int topw = 0, int row_div = 1) {
    RFN_TRY(check_dims(d));
    if (B < 1 || step < 0 || row_div < 1 || B % row_div) return RFN_ERR_SHAPE;
    if (!prm || !comb || !cproj || (!ids && !xt) || !h || !c || !ws) return RFN_ERR_ARG;
    if (xt && ld_xt < d->E) return RFN_ERR_SHAPE;
    if (ws_bytes < rfn_decoder_step_ws_bytes(d, B)) return RFN_ERR_WORKSPACE;
    const PIdx P(d);
    const int R = d->R, A = d->A, E = d->E, T2 = d->T2, V1 = d->V1;
    const int GD = gate_width(d->decoder_maxout, R);
    Bump b;
    float* W = (float*)ws;
    const GemmCtx gx{st, W + b.take(STEP_GEMM_WS_FLOATS), STEP_GEMM_WS_FLOATS * sizeof(float), d->gemm_flags};  // split-K scratch
    const GemmCtx gx_whole{st, nullptr, 0, d->gemm_flags};
    float* x = W + b.take((size_t)B * E);
    float* hp = W + b.take((size_t)B * A);
    float* al = W + b.take((size_t)B * T2);
    float* z = W + b.take((size_t)B * R);
    float* g = W + b.take((size_t)B * GD);
    float* lg = logits ? logits : W + b.take((size_t)B * V1);
    if (!xt) RFN_TRY(rfn_embed_fwd(prm[P.embed()], E, V1, ids, B, 1, 0, B, x, E, st));
    RFN_TRY(gemm1(B, GD, xt ? seg_lin(xt, ld_xt, prm[P.dec(0)], E, E, prm[P.dec(1)]) : seg_lin(x, E, prm[P.dec(0)], E, E, prm[P.dec(1)]),
                  g, GD, 0, gx_whole));
    RFN_TRY(decoder_cell_core(d, B, prm, comb, cproj, cproj + cproj_u_off(d, B / row_div), row_div, h, c, h, c, hp, al, z, g, gx,
                              seed, step, st));
    if (logits || logp || topv) {
        RFN_TRY(gemm_logits(B, V1, h, R, prm[P.logit_w()], prm[P.logit_b()], lg, gx_whole));
        if (logp) {
            if (ld_logp < V1) return RFN_ERR_SHAPE;
            RFN_TRY(rfn_log_softmax_fwd(lg, V1, B, V1, B, ld_logp, 0, logp, st));
        }
        if (topv) RFN_TRY(rfn_log_softmax_topk(lg, V1, B, V1, topw, topv, topi, st));   // beam search: W best per row, no full rows
    }
    return RFN_OK;
}

extern "C" int rfn_decoder_step(const rfn_dims* d, int B, const float* const* prm, const float* comb,
                                const float* cproj, const int64_t* ids, float* h, float* c, float* logits, float* logp,
                                int64_t ld_logp, void* ws, size_t ws_bytes, uint64_t seed, int step, void* st) {
    if (!ids) return RFN_ERR_ARG;
    return decoder_step_impl(d, B, prm, comb, cproj, ids, nullptr, 0, h, c, logits, logp, ld_logp, ws, ws_bytes, seed,
                             step, st);
}
// the reference's one_time_step signature: the caller has already embedded the token (xt = model.embed(it))
extern "C" int rfn_decoder_step_embedded(const rfn_dims* d, int B, const float* const* prm, const float* comb,
                                         const float* cproj, const float* xt, int64_t ld_xt, float* h, float* c,
                                         float* logits, float* logp, int64_t ld_logp, void* ws, size_t ws_bytes,
                                         uint64_t seed, int step, void* st) {
    if (!xt) return RFN_ERR_ARG;
    return decoder_step_impl(d, B, prm, comb, cproj, nullptr, xt, ld_xt, h, c, logits, logp, ld_logp, ws, ws_bytes,
                             seed, step, st);
}

// =============================================================================================
// whole decode loops queued by one call (no host work between steps)
// =============================================================================================
// sample() free-running decode (misc/RecurrentFusionModel.py:616-653): step t = 0 feeds BOS; step t >= 1 feeds the token
// picked from step t-1's distribution (mode 0: argmax; mode 1: inverse-CDF draw with the caller's uniform u[t-1][b]).
// Every step is rfn_decoder_step (embedding K10, cell a5, logit + log-softmax K11) on the same buffers, so the result is
// bit for bit what the step-by-step host loop gives -- only the host is gone from the loop.  `unf` keeps one row of
// unfinished flags per step so that the caller applies the reference's early exit (:645) with ONE read-back afterwards.
extern "C" int rfn_decoder_loop(const rfn_dims* d, int B, int steps, const float* const* prm, const float* comb,
                                const float* cproj, float* h, float* c, int mode, float inv_temperature, const float* u,
                                float* logp_all, int64_t ld_b, int64_t ld_t, int64_t* seq, int64_t ld_seq, float* seq_lp,
                                int64_t ld_lp, int32_t* unf, int64_t* ids, void* ws, size_t ws_bytes, uint64_t seed,
                                void* st) {
    RFN_TRY(check_dims(d));
    if (B < 1 || steps < 1 || (mode != 0 && mode != 1)) return RFN_ERR_SHAPE;
    if (!prm || !comb || !cproj || !h || !c || !logp_all || !seq || !seq_lp || !unf || !ids || !ws) return RFN_ERR_ARG;
    if (mode == 1 && !u) return RFN_ERR_ARG;
    const int V1 = d->V1;
    if (hipMemsetAsync(ids, 0, (size_t)B * sizeof(int64_t), (hipStream_t)st) != hipSuccess) return RFN_ERR_LAUNCH;   // BOS
    for (int t = 0; t < steps; ++t) {
        if (t >= 1) {
            const float* prev = logp_all + (long)(t - 1) * ld_t;
            if (mode == 1)   // the draw first: the greedy-pick kernel below then records ITS log-prob and finished flags
                RFN_TRY(rfn_multinomial_pick(prev, ld_b, B, V1, inv_temperature, u + (long)(t - 1) * B, nullptr, 1.f, ids, 1, st));
            RFN_TRY(rfn_pick_record(prev, ld_b, B, V1, t, mode == 1 ? ids : nullptr, ids, seq + (t - 1), ld_seq, seq_lp + (t - 1),
                                    ld_lp, t > 1 ? unf + (long)(t - 1) * B : nullptr, unf + (long)t * B, st));
        }
        RFN_TRY(rfn_decoder_step(d, B, prm, comb, cproj, ids, h, c, nullptr, logp_all + (long)t * ld_t, ld_b, ws, ws_bytes, seed, t,
                                 st));
    }
    return RFN_OK;
}

// The step-wise training decoder with draws between the steps (scheduled sampling :260-270, multinomial sample() with
// grad :623-631), queued by one call: begin, then for every step s >= 1 rows whose coin u_coin[s][b] < ss_prob get a
// token drawn from step s-1's distribution (uniform u_draw[s][b]), then rfn_decoder_fwd_step.  Leaves workspace, ids and
// log_prob exactly as the host loop over rfn_multinomial_pick / rfn_decoder_fwd_step does.
extern "C" int rfn_decoder_fwd_sampled(const rfn_dims* d, int B, int S, const float* const* prm, const float* comb,
                                       const float* h0, const float* c0, int64_t* ids, int64_t ld_ids, float ss_prob,
                                       float inv_temperature, const float* u_draw, const float* u_coin, float* log_prob,
                                       void* ws, size_t ws_bytes, int train, uint64_t seed, void* st) {
    RFN_TRY(check_dims(d));
    if (B < 1 || S < 1) return RFN_ERR_SHAPE;
    if (!prm || !comb || !h0 || !c0 || !ids || !u_draw || !u_coin || !log_prob || !ws) return RFN_ERR_ARG;
    RFN_TRY(rfn_decoder_fwd_begin(d, B, S, prm, comb, h0, c0, ws, ws_bytes, train, st));
    const long ld_b = (long)S * d->V1;
    for (int s = 0; s < S; ++s) {
        if (s >= 1)
            RFN_TRY(rfn_multinomial_pick(log_prob + (long)(s - 1) * d->V1, ld_b, B, d->V1, inv_temperature, u_draw + (long)s * B,
                                         u_coin + (long)s * B, ss_prob, ids + s, ld_ids, st));
        RFN_TRY(rfn_decoder_fwd_step(d, B, S, s, prm, comb, ids + s, ld_ids, log_prob, ws, ws_bytes, train, seed, st));
    }
    return RFN_OK;
}

// sample_beam's search (misc/RecurrentFusionModel.py:451-531) for all images at once, queued by one call: per step the
// device-side bookkeeping (rfn_beam_step), the re-gather of the recurrent state rows and one decoder step on the
// NB * W beam rows.  h / c are ping-ponged with h_alt / c_alt; on return the live state is in h / c again.
extern "C" int rfn_beam_loop(const rfn_dims* d, int NB, int W, int S, const float* const* prm, const float* comb,
                             const float* cproj, float* h, float* c, float* h_alt, float* c_alt, float* logp,
                             int64_t* beam_seq, float* beam_lp, float* beam_sum, int32_t* order, int64_t* ids,
                             int64_t* done_seq, float* done_lp, float* done_p, int32_t* done_n, int32_t* active, int max_done,
                             void* ws, size_t ws_bytes, uint64_t seed, void* st) {
    // `logp` holds, per beam row, its W best log-probs and their tokens (2 * W values): the search never looks at more
    // (:463-466), so the full (rows, V+1) log-prob matrix is neither written nor read back
    RFN_TRY(check_dims(d));
    if (NB < 1 || W < 1 || S < 1) return RFN_ERR_SHAPE;
    if (!prm || !comb || !cproj || !h || !c || !h_alt || !c_alt || !logp || !ids || !order || !ws) return RFN_ERR_ARG;
    const int rows = NB * W, V1 = d->V1, R = d->R;
    if (W > 32) return RFN_ERR_SHAPE;
    float* topv = logp;
    int32_t* topi = (int32_t*)(logp + (size_t)rows * W);
    float *hc = h, *cc = c, *ha = h_alt, *ca = c_alt;
    if (hipMemsetAsync(ids, 0, (size_t)rows * sizeof(int64_t), (hipStream_t)st) != hipSuccess) return RFN_ERR_LAUNCH;
    for (int t = 0; t <= S; ++t) {
        if (t >= 1) {
            RFN_TRY(rfn_beam_step_topk(topv, topi, V1, W, S, t, NB, max_done, beam_seq, beam_lp, beam_sum, order, ids, done_seq,
                                       done_lp, done_p, done_n, active, st));
            if (t == S) break;   // the reference still runs one more decoder step whose output is never used
            RFN_TRY(rfn_gather_rows(hc, ha, order, rows, R, st));
            RFN_TRY(rfn_gather_rows(cc, ca, order, rows, R, st));
            float* x = hc; hc = ha; ha = x;
            x = cc; cc = ca; ca = x;
        }
        RFN_TRY(decoder_step_impl(d, rows, prm, comb, cproj, ids, nullptr, 0, hc, cc, nullptr, nullptr, 0, ws, ws_bytes, seed, t, st,
                                  topv, topi, W, W));   // the W rows of an image share its thought vectors (comb / cproj: NB rows)
    }
    if (hc != h) {   // an odd number of swaps: bring the live state home
        RFN_TRY(mem_batch({{h, hc, (long)rows * R}, {c, cc, (long)rows * R}}, st));
    }
    return RFN_OK;
}
