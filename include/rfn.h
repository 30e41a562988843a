/*
 * rfn.h -- C ABI of librfn_hip.so: the MI355X (gfx950) implementation of the recurrent-fusion
 * caption-decoder hot path of cswhjiang/Recurrent_Fusion_Network.
 *
 * The reference has no FFI for this path: its seam is the Python duck type returned by
 * models.setup(opt) (reference models.py:14-38) whose forward()/sample() call nothing but ATen.
 * Each entry point below names the reference code it replaces (paths relative to the reference
 * checkout).  INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *   - plain C types only; every pointer is a DEVICE pointer unless its name ends in _host;
 *   - all floating-point data is IEEE fp32, row-major; token / target ids are int64;
 *   - the caller owns every buffer, including workspaces (sizes from the *_ws_bytes queries);
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous on it, never allocate,
 *     never synchronise, and are re-entrant.  The library reads no environment variable and keeps
 *     no mutable state except write-once, per-device launch attributes of its kernels (dynamic-LDS
 *     opt-in, blocks per CU), so one process may drive several devices; every tuning choice
 *     travels with the call (rfn_dims.gemm_flags, rfn_gemm_f32_opt);
 *   - return RFN_OK (0) or a negative RFN_ERR_* code; nothing is launched on a shape error.
 */
#ifndef RFN_H_
#define RFN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RFN_OK 0
#define RFN_ERR_SHAPE (-1)       /* unsupported or inconsistent dimensions */
#define RFN_ERR_UNSUPPORTED (-2) /* configuration the path does not implement (e.g. maxout) */
#define RFN_ERR_LAUNCH (-3)      /* HIP launch failure */
#define RFN_ERR_WORKSPACE (-4)   /* workspace too small */
#define RFN_ERR_ARG (-5)         /* null / misaligned pointer */

#define RFN_MAX_ENC 8
#define RFN_ABI_VERSION 7

/* Model dimensions: the fields RecurrentFusionModel.__init__ reads from `opt`
 * (misc/RecurrentFusionModel.py:118-151).  Limits (RFN_ERR_SHAPE otherwise): M <= RFN_MAX_ENC,
 * T1*M <= 64 and T2*M <= 64 (the reference ships T1 = T2 = 8, M <= 5), dropout probabilities in [0, 1). */
typedef struct rfn_dims {
    int32_t M;                 /* number of encoders, len(opt.feat_array_info)            */
    int32_t R;                 /* opt.rnn_size                                            */
    int32_t A;                 /* opt.att_hid_size                                        */
    int32_t E;                 /* opt.input_encoding_size                                 */
    int32_t T1;                /* opt.num_review_steps_0 (fusion stage I steps)           */
    int32_t T2;                /* opt.num_review_steps   (fusion stage II steps)          */
    int32_t K;                 /* opt.top_words_count                                     */
    int32_t V1;                /* opt.vocab_size + 1                                      */
    int32_t L[RFN_MAX_ENC];    /* feat_array_info[i]['att_num']                           */
    int32_t D[RFN_MAX_ENC];    /* feat_array_info[i]['att_feat_size']                     */
    int32_t F[RFN_MAX_ENC];    /* feat_array_info[i]['fc_feat_size']                      */
    int32_t review_maxout;     /* opt.review_maxout (stage II gates are 5R wide)          */
    int32_t decoder_maxout;    /* opt.maxout        (decoder gates are 5R wide)           */
    float drop_fusion;         /* opt.drop_prob_fusion (stage I)                          */
    float drop_reason;         /* opt.drop_prob_reason (stage II)                         */
    float drop_lm;             /* opt.drop_prob_lm     (decoder)                          */
    uint32_t gemm_flags;       /* RFN_GEMM_OPT_* bits applied to every GEMM of the path (0 = defaults)  */
    uint32_t path_flags;       /* RFN_PATH_OPT_* bits (0 = defaults)                                    */
    /* Measurement hook (NULL = off; bench.py's in-step roofline): an array of 4*M hipEvent_t created by the caller
     * with timing enabled.  rfn_prefix_fwd records events [2i] / [2i+1] on the launch stream right before / after
     * encoder i's hoisted att_2_att_h projection launch (the path's dominant kernel), rfn_prefix_bwd_wgrad records
     * [2M+2i] / [2M+2i+1] around encoder i's att_2_att_h weight-gradient launch.  Recording is asynchronous; the
     * caller reads the pairs after its own synchronisation.  Nothing else in the library looks at it. */
    void* const* probe_events;
} rfn_dims;
/* rfn_dims.path_flags: run the named recurrence -- all of its steps -- inside ONE persistent launch (csrc/rfn_chain.hip: the
 * per-step launches' device bodies, a grid barrier between dependent phases; bit-identical results) instead of three launches
 * per step.  Opt-in: measured on MI355X the in-launch hand-off costs what the launch boundary costs (profiles/r05_chain.md).
 * A/B hooks only: the persistent grid is launched non-cooperatively and relies on all of its blocks (<= CU count, 128 KB of LDS each)
 * being resident at once -- do not set them beside other streams, processes or collectives on the same device (a block that cannot
 * become resident makes its peers spin into a trap).  The decoder bits imply RFN_PATH_OPT_DEC_UNHOISTED (the chains are built
 * from the three-launch decoder cell). */
#define RFN_PATH_OPT_PERSIST_DEC_FWD 1u   /* teacher-forced decoder steps (rfn_decoder_fwd)                   */
#define RFN_PATH_OPT_PERSIST_S2_FWD 2u    /* stage-II steps (rfn_prefix_fwd)                                  */
#define RFN_PATH_OPT_PERSIST_DEC_BWD 4u   /* decoder backward sweep, steps S-1 ... 1 (rfn_decoder_bwd)        */
#define RFN_PATH_OPT_PERSIST_S2_BWD 8u    /* stage-II backward sweep, steps T2-1 ... 1 (rfn_prefix_bwd)       */
#define RFN_PATH_OPT_PERSIST_ALL 15u
#define RFN_PATH_OPT_NO_SMALL_TILES 32u   /* A/B hook: per-step products keep 32-row tiles even when the launch has so few of them
                                           * that the library would take 16-row tiles (rfn_cell_gemm variant 6; same results)  */
#define RFN_PATH_OPT_SHARED_SMALL_TILES 64u /* A/B hook: those 16-row tiles on ring slots shared by the block (rfn_cell_gemm variants
                                           * 4 / 5) instead of wave-private ones (variant 6, the default); same results        */
#define RFN_PATH_OPT_DEC_UNHOISTED 128u   /* A/B hook: the decoder cell of rounds 3-5 (z2h(z) as a per-step product: three
                                           * dependent launches per step each way) instead of the hoisted form (two).  Same
                                           * mathematics, different rounding.  Implied by the PERSIST_DEC_* bits; the beam loop,
                                           * whose rows share their image's thought vectors, has no unhoisted form          */
#define RFN_PATH_OPT_DEEP_CELLS 16u       /* A/B hook: per-step products with no more tiles than CUs on the deep-ring kernel
                                           * (rfn_cell_gemm, RFN_CELL_VARIANT_DEEP) instead of the 3-slot one; not faster      */

int rfn_abi_version(void);
const char* rfn_error_string(int code);

/* ---- parameter table ------------------------------------------------------------------------
 * Parameters and their gradients are passed as arrays of device pointers in ONE canonical order.
 * rfn_param_name() returns the reference state_dict key of slot `idx` (e.g.
 * "review_steps_individual.3.lstm.1.att_model.att_2_att_h.weight"), so a host binds the table by
 * name and never hard-codes the order.  nn.Linear layout (out, in). */
int rfn_param_count(const rfn_dims* d);
int rfn_param_name(const rfn_dims* d, int idx, char* buf, size_t buflen);
/* rows/cols of slot idx (cols = 1 for a bias) */
int rfn_param_shape(const rfn_dims* d, int idx, int64_t* rows, int64_t* cols);

/* ---- primitive operators (exported for unit parity tests and for other hosts) ---------------- */

/* C[M,N] (+)= sum_s op(A_s)[M,K_s] * op(B_s)[N,K_s]^T + sum_s bias_s[N]; exact-fp32 MFMA
 * (v_mfma_f32_32x32x2_f32: each output is a k-ordered fp32 fma chain).  Replaces every nn.Linear
 * / mm / addmm / bmm on the path and their backward
 * matmuls.  *_kfast = 1: the reduction index is the contiguous one (A is [M,K] with row stride
 * lda; B is [N,K] = an nn.Linear weight).  *_kfast = 0: the output index is contiguous (A is
 * [K,M] with row stride lda; B is [K,N]).  Up to RFN_GEMM_MAXSEG K-segments accumulate into one C
 * (gate sums such as H2h(H) + z2h(z), misc/RecurrentFusionModel.py:53); up to RFN_GEMM_MAXGROUP
 * same-shape problems run in one launch (the per-step weights of one encoder). */
#define RFN_GEMM_MAXSEG 8
#define RFN_GEMM_MAXGROUP 8
typedef struct rfn_gemm_seg {
    const float* A;
    const float* B;
    const float* bias; /* may be NULL */
    int64_t lda, ldb;
    int32_t K;
    int32_t a_kfast, b_kfast;
    int32_t pad_;
} rfn_gemm_seg;
typedef struct rfn_gemm_problem {
    float* C;
    int64_t ldc;
    int32_t nseg;
    int32_t pad_;
    /* optional, only with a_kfast = 0: a_colsum[m] (=|+=) sum_s sum_k A_s[k, m].  For dW = dY^T X this is
     * the bias gradient colsum(dY), produced by the GEMM that already streams dY (no extra pass). */
    float* a_colsum;
    rfn_gemm_seg seg[RFN_GEMM_MAXSEG];
} rfn_gemm_problem;
int rfn_gemm_f32(int M, int N, int ngroups, const rfn_gemm_problem* problems_host, int accumulate,
                 void* stream);
/* Same, with a scratch buffer (>= 1 MiB, 16-B aligned).  Skinny problems (M = batch rows, few output tiles)
 * are then cut along K across thread blocks; the partial tiles are summed in a fixed order by a second
 * kernel, so the result is deterministic (it differs from the unsplit result only by fp32 re-association). */
int rfn_gemm_f32_ws(int M, int N, int ngroups, const rfn_gemm_problem* problems_host, int accumulate,
                    void* ws, size_t ws_bytes, void* stream);
/* Same, with option bits (every tuning choice travels with the call: the library reads no environment variable and
 * keeps no mutable process state beyond write-once, per-device launch attributes of its kernels).
 *   RFN_GEMM_OPT_LDS_LEAN  the long big-tile GEMMs take 32 KB of LDS per block (64 KB per CU for a one-round
 *                          weight-gradient launch) instead of 64 KB per block, so that kernels of other streams --
 *                          RCCL's under data parallelism -- can co-reside instead of waiting for a
 *                          multi-millisecond GEMM to drain; same results, same speed within 1 %;
 *   RFN_GEMM_OPT_NO_DMA    interior big tiles use the register-staged kernel instead of the LDS-DMA one (A/B hook;
 *                          both give bit-identical results: same k order per output element);
 *   RFN_GEMM_OPT_BF16X3    (rfn_dims.gemm_flags only) the two long products of the path -- the hoisted stage-I feature
 *                          projection and its weight gradient -- run on the bf16 matrix cores with every f32 operand
 *                          held as three bf16 planes and six plane products accumulated in f32 (rfn_x3_*, below):
 *                          f32-level accuracy (at least as close to an f64 product as the f32 MFMA chain), not
 *                          bit-identical to it.  Off by default.  Products too short to pay for the split passes
 *                          (< 2e10 multiply-adds x 2) stay on the exact kernels unless
 *   RFN_GEMM_OPT_BF16X3_ANY_SIZE is also set (parity tests run the small reference-generated tiers through the plane
 *                          GEMM with it). */
#define RFN_GEMM_OPT_LDS_LEAN 1u
#define RFN_GEMM_OPT_NO_DMA 2u
#define RFN_GEMM_OPT_BF16X3 4u
#define RFN_GEMM_OPT_BF16X3_ANY_SIZE 8u
/* (rfn_dims.gemm_flags only) A/B hook: the split-K GEMMs of the path finish inside their launch (rfn_gemm_f32_tk, below)
 * instead of through rfn_gemm_reduce_k.  Bit-identical results; MEASURED SLOWER on MI355X (C3 step 65.6 -> 68.9 ms, C2
 * 5.4 -> 6.6 ms: every K-range block pays an agent-scope release = a write-back of its XCD's L2 before its ticket), which
 * is what the MI355X guide reports for split-K seams kept inside a launch.  Off by default. */
#define RFN_GEMM_OPT_SPLITK_IN_KERNEL 16u
/* Diagnostics (tools/split_probe.py): bits 8-12 force the K split of a medium big-tile product (1 = unsplit). */
#define RFN_GEMM_OPT_FORCE_SPLIT(n) (((unsigned)(n) & 31u) << 8)
int rfn_gemm_f32_opt(int M, int N, int ngroups, const rfn_gemm_problem* problems_host, int accumulate,
                     void* ws, size_t ws_bytes, unsigned flags, void* stream);
/* Same, with split-K finished INSIDE the launch: `tickets` points to n_tickets int32 counters that are ZERO on entry
 * (the library leaves them zero: a caller zeroes them once per workspace and keeps them for its GEMMs).  The K ranges of
 * an output tile draw tickets on the tile's counter; the last one to arrive adds the partial tiles in K-range order
 * (the order the separate reduce kernel uses: bit-identical results) -- no second launch.  Launches with more output
 * tiles than counters use the separate reduce kernel. */
int rfn_gemm_f32_tk(int M, int N, int ngroups, const rfn_gemm_problem* problems_host, int accumulate, void* ws,
                    size_t ws_bytes, unsigned flags, int32_t* tickets, int n_tickets, void* stream);
/* Gate GEMM + LSTM update in one call (stage I: misc/RecurrentFusionModel.py:53-73): gates_g[M, 4R] = sum_s A_s W_s^T + b
 * for ngroups cells (problems_host[g].C = cell g's gate buffer, equally spaced), then rfn_lstm_fwd_grouped on them.  When
 * the product is cut along K the update rides on the fixed-order reduce (one launch less); results are those of
 * rfn_gemm_f32_ws followed by rfn_lstm_fwd_grouped either way. */
typedef struct rfn_gemm_lstm {
    const float* c_prev;   /* cell g at c_prev + g * gs_cprev, (M, R) with row stride ldcp; likewise c_next, h_next */
    float* c_next;
    float* h_next;
    int64_t ldcp, ldcn, ldh, gs_cprev, gs_cnext, gs_h;
    float drop_p;
    int32_t pad_;
    uint64_t seed, offset; /* dropout stream of cell g: (seed, offset + g) */
} rfn_gemm_lstm;
int rfn_gemm_f32_lstm(int M, int R, int ngroups, const rfn_gemm_problem* problems_host, void* ws, size_t ws_bytes,
                      unsigned flags, const rfn_gemm_lstm* lstm, void* stream);

/* ---- f32 GEMM on the bf16 matrix cores (csrc/rfn_gemm_x3.hip) -------------------------------------------------------
 * Replaces, when RFN_GEMM_OPT_BF16X3 is set, the two nn.Linear products the reference spends most of its time in:
 * att_2_att_h over all B*L regions (misc/AttentionModelCore.py:33-35, hoisted over the T1 review steps) and its
 * weight gradient (autograd of the same line).
 *
 * A logical operand Y[rows][K] (row = output index, K = reduction index) is held as a PLANE IMAGE: x = x0 + x1 + x2 with
 * x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1) (round to nearest even; exact for finite x below 2^127), stored
 * as 1-KiB pieces in the operand order of v_mfma_f32_16x16x32_bf16: piece (kc, rb, p) at byte
 * ((kc * nrb + rb) * 3 + p) * 1024 holds, at 16 * l, plane p of Y[rb * 16 + l % 16][kc * 32 + 8 * (l / 16) + 0..7];
 * nrb = rows padded to 256, over 16; K is padded to 32; pad rows / columns are zero.  rfn_x3_image_bytes gives the size
 * (6 bytes per padded element). */
size_t rfn_x3_image_bytes(int rows, int K);
/* Image of Y[ngroups * rows][K] whose row block g is the f32 matrix srcs_host[g] (device pointers in a host array;
 * k_fast = 1: element (row, k) at src[row * ld + k], k_fast = 0: at src[k * ld + row]).  rows % 32 == 0 unless
 * ngroups == 1; ngroups <= 64. */
int rfn_x3_split(const float* const* srcs_host, int ngroups, int64_t ld, int rows, int K, int k_fast, void* image,
                 void* stream);
/* C (+)= A . B^T from two images (A: M rows, B: N rows, both with reduction length K): the six plane products
 * a0.b0, a0.b1, a1.b0, a0.b2, a1.b1, a2.b0, smallest first, accumulated in f32 by v_mfma_f32_16x16x32_bf16.
 * The output is cut into groups of gm rows x gn columns, group (i, j) written to C_host[i * ceil(N/gn) + j] with leading
 * dimension ldc (+ bias_host[..][column inside the group] when bias_host and the entry are non-NULL); gm and gn must be
 * multiples of 256 unless they cover the whole dimension; at most 64 groups.  splitk > 1 cuts K over blocks; the partial
 * tiles (part: rfn_x3_part_floats floats) are summed in slice order by a second kernel (deterministic; N % 4 == 0). */
int rfn_x3_gemm(int M, int N, int K, const void* imageA, const void* imageB, int gm, int gn, float* const* C_host,
                const float* const* bias_host, int64_t ldc, int accumulate, int splitk, float* part, void* stream);
/* K-SLOW images: a logical operand stored reduction-index-major in memory, Y[K][cols] (the weight gradient's two operands:
 * dproj[(b,l)][a] and att[(b,l)][d]), keeps that orientation: element (k, plane, m) at ((k * 3 + plane) * Mp + m) * 2
 * bytes, Mp = ngroups * cols padded to 256, k padded to 32 with zero rows; rfn_x3_image_bytes(ngroups * cols, K) bytes.
 * Column block g comes from srcs_host[g][k * ld + c] (cols % 4 == 0, ld % 4 == 0, 16-B aligned).  No transposing pass:
 * rfn_x3_gemm_ks forms the MFMA operands with the transposing LDS read (ds_read_b64_tr_b16). */
int rfn_x3_split_ks(const float* const* srcs_host, int ngroups, int64_t ld, int K, int cols, void* image, void* stream);
/* rfn_x3_gemm with BOTH operands given as k-slow images (A: M columns, B: N columns, K rows each). */
int rfn_x3_gemm_ks(int M, int N, int K, const void* imageA, const void* imageB, int gm, int gn, float* const* C_host,
                   const float* const* bias_host, int64_t ldc, int accumulate, int splitk, float* part, void* stream);
size_t rfn_x3_part_floats(int M, int N, int splitk);
/* the number of K slices with which one round of blocks covers the chip (1 for launches of many tiles) */
int rfn_x3_splitk_for(int M, int N, int K);

/* out[n] (+)= sum_r X[r*ldx + n]   (bias gradients) */
int rfn_colsum_f32(const float* X, int64_t ldx, int rows, int cols, float* out, int accumulate,
                   void* stream);
/* the same for `ngroups` matrices X + g*group_stride, each into its own outs_host[g] (one launch) */
int rfn_colsum_grouped_f32(const float* X, int64_t group_stride, int64_t ldx, int rows, int cols,
                           float* const* outs_host, int ngroups, void* stream);
/* the same, every sum also written to outs2_host[g] where that is not NULL (two parameters that enter one pre-activation
 * share one bias gradient: H2h.bias / z2h.bias, LSTMFusionNoInputCore.py:42; att_2_att_h.bias / h_2_att_h.bias,
 * AttentionModelCore.py:36-38) */
int rfn_colsum_grouped2_f32(const float* X, int64_t group_stride, int64_t ldx, int rows, int cols,
                            float* const* outs_host, float* const* outs2_host, int ngroups, void* stream);

/* outs_host[g][0..n) = value for `ngroups` small device buffers in one launch */
int rfn_fill_small_f32(float* const* outs_host, int ngroups, int n, float value, void* stream);
/* dst_host[g][0..n) = src_host[g][0..n) for `ngroups` small device buffer pairs in one launch */
int rfn_copy_small_f32(float* const* dst_host, const float* const* src_host, int ngroups, int n, void* stream);

/* Additive soft attention, AttentionModelCore.forward (misc/AttentionModelCore.py:31-48; inlined
 * copy misc/LSTMSoftAttentionCore.py:64-79), split at the GEMM boundary:
 *   proj[b,l,:] = att_2_att_h(att_seq[b,l,:])  (hoisted GEMM, :32-34)
 *   hproj[b,:]  = h_2_att_h(pre_h[b,:])        (GEMM, :36)
 * scores:  alpha[b,:] = softmax_l( w . tanh(proj[b,l,:] + hproj[b,:]) + b_o )      (:39-44)
 * context: z[b,:]     = sum_l alpha[b,l] * att_seq[b,l,:]                          (:45-47)
 * Element (b,l,x) of proj / att_seq lives at base + b*stride_b + l*stride_l + x. */
int rfn_attn_scores_fwd(const float* proj, int64_t proj_sb, int64_t proj_sl, const float* hproj,
                        const float* w_out, const float* b_out, int B, int L, int A, float* alpha,
                        void* stream);
int rfn_attn_context_fwd(const float* att_seq, int64_t sb, int64_t sl, const float* alpha, int B,
                         int L, int D, float* z, int64_t ldz, void* stream);
/* rfn_attn_scores_fwd + rfn_attn_context_fwd in two launches instead of three: the context kernel normalises the
 * raw scores itself.  scores_scratch: B*L floats, must not alias alpha.  Results are bit-identical to the pair. */
int rfn_attn_fwd(const float* proj, int64_t proj_sb, int64_t proj_sl, const float* hproj, const float* w_out,
                 const float* b_out, const float* att_seq, int64_t sb, int64_t sl, int B, int L, int A, int D,
                 float* scores_scratch, float* alpha, float* z, int64_t ldz, void* stream);
/* The same for `ngroups` encoders that share (L, A, D) and every stride, two launches in total; arrays are host
 * arrays of device pointers, one entry per encoder. */
int rfn_attn_fwd_grouped(int ngroups, const float* const* proj, int64_t proj_sb, int64_t proj_sl,
                         const float* const* hproj, const float* const* w_out, const float* const* b_out,
                         const float* const* att_seq, int64_t sb, int64_t sl, int B, int L, int A, int D,
                         float* const* scores_scratch, float* const* alpha, float* const* z, int64_t ldz,
                         void* stream);
/* dalpha[b,l] = <dz[b,:], att_seq[b,l,:]> */
int rfn_attn_context_bwd_dalpha(const float* att_seq, int64_t sb, int64_t sl, const float* dz,
                                int64_t lddz, int B, int L, int D, float* dalpha, void* stream);
/* datt_seq[b,l,:] += alpha[b,l] * dz[b,:]   (only for differentiable att_seq: stages II, decoder) */
int rfn_attn_context_bwd_dseq(const float* alpha, const float* dz, int64_t lddz, int B, int L,
                              int D, float* datt_seq, int64_t sb, int64_t sl, void* stream);
/* softmax + tanh backward.  ds = alpha*(dalpha - <alpha,dalpha>); dproj[b,l,a] (=|+=)
 * ds[b,l]*w[a]*(1-e^2), e = tanh(proj+hproj); dhproj[b,a] = sum_l dproj[b,l,a];
 * dw_part[b,a] = sum_l ds[b,l]*e[b,l,a]  (caller column-sums dw_part over b).
 * dproj may alias proj (in-place).  */
int rfn_attn_scores_bwd(const float* proj, int64_t proj_sb, int64_t proj_sl, const float* hproj,
                        const float* w_out, const float* alpha, const float* dalpha, int B, int L,
                        int A, float* dproj, int64_t dproj_sb, int64_t dproj_sl,
                        int accumulate_dproj, float* dhproj, float* dw_part, void* stream);

/* rfn_attn_context_bwd_dalpha + rfn_attn_scores_bwd in one launch, one block per batch row: dalpha never leaves
 * the CU.  Results are bit-identical to the pair.  (d att_seq is not produced: stage-I features need no gradient.) */
int rfn_attn_bwd(const float* proj, int64_t proj_sb, int64_t proj_sl, const float* hproj, const float* w_out,
                 const float* alpha, const float* att_seq, int64_t sb, int64_t sl, const float* dz, int64_t lddz,
                 int B, int L, int A, int D, float* dproj, int64_t dproj_sb, int64_t dproj_sl,
                 int accumulate_dproj, float* dhproj, float* dw_part, void* stream);
/* rfn_attn_fwd_grouped / rfn_attn_bwd_grouped for encoders whose feature maps differ in (L_g, D_g) -- the reference ships
 * 196 x 2048, 64 x 1536, 64 x 1280 and 49 x 2208 maps side by side (feat_array.py:240-244, one AttentionModelCore per
 * encoder: misc/RecurrentFusionModel.py:139-146) -- in ONE pair of launches / one launch: the M cells of a stage-I step are
 * independent (misc/RecurrentFusionModel.py:101-114).  Contiguous layouts: proj / dproj (B, L_g, A), att_seq (B, L_g, D_g),
 * z / dz (B, D_g), alpha and the raw-score scratch (B, L_g).  Same kernels and per-row arithmetic as the per-encoder calls:
 * bit-identical results.  L_host / D_host: host arrays of ngroups ints. */
int rfn_attn_fwd_het(int ngroups, const float* const* proj, const float* const* hproj, const float* const* w_out,
                     const float* const* b_out, const float* const* att_seq, int B, const int* L_host, int A,
                     const int* D_host, float* const* scores_scratch, float* const* alpha, float* const* z, void* stream);
int rfn_attn_bwd_het(int ngroups, const float* const* proj, const float* const* hproj, const float* const* w_out,
                     const float* const* alpha, const float* const* att_seq, const float* const* dz, int B,
                     const int* L_host, int A, const int* D_host, float* const* dproj, int accumulate_dproj,
                     float* const* dhproj, float* const* dw_part, void* stream);
/* The same for `ngroups` encoders that share (L, A, D) and every stride, one launch (grid = B x ngroups). */
int rfn_attn_bwd_grouped(int ngroups, const float* const* proj, int64_t proj_sb, int64_t proj_sl,
                         const float* const* hproj, const float* const* w_out, const float* const* alpha,
                         const float* const* att_seq, int64_t sb, int64_t sl, const float* const* dz, int64_t lddz,
                         int B, int L, int A, int D, float* const* dproj, int64_t dproj_sb, int64_t dproj_sl,
                         int accumulate_dproj, float* const* dhproj, float* const* dw_part, void* stream);
/* The same launch with dproj delivered as bf16 planes into k-slow plane images (rfn_x3_split_ks layout, one image per
 * encoder, row pitch ks_mp, this call's A columns at column ks_col0; k = b * L + l) instead of f32 -- the operand of
 * rfn_x3_gemm_ks for d att_2_att_h.weight.  A % 4 == 0 and 16-B aligned contiguous rows only. */
int rfn_attn_bwd_grouped_ks(int ngroups, const float* const* proj, int64_t proj_sb, int64_t proj_sl,
                            const float* const* hproj, const float* const* w_out, const float* const* alpha,
                            const float* const* att_seq, int64_t sb, int64_t sl, const float* const* dz, int64_t lddz,
                            int B, int L, int A, int D, void* const* ks_images, int ks_mp, int ks_col0,
                            float* const* dhproj, float* const* dw_part, void* stream);

/* Fused small-L (L <= 1024; meant for a handful) attention of up to RFN_MAX_ENC encoders in one launch: scores + softmax + context
 * (forward) and dalpha + softmax/tanh backward + d att_seq (backward), one block per (batch row, encoder).
 * Used by stage II and the decoder, which attend over the T1 / T2 thought vectors.  Arrays are host arrays of
 * device pointers, one entry per encoder; all encoders share strides and (L, A, D). */
int rfn_attn_small_fwd(int ngroups, const float* const* proj, int64_t proj_sb, int64_t proj_sl,
                       const float* const* hproj, const float* const* w_out, const float* const* b_out,
                       const float* const* att_seq, int64_t sb, int64_t sl, int B, int L, int A, int D,
                       float* const* alpha, float* const* z, int64_t ldz, void* stream);
/* datt_seq (may be NULL) is ACCUMULATED into: datt_seq[g][b,l,:] += alpha*dz; dproj/dhproj/dw_part as in
 * rfn_attn_scores_bwd. */
int rfn_attn_small_bwd(int ngroups, const float* const* proj, int64_t proj_sb, int64_t proj_sl,
                       const float* const* hproj, const float* const* w_out, const float* const* alpha,
                       const float* const* att_seq, int64_t sb, int64_t sl, const float* const* dz,
                       int64_t lddz, int B, int L, int A, int D, float* const* dproj, int64_t dproj_sb,
                       int64_t dproj_sl, int accumulate_dproj, float* const* dhproj, float* const* dw_part,
                       float* const* datt_seq, void* stream);

/* ---- decoder cell with z2h hoisted through the attention (csrc/rfn_deccell.hip, round 6) -------------------------------
 * misc/LSTMSoftAttentionCore.py:64-81 applies z2h to z = sum_l alpha_l v_l over the SAME thought vectors v with the SAME
 * weights on every step; z2h is linear, so z2h(z) = b_z + sum_l alpha_l U_l with U = v . W_z^T computed once per call.
 * rfn_dec_cell_fwd: one launch = scores (:64-75), softmax (:76), gates = (sum_l alpha_l U_l + b_z) + gates_in (:81; gates_in =
 *   i2h(x) + h2h(h) already in `gates`), LSTM update (:83-101) with the dropout mask of (seed, drop_offset); the gate
 *   activations overwrite `gates` (4R wide, 5R with maxout) as rfn_lstm_fwd leaves them.  proj / U rows of batch row b are
 *   those of row b / row_div (beam search: the rows of one image share its thought vectors).  alpha (B, L) out.
 * rfn_dec_attn_bwd: the attention backward of one step from that step's gate gradients: d alpha_l = <dgates, U_l>, softmax
 *   and tanh backward; dproj (accumulated when `accumulate`), dhproj, dw_part as rfn_attn_small_bwd.  GD = 4R or 5R.
 * rfn_dec_du: after the loop, dU[b, l, :] = sum_s alpha[s, b, l] * dgates[s, b, :] (alpha (S, B, L), dgates (S, B, GD)
 *   contiguous; dU strides as U). */
int rfn_dec_cell_fwd(const float* proj, int64_t psb, int64_t psl, const float* hproj, const float* w_out,
                     const float* b_out, const float* U, int64_t usb, int64_t usl, const float* bz, float* gates,
                     int64_t ldg, const float* c_prev, int64_t ldcp, float* c_next, int64_t ldcn, float* h_next,
                     int64_t ldh, float* alpha, int B, int L, int A, int R, int maxout, int row_div, float drop_p,
                     uint64_t seed, uint64_t drop_offset, void* stream);
int rfn_dec_attn_bwd(const float* proj, int64_t psb, int64_t psl, const float* hproj, const float* w_out,
                     const float* alpha, const float* U, int64_t usb, int64_t usl, const float* dgates, int64_t ldg,
                     int B, int L, int A, int GD, float* dproj, int64_t dpsb, int64_t dpsl, int accumulate,
                     float* dhproj, float* dw_part, void* stream);
int rfn_dec_du(const float* alpha, const float* dgates, int S, int B, int L, int GD, float* dU, int64_t usb,
               int64_t usl, void* stream);

/* ---- row-panel GEMM of the recurrences (csrc/rfn_cellgemm.hip) ------------------------------------------------------
 * One launch computes up to RFN_CELL_MAXOUT outputs over the same M (= batch) rows, each with its own destination,
 * weights and K segments:  C_o[M, N_o] (+)= sum_s A_os[M, K_os] * op(B_os) + sum_s bias_os.  Segments are rfn_gemm_seg
 * with a_kfast = 1; b_kfast = 1: B is an nn.Linear weight [N][K] (forward products, K1-K2 / K5 / K8 / K9 of SURVEY.md
 * 2.2); b_kfast = 0: B is [K][N] (dX = dY . W of the backward recurrences).  The K range of an output tile is cut across
 * the waves of ONE block and the partial tiles are added in LDS in wave order: deterministic, no scratch, and an
 * element's k order depends only on K.
 * epilogue RFN_CELL_EPI_STORE: plain store / accumulate (16-B coalesced).
 * epilogue RFN_CELL_EPI_LSTM (b_kfast = 1, N = 4R): C is the gate buffer (B, 4R) [in | forget | out | g]; the sums
 *   (+ C when accumulate: e.g. i2h(x) + h2h(h) produced earlier) go through the LSTM gate math of rfn_lstm_fwd inside
 *   the same launch: activations written back to C, c_next = f c_prev + i g, h_next = dropout(o tanh c_next) with the
 *   mask of (seed, drop_offset) (rfn_dropout_mask).  Replaces misc/RecurrentFusionModel.py:53-73,
 *   misc/LSTMSoftMultiAttentionFeatArrayNoInputCore.py:50-72, misc/LSTMSoftAttentionCore.py:81-101 after the attention.
 * epilogue RFN_CELL_EPI_LSTM_BWD (b_kfast = 0, N = R, one output): the product is the recurrent part of d h of an
 *   EARLIER cell call (the one the backward sweep processes next); the launch finishes that gradient,
 *   dh = product (+ C when accumulate) + dh_ext, and runs rfn_lstm_bwd of that call on it: `gates` holds its activations
 *   on entry and its gate gradients on exit, c_prev / c_next are its cell states, dc_next (may be NULL, may alias
 *   dc_prev) the incoming d c, dc_prev the outgoing one, (seed, drop_offset) its dropout mask.  C (may be NULL unless
 *   accumulate) receives dh.  Replaces the axpby + rfn_lstm_bwd pair at the head of every backward step.
 * Requirements (rfn_cell_gemm_supported; otherwise RFN_ERR_UNSUPPORTED and the caller uses rfn_gemm_f32 + rfn_lstm_*):
 * every K and N a multiple of 32, 16-B aligned operands with leading dimensions that are multiples of 4, R a multiple
 * of 8 for the gate epilogue; any M.  `variant` 0 lets the library pick: the K step and K-wave count -- which fix the k order
 * of every output element -- from the K counts alone, never from M, so a row's arithmetic does not depend on the batch it
 * sits in; the tile height may follow M (16-row tiles on wave-private ring slots, variant 6, for launches with few tiles:
 * the same fma chains, identical bits).  1..6 and 8 force a variant; 7 / 9 = the library's choice restricted to the
 * shared-slot forms 1..5 / to the 64- and 32-row shared-slot forms 1..3 (tests, tools, A/B); bits 4-7 force the ring depth
 * of variants 1..5 (tools).  RFN_CELL_VARIANT_DEEP in `variant` (A/B hook): a launch of the 32-row variant whose tiles do not
 * outnumber the device's CUs runs on the deep-ring kernel (8 ring slots, the whole K range of K <= 512 in flight, a K loop
 * without barriers) -- bit-identical results, measured slower inside the step (profiles/r05_chain.md). */
#define RFN_CELL_VARIANT_DEEP 256
#define RFN_CELL_MAXOUT 10
#define RFN_CELL_MAXSEG 8
#define RFN_CELL_EPI_STORE 0
#define RFN_CELL_EPI_LSTM 1
#define RFN_CELL_EPI_LSTM_BWD 2
typedef struct rfn_cell_out {
    float* C;
    int64_t ldc;
    int32_t N;
    int32_t accumulate;
    int32_t nseg;
    int32_t epilogue;            /* the same for every output of a launch */
    rfn_gemm_seg seg[RFN_CELL_MAXSEG];
    /* RFN_CELL_EPI_LSTM and _LSTM_BWD */
    const float* c_prev;         /* (M, R); forward: may alias c_next */
    int64_t ldcp;
    float* c_next;               /* forward: written; backward: read */
    int64_t ldcn;
    float* h_next;               /* forward only */
    int64_t ldh;
    uint64_t drop_offset;
    /* RFN_CELL_EPI_LSTM_BWD only */
    float* gates;                /* (M, 4R) activations in, gate gradients out */
    int64_t ldg;
    const float* dh_ext;         /* (M, R) gradient of that call's h from outside the recurrence, may be NULL */
    int64_t lddh;
    const float* dc_next;        /* may be NULL */
    int64_t lddcn;
    float* dc_prev;
    int64_t lddcp;
    /* RFN_CELL_EPI_STORE and _LSTM_BWD with accumulate: the sums start from C + S_0 + ... + S_{acc_parts-1}, added in that
     * order, where slab S_p has C's shape and leading dimension and starts at acc_slabs + p * acc_stride (acc_parts = 0: C
     * alone) -- the partial products of an earlier launch: the decoder's K-split d gates . W_hh computed beside the attention
     * backward (csrc/rfn_deccell.hip), the stage-I cells' d gates_j . W_H_j of small batches */
    const float* acc_slabs;
    int32_t acc_parts;
    int64_t acc_stride;
} rfn_cell_out;
int rfn_cell_gemm_supported(int M, int nout, const rfn_cell_out* outs_host, int R);
int rfn_cell_gemm(int M, int nout, const rfn_cell_out* outs_host, int R, float drop_p, uint64_t seed, int variant,
                  void* stream);

/* Dropout masks of the path (nn.Dropout of the three cells: misc/RecurrentFusionModel.py:70,
 * misc/LSTMSoftMultiAttentionFeatArrayNoInputCore.py:69, misc/LSTMSoftAttentionCore.py:98).  A mask is never stored:
 * unit j of batch row b of one cell call is kept iff philox4x32-10(key = seed, counter = (b * R + j, offset)) >= p, with
 * `seed` the 64-bit seed the caller hands to rfn_prefix_fwd / rfn_decoder_fwd / rfn_decoder_step and `offset` the call
 * site: stage-I cell (step t, encoder i) -> t * M + i (p = drop_fusion); stage-II cell t -> RFN_DROP_OFFSET_STAGE2 + t
 * (p = drop_reason); decoder step s -> RFN_DROP_OFFSET_DECODER + s (p = drop_lm).  Forward and backward regenerate it.
 * rfn_dropout_mask writes the keep mask of one call site (n = B * R values, 1.0f = kept, 0.0f = dropped) so that a
 * test can hand the product's own masks to the CPU oracle. */
#define RFN_DROP_OFFSET_STAGE2 (1ull << 20)
#define RFN_DROP_OFFSET_DECODER (1ull << 21)
int rfn_dropout_mask(uint64_t seed, uint64_t offset, int64_t n, float drop_p, float* keep_out, void* stream);

/* LSTM gate epilogue shared by the three cells (misc/RecurrentFusionModel.py:55-73,
 * misc/LSTMSoftMultiAttentionFeatArrayNoInputCore.py:54-72, misc/LSTMSoftAttentionCore.py:83-101):
 * gates[b, 0:4R] = [in | forget | out | g] pre-activations on entry, activations on exit;
 * c_next = f*c_prev + i*g; h_next = dropout(o*tanh(c_next)).  maxout != 0: gates are 5R wide,
 * g = max(chunk 3, chunk 4) without tanh (LSTMSoftMultiAttention...py:60-62, LSTMSoftAttentionCore.py:89-91).  drop_p > 0 draws a Philox mask from
 * (seed, offset) that rfn_lstm_bwd regenerates. */
int rfn_lstm_fwd(float* gates, int64_t ldg, const float* c_prev, int64_t ldcp, float* c_next,
                 int64_t ldcn, float* h_next, int64_t ldh, int B, int R, int maxout, float drop_p,
                 uint64_t seed, uint64_t offset, void* stream);
/* G independent cells in one launch (the M encoders of a stage-I step): group g uses gates + g*gs_gates,
 * c_prev + g*gs_cprev, c_next + g*gs_cnext, h_next + g*gs_h and dropout stream offset + g. */
int rfn_lstm_fwd_grouped(float* gates, int64_t ldg, const float* c_prev, int64_t ldcp, float* c_next,
                         int64_t ldcn, float* h_next, int64_t ldh, int B, int R, int maxout, float drop_p,
                         uint64_t seed, uint64_t offset, int G, int64_t gs_gates, int64_t gs_cprev,
                         int64_t gs_cnext, int64_t gs_h, void* stream);
/* dgates (in place over the activations), dc_prev = dc_next_total * f.
 * dh / dc_next are the TOTAL incoming gradients of h_next / c_next (dc_next may be NULL). */
int rfn_lstm_bwd(float* gates, int64_t ldg, const float* c_prev, int64_t ldcp, const float* c_next,
                 int64_t ldcn, const float* dh, int64_t lddh, const float* dc_next, int64_t lddcn,
                 float* dc_prev, int64_t lddcp, int B, int R, int maxout, float drop_p, uint64_t seed,
                 uint64_t offset, void* stream);

/* grouped backward: c_prev and c_next share gs_c; dc_next and dc_prev share gs_dc */
int rfn_lstm_bwd_grouped(float* gates, int64_t ldg, const float* c_prev, int64_t ldcp, const float* c_next,
                         int64_t ldcn, const float* dh, int64_t lddh, const float* dc_next, int64_t lddcn,
                         float* dc_prev, int64_t lddcp, int B, int R, int maxout, float drop_p,
                         uint64_t seed, uint64_t offset, int G, int64_t gs_gates, int64_t gs_c, int64_t gs_dh,
                         int64_t gs_dc,
                         void* stream);

/* nn.Embedding gather (misc/RecurrentFusionModel.py:276): out[r,:] = W[id(r), :] with
 * id(r) = ids[(r % inner)*ids_s_inner + (r / inner)*ids_s_outer]  (row r = (step, batch) reads
 * ids[batch, step] with inner = B, ids_s_inner = ld_ids, ids_s_outer = 1). */
int rfn_embed_fwd(const float* W, int E, int64_t V1, const int64_t* ids, int inner,
                  int64_t ids_s_inner, int64_t ids_s_outer, int rows, float* out, int64_t ldo,
                  void* stream);
/* dW[v,:] = sum_{r: id(r)==v} dout[r,:]  (fixed order, overwrites all of dW) */
int rfn_embed_bwd(const float* dout, int64_t ldo, const int64_t* ids, int inner,
                  int64_t ids_s_inner, int64_t ids_s_outer, int rows, int E, int64_t V1, float* dW,
                  void* stream);

/* F.log_softmax over the vocabulary (misc/RecurrentFusionModel.py:278).  Row r of `logits`
 * (ld = ldl) is written to out + (r % inner)*out_s_inner + (r / inner)*out_s_outer, which turns
 * the path's time-major (step, batch) rows into the reference's (batch, step, V+1) layout. */
int rfn_log_softmax_fwd(const float* logits, int64_t ldl, int rows, int V1, int inner,
                        int64_t out_s_inner, int64_t out_s_outer, float* out, void* stream);
/* dlogits[r,:] = g[r',:] - exp(logp[r',:]) * sum_v g[r',v]   (r' = the same row mapping) */
int rfn_log_softmax_bwd(const float* g, const float* logp, int rows, int V1, int inner,
                        int64_t s_inner, int64_t s_outer, float* dlogits, int64_t ldd, void* stream);

/* max over steps of the reason heads (misc/RecurrentFusionModel.py:229,253):
 * out[b,k] = max_t X[t,b,k]; arg[b,k] = first t attaining it. */
int rfn_max_over_steps_fwd(const float* X, int T, int B, int K, float* out, int32_t* arg,
                           void* stream);
/* dX[t,b,k] = (arg[b,k]==t) ? dout[b,k] : 0 */
int rfn_max_over_steps_bwd(const float* dout, const int32_t* arg, int T, int B, int K, float* dX,
                           void* stream);
/* The same for `ngroups` heads in one launch: X slabs (T,B,K) back to back, out / arg / dout (B,K) back to back. */
int rfn_max_over_steps_fwd_grouped(const float* X, int T, int B, int K, float* out, int32_t* arg, int ngroups,
                                   void* stream);
int rfn_max_over_steps_bwd_grouped(const float* dout, const int32_t* arg, int T, int B, int K, float* dX,
                                   int ngroups, void* stream);

/* y[r,c] = alpha*x[r,c] + beta*y[r,c]  (state mean misc/RecurrentFusionModel.py:233-235,
 * gradient fan-in) */
int rfn_axpby_2d(float alpha, const float* x, int64_t ldx, float beta, float* y, int64_t ldy,
                 int rows, int cols, void* stream);

/* y[r,c] = y[r,c] / divisor (IEEE division, as `sum / num_feat_array` in the reference) */
int rfn_div_2d(float* y, int64_t ldy, int rows, int cols, float divisor, void* stream);
/* State mean over the M encoder slices of a concatenated (rows, G*gstride) state (misc/RecurrentFusionModel.py:233-235):
 * y_p[r,c] = (x_p[r,c] + x_p[r,gstride+c] + ...) / G, summed in slice order then divided; up to two (x, y) pairs
 * (h and c) per launch.  Arrays are host arrays of device pointers. */
int rfn_mean_over_groups(int npairs, const float* const* x, int64_t ldx, int64_t gstride, int G, float* const* y,
                         int64_t ldy, int rows, int cols, void* stream);
/* Its backward: y_p[r, g*gstride + c] = alpha*x_p[r,c] + beta[p]*y_p[r, g*gstride + c] for every slice g. */
int rfn_bcast_to_groups(int npairs, float alpha, const float* const* x, int64_t ldx, const float* beta,
                        float* const* y, int64_t ldy, int64_t gstride, int G, int rows, int cols, void* stream);

/* ---- criteria (misc/utils.py) --------------------------------------------------------------- */
/* ReviewNetEnsembleCriterion language term (misc/utils.py:163-184): loss_out[0] (+)=
 * -sum_{b,t} mask[b,t]*q[b,t,:].logp[b,t,:] / B with q one-hot (eps = 0) or label-smoothed, and
 * dlogp (same layout as logp, overwritten) = d loss / d logp * gscale.  Either output may be NULL. */
int rfn_xe_loss(const float* logp, int B, int T, int V1, const int64_t* target, int64_t ld_target,
                const float* mask, int64_t ld_mask, float eps, float gscale,
                float* scratch /* B*T floats, needed when loss_out != NULL */, float* loss_out,
                int accumulate_loss, float* dlogp, void* stream);
/* Same, with the upstream gradient also read from a device scalar (autograd hands d loss as a 0-dim tensor):
 * dlogp is scaled by gscale * gscale_dev[0]; gscale_dev may be NULL. */
int rfn_xe_loss_ex(const float* logp, int B, int T, int V1, const int64_t* target, int64_t ld_target,
                   const float* mask, int64_t ld_mask, float eps, float gscale, const float* gscale_dev,
                   float* scratch, float* loss_out, int accumulate_loss, float* dlogp, void* stream);
/* The same language term straight from the LOGITS (log-softmax / NLL / label smoothing and their backward
 * without the (B, T, V+1) log_prob and d log_prob tensors; F.log_softmax of misc/RecurrentFusionModel.py:276 + misc/utils.py:
 * 163-184).  `logits`: time-major rows r = t * B + b with row stride ldl (the layout of the decoder workspace,
 * rfn_decoder_logits).  _fwd: lse[r] (T*B floats) = the row's logsumexp, loss_out[0] (+)= -sum mask * q . (x - lse) / B;
 * scratch: B*T floats.  _bwd overwrites the logits with d loss / d logits * gscale * gscale_dev[0] (gscale_dev may be NULL). */
int rfn_xe_logits_fwd(const float* logits, int64_t ldl, int B, int T, int V1, const int64_t* target, int64_t ld_target,
                      const float* mask, int64_t ld_mask, float eps, float* lse, float* scratch, float* loss_out,
                      int accumulate_loss, void* stream);
int rfn_xe_logits_bwd(float* logits, int64_t ldl, int B, int T, int V1, const int64_t* target, int64_t ld_target,
                      const float* mask, int64_t ld_mask, float eps, const float* lse, float gscale,
                      const float* gscale_dev, void* stream);
/* ReviewNetRewardCriterion policy + entropy terms (misc/utils.py:50-72):
 * loss_out[0] (+)= [ -sum pol(b,t)*mask(b,t) + entropy_reg * sum mask0(b,t) * sum_v lp*exp(lp) ] / B with
 * mask0 = seq > 0, mask = [1, mask0[:, :-1]], pol = input*reward or the reference's PPO-clip surrogate.
 * d_input (B,T) and d_logprobs_all rows t < T (same strides convention) are overwritten when non-NULL. */
int rfn_rl_loss(const float* input, int64_t ld_in, const int64_t* seq, int64_t ld_seq, const float* reward,
                int64_t ld_rw, const float* logprobs_all, int64_t lp_sb, int64_t lp_st, int B, int T, int V1,
                float entropy_reg, const float* old_logprobs, int64_t ld_old, int use_ppo, float ppo_clip,
                float* scratch /* B*T floats when loss_out != NULL */, float* loss_out, int accumulate_loss,
                float* d_input, int64_t ld_din, float* d_logprobs_all, int64_t dlp_sb, int64_t dlp_st,
                void* stream);
/* Same, for an autograd host: the upstream gradient is read from a device scalar (d_input and d_logprobs_all are scaled
 * by gscale_dev[0]; NULL = 1), and logprobs_all / d_logprobs_all hold T_all >= T steps per caption (sample() returns
 * seq_length + 1 of them, misc/RecurrentFusionModel.py:655-658): rows t in [T, T_all) of d_logprobs_all are written as
 * zeros by the same launch, so the caller needs neither a zero fill nor a scaling pass over the (B, T_all, V+1) tensor. */
int rfn_rl_loss_ex(const float* input, int64_t ld_in, const int64_t* seq, int64_t ld_seq, const float* reward,
                   int64_t ld_rw, const float* logprobs_all, int64_t lp_sb, int64_t lp_st, int B, int T, int T_all, int V1,
                   float entropy_reg, const float* old_logprobs, int64_t ld_old, int use_ppo, float ppo_clip,
                   const float* gscale_dev, float* scratch, float* loss_out, int accumulate_loss, float* d_input,
                   int64_t ld_din, float* d_logprobs_all, int64_t dlp_sb, int64_t dlp_st, void* stream);
/* nn.MultiLabelMarginLoss, mean reduction (misc/utils.py:186-190), scaled by `scale`:
 * loss_out[0] (+)= scale * MLM(pred, target); dpred (overwritten) = scale * gscale * dMLM/dpred. */
int rfn_multilabel_margin(const float* pred, int B, int K, const int64_t* target, float scale,
                          float gscale, float* scratch /* B floats when loss_out != NULL */,
                          float* loss_out, int accumulate_loss, float* dpred, void* stream);

/* All M+1 reasoning heads of ReviewNetEnsembleCriterion in one launch (misc/utils.py:186-190: the per-head
 * losses are added one after the other, which the fixed-order finish reproduces).  preds / dpreds: host arrays of
 * nheads device pointers ((B,K) each; dpreds or its entries may be NULL); scratch: nheads*B floats when loss_out. */
int rfn_multilabel_margin_grouped(int nheads, const float* const* preds, int B, int K, const int64_t* target,
                                  float scale, float gscale, const float* gscale_dev, float* scratch,
                                  float* loss_out, int accumulate_loss, float* const* dpreds, void* stream);

/* clip_gradient + Adam with L2 weight decay (misc/utils.py:292-296, train.py:69-71,162-163),
 * one pass: g = clamp(g, +-clip) + wd*p; m,v update; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps).
 * grad_scale multiplies g first (1/world_size after a sum all-reduce). */
int rfn_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, float grad_clip, float grad_scale,
                  int step, void* stream);
/* The same update of up to RFN_ADAM_MAXBUCKET flat buckets in ONE launch (same hyper-parameters and step for all, as
 * torch.optim.Adam applies them to the reference's single param group, train.py:69-71): bit-identical to one rfn_adam_step
 * per bucket, without the per-launch ramp and tail (2.15 -> 1.95 ms at C3).  Host arrays of nbuckets entries. */
#define RFN_ADAM_MAXBUCKET 16
int rfn_adam_step_multi(int nbuckets, float* const* p, const float* const* g, float* const* m, float* const* v,
                        const int64_t* n_host, float lr, float beta1, float beta2, float eps, float weight_decay,
                        float grad_clip, float grad_scale, int step, void* stream);
/* The same launch with its two step-dependent scalars read from DEVICE memory: coef_dev[0] = lr / (1 - beta1^step),
 * coef_dev[1] = 1 / sqrt(1 - beta2^step), computed by the host exactly as rfn_adam_step_multi computes them (in double, then
 * rounded to float) and written there before the launch.  Everything else of the call is step-independent, so a HIP graph
 * that captured it can be replayed for every step (kernel arguments are frozen by a capture; train.py:162-163 steps the
 * optimizer once per iteration).  Bit-identical to rfn_adam_step_multi for the same coefficients. */
int rfn_adam_step_multi_coef(int nbuckets, float* const* p, const float* const* g, float* const* m, float* const* v,
                             const int64_t* n_host, const float* coef_dev, float beta1, float beta2, float eps,
                             float weight_decay, float grad_clip, float grad_scale, void* stream);

/* greedy pick of sample() (misc/RecurrentFusionModel.py:619-649) for one step t >= 1:
 * it = argmax_v logp[b,:] (first maximum), lp_out[b] = that value,
 * unf_out[b] = (t==1 ? 1 : unf_prev[b]) & (it>0), seq_out[b] = it*unf_out[b],
 * next_ids[b] = it (UNMASKED: the reference embeds the raw argmax, :637).
 * Keeping one unf_out row per step lets the host apply the reference's early exit (:645) with a
 * single device read after the loop instead of one sync per step. */
int rfn_greedy_pick(const float* logp, int64_t ldl, int B, int V1, int t, int64_t* next_ids,
                    int64_t* seq_out, int64_t ld_seq, float* lp_out, int64_t ld_lp,
                    const int32_t* unf_prev, int32_t* unf_out, void* stream);

/* The same bookkeeping for a token chosen elsewhere (given != NULL, e.g. a multinomial draw; may alias next_ids):
 * it = given[b], lp_out[b] = logp[b, it], unf / seq / next_ids as above.  given == NULL: rfn_greedy_pick. */
int rfn_pick_record(const float* logp, int64_t ldl, int B, int V1, int t, const int64_t* given, int64_t* next_ids,
                    int64_t* seq_out, int64_t ld_seq, float* lp_out, int64_t ld_lp, const int32_t* unf_prev,
                    int32_t* unf_out, void* stream);

/* multinomial pick of sample() (misc/RecurrentFusionModel.py:623-631) and of scheduled sampling (:260-270), on the
 * device: ids[b] = inverse-CDF draw from p[v] ~ exp(logp[b,v] * inv_temperature) with the caller's uniform u[b] in
 * [0,1) -- a deterministic function of (logp, u).  With coin != NULL only rows with coin[b] < keep_prob are redrawn,
 * the others keep the token ids[b] already holds (the scheduled-sampling mask).  Element b of ids lives at
 * ids[b * ld_ids].  (The reference draws on the host with torch.multinomial; RNG streams are not portable, the
 * distribution is the same.) */
int rfn_multinomial_pick(const float* logp, int64_t ldl, int B, int V1, float inv_temperature, const float* u,
                         const float* coin, float keep_prob, int64_t* ids, int64_t ld_ids, void* stream);

/* Batched beam-search bookkeeping of sample_beam (misc/RecurrentFusionModel.py:451-531) for step t in [1, S]:
 * one block per image consumes the log-probs of its W beam rows (rows k*W .. k*W+W-1 of `logp`), reproduces the
 * reference's candidate order / stable sort / fork / done-beam rules on the device and emits
 *   order[r]    source row whose recurrent state row r continues (feed rfn_gather_rows),
 *   next_ids[r] token row r feeds to the next decoder step,
 * updating beam_seq / beam_lp (S, NB, W), beam_sum (NB, W), the done-beam arrays (NB, max_done, ...) and
 * active[k] (0 once image k has no candidate left, :480).  W <= 16 (rfn_beam_step_topk: W <= 32), S <= 64. */
int rfn_beam_step(const float* logp, int64_t ldl, int V1, int W, int S, int t, int NB, int max_done,
                  int64_t* beam_seq, float* beam_lp, float* beam_sum, int32_t* order, int64_t* next_ids,
                  int64_t* done_seq, float* done_lp, float* done_p, int32_t* done_n, int32_t* active,
                  void* stream);
/* The same step fed with every beam row's W best log-probs and tokens (rfn_log_softmax_topk, (NB * W, W) each) instead
 * of the full rows -- all the reference ever reads of them (:463-466). */
int rfn_beam_step_topk(const float* topv, const int32_t* topi, int V1, int W, int S, int t, int NB, int max_done,
                       int64_t* beam_seq, float* beam_lp, float* beam_sum, int32_t* order, int64_t* next_ids,
                       int64_t* done_seq, float* done_lp, float* done_p, int32_t* done_n, int32_t* active, void* stream);
/* log-softmax of every row + its W <= 32 best entries, ordered (value descending, token ascending) as a descending
 * sort lists them; the log-prob bits are those rfn_log_softmax_fwd writes, which are not materialised here. */
int rfn_log_softmax_topk(const float* logits, int64_t ldl, int rows, int V1, int W, float* topv, int32_t* topi,
                         void* stream);
/* dst[r,:] = src[order[r],:]  (src != dst) */
int rfn_gather_rows(const float* src, float* dst, const int32_t* order, int rows, int R, void* stream);

/* ---- whole-path entry points ---------------------------------------------------------------- */
/* Phase 1 = get_init_state + get_thought_vectors (misc/RecurrentFusionModel.py:333-343, 283-331;
 * the same code is inlined in forward :199-255 and sample :557-612): fc2h, T1 fusion-stage-I steps
 * over M encoders, reason heads, state mean, T2 stage-II steps.
 * Inputs : params (rfn_param_count pointers), fc_feats[i] (B,F_i), att_feats[i] (B,L_i,D_i).
 * Outputs: comb (T2,B,R) time-major thought_vectors_comb; h_out,c_out (B,R) = state_review;
 *          reason_pred (M+1,B,K).
 * `ws` keeps every activation rfn_prefix_bwd needs (size rfn_prefix_ws_bytes; with train = 0 a
 * smaller inference workspace suffices and rfn_prefix_bwd must not be called). */
size_t rfn_prefix_ws_bytes(const rfn_dims* d, int B, int train);
/* `train` selects the workspace layout (activations kept for rfn_prefix_bwd); dropout is applied whenever the
 * probabilities in `d` are non-zero (masks from `seed`), with or without `train`. */
int rfn_prefix_fwd(const rfn_dims* d, int B, const float* const* params,
                   const float* const* fc_feats, const float* const* att_feats, float* comb,
                   float* h_out, float* c_out, float* reason_pred, void* ws, size_t ws_bytes,
                   int train, uint64_t seed, void* stream);
/* get_thought_vectors(fc_feats, att_feats, state_list) with the caller's stage-I state (:283-331): init_h[i],
 * init_c[i] are (B,R) contiguous, one pair per encoder; inference only (workspace of train = 0). */
int rfn_prefix_fwd_from_state(const rfn_dims* d, int B, const float* const* params,
                              const float* const* init_h, const float* const* init_c,
                              const float* const* att_feats, float* comb, float* h_out, float* c_out,
                              float* reason_pred, void* ws, size_t ws_bytes, void* stream);
/* Gradients of phase 1.  d_comb (T2,B,R), d_h, d_c (B,R), d_reason_pred (M+1,B,K) are the incoming
 * gradients (any may be NULL = zero).  Every slot of grads[] (same table as params) that belongs
 * to phase 1 is OVERWRITTEN exactly once (no zero-fill needed); phase-2 slots are not touched. */
int rfn_prefix_bwd(const rfn_dims* d, int B, const float* const* params,
                   const float* const* fc_feats, const float* const* att_feats,
                   const float* d_comb, const float* d_h, const float* d_c,
                   const float* d_reason_pred, float* const* grads, void* ws, size_t ws_bytes,
                   uint64_t seed, int defer_wgrad, void* stream);
/* With defer_wgrad != 0, rfn_prefix_bwd leaves out the stage-I weight gradients of the encoders
 * (review_steps_individual.{t}.lstm.{enc}.{att_model.att_2_att_h, att_model.h_2_att_h, H2h, z2h}.{weight,bias}
 * for all t); this call produces them for one encoder from the same workspace.  `parts` bit 0: H2h, z2h,
 * h_2_att_h (0.27 GB of gradients, short GEMMs); bit 1: att_2_att_h (34 MB, the longest GEMM of backward).
 * A data-parallel host issues part 1, all-reduces that bucket under part 2's GEMM, and so on: only the last
 * encoder's 34 MB bucket remains exposed at the end of backward.  Part 1 of an encoder must be issued before its
 * part 2: att_2_att_h.bias and h_2_att_h.bias have the same gradient, which part 1 computes once and writes to both. */
int rfn_prefix_bwd_wgrad(const rfn_dims* d, int B, const float* const* att_feats, float* const* grads,
                         void* ws, size_t ws_bytes, int enc, int parts, void* stream);

/* Phase 2 = the teacher-forced decoder loop of forward() (misc/RecurrentFusionModel.py:257-281):
 * for s < S: xt = embed(ids[b,s]); decoder cell (misc/LSTMSoftAttentionCore.py:60-102);
 * log_prob[b,s,:] = log_softmax(logit(h)).  `ids` is (B, ld_ids) int64, column s feeds step s.
 * The caller derives S from the reference's break rule (:274).
 * log_prob may be NULL: the pass then stops at the logits, which stay in the workspace as time-major rows r = s * B + b of
 * V1 floats (rfn_decoder_logits returns their address) -- the operand of rfn_xe_logits_fwd / _bwd, the loss-only form of
 * the step (RecurrentFusionModel.forward_loss). */
size_t rfn_decoder_ws_bytes(const rfn_dims* d, int B, int S, int train);
float* rfn_decoder_logits(const rfn_dims* d, int B, int S, int train, void* ws);
int rfn_decoder_fwd(const rfn_dims* d, int B, int S, const float* const* params, const float* comb,
                    const float* h0, const float* c0, const int64_t* ids, int64_t ld_ids,
                    float* log_prob /* (B,S,V1) */, void* ws, size_t ws_bytes, int train,
                    uint64_t seed, void* stream);
/* The same pass one step at a time, for scheduled sampling (misc/RecurrentFusionModel.py:260-270: the token fed at
 * step s may be drawn from the distribution of step s-1, so the host interleaves its draws with the steps).
 * rfn_decoder_fwd_begin + rfn_decoder_fwd_step for s = 0 .. S-1 leave `ws` and log_prob[:, 0..S-1, :] exactly as
 * rfn_decoder_fwd on the final ids does (bit for bit), so rfn_decoder_bwd runs on the result unchanged: the sampled
 * pass IS the differentiated pass.  ids_s points at the B tokens of step s (element b at ids_s[b * ld_ids]). */
int rfn_decoder_fwd_begin(const rfn_dims* d, int B, int S, const float* const* params, const float* comb,
                          const float* h0, const float* c0, void* ws, size_t ws_bytes, int train, void* stream);
int rfn_decoder_fwd_step(const rfn_dims* d, int B, int S, int s, const float* const* params, const float* comb,
                         const int64_t* ids_s, int64_t ld_ids, float* log_prob /* (B,S,V1) base */, void* ws,
                         size_t ws_bytes, int train, uint64_t seed, void* stream);
/* d_log_prob (B,S,V1) in; d_comb (T2,B,R), d_h0, d_c0 (B,R) out (overwritten); the phase-2 slots of
 * grads[] (embed, logit, decoder.*) are overwritten.  log_prob == d_log_prob == NULL: the workspace's logits rows
 * already hold d loss / d logits (rfn_xe_logits_bwd wrote them there). */
int rfn_decoder_bwd(const rfn_dims* d, int B, int S, const float* const* params, const float* comb,
                    const float* h0, const float* c0, const int64_t* ids, int64_t ld_ids,
                    const float* log_prob, const float* d_log_prob, float* d_comb, float* d_h0,
                    float* d_c0, float* const* grads, void* ws, size_t ws_bytes, uint64_t seed,
                    void* stream);

/* One free-running decoder step for sample()/sample_beam()/one_time_step
 * (misc/RecurrentFusionModel.py:345-350, 616-653, 526-527): state (h,c) is updated in place,
 * logits (B,V1) pre-softmax and/or logp (B,V1) are written when non-NULL.
 * cproj (rfn_decoder_cproj_floats(d, B) floats) must have been filled by rfn_decoder_prepare: the loop-invariant products of
 * the thought vectors, [att_2_att_h(comb) (T2*B, A) | U = comb . z2h.weight^T (T2*B, 4R or 5R), 256-B aligned] -- the
 * reference recomputes the first every step (misc/LSTMSoftAttentionCore.py:64-66) and applies z2h to the context every step
 * (:81); z2h(sum_l alpha_l v_l) = z2h.bias + sum_l alpha_l U_l (csrc/rfn_deccell.hip).
 * The step is computed with the operation sequence of step `step` of rfn_decoder_fwd: with d->drop_lm > 0 it applies
 * the dropout mask of (seed, step) that rfn_decoder_fwd(train, seed) applies at that step, and its log-probs are
 * bit-identical to the teacher-forced pass fed the same tokens -- the reference samples (scheduled sampling :260-270,
 * multinomial sample :623-631) from the very distribution it differentiates.  Hosts pass drop_lm = 0 outside
 * training mode. */
size_t rfn_decoder_step_ws_bytes(const rfn_dims* d, int B);
size_t rfn_decoder_cproj_floats(const rfn_dims* d, int B);
int rfn_decoder_prepare(const rfn_dims* d, int B, const float* const* params, const float* comb,
                        float* cproj, void* stream);
int rfn_decoder_step(const rfn_dims* d, int B, const float* const* params, const float* comb,
                     const float* cproj, const int64_t* ids, float* h, float* c, float* logits,
                     float* logp, int64_t ld_logp /* row stride of logp, >= V1 */, void* ws,
                     size_t ws_bytes, uint64_t seed, int step, void* stream);
/* The same step fed with an already embedded token xt (B, E) -- the literal signature of the reference's
 * one_time_step(xt, thought_vectors_comb, state) (misc/RecurrentFusionModel.py:345-350), whose callers do
 * xt = model.embed(it) themselves (eval_utils.py:368,516,539). */
int rfn_decoder_step_embedded(const rfn_dims* d, int B, const float* const* params, const float* comb,
                              const float* cproj, const float* xt, int64_t ld_xt, float* h, float* c,
                              float* logits, float* logp, int64_t ld_logp, void* ws, size_t ws_bytes,
                              uint64_t seed, int step, void* stream);

/* ---- whole decode loops queued by ONE call: the device-resident decoder loop (SURVEY.md 8b "decoder_loop") ------------
 * Each is the sequence of per-step launches its host loop would issue (embedding K10, decoder cell a5, logit +
 * log-softmax K11, the pick / beam bookkeeping between steps) with the host removed from the loop: nothing is read back,
 * nothing is decided on the host between steps, results are bit for bit those of the step-by-step calls.
 * They are NOT one persistent kernel, on purpose: a decoder step is three dependent all-to-all products, and on MI355X a
 * grid-wide barrier inside a launch (4-7 us) costs more than the kernel boundary it replaces (1.5-2 us) -- the same
 * trade the in-launch split-K finish lost when it was measured (RFN_GEMM_OPT_SPLITK_IN_KERNEL); DESIGN.md section 13.
 *
 * rfn_decoder_loop: sample() free-running decode (misc/RecurrentFusionModel.py:616-653), `steps` = seq_length + 1 steps.
 *   mode 0 greedy (argmax of the previous step's log-probs, first maximum), mode 1 multinomial (inverse-CDF draw with the
 *   caller's uniforms u[(t-1) * B + b], temperature 1 / inv_temperature).  h, c: decoder state in / out (B, R).
 *   logp_all[b * ld_b + t * ld_t + v]: log-probs of step t.  seq / seq_lp (B, steps - 1) with row strides ld_seq / ld_lp:
 *   token (0 once the row has finished) and its log-prob per step, as :647-649.  unf (steps, B) int32: unfinished flags
 *   per step (row 0 unused), for the caller's single early-exit read-back (:645).  ids: B int64 of scratch (last fed
 *   tokens).  ws: rfn_decoder_step_ws_bytes. */
int rfn_decoder_loop(const rfn_dims* d, int B, int steps, const float* const* params, const float* comb,
                     const float* cproj, float* h, float* c, int mode, float inv_temperature, const float* u,
                     float* logp_all, int64_t ld_b, int64_t ld_t, int64_t* seq, int64_t ld_seq, float* seq_lp,
                     int64_t ld_lp, int32_t* unf, int64_t* ids, void* ws, size_t ws_bytes, uint64_t seed, void* stream);
/* rfn_decoder_fwd_sampled: the step-wise TRAINING decoder with draws between the steps (scheduled sampling :260-270 with
 *   probability ss_prob; ss_prob = 1: the multinomial sample() with grad, train_rl.py:160) = rfn_decoder_fwd_begin + S x
 *   (rfn_multinomial_pick, rfn_decoder_fwd_step).  ids (B, S) in / out (column 0 = BOS is never redrawn); u_draw, u_coin
 *   (S, B) uniforms in [0, 1) (row 0 unused).  Leaves the workspace ready for rfn_decoder_bwd. */
int rfn_decoder_fwd_sampled(const rfn_dims* d, int B, int S, const float* const* params, const float* comb,
                            const float* h0, const float* c0, int64_t* ids, int64_t ld_ids, float ss_prob,
                            float inv_temperature, const float* u_draw, const float* u_coin, float* log_prob, void* ws,
                            size_t ws_bytes, int train, uint64_t seed, void* stream);
/* rfn_beam_loop: sample_beam's search (:451-531) for NB images x W beams: S x (rfn_beam_step, two rfn_gather_rows,
 *   rfn_decoder_step on the NB * W rows, ending in rfn_log_softmax_topk instead of the full log-softmax).  comb (T2, NB, R)
 *   and cproj = rfn_decoder_prepare(d, NB, ...) are per IMAGE: the W beam rows of image k read row k (the reference feeds
 *   beam_size copies of the image's thought vectors, :417-420).  h / c
 *   (NB * W, R) in / out, h_alt / c_alt same-size scratch, logp: 2 * NB * W * W floats of scratch (the rows' top-W lists);
 *   the beam / done arrays as rfn_beam_step (zero-initialised by the caller; active = 1). */
int rfn_beam_loop(const rfn_dims* d, int NB, int W, int S, const float* const* params, const float* comb,
                  const float* cproj, float* h, float* c, float* h_alt, float* c_alt, float* logp, int64_t* beam_seq,
                  float* beam_lp, float* beam_sum, int32_t* order, int64_t* ids, int64_t* done_seq, float* done_lp,
                  float* done_p, int32_t* done_n, int32_t* active, int max_done, void* ws, size_t ws_bytes, uint64_t seed,
                  void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RFN_H_ */
