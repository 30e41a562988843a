// Row-panel GEMM of the recurrences: the per-step products of the three cells (M = batch rows, a few hundred) with their
// epilogues fused, so that one cell step is three launches instead of six or seven.
//
// Replaces, inside the step loops of rfn_path.hip, the sequences
//     gemm (64x64 tiles, K cut across blocks) -> rfn_gemm_reduce_k -> lstm_fwd_k               (forward)
//     axpby -> lstm_bwd_k -> gemm -> rfn_gemm_reduce_k -> ... -> gemm -> rfn_gemm_reduce_k      (backward)
// of misc/RecurrentFusionModel.py:53-73 (stage I gates), misc/LSTMSoftMultiAttentionFeatArrayNoInputCore.py:46-72
// (stage II) and misc/LSTMSoftAttentionCore.py:76-101 (decoder).
//
// What is different from rfn_gemm.hip (which stays for the long products):
//   * ONE launch computes several outputs that have different weights and destinations (h_2_att_h of every encoder and
//     h2h share the operand h; dz of every encoder and the recurrent dh share the gate gradients);
//   * a per-step product has too few output tiles for 256 CUs and a K chain that is all latency (a wave's 32x32 tile
//     takes 32 cycles per k on the f32 matrix pipe): the K range is cut across the WAVES OF ONE BLOCK (k-groups of 8
//     dealt to the waves in adjacent pairs, cg_kgroup()), partial tiles meet in LDS and are added in wave order -- no partial slabs in HBM, no second
//     kernel, no atomics; the k order of an output element depends on nothing but the K step and the wave count of the
//     tile variant, which the host picks from (N, K) only -- never from the batch size -- so the free-running, step-wise
//     and batched decoder passes stay bit-identical to each other and rows stay independent of their batch;
//   * the block's finished tile sits in LDS, so the epilogue is ordinary row-major code: 16-B coalesced stores, the
//     LSTM gate math itself (tile columns = [in | forget | out | g] x the tile's units, so a thread sees all four gates
//     of its unit) with the Philox dropout mask of rfn_cell.hip, or -- in the backward recurrences -- the LSTM backward
//     of the NEXT step to be processed (the product is the recurrent dh that step is waiting for).  Everything an
//     epilogue reads from global memory is requested before the K loop, so it arrives under the loop;
//   * operands go global -> LDS by LDS-DMA (1-KiB pieces, XOR swizzle on the source address, 3-slot ring, one barrier
//     per K step) as in gemm_tile_dma; ragged row counts are handled by clamping the source row (the extra rows are
//     computed and dropped), so any batch size takes this path.
// Numerics: v_mfma_f32_32x32x2_f32, i.e. fp32 fma chains; sums of WK chains + bias (+ previous C), fixed order.
#include <string.h>

#include "rfn_cellgemm_body.h"
#include "rfn_internal.h"


struct CgDevStateRows {
    bool set[16] = {};
};

template <int BM, int BK, int WK, bool BKF, int EPI, bool WP = false>
__global__ __launch_bounds__(64 * (BM >= 32 ? BM / 32 : 1) * WK) void cell_gemm_k(const CgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    cg_tile<BM, BK, WK, BKF, EPI, false, false, WP>(a, blockIdx.x, smem, [] {});
}
// The deep-ring form of the 32-row variant (rfn_cellgemm_body.h, DEEP; opt-in, RFN_CELL_VARIANT_DEEP): for launches whose tiles
// do not outnumber the CUs -- every per-step product of the recurrences at B <= 64, the backward ones up to B = 256.
// Bit-identical to cell_gemm_k<32, 64, 4, ...>; measured SLOWER in the step (C2 4.99-5.02 against 4.87-4.93 ms, C3 64.7 against
// 64.3-64.6, profiles/r05_chain.md): the K loop itself is shorter (3.5 -> 2.4 us), but a 128 KB block no longer shares its CU
// with the next launch's early blocks and the 28 requests per wave it puts in flight at once queue behind each other.
#define CG_DEEP_SLOTS 8
template <bool BKF, int EPI>
__global__ __launch_bounds__(256) void cell_gemm_deep_k(const CgArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    cg_tile<32, 64, 4, BKF, EPI, false, true, false>(a, blockIdx.x, smem, [] {});
}

// A store-epilogue dX product and the rows of the decoder's hoisted attention backward in ONE launch (round 6): the two are
// independent -- both consume the step's gate gradients -- so they run beside each other instead of one after the other.
// Blocks [0, rows) take the attention rows (the critical path: the product after this launch reads their d hproj), the rest
// the GEMM tiles.  Same device bodies as the separate launches: identical bits.
template <int BM, bool WP>
__global__ __launch_bounds__(256) void cell_gemm_rows_k(const CgArgs a, const DecAttnBwdArgs d, const int rows) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if ((int)blockIdx.x < rows) dec_attn_bwd_fast_body(d, blockIdx.x);
    else cg_tile<BM, 64, 4, false, CG_EPI_STORE, false, false, WP>(a, blockIdx.x - rows, smem, [] {});
}
template <int BM, bool WP>
static int cg_launch_rows(const CgArgs& a, int blocks, const DecAttnBwdArgs& d, int rows, hipStream_t st) {
    auto k = cell_gemm_rows_k<BM, WP>;
    constexpr size_t slot = (size_t)(BM + CG_BN) * 64 * sizeof(float);
    static CgDevStateRows ds;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return RFN_ERR_LAUNCH;
    if (!ds.set[dev & 15]) {
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(CG_MAX_SLOTS * slot)) != hipSuccess)
            return RFN_ERR_LAUNCH;
        ds.set[dev & 15] = true;
    }
    hipLaunchKernelGGL(k, dim3(rows + blocks), dim3(256), a.slots * slot, st, a, d, rows);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}

// ---- host side -----------------------------------------------------------------------------------------------------------
struct CgDevState {
    bool set[16] = {};
};
template <int BM, int BK, int WK, bool BKF, int EPI, bool WP = false>
static int cg_launch(const CgArgs& a, int blocks, hipStream_t st) {
    auto k = cell_gemm_k<BM, BK, WK, BKF, EPI, WP>;
    constexpr size_t slot = (size_t)(BM + CG_BN) * BK * sizeof(float);
    static CgDevState ds;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return RFN_ERR_LAUNCH;
    if (!ds.set[dev & 15]) {
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(CG_MAX_SLOTS * slot)) !=
            hipSuccess)
            return RFN_ERR_LAUNCH;
        ds.set[dev & 15] = true;
    }
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * (BM >= 32 ? BM / 32 : 1) * WK), a.slots * slot, st, a);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}

static int cg_device_cus() {   // CU count of the current device (cached per device index)
    static int cus[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    int& c = cus[dev & 15];
    if (c == 0 && hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) c = 0;
    return c;
}

// Tile variants: 1 = 64 rows, K step 32, 2 K-waves; 2 = 64 rows, K step 64, 4 K-waves; 3 = 32 rows, K step 64, 4 K-waves;
// 4 = 16 rows, K step 64, 4 K-waves on the 16x16x4 MFMA shape; 5 = the same on K steps of 128; 6 = 4 with wave-private ring
// slots; 8 = 3 with wave-private ring slots (all bit-identical to 2 and 3); 7 / 9 = the library's choice among 1-5 / 1-3 (A/B hooks).
// variant 0: the K step / K-wave count -- which fix the k order of every output element -- are chosen FROM THE K COUNTS
// ONLY, so a row's arithmetic never depends on the batch it sits in (variants 2 and 3 give bit-identical results).
#ifndef CG_T16_PER_CU
#define CG_T16_PER_CU 1     /* 16-row tiles are taken while a CU gets at most this many of them (A/B knob, tools/build_variant.sh) */
#endif
static int cg_plan(CgPrepared& pz, int variant) {
    CgArgs& a = pz.a;
    const int force_slots = (variant >> 4) & 15;   // tools: bits 4-7 of `variant` force the ring depth
    const bool deep = (variant & RFN_CELL_VARIANT_DEEP) != 0;
    variant &= 15;
    bool k64 = true, k128 = true;
    int max_iters = 0;
    for (int o = 0; o < a.nout; ++o) {
        int it = 0;
        for (int s = 0; s < a.out[o].nseg; ++s) {
            k64 = k64 && (a.seg[a.out[o].seg0 + s].K % 64 == 0);
            k128 = k128 && (a.seg[a.out[o].seg0 + s].K % 128 == 0);
            it += a.seg[a.out[o].seg0 + s].K;
        }
        max_iters = it > max_iters ? it : max_iters;
    }
    // 32-row tiles whenever the K step of 64 applies: measured fastest at every per-step shape of the path (B = 64 ... 640),
    // because they put two to three independent blocks on a CU (profiles/r03_cellgemm.md)
    if (variant == 0 || variant == 7 || variant == 9) {   // A/B hooks: 7 = the choice with the shared-slot 16-row forms of round 5's
        const bool shared16 = variant == 7;               // first half; 9 = the choice among the 64- / 32-row shared-slot forms only
        const bool tall_only = variant == 9;
        variant = k64 ? 3 : 1;
        // few tiles: 16-row tiles put the launch on twice the CUs with half the MFMA chain per block (variant 4, bit-identical
        // to variant 3: rfn_cellgemm_body.h) -- taken when even the 16-row tiles do not outnumber the CUs
        if (k64 && !tall_only) {
            long cols = 0;
            for (int o = 0; o < a.nout; ++o) cols += a.out[o].N / CG_BN;
            // 6: those tiles with wave-private ring slots -- no block barrier in the K loop, fragments read one step ahead
            // (tools/bench_cellgemm.py --small --variants 3,4,5,6, B = 64: Kb1 17.4 / 15.7 / 13.7 / 10.1 us, stage-II K3 + LSTM
            // 21.0 / 20.8 / 19.8 / 16.6, Kb2 6.9 / 6.5 / 6.2 / 5.4); 4 and 5 (shared slots, K steps of 64 / 128) stay as A/B forms
            const long t16 = (long)rfn_cdiv(a.M, 16) * cols, t32 = (long)rfn_cdiv(a.M, 32) * cols, cus = cg_device_cus();
            if (t16 <= CG_T16_PER_CU * cus) variant = !shared16 ? 6 : (k128 && max_iters >= 1024) ? 5 : 4;
            // 8: 32-row tiles on wave-private slots while a CU gets at most one of them (B = 64: decoder K1 6.0 against 6.9 us,
            // stage-II backward 14.6 against 17.8; B = 256 Kb1 14.3 against 17.6); with several blocks per CU the shared-slot
            // form's waits are covered by the other blocks' waves and it stays ahead (B = 256 stage-II K3: 28.8 against 31.2)
            else if (!shared16 && t32 <= cus) variant = 8;
        }
    }
    if ((variant == 2 || variant == 3 || variant == 4 || variant == 6 || variant == 8) && !k64) return RFN_ERR_SHAPE;
    if (variant == 5 && !k128) return RFN_ERR_SHAPE;
    if (variant < 1 || variant > 8 || variant == 7) return RFN_ERR_SHAPE;
    const int bm = (variant == 3 || variant == 8) ? 32 : (variant >= 4) ? 16 : 64;
    max_iters /= (variant == 1) ? 32 : (variant == 5) ? 128 : 64;
    a.tiles_m = rfn_cdiv(a.M, bm);
    int t0 = 0;
    for (int o = 0; o < a.nout; ++o) {
        a.out[o].tiles_n = a.out[o].N / CG_BN;
        a.out[o].tile0 = t0;
        t0 += a.tiles_m * a.out[o].tiles_n;
    }
    // Ring depth 3 (two K steps in flight).  Measured on MI355X (tools/bench_cellgemm.py --slots 2,3,4,6): 2 and 3 slots tie,
    // deeper rings LOSE 10-40 % -- the kernel is bound by the rate of the L2 -> LDS path at 8 flops per operand byte, not by
    // its latency, and what helps is more co-resident blocks per CU (each with its own barrier), i.e. LESS LDS per block.
    int slots = 3;
    if (force_slots > 0) slots = force_slots;   // tools only
    if (slots > CG_MAX_SLOTS) slots = CG_MAX_SLOTS;
    if (slots > max_iters + 1) slots = max_iters + 1;
    if (slots < 2) slots = 2;
    if (variant == 6 || variant == 8) slots = CG_WP_SLOTS;   // the wave-private form's ring depth is a compile-time constant
    a.slots = slots;
    pz.variant = variant;
    pz.blocks = t0;
    pz.deep = deep && force_slots == 0;     // opt-in (A/B hook); tools that sweep the ring depth measure the shallow form
    return RFN_OK;
}

struct CgDeepDev {
    bool set[16] = {};
    int cus[16] = {};
};
template <bool BKF, int EPI>
static int cg_launch_deep(const CgPrepared& pz, hipStream_t st, bool* taken) {
    static CgDeepDev ds;
    *taken = false;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return RFN_ERR_LAUNCH;
    const int di = dev & 15;
    constexpr size_t lds = (size_t)CG_DEEP_SLOTS * (32 + CG_BN) * 64 * sizeof(float);
    auto k = cell_gemm_deep_k<BKF, EPI>;
    if (!ds.set[di]) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return RFN_ERR_LAUNCH;
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return RFN_ERR_LAUNCH;
        ds.cus[di] = cus;
        ds.set[di] = true;
    }
    if (pz.blocks > ds.cus[di]) return RFN_OK;     // more tiles than CUs: co-resident shallow blocks hide latency better
    CgArgs a = pz.a;
    a.slots = CG_DEEP_SLOTS;
    hipLaunchKernelGGL(k, dim3(pz.blocks), dim3(256), lds, st, a);
    RFN_CHECK_LAUNCH();
    *taken = true;
    return RFN_OK;
}

// The same product on 32-row tiles (bit-identical): what the persistent recurrence kernels are built from.
int rfn_cg_replan32(CgPrepared* pz) { return pz->variant == 3 ? RFN_OK : cg_plan(*pz, 3); }

int rfn_cg_launch(const CgPrepared& pz, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (pz.variant == 3 && pz.deep) {
        bool taken = false;
        int rc;
        if (pz.epi == CG_EPI_LSTM) rc = cg_launch_deep<true, CG_EPI_LSTM>(pz, st, &taken);
        else if (pz.epi == CG_EPI_LSTM_BWD) rc = cg_launch_deep<false, CG_EPI_LSTM_BWD>(pz, st, &taken);
        else if (pz.bkf) rc = cg_launch_deep<true, CG_EPI_STORE>(pz, st, &taken);
        else rc = cg_launch_deep<false, CG_EPI_STORE>(pz, st, &taken);
        if (rc != RFN_OK || taken) return rc;
    }
#define CG_CASE(BKF_, EPI_)                                                                 \
    switch (pz.variant) {                                                                   \
        case 1: return cg_launch<64, 32, 2, BKF_, EPI_>(pz.a, pz.blocks, st);                \
        case 2: return cg_launch<64, 64, 4, BKF_, EPI_>(pz.a, pz.blocks, st);                \
        case 3: return cg_launch<32, 64, 4, BKF_, EPI_>(pz.a, pz.blocks, st);                \
        case 4: return cg_launch<16, 64, 4, BKF_, EPI_>(pz.a, pz.blocks, st);                \
        case 5: return cg_launch<16, 128, 4, BKF_, EPI_>(pz.a, pz.blocks, st);               \
        case 6: return cg_launch<16, 64, 4, BKF_, EPI_, true>(pz.a, pz.blocks, st);          \
        case 8: return cg_launch<32, 64, 4, BKF_, EPI_, true>(pz.a, pz.blocks, st);          \
        default: return RFN_ERR_SHAPE;                                                      \
    }
    if (pz.epi == CG_EPI_LSTM) { CG_CASE(true, CG_EPI_LSTM) }
    if (pz.epi == CG_EPI_LSTM_BWD) { CG_CASE(false, CG_EPI_LSTM_BWD) }
    if (pz.bkf) { CG_CASE(true, CG_EPI_STORE) }
    CG_CASE(false, CG_EPI_STORE)
#undef CG_CASE
}

// rfn_cg_launch(pz) with `rows` rows of the decoder's attention backward in the same launch.  RFN_ERR_UNSUPPORTED (nothing
// launched) when the prepared product is not a 256-thread store-epilogue dX launch or the rows do not take the fast body.
int rfn_cg_launch_with_rows(const CgPrepared& pz, const DecAttnBwdArgs& d, int rows, void* stream) {
    if (pz.epi != CG_EPI_STORE || pz.bkf || pz.deep || rows < 1 || !rfn_dec_attn_bwd_fast_ok(d)) return RFN_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    switch (pz.variant) {
        case 3: return cg_launch_rows<32, false>(pz.a, pz.blocks, d, rows, st);
        case 6: return cg_launch_rows<16, true>(pz.a, pz.blocks, d, rows, st);
        case 8: return cg_launch_rows<32, true>(pz.a, pz.blocks, d, rows, st);
        default: return RFN_ERR_UNSUPPORTED;
    }
}

extern "C" int rfn_cell_gemm_supported(int M, int nout, const rfn_cell_out* outs, int R) {
    if (M < 1 || nout < 1 || nout > RFN_CELL_MAXOUT || !outs) return 0;
    int nseg = 0;
    const int bkf = outs[0].seg[0].b_kfast, epi = outs[0].epilogue;
    if (epi != RFN_CELL_EPI_STORE && epi != RFN_CELL_EPI_LSTM && epi != RFN_CELL_EPI_LSTM_BWD) return 0;
    for (int o = 0; o < nout; ++o) {
        const rfn_cell_out& t = outs[o];
        if (t.nseg < 1 || t.nseg > RFN_CELL_MAXSEG || t.N < 32 || t.N % 32 || t.epilogue != epi) return 0;
        if (epi != RFN_CELL_EPI_LSTM_BWD || t.C) {
            if (!t.C || t.ldc % 4 || !rfn_aligned16(t.C)) return 0;
        }
        if (t.acc_parts < 0 || (t.acc_parts > 0 && (!t.accumulate || !t.C || !t.acc_slabs || !rfn_aligned16(t.acc_slabs) ||
                                                    epi == RFN_CELL_EPI_LSTM || t.acc_parts > 8 || t.acc_stride % 4)))
            return 0;
        if (epi == RFN_CELL_EPI_LSTM) {
            // gate-major tiles: 8 units x 4 gates per 32 columns
            if (R < 8 || R % 8 || t.N != 4 * R || !t.c_prev || !t.c_next || !t.h_next || !bkf) return 0;
        }
        if (epi == RFN_CELL_EPI_LSTM_BWD) {
            if (R < 32 || t.N != R || !t.gates || !t.c_prev || !t.c_next || !t.dc_prev || bkf) return 0;
            if (t.accumulate && !t.C) return 0;
        }
        for (int s = 0; s < t.nseg; ++s) {
            const rfn_gemm_seg& sg = t.seg[s];
            if (!sg.A || !sg.B || sg.K < 32 || sg.K % 32 || !sg.a_kfast || (sg.b_kfast != 0) != (bkf != 0)) return 0;
            if (!rfn_aligned16(sg.A) || !rfn_aligned16(sg.B) || sg.lda % 4 || sg.ldb % 4 || sg.lda < 0 || sg.ldb < 0) return 0;
            if (sg.bias && (!rfn_aligned16(sg.bias) || epi == RFN_CELL_EPI_LSTM_BWD)) return 0;
            // 32-bit lane offsets inside an operand
            const double ea = (double)M * sg.lda * 4;
            const double eb = (bkf ? (double)(epi == RFN_CELL_EPI_LSTM ? 4 * R : t.N) : (double)sg.K) * sg.ldb * 4;
            if (ea >= 4.0e9 || eb >= 4.0e9) return 0;
        }
        nseg += t.nseg;
    }
    return nseg <= CG_MAXSEG;
}

// The launch arguments of one cell product without launching it: rfn_cell_gemm = prepare + launch; the persistent recurrence
// kernels (rfn_chain.hip) take the prepared form of every step and run them inside one launch.
int rfn_cg_prepare(int M, int nout, const rfn_cell_out* outs, int R, float drop_p, uint64_t seed, int variant, CgPrepared* pz) {
    if (!rfn_cell_gemm_supported(M, nout, outs, R)) return RFN_ERR_UNSUPPORTED;
    if (drop_p < 0.f || drop_p >= 1.f) return RFN_ERR_SHAPE;
    CgArgs& a = pz->a;
    memset(pz, 0, sizeof(*pz));
    a.M = M;
    a.nout = nout;
    a.R = R;
    a.drop_p = drop_p;
    a.seed = seed;
    int ns = 0;
    for (int o = 0; o < nout; ++o) {
        const rfn_cell_out& t = outs[o];
        CgOut& d = a.out[o];
        d.C = t.C; d.ldc = t.ldc; d.N = t.N; d.accumulate = t.accumulate; d.seg0 = ns; d.nseg = t.nseg;
        d.c_prev = t.c_prev; d.c_next = t.c_next; d.h_next = t.h_next;
        d.ldcp = t.ldcp; d.ldcn = t.ldcn; d.ldh = t.ldh; d.drop_offset = t.drop_offset;
        d.gates = t.gates; d.ldg = t.ldg; d.dh_ext = t.dh_ext; d.lddh = t.lddh;
        d.dc_next = t.dc_next; d.lddcn = t.lddcn; d.dc_prev = t.dc_prev; d.lddcp = t.lddcp;
        d.acc_slabs = t.acc_slabs; d.acc_parts = t.acc_parts; d.acc_stride = t.acc_stride;
        for (int s = 0; s < t.nseg; ++s) {
            CgSeg& g = a.seg[ns++];
            g.A = t.seg[s].A; g.B = t.seg[s].B; g.bias = t.seg[s].bias;
            g.lda = t.seg[s].lda; g.ldb = t.seg[s].ldb; g.K = t.seg[s].K;
        }
    }
    pz->bkf = outs[0].seg[0].b_kfast != 0;
    pz->epi = outs[0].epilogue == RFN_CELL_EPI_LSTM ? CG_EPI_LSTM : outs[0].epilogue == RFN_CELL_EPI_LSTM_BWD ? CG_EPI_LSTM_BWD : CG_EPI_STORE;
    return cg_plan(*pz, variant);
}

extern "C" int rfn_cell_gemm(int M, int nout, const rfn_cell_out* outs, int R, float drop_p, uint64_t seed, int variant,
                             void* stream) {
    CgPrepared pz;
    RFN_TRY(rfn_cg_prepare(M, nout, outs, R, drop_p, seed, variant, &pz));
    return rfn_cg_launch(pz, stream);
}
