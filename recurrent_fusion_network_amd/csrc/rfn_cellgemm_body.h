// Device body of the row-panel GEMM of the recurrences (rfn_cellgemm.hip has the story) as a function, so that two kernels
// can run it: cell_gemm_k (one launch per product, rfn_cellgemm.hip) and the persistent recurrence kernels (rfn_chain.hip),
// which walk the products of ALL steps of a chain inside one launch with a grid barrier between dependent phases.
//
// XB ("cross-block") = the second use.  Inside one launch the caches do not keep blocks coherent: a CU's vector L1 is never
// refreshed by another CU's stores and the per-XCD L2s are not coherent with each other.  So in XB mode EVERY global access
// to a buffer that some block of the launch writes takes the sc1 form -- loads bypass L1 and are served coherently
// (global_load / buffer_load / global_load_lds ... sc1), stores write through (sc1) -- parameters (weights, biases) stay
// plain; each storing wave drains its stores (s_waitcnt vmcnt(0)) before the block's arrive at the grid barrier
// (rfn_chain.hip).  The arithmetic is untouched: a product computed in either mode is the same bits.
// In XB mode the descriptor sits in LDS (built per step by the persistent kernel), so what steers control flow or forms a
// DMA base is made wave-uniform again with readfirstlane.
#pragma once
#include "rfn_common.h"
#include "rfn_xb.h"

typedef float cg_f32x16 __attribute__((ext_vector_type(16)));
typedef float cg_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void cg_lds_void;
typedef const __attribute__((address_space(1))) void cg_gbl_void;

#define CG_MAXSEG 24
#ifndef CG_WP_SLOTS
#define CG_WP_SLOTS 3             /* ring depth of the wave-private form (cg_tile WP): compile-time, the K loop counts on it */
#endif
#define CG_MAX_SLOTS 6            /* ring slots are a launch parameter: as many as LDS allows for the blocks a CU hosts */
#define CG_BN 32                   /* tile width: 8 units x 4 gates in the gate epilogue */
enum { CG_EPI_STORE = 0, CG_EPI_LSTM = 1, CG_EPI_LSTM_BWD = 2 };

struct CgSeg {
    const float* A;
    const float* B;
    const float* bias;
    long lda, ldb;
    int K, pad;
};
struct CgOut {
    float* C;
    long ldc;
    // gate epilogues (forward: c_prev, c_next, h_next; backward: gates, c_prev, c_next, dh_ext, dc_next, dc_prev)
    const float* c_prev;
    float* c_next;
    float* h_next;
    float* gates;
    const float* dh_ext;
    const float* dc_next;
    float* dc_prev;
    long ldcp, ldcn, ldh, ldg, lddh, lddcn, lddcp;
    unsigned long long drop_offset;
    int N, accumulate, seg0, nseg, tile0, tiles_n;
    const float* acc_slabs;   // accumulate: C + acc_parts further slabs (C's shape and ld) at acc_slabs + p * acc_stride
    int acc_parts;            // (store / gate-gradient epilogue)
    long acc_stride;
};
struct CgArgs {
    int M, nout, R, tiles_m;
    float drop_p;
    int slots;
    unsigned long long seed;
    CgOut out[RFN_CELL_MAXOUT];
    CgSeg seg[CG_MAXSEG];
};


#ifndef CG_ISSUE_AFTER
#define CG_ISSUE_AFTER 0    /* A/B: request the next K step after (1) or before (0) this step's MFMAs are issued */
#endif
// -DCG_LOOP_STAMPS (tools/cg_loop_probe.hip only): block 0, lane 0 of wave 0 stamps the shader clock inside every K step of
// the launch kernel's shallow loop: [it][0] top, [1] own pieces landed, [2] block barrier passed, [3] next step requested,
// [4] fragments read + MFMAs issued
#ifdef CG_LOOP_STAMPS
__device__ unsigned long long g_cg_loop_stamps[64 * 8];
#define CG_LSTAMP(it_, k_) do { if (!XB && !DEEP && blockIdx.x == 0 && threadIdx.x == 0 && (it_) < 64) g_cg_loop_stamps[(it_) * 8 + (k_)] = __builtin_readcyclecounter(); } while (0)
#else
#define CG_LSTAMP(it_, k_)
#endif

// RFN_CHAIN_TIMING builds (tools/chain_timing.py): lane 0 of a block stamps the 100 MHz clock inside the XB tile
#ifdef RFN_CHAIN_TIMING
#define CG_STAMP(i) do { if (XB && g_cg_stamp_local && threadIdx.x == 0) g_cg_stamp_local[i] = wall_clock64(); } while (0)
#else
#define CG_STAMP(i)
#endif

// Which k-group (8 consecutive k) of a K step is the t-th one wave `wk` of WK K-waves takes.  Round 6: with four K-waves a wave
// owns PAIRS of adjacent k-groups -- groups {2 wk, 2 wk + 1} of every 64 k -- so that what it requests of an operand row in the
// wave-private form is one 64-B run instead of two 32-B runs a quarter line apart (every 128-B line of a [row][k] operand used
// to be asked for by all four waves).  The assignment depends on nothing but the absolute k index, so every tile variant and
// K step (64 / 128) still adds the same k's in the same order per wave: the variants stay bit-identical to each other.
// -DCG_KGROUP_INTERLEAVED=1: the round-3..5 assignment (group g to wave g % WK), A/B.
#ifndef CG_KGROUP_INTERLEAVED
#define CG_KGROUP_INTERLEAVED 0
#endif
template <int WK>
__device__ __forceinline__ constexpr int cg_kgroup(int wk, int t) {
    return (WK == 4 && !CG_KGROUP_INTERLEAVED) ? 2 * wk + (t & 1) + 8 * (t >> 1) : wk + WK * t;
}

template <int N>
__device__ __forceinline__ void cg_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// at most `n` (wave-uniform, run-time) vector-memory operations of this wave still in flight; stricter when n > 30
__device__ __forceinline__ void cg_wait_vmcnt_dyn(int n) {
    switch (n >> 1) {   // counts are even here: two pieces per wave, operand and K step
        case 0: cg_wait_vmcnt<0>(); break;
        case 1: cg_wait_vmcnt<2>(); break;
        case 2: cg_wait_vmcnt<4>(); break;
        case 3: cg_wait_vmcnt<6>(); break;
        case 4: cg_wait_vmcnt<8>(); break;
        case 5: cg_wait_vmcnt<10>(); break;
        case 6: cg_wait_vmcnt<12>(); break;
        case 7: cg_wait_vmcnt<14>(); break;
        case 8: cg_wait_vmcnt<16>(); break;
        case 9: cg_wait_vmcnt<18>(); break;
        case 10: cg_wait_vmcnt<20>(); break;
        case 11: cg_wait_vmcnt<22>(); break;
        case 12: cg_wait_vmcnt<24>(); break;
        case 13: cg_wait_vmcnt<26>(); break;
        case 14: cg_wait_vmcnt<28>(); break;
        default: cg_wait_vmcnt<30>(); break;
    }
}


// BM x 32 output tile, K steps of BK, WK waves per 32x32 sub-tile (each takes every WK-th k-group of 8).
// BKF: B is an nn.Linear weight [n][k] (forward products); !BKF: B is [k][n] (dX = dY . W: the reduction index is W's row).
// bid: the tile this call computes (cell_gemm_k: blockIdx.x; the persistent kernels deal tiles to blocks themselves).
// The caller provides the ring (`smem`, a.slots slots) and, between two calls of one block, a __syncthreads().
//
// XB adds a software pipeline across the grid barrier that precedes the tile (rfn_chain.hip): the weights (B) of the first ring
// slots do not depend on any other block, so their DMAs are issued BEFORE `hook()` -- the wait half of the barrier -- and
// everything another block produced (the A operand, the epilogue's previous values) is requested right after it, all K steps
// the ring holds at once: after the barrier a tile costs one round trip for its activations instead of one for the weights
// plus one per pair of K steps.  Piece-to-slot layout, k order and epilogue are those of the launch form.
//
// DEEP (implied by XB; also used by the launch form when its tiles do not outnumber the CUs, i.e. when a block has a CU's LDS
// to itself anyway): the ring is as deep as the launch's LDS allows (a.slots, up to 8), the two operands have cursors of their
// own, and when the ring holds the whole K range every K step is requested up front and the K loop runs without counted
// waits or barriers.  Same pieces in the same slots, same k order: bit-identical to the shallow form.
//
// WP (round 5; 16-row tiles, the launch kernel of few-tile products): every wave requests exactly the k-groups IT consumes
// into a region of the slot that is its own, so the K loop has no block barrier and no wave waits for another one's requests;
// the fragments of step it + 1 are read from LDS while the MFMAs of step it run, and the slot they came from is requested
// again right away.  Same k-groups per wave in the same order on the same MFMA shape: bit-identical to the shared-slot forms.
template <int BM, int BK, int WK, bool BKF, int EPI, bool XB, bool DEEP, bool WP, typename Hook>
__device__ __forceinline__ void cg_tile(const CgArgs& a, const int bid, float* smem, Hook hook) {
    static_assert(DEEP || !XB, "the cross-block form is built on the deep pipeline");
    static_assert(!WP || ((BM == 16 || BM == 32) && BK == 64 && WK == 4 && !XB && !DEEP), "wave-private slots: 16- / 32-row tiles, K step 64");
    constexpr int BN = CG_BN;
    // BM = 16 (round 5): a 16-row tile = two 16 x 16 MFMA tiles side by side per wave (v_mfma_f32_16x16x4_f32), for launches
    // with so few 32-row tiles that most CUs would idle: twice the blocks, half the MFMA chain per block.  The 16x16x4 shape is
    // bit for bit the k-ordered fma chain of the 32x32x2 shape (tools/mfma16_order_probe.hip) and the k's of a k-group are fed in
    // the order the 32-row form feeds them (0 4 1 5 | 2 6 3 7), so a 16-row tile's outputs are the 32-row tile's bits.
    constexpr bool SMALL = (BM == 16);
    static_assert(!SMALL || ((BK == 64 || BK == 128) && !XB && !DEEP), "the 16-row form exists for the launch kernel, K step 64 / 128");
    constexpr int WM = SMALL ? 1 : BM / 32, W = WM * WK, T = 64 * W;
    constexpr int A_FL = BM * BK, B_FL = BN * BK, SLOT_FL = A_FL + B_FL;
    constexpr int PA = A_FL / 256, PB = B_FL / 256, P = PA + PB;   // 1-KiB pieces per K step
    static_assert(P % W == 0, "pieces must divide evenly over the waves");
    constexpr int NIW = P / W;
    constexpr int CPR = BK / 4;      // 16-B chunks per [row][k] row
    constexpr int RPP = 64 / CPR;    // rows per piece
    constexpr int KG = BK / 8;       // k-groups per K step
    // BK = 128 (round 5, few-tile launches): half the K steps of BK = 64 -- a K step costs ~0.45 us of waits, barrier and request
    // issue whatever it computes -- and the SAME k order: wave wk takes the k-groups cg_kgroup() gives it, in ascending order,
    // whatever the K step is, as long as it holds a multiple of WK groups.
    static_assert(KG % WK == 0 && (BK == 32 || BK == 64 || BK == 128), "unsupported K step");
    static_assert(WK * BM * BN <= 2 * SLOT_FL, "the partial tiles reuse the ring (at least two slots)");
    static_assert(NIW * (CG_MAX_SLOTS - 1) <= 63, "vmcnt is a 6-bit counter");
    static_assert((EPI == CG_EPI_LSTM) ? BKF : true, "the gate epilogue belongs to forward products");
    static_assert((EPI == CG_EPI_LSTM_BWD) ? !BKF : true, "the gate-gradient epilogue belongs to dX products");
    constexpr int U = BN / 4;                       // units per tile of the gate epilogue
    constexpr int NV = (BM * BN / 4 + T - 1) / T;   // float4 of the tile per thread       (store epilogue)
    constexpr int NP = (BM * U + T - 1) / T;        // (row, unit) pairs per thread         (gate epilogue)
    constexpr int NE = BM * BN / T;                 // (row, unit) elements per thread      (gate-gradient epilogue)
    // 32- / 64-row tiles give every thread whole shares; a 16-row tile has work for half the threads of the float4 / pair loops
    constexpr bool PART = (BM * BN / 4) % T != 0;
    static_assert(NE >= 1 && BM * BN % T == 0 && (PART ? SMALL : (BM * BN % (4 * T) == 0 && BM * U % T == 0)), "epilogue tiling");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave / WM, wm = wave - wk * WM;
    const int l31 = lane & 31, h = lane >> 5;

    int r = 0;
    const int nout = XB ? xb_uni(a.nout) : a.nout;
    for (int i = 1; i < nout; ++i)
        if (bid >= a.out[i].tile0) r = i;
    if constexpr (XB) r = xb_uni(r);
    const CgOut& O = a.out[r];
    const int lt = bid - (XB ? xb_uni(O.tile0) : O.tile0);
    const int tiles_n = XB ? xb_uni(O.tiles_n) : O.tiles_n;
    const int tm = lt / tiles_n, tn = lt - tm * tiles_n;
    const int row0 = tm * BM, col0 = tn * BN;
    const int M = XB ? xb_uni(a.M) : a.M, R = XB ? xb_uni(a.R) : a.R;
    const int nseg = XB ? xb_uni(O.nseg) : O.nseg, seg0 = XB ? xb_uni(O.seg0) : O.seg0;

    auto swz = [](int row) -> int { return BK == 32 ? ((row >> 1) & 7) : (BK == 64 ? (row & 15) : (row & 31)); };

    cg_f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    cg_f32x4 acc16[2] = {cg_f32x4{0.f, 0.f, 0.f, 0.f}, cg_f32x4{0.f, 0.f, 0.f, 0.f}};   // SMALL: the two column halves
    const int l15 = lane & 15, kq = lane >> 4;

    int total_iters = 0;
    for (int s = 0; s < nseg; ++s) total_iters += a.seg[seg0 + s].K / BK;
    if constexpr (XB) total_iters = xb_uni(total_iters);

    // per-lane byte offsets of this wave's pieces inside the current segment's operands; piece p = wave + W * j
    uint32_t off[NIW];
    const char* baseA = nullptr;
    const char* baseB = nullptr;
    long stepB = 0;
    int seg = 0, k0 = 0, segK = 0;
    auto setup = [&]() {
        if (seg >= nseg) return;
        const CgSeg& sg = a.seg[seg0 + seg];
        segK = XB ? xb_uni(sg.K) : sg.K;
        baseA = (const char*)sg.A;
        baseB = (const char*)sg.B;
        stepB = BKF ? (long)BK * 4 : (long)BK * sg.ldb * 4;
        if constexpr (XB) {
            baseA = xb_uni_ptr(baseA);
            baseB = xb_uni_ptr(baseB);
            stepB = BKF ? (long)BK * 4 : (long)BK * 4 * xb_uni((int)sg.ldb);
        }
#pragma unroll
        for (int j = 0; j < NIW; ++j) {
            const int p = wave + W * j;
            if (p < PA) {
                const int rt = p * RPP + lane / CPR;
                int gr = row0 + rt;
                gr = gr < M ? gr : M - 1;     // rows past the batch: a valid row is fetched, its results are dropped
                off[j] = (uint32_t)(((long)gr * sg.lda + 4 * ((lane % CPR) ^ swz(rt))) * 4);
            } else {
                const int pb = p - PA;
                if constexpr (BKF) {
                    const int rt = pb * RPP + lane / CPR;   // tile column = row of the [n][k] weight
                    long n;
                    if constexpr (EPI == CG_EPI_LSTM) n = (long)(rt / U) * R + tn * U + rt % U;   // gate-major columns
                    else n = col0 + rt;
                    off[j] = (uint32_t)((n * sg.ldb + 4 * ((lane % CPR) ^ swz(rt))) * 4);
                } else {
                    constexpr int CQ = BN / 4, KPP = 64 / CQ;
                    const int kr = pb * KPP + lane / CQ;
                    off[j] = (uint32_t)(((long)kr * sg.ldb + col0 + 4 * (lane % CQ)) * 4);
                }
            }
        }
    };
    if constexpr (!DEEP && !WP) setup();
    auto issue = [&](int slot) {
        float* st = smem + slot * SLOT_FL;
#pragma unroll
        for (int j = 0; j < NIW; ++j) {
            const int p = wave + W * j;
            // XB: the activations (A) may be another block's output of this very launch: sc1; the weights (B) are parameters
            if (XB && p < PA)
                __builtin_amdgcn_global_load_lds((cg_gbl_void*)(baseA + off[j]), (cg_lds_void*)(st + p * 256), 16, 0, 16);
            else
                __builtin_amdgcn_global_load_lds((cg_gbl_void*)(((p < PA) ? baseA : baseB) + off[j]), (cg_lds_void*)(st + p * 256), 16, 0, 0);
        }
        baseA += BK * 4;
        baseB += stepB;
        k0 += BK;
        if (k0 >= segK) {
            k0 = 0;
            ++seg;
            setup();
        }
    };

    // ---- XB: separate cursors for the two operands (the weights run ahead of the barrier) ------------------------------------
    constexpr int NA = PA / W, NB = PB / W;
    static_assert(!DEEP || (PA % W == 0 && PB % W == 0), "DEEP: whole pieces per wave and operand");
    uint32_t offA[DEEP ? NA : 1], offB[DEEP ? NB : 1];
    const char* curA = nullptr;
    const char* curB = nullptr;
    long curStepB = 0;
    int segA = 0, segB = 0, kA = 0, kB = 0, segKA = 0, segKB = 0;
    auto setupA = [&]() {
        if (segA >= nseg) return;
        const CgSeg& sg = a.seg[seg0 + segA];
        segKA = xb_uni(sg.K);
        curA = xb_uni_ptr((const char*)sg.A);
        const long lda = sg.lda;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int rt = (wave + W * j) * RPP + lane / CPR;
            int gr = row0 + rt;
            gr = gr < M ? gr : M - 1;
            offA[j] = (uint32_t)(((long)gr * lda + 4 * ((lane % CPR) ^ swz(rt))) * 4);
        }
    };
    auto setupB = [&]() {
        if (segB >= nseg) return;
        const CgSeg& sg = a.seg[seg0 + segB];
        segKB = xb_uni(sg.K);
        curB = xb_uni_ptr((const char*)sg.B);
        const long ldb = sg.ldb;
        curStepB = BKF ? (long)BK * 4 : (long)BK * 4 * xb_uni((int)ldb);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int pb = wave + W * j;
            if constexpr (BKF) {
                const int rt = pb * RPP + lane / CPR;
                long n;
                if constexpr (EPI == CG_EPI_LSTM) n = (long)(rt / U) * R + tn * U + rt % U;
                else n = col0 + rt;
                offB[j] = (uint32_t)((n * ldb + 4 * ((lane % CPR) ^ swz(rt))) * 4);
            } else {
                constexpr int CQ = BN / 4, KPP = 64 / CQ;
                const int kr = pb * KPP + lane / CQ;
                offB[j] = (uint32_t)(((long)kr * ldb + col0 + 4 * (lane % CQ)) * 4);
            }
        }
    };
    auto issueA = [&](int slot) {   // XB: the activations may be another block's output of this very launch: sc1
        float* st = smem + slot * SLOT_FL;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            if constexpr (XB)
                __builtin_amdgcn_global_load_lds((cg_gbl_void*)(curA + offA[j]), (cg_lds_void*)(st + (wave + W * j) * 256), 16, 0, 16);
            else
                __builtin_amdgcn_global_load_lds((cg_gbl_void*)(curA + offA[j]), (cg_lds_void*)(st + (wave + W * j) * 256), 16, 0, 0);
        }
        curA += BK * 4;
        kA += BK;
        if (kA >= segKA) {
            kA = 0;
            ++segA;
            setupA();
        }
    };
    auto issueB = [&](int slot) {   // the weights are parameters: default policy
        float* st = smem + slot * SLOT_FL;
#pragma unroll
        for (int j = 0; j < NB; ++j)
            __builtin_amdgcn_global_load_lds((cg_gbl_void*)(curB + offB[j]), (cg_lds_void*)(st + (PA + wave + W * j) * 256), 16, 0, 0);
        curB += curStepB;
        kB += BK;
        if (kB >= segKB) {
            kB = 0;
            ++segB;
            setupB();
        }
    };
    // ---- WP: this wave's own pieces of a K step: A rows x its 16 k's (one piece per 16 rows) | B, two pieces -----------------
    // LDS image of a [16 rows][4 chunks] piece: chunk slot j of row r holds the wave's chunk j ^ wp_g(r) (chunks 0 1 = its first
    // k-group of the step, 2 3 = its second), which makes every ds_read_b128 lane group of the fragment reads cover the 64
    // banks once (16-row tiles: the lanes of a group ask for two different chunks; 32-row tiles: for one).  [k][n] weights:
    // 8 k-rows x 8 column chunks per piece, rows 4-7 with the column halves swapped (the two 16-lane halves of a ds_read_b32
    // group of the 16-row form read k-rows 4 apart).
    constexpr int WPA = BM / 16, WPN = WPA + 2, WP_FL = WPN * 256;     // pieces (1 KiB) per wave and K step; floats
    static_assert(!WP || WK * WP_FL == SLOT_FL, "the waves' regions tile the slot");
    auto wp_g = [](int r) -> int { return SMALL ? ((r >> 3) & 1) * 3 : (r >> 2) & 3; };
    uint32_t offW[WPN] = {};
    const char* wpA = nullptr;
    const char* wpB = nullptr;
    long wpStepB = 0;
    int wpSeg = 0, wpK = 0, wpSegK = 0;
    auto setupW = [&]() {
        if (wpSeg >= nseg) return;
        const CgSeg& sg = a.seg[seg0 + wpSeg];
        wpSegK = sg.K;
        wpA = (const char*)sg.A;
        wpB = (const char*)sg.B;
        wpStepB = BKF ? (long)BK * 4 : (long)BK * sg.ldb * 4;
        const int rr = lane >> 2, j = (lane & 3) ^ wp_g(rr);
        const int ck = 2 * cg_kgroup<WK>(wk, j >> 1) + (j & 1);     // 16-B chunk of the K step's 64 k's
#pragma unroll
        for (int pa = 0; pa < WPA; ++pa) {
            int gr = row0 + 16 * pa + rr;
            gr = gr < M ? gr : M - 1;
            offW[pa] = (uint32_t)(((long)gr * sg.lda + 4 * ck) * 4);
        }
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            if constexpr (BKF) {
                const int rt = 16 * pc + rr;     // tile column = row of the [n][k] weight
                long n;
                if constexpr (EPI == CG_EPI_LSTM) n = (long)(rt / U) * R + tn * U + rt % U;
                else n = col0 + rt;
                offW[WPA + pc] = (uint32_t)((n * sg.ldb + 4 * ck) * 4);
            } else {
                const int kr = lane >> 3, cq = (lane & 7) ^ (4 * ((kr >> 2) & 1));
                offW[WPA + pc] = (uint32_t)(((long)(8 * cg_kgroup<WK>(wk, pc) + kr) * sg.ldb + col0 + 4 * cq) * 4);
            }
        }
    };
    auto wp_dma = [&](int slot) {     // request this wave's pieces of its next K step; wp_seg() must follow before the next one
        float* st = smem + slot * SLOT_FL + wk * WP_FL;
#pragma unroll
        for (int pa = 0; pa < WPA; ++pa)
            __builtin_amdgcn_global_load_lds((cg_gbl_void*)(wpA + offW[pa]), (cg_lds_void*)(st + 256 * pa), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((cg_gbl_void*)(wpB + offW[WPA]), (cg_lds_void*)(st + 256 * WPA), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((cg_gbl_void*)(wpB + offW[WPA + 1]), (cg_lds_void*)(st + 256 * WPA + 256), 16, 0, 0);
        wpA += BK * 4;
        wpB += wpStepB;
        wpK += BK;
    };
    auto wp_seg = [&]() {             // kept out of wp_dma so that the K loop's body is one basic block
        if (wpK >= wpSegK) {
            wpK = 0;
            ++wpSeg;
            setupW();
        }
    };
    if constexpr (WP) setupW();
    const int SLx = DEEP ? xb_uni(a.slots) : 0;
    // K steps whose operands are requested up front: all of them when the ring holds the whole K range (every slot is then
    // used once and the K loop needs neither counted waits nor barriers), else what the ring has in flight
    const bool ALLIN = DEEP && total_iters <= SLx;
    const int PRE = DEEP ? (ALLIN ? total_iters : SLx - 1) : 0;
    if constexpr (DEEP) {
        setupA();
        setupB();
        for (int s = 0; s < PRE; ++s) issueB(s);
        hook();   // the grid barrier's wait: from here on the other blocks' stores of the previous phase are visible
    }
#ifdef RFN_CHAIN_TIMING
    uint64_t* g_cg_stamp_local = XB ? reinterpret_cast<uint64_t*>(smem + CG_MAX_SLOTS * 0 + 8 * SLOT_FL) : nullptr;   // behind the 8-slot ring
#endif
    CG_STAMP(0);

    // ---- everything the epilogue reads from global memory is requested first (oldest in the vector-memory queue: the
    // counted waits of the K loop then never wait for more than the K step they need) and lands under the loop -----------
    cg_f32x4 e_prev[NV], e_bias[NV];                       // store epilogue
    float g_prev[NP][4], g_bias[NP][4], g_cprev[NP];       // gate epilogue
    float b_in[(EPI == CG_EPI_LSTM_BWD) ? NE : 1][9];      // gate-gradient epilogue: prev dh, dh_ext, i f o g, c_prev, c_next, dc_next
    if constexpr (EPI == CG_EPI_STORE) {
        constexpr int C4 = BN / 4;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int idx = tid + v * T;
            const int row = idx / C4, c4 = idx - row * C4;
            int grow = row0 + row;
            grow = grow < M ? grow : M - 1;
            const int col = col0 + 4 * c4;
            cg_f32x4 b = {0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < nseg; ++s) {
                const float* bp = a.seg[seg0 + s].bias;
                if (bp) b += *reinterpret_cast<const cg_f32x4*>(bp + col);
            }
            e_bias[v] = b;
            e_prev[v] = cg_f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (XB) {
                if (O.accumulate) e_prev[v] = xb_buf_ld4_sc1(xb_rsrc(xb_uni_ptr(O.C)), (uint32_t)(((long)grow * O.ldc + col) * 4));
            } else {
                if (O.accumulate) {
                    e_prev[v] = *reinterpret_cast<const cg_f32x4*>(O.C + (long)grow * O.ldc + col);
                    for (int pp = 0; pp < O.acc_parts; ++pp)   // partial slabs of an earlier launch, in order
                        e_prev[v] += *reinterpret_cast<const cg_f32x4*>(O.acc_slabs + pp * O.acc_stride + (long)grow * O.ldc + col);
                }
            }
        }
    } else if constexpr (EPI == CG_EPI_LSTM) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int idx = tid + p * T;
            const int row = idx / U, u = idx - row * U;
            int grow = row0 + row;
            grow = grow < M ? grow : M - 1;
            const int unit = tn * U + u;
            const float* G = O.C + (long)grow * O.ldc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float b = 0.f;
                for (int s = 0; s < nseg; ++s) {
                    const float* bp = a.seg[seg0 + s].bias;
                    if (bp) b += bp[g * R + unit];
                }
                g_bias[p][g] = b;
                g_prev[p][g] = O.accumulate ? xb_ld1<XB>(G + g * R + unit) : 0.f;
            }
            g_cprev[p] = xb_ld1<XB>(O.c_prev + (long)grow * O.ldcp + unit);
        }
    } else {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int idx = tid + e * T;
            const int row = idx / BN, cc = idx - row * BN;
            int grow = row0 + row;
            grow = grow < M ? grow : M - 1;
            const int unit = col0 + cc;
            const float* G = O.gates + (long)grow * O.ldg;
            b_in[e][0] = O.accumulate ? xb_ld1<XB>(O.C + (long)grow * O.ldc + unit) : 0.f;
            if constexpr (!XB) {
                if (O.accumulate)
                    for (int pp = 0; pp < O.acc_parts; ++pp) b_in[e][0] += O.acc_slabs[pp * O.acc_stride + (long)grow * O.ldc + unit];
            }
            b_in[e][1] = O.dh_ext ? xb_ld1<XB>(O.dh_ext + (long)grow * O.lddh + unit) : 0.f;
            b_in[e][2] = xb_ld1<XB>(G + unit);
            b_in[e][3] = xb_ld1<XB>(G + R + unit);
            b_in[e][4] = xb_ld1<XB>(G + 2 * R + unit);
            b_in[e][5] = xb_ld1<XB>(G + 3 * R + unit);
            b_in[e][6] = xb_ld1<XB>(O.c_prev + (long)grow * O.ldcp + unit);
            b_in[e][7] = xb_ld1<XB>(O.c_next + (long)grow * O.ldcn + unit);
            b_in[e][8] = O.dc_next ? xb_ld1<XB>(O.dc_next + (long)grow * O.lddcn + unit) : 0.f;
        }
    }

    // Ring of SL slots, SL - 1 K steps in flight: a step is a few hundred matrix-pipe cycles but a microsecond of L2 / fabric
    // latency under load, so the ring is as deep as the LDS of the blocks sharing a CU allows (host: cg_dispatch).
    const int SL = DEEP ? SLx : a.slots;
    int issued = 0;
    if constexpr (DEEP) {
        for (int s = 0; s < PRE; ++s) issueA(s);
        issued = PRE;
    } else if constexpr (WP) {
        for (int s = 0; s < CG_WP_SLOTS && s < total_iters; ++s) {   // every slot: a step's slot is requested again as soon as
            wp_dma(s);                                                // its fragments are in registers
            wp_seg();
            ++issued;
        }
    } else {
        for (int s = 0; s < SL - 1 && s < total_iters; ++s) {
            issue(s);
            ++issued;
        }
    }

    CG_STAMP(1);
    const int swa = swz(l31), swb = swz(l31);   // tile rows are l31 + multiples of 32
    auto k_step = [&](const float* a_l) {   // the MFMAs of one K step on the slot at a_l
        const float* b_l = a_l + A_FL;
        if constexpr (SMALL) {
#pragma unroll
            for (int t = 0; t < KG / WK; ++t) {
                const int q = cg_kgroup<WK>(wk, t);
                // lane (row l15, k-quarter kq) feeds k = 8q + {0 4 1 5}[kq] to the first MFMA of the k-group and
                // 8q + {2 6 3 7}[kq] to the second: 16-B chunk 2q + (kq & 1) of its row, elements kq >> 1 and (kq >> 1) + 2
                const int ch = 2 * q + (kq & 1);
                const cg_f32x4 af = *reinterpret_cast<const cg_f32x4*>(a_l + l15 * BK + 4 * (ch ^ swz(l15)));
                const float a0 = (kq < 2) ? af[0] : af[1], a1 = (kq < 2) ? af[2] : af[3];
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) {
                    const int n = 16 * nh + l15;
                    float b0, b1;
                    if constexpr (BKF) {
                        const cg_f32x4 bf = *reinterpret_cast<const cg_f32x4*>(b_l + n * BK + 4 * (ch ^ swz(n)));
                        b0 = (kq < 2) ? bf[0] : bf[1];
                        b1 = (kq < 2) ? bf[2] : bf[3];
                    } else {
                        const int k0 = 8 * q + 4 * (kq & 1) + (kq >> 1);
                        b0 = b_l[k0 * BN + n];
                        b1 = b_l[(k0 + 2) * BN + n];
                    }
                    acc16[nh] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc16[nh], 0, 0, 0);
                    acc16[nh] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc16[nh], 0, 0, 0);
                }
            }
            return;
        }
#pragma unroll
        for (int t = 0; t < KG / WK; ++t) {
            const int q = cg_kgroup<WK>(wk, t);
            const cg_f32x4 af = *reinterpret_cast<const cg_f32x4*>(a_l + (wm * 32 + l31) * BK + 4 * ((2 * q + h) ^ swa));
            cg_f32x4 bf;
            if constexpr (BKF) {
                bf = *reinterpret_cast<const cg_f32x4*>(b_l + l31 * BK + 4 * ((2 * q + h) ^ swb));
            } else {
                const float* pb = b_l + (8 * q + 4 * h) * BN + l31;
                bf[0] = pb[0];
                bf[1] = pb[BN];
                bf[2] = pb[2 * BN];
                bf[3] = pb[3 * BN];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c], bf[c], acc, 0, 0, 0);
        }
    };
    if constexpr (WP) {
        struct WpFrag {
            cg_f32x4 af[2];
            cg_f32x4 bf[BKF ? (SMALL ? 4 : 2) : 1];
            float bs[BKF ? 1 : 8];
        };
        auto wp_wait = [&](int younger_steps) {   // this wave's requests of all but the `younger_steps` newest steps have landed
            switch (younger_steps) {
                case 0: cg_wait_vmcnt<0>(); break;
                case 1: cg_wait_vmcnt<WPN>(); break;
                case 2: cg_wait_vmcnt<2 * WPN>(); break;
                case 3: cg_wait_vmcnt<3 * WPN>(); break;
                case 4: cg_wait_vmcnt<4 * WPN>(); break;
                case 5: cg_wait_vmcnt<5 * WPN>(); break;
                default: cg_wait_vmcnt<6 * WPN>(); break;
            }
        };
        auto wp_read = [&](int slot, WpFrag& f) {
            const float* w_l = smem + slot * SLOT_FL + wk * WP_FL;
            const float* wb_l = w_l + 256 * WPA;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if constexpr (SMALL) {
                    const int js = (2 * t + (kq & 1)) ^ wp_g(l15);
                    f.af[t] = *reinterpret_cast<const cg_f32x4*>(w_l + (l15 * 4 + js) * 4);
                    if constexpr (BKF) {
                        f.bf[2 * t] = *reinterpret_cast<const cg_f32x4*>(wb_l + (l15 * 4 + js) * 4);
                        f.bf[2 * t + 1] = *reinterpret_cast<const cg_f32x4*>(wb_l + 256 + (l15 * 4 + js) * 4);
                    } else {
                        const int kr = 4 * (kq & 1) + (kq >> 1);      // k-row of the first MFMA's k inside the k-group; + 2: the second's
                        const float* pb = wb_l + 256 * t + kr * 32;
                        const int nx = 16 * (kq & 1);
#pragma unroll
                        for (int nh = 0; nh < 2; ++nh) {
                            f.bs[4 * t + 2 * nh] = pb[(16 * nh + l15) ^ nx];
                            f.bs[4 * t + 2 * nh + 1] = pb[64 + ((16 * nh + l15) ^ nx)];
                        }
                    }
                } else {   // 32 rows: lane (row l31, k-half h) feeds k = 8q + 4h + c to the c-th MFMA of k-group q = wk + 4t
                    const int rp = l31 & 15, js = (2 * t + h) ^ wp_g(rp), o = (l31 >> 4) * 256 + (rp * 4 + js) * 4;
                    f.af[t] = *reinterpret_cast<const cg_f32x4*>(w_l + o);
                    if constexpr (BKF) {
                        f.bf[t] = *reinterpret_cast<const cg_f32x4*>(wb_l + o);
                    } else {
                        const float* pb = wb_l + 256 * t + (4 * h) * 32 + (l31 ^ (16 * h));
#pragma unroll
                        for (int c = 0; c < 4; ++c) f.bs[4 * t + c] = pb[c * 32];
                    }
                }
            }
        };
        // one v_cndmask per operand (written as `lo ? v[0] : v[1]` the compiler makes it a dynamic element extract: three)
        const bool wp_lo = kq < 2;
        auto wp_sel = [&](float x, float y) -> float {
            asm("" : "+v"(x), "+v"(y));
            return wp_lo ? x : y;
        };
        auto wp_mfma = [&](const WpFrag& f) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if constexpr (SMALL) {
                    const float a0 = wp_sel(f.af[t][0], f.af[t][1]), a1 = wp_sel(f.af[t][2], f.af[t][3]);
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh) {
                        float b0, b1;
                        if constexpr (BKF) {
                            b0 = wp_sel(f.bf[2 * t + nh][0], f.bf[2 * t + nh][1]);
                            b1 = wp_sel(f.bf[2 * t + nh][2], f.bf[2 * t + nh][3]);
                        } else {
                            b0 = f.bs[4 * t + 2 * nh];
                            b1 = f.bs[4 * t + 2 * nh + 1];
                        }
                        acc16[nh] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc16[nh], 0, 0, 0);
                        acc16[nh] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc16[nh], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.af[t][c], BKF ? f.bf[BKF ? t : 0][c] : f.bs[BKF ? 0 : 4 * t + c], acc, 0, 0, 0);
                }
            }
        };
        constexpr int WSL = CG_WP_SLOTS;
        WpFrag fc, fn;
        if (total_iters >= WSL) {
            // steady state: step it's fragments are in registers, steps it + 1 ... it + WSL - 1 are in flight or landed
            wp_wait(WSL - 1);
            wp_read(0, fc);
            int rd = 1, fl = 0;     // slot of step it + 1; slot of step it (free: its fragments are in registers)
            [[maybe_unused]] int wp_it = 0;   // probe builds only (CG_LOOP_STAMPS)
            for (int it = total_iters - WSL; it > 0; --it) {
                CG_LSTAMP(wp_it, 0);
                cg_wait_vmcnt<(WSL - 2) * WPN>();
                CG_LSTAMP(wp_it, 1);
                ++wp_it;
                wp_read(rd, fn);
                wp_mfma(fc);
                wp_dma(fl);
                fc = fn;
                rd = (rd + 1 == WSL) ? 0 : rd + 1;
                fl = (fl + 1 == WSL) ? 0 : fl + 1;
                wp_seg();
            }
#pragma unroll
            for (int j = 0; j < WSL; ++j) {   // the last WSL steps: nothing left to request
                if (j < WSL - 1) {
                    wp_wait(WSL - 2 - j);
                    wp_read(rd, fn);
                }
                wp_mfma(fc);
                fc = fn;
                rd = (rd + 1 == WSL) ? 0 : rd + 1;
            }
        } else {   // fewer K steps than slots: all requested above
            for (int it = 0; it < total_iters; ++it) {
                wp_wait(total_iters - it - 1);
                wp_read(it, fc);
                wp_mfma(fc);
            }
        }
    } else {
    int cur = 0, fill = SL - 1;
    if (ALLIN) {   // DEEP, whole K range resident: one wait, one barrier, then a loop the compiler can pipeline (same k order)
        cg_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        for (int it = 0; it < total_iters; ++it) k_step(smem + it * SLOT_FL);
    } else
    for (int it = 0; it < total_iters; ++it) {
        CG_LSTAMP(it, 0);
        // this wave's pieces of step `it` have landed; the `younger` steps issued after it stay in flight
        if constexpr (DEEP) {
            // issue order: B of the PRE up-front steps | A of those steps | then whole steps (A, B) from inside the loop
            const int inloop = issued - PRE;                                   // whole steps issued from inside the loop
            const int younger = it < PRE ? (PRE - 1 - it) * NA + inloop * (NA + NB) : (issued - 1 - it) * (NA + NB);
            cg_wait_vmcnt_dyn(younger);
        } else {
            switch (issued - it - 1) {
                case 0: cg_wait_vmcnt<0>(); break;
                case 1: cg_wait_vmcnt<NIW>(); break;
                case 2: cg_wait_vmcnt<2 * NIW>(); break;
                case 3: cg_wait_vmcnt<3 * NIW>(); break;
                case 4: cg_wait_vmcnt<4 * NIW>(); break;
                default: cg_wait_vmcnt<5 * NIW>(); break;
            }
        }
        CG_LSTAMP(it, 1);
        __builtin_amdgcn_s_barrier();                     // ... everyone's have, and slot (it - 1) % SL is free
        CG_LSTAMP(it, 2);
#if CG_ISSUE_AFTER
        // this step's fragment reads and MFMAs go out first: the matrix pipe then works while the wave does the address
        // arithmetic and the issue of the next step's requests (a K step is one serial chain on a wave that has its SIMD to
        // itself; tools/cg_loop_probe.hip)
        k_step(smem + cur * SLOT_FL);
        CG_LSTAMP(it, 3);
#endif
        if (issued < total_iters) {
            if constexpr (DEEP) {
                issueA(fill);
                issueB(fill);
            } else {
                issue(fill);
            }
            ++issued;
        }
#if !CG_ISSUE_AFTER
        CG_LSTAMP(it, 3);
        k_step(smem + cur * SLOT_FL);
#endif
        CG_LSTAMP(it, 4);
        cur = (cur + 1 == SL) ? 0 : cur + 1;
        fill = (fill + 1 == SL) ? 0 : fill + 1;
    }
    }

    // ---- the WK partial tiles meet in LDS (the ring is free: every DMA has been waited for) ----------------------------
    CG_STAMP(2);
    __syncthreads();
    float* slab = smem;   // [WK][BM][BN]
    if constexpr (SMALL) {
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int i = 0; i < 4; ++i) slab[(wk * BM + 4 * kq + i) * BN + 16 * nh + l15] = acc16[nh][i];
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            slab[(wk * BM + row) * BN + l31] = acc[i];
        }
    }
    __syncthreads();

    if constexpr (EPI == CG_EPI_STORE) {
        constexpr int C4 = BN / 4;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int idx = tid + v * T;
            const int row = idx / C4, c4 = idx - row * C4, grow = row0 + row;
            if (grow >= M || (PART && row >= BM)) continue;
            cg_f32x4 x = *reinterpret_cast<const cg_f32x4*>(slab + row * BN + 4 * c4);
#pragma unroll
            for (int w = 1; w < WK; ++w) x += *reinterpret_cast<const cg_f32x4*>(slab + (w * BM + row) * BN + 4 * c4);
            x += e_bias[v];
            if (O.accumulate) x += e_prev[v];
            if constexpr (XB) xb_buf_st4_sc1(xb_rsrc(xb_uni_ptr(O.C)), (uint32_t)(((long)grow * O.ldc + col0 + 4 * c4) * 4), x);
            else *reinterpret_cast<cg_f32x4*>(O.C + (long)grow * O.ldc + col0 + 4 * c4) = x;
        }
    } else if constexpr (EPI == CG_EPI_LSTM) {
        // LSTM gate epilogue (rfn_cell.hip lstm_fwd_k, same formulas): tile column g * U + u = gate g of unit tn * U + u
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int idx = tid + p * T;
            const int row = idx / U, u = idx - row * U, grow = row0 + row;
            if (grow >= M || (PART && row >= BM)) continue;
            const int unit = tn * U + u;
            float* G = O.C + (long)grow * O.ldc;
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float s = slab[row * BN + g * U + u];
#pragma unroll
                for (int w = 1; w < WK; ++w) s += slab[(w * BM + row) * BN + g * U + u];
                pre[g] = (s + g_bias[p][g]) + g_prev[p][g];
            }
            const float ig = rfn_sigmoid(pre[0]), fg = rfn_sigmoid(pre[1]), og = rfn_sigmoid(pre[2]);
            const float gg = tanhf(pre[3]);
            xb_st1<XB>(G + unit, ig);
            xb_st1<XB>(G + R + unit, fg);
            xb_st1<XB>(G + 2 * R + unit, og);
            xb_st1<XB>(G + 3 * R + unit, gg);
            const float c = fg * g_cprev[p] + ig * gg;
            xb_st1<XB>(O.c_next + (long)grow * O.ldcn + unit, c);
            float hv = og * tanhf(c);
            if (a.drop_p > 0.f) {
                const float uu = rfn_philox_uniform(a.seed, O.drop_offset, (uint64_t)((long)grow * R + unit));
                hv = (uu >= a.drop_p) ? hv * (1.0f / (1.0f - a.drop_p)) : 0.f;
            }
            xb_st1<XB>(O.h_next + (long)grow * O.ldh + unit, hv);
        }
    } else {
        // The product is the recurrent part of d h of the cell call that produced `gates` (the next one the backward
        // sweep processes): finish that gradient and run its LSTM backward here (rfn_cell.hip lstm_bwd_k, same formulas).
        // Contraction is off in this block: whether `dc = dhv * og * (1 - tc^2)` and the conditional `dc += dc_next` fuse into
        // an fma depended on how the compiler shaped the branch around them, which differs between the launch form (uniform
        // kernel arguments) and the persistent form (descriptor in LDS) -- one rounding, but the two forms must agree bit for bit.
#pragma clang fp contract(off)
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int idx = tid + e * T;
            const int row = idx / BN, cc = idx - row * BN, grow = row0 + row;
            if (grow >= M) continue;
            const int unit = col0 + cc;
            float s = slab[row * BN + cc];
#pragma unroll
            for (int w = 1; w < WK; ++w) s += slab[(w * BM + row) * BN + cc];
            float dhv = (s + b_in[e][0]) + b_in[e][1];
            if (O.C) xb_st1<XB>(O.C + (long)grow * O.ldc + unit, dhv);   // total d h of that call (kept for the caller's bookkeeping)
            if (a.drop_p > 0.f) {
                const float uu = rfn_philox_uniform(a.seed, O.drop_offset, (uint64_t)((long)grow * R + unit));
                dhv = (uu >= a.drop_p) ? dhv * (1.0f / (1.0f - a.drop_p)) : 0.f;
            }
            const float ig = b_in[e][2], fg = b_in[e][3], og = b_in[e][4], gg = b_in[e][5];
            const float tc = tanhf(b_in[e][7]);
            float dc = dhv * og * (1.0f - tc * tc);
            if (O.dc_next) dc += b_in[e][8];
            const float d_o = dhv * tc;
            const float d_i = dc * gg;
            const float d_f = dc * b_in[e][6];
            const float d_g = dc * ig;
            float* G = O.gates + (long)grow * O.ldg;
            xb_st1<XB>(G + unit, d_i * ig * (1.0f - ig));
            xb_st1<XB>(G + R + unit, d_f * fg * (1.0f - fg));
            xb_st1<XB>(G + 2 * R + unit, d_o * og * (1.0f - og));
            xb_st1<XB>(G + 3 * R + unit, d_g * (1.0f - gg * gg));
            xb_st1<XB>(O.dc_prev + (long)grow * O.lddcp + unit, dc * fg);
        }
    }
    CG_STAMP(3);
}

