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
//     dealt round-robin), partial tiles meet in LDS and are added in wave order -- no partial slabs in HBM, no second
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

#include "rfn_common.h"

typedef float cg_f32x16 __attribute__((ext_vector_type(16)));
typedef float cg_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void cg_lds_void;
typedef const __attribute__((address_space(1))) void cg_gbl_void;

#define CG_MAXSEG 24
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
};
struct CgArgs {
    int M, nout, R, tiles_m;
    float drop_p;
    int slots;
    unsigned long long seed;
    CgOut out[RFN_CELL_MAXOUT];
    CgSeg seg[CG_MAXSEG];
};


template <int N>
__device__ __forceinline__ void cg_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// BM x 32 output tile, K steps of BK, WK waves per 32x32 sub-tile (each takes every WK-th k-group of 8).
// BKF: B is an nn.Linear weight [n][k] (forward products); !BKF: B is [k][n] (dX = dY . W: the reduction index is W's row).
template <int BM, int BK, int WK, bool BKF, int EPI>
__global__ __launch_bounds__(64 * (BM / 32) * WK) void cell_gemm_k(const CgArgs a) {
    constexpr int BN = CG_BN;
    constexpr int WM = BM / 32, W = WM * WK, T = 64 * W;
    constexpr int A_FL = BM * BK, B_FL = BN * BK, SLOT_FL = A_FL + B_FL;
    constexpr int PA = A_FL / 256, PB = B_FL / 256, P = PA + PB;   // 1-KiB pieces per K step
    static_assert(P % W == 0, "pieces must divide evenly over the waves");
    constexpr int NIW = P / W;
    constexpr int CPR = BK / 4;      // 16-B chunks per [row][k] row
    constexpr int RPP = 64 / CPR;    // rows per piece
    constexpr int KG = BK / 8;       // k-groups per K step
    static_assert(KG % WK == 0 && (BK == 32 || BK == 64), "unsupported K step");
    static_assert(WK * BM * BN <= 2 * SLOT_FL, "the partial tiles reuse the ring (at least two slots)");
    static_assert(NIW * (CG_MAX_SLOTS - 1) <= 63, "vmcnt is a 6-bit counter");
    static_assert((EPI == CG_EPI_LSTM) ? BKF : true, "the gate epilogue belongs to forward products");
    static_assert((EPI == CG_EPI_LSTM_BWD) ? !BKF : true, "the gate-gradient epilogue belongs to dX products");
    constexpr int U = BN / 4;                       // units per tile of the gate epilogue
    constexpr int NV = BM * BN / 4 / T;             // float4 of the tile per thread       (store epilogue)
    constexpr int NP = BM * U / T;                  // (row, unit) pairs per thread         (gate epilogue)
    constexpr int NE = BM * BN / T;                 // (row, unit) elements per thread      (gate-gradient epilogue)
    static_assert(NV >= 1 && NP >= 1 && BM * BN % (4 * T) == 0 && BM * U % T == 0, "epilogue tiling");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave / WM, wm = wave - wk * WM;
    const int l31 = lane & 31, h = lane >> 5;

    int r = 0;
    const int bid = blockIdx.x;
    for (int i = 1; i < a.nout; ++i)
        if (bid >= a.out[i].tile0) r = i;
    const CgOut& O = a.out[r];
    const int lt = bid - O.tile0;
    const int tm = lt / O.tiles_n, tn = lt - tm * O.tiles_n;
    const int row0 = tm * BM, col0 = tn * BN;
    const int M = a.M, R = a.R;
    const int nseg = O.nseg, seg0 = O.seg0;

    auto swz = [](int row) -> int { return BK == 32 ? ((row >> 1) & 7) : (row & 15); };

    cg_f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    int total_iters = 0;
    for (int s = 0; s < nseg; ++s) total_iters += a.seg[seg0 + s].K / BK;

    // per-lane byte offsets of this wave's pieces inside the current segment's operands; piece p = wave + W * j
    uint32_t off[NIW];
    const char* baseA = nullptr;
    const char* baseB = nullptr;
    long stepB = 0;
    int seg = 0, k0 = 0, segK = 0;
    auto setup = [&]() {
        if (seg >= nseg) return;
        const CgSeg& sg = a.seg[seg0 + seg];
        segK = sg.K;
        baseA = (const char*)sg.A;
        baseB = (const char*)sg.B;
        stepB = BKF ? (long)BK * 4 : (long)BK * sg.ldb * 4;
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
    setup();
    auto issue = [&](int slot) {
        float* st = smem + slot * SLOT_FL;
#pragma unroll
        for (int j = 0; j < NIW; ++j) {
            const int p = wave + W * j;
            const char* base = (p < PA) ? baseA : baseB;
            __builtin_amdgcn_global_load_lds((cg_gbl_void*)(base + off[j]), (cg_lds_void*)(st + p * 256), 16, 0, 0);
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
            if (O.accumulate) e_prev[v] = *reinterpret_cast<const cg_f32x4*>(O.C + (long)grow * O.ldc + col);
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
                g_prev[p][g] = O.accumulate ? G[g * R + unit] : 0.f;
            }
            g_cprev[p] = O.c_prev[(long)grow * O.ldcp + unit];
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
            b_in[e][0] = O.accumulate ? O.C[(long)grow * O.ldc + unit] : 0.f;
            b_in[e][1] = O.dh_ext ? O.dh_ext[(long)grow * O.lddh + unit] : 0.f;
            b_in[e][2] = G[unit];
            b_in[e][3] = G[R + unit];
            b_in[e][4] = G[2 * R + unit];
            b_in[e][5] = G[3 * R + unit];
            b_in[e][6] = O.c_prev[(long)grow * O.ldcp + unit];
            b_in[e][7] = O.c_next[(long)grow * O.ldcn + unit];
            b_in[e][8] = O.dc_next ? O.dc_next[(long)grow * O.lddcn + unit] : 0.f;
        }
    }

    // Ring of SL slots, SL - 1 K steps in flight: a step is a few hundred matrix-pipe cycles but a microsecond of L2 / fabric
    // latency under load, so the ring is as deep as the LDS of the blocks sharing a CU allows (host: cg_dispatch).
    const int SL = a.slots;
    int issued = 0;
    for (int s = 0; s < SL - 1 && s < total_iters; ++s) {
        issue(s);
        ++issued;
    }

    const int swa = swz(l31), swb = swz(l31);   // tile rows are l31 + multiples of 32
    int cur = 0, fill = SL - 1;
    for (int it = 0; it < total_iters; ++it) {
        // this wave's pieces of step `it` have landed; the `younger` steps issued after it stay in flight
        switch (issued - it - 1) {
            case 0: cg_wait_vmcnt<0>(); break;
            case 1: cg_wait_vmcnt<NIW>(); break;
            case 2: cg_wait_vmcnt<2 * NIW>(); break;
            case 3: cg_wait_vmcnt<3 * NIW>(); break;
            case 4: cg_wait_vmcnt<4 * NIW>(); break;
            default: cg_wait_vmcnt<5 * NIW>(); break;
        }
        __builtin_amdgcn_s_barrier();                     // ... everyone's have, and slot (it - 1) % SL is free
        if (issued < total_iters) {
            issue(fill);
            ++issued;
        }
        const float* a_l = smem + cur * SLOT_FL;
        const float* b_l = a_l + A_FL;
#pragma unroll
        for (int t = 0; t < KG / WK; ++t) {
            const int q = wk + WK * t;
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
        cur = (cur + 1 == SL) ? 0 : cur + 1;
        fill = (fill + 1 == SL) ? 0 : fill + 1;
    }

    // ---- the WK partial tiles meet in LDS (the ring is free: every DMA has been waited for) ----------------------------
    __syncthreads();
    float* slab = smem;   // [WK][BM][BN]
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        slab[(wk * BM + row) * BN + l31] = acc[i];
    }
    __syncthreads();

    if constexpr (EPI == CG_EPI_STORE) {
        constexpr int C4 = BN / 4;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int idx = tid + v * T;
            const int row = idx / C4, c4 = idx - row * C4, grow = row0 + row;
            if (grow >= M) continue;
            cg_f32x4 x = *reinterpret_cast<const cg_f32x4*>(slab + row * BN + 4 * c4);
#pragma unroll
            for (int w = 1; w < WK; ++w) x += *reinterpret_cast<const cg_f32x4*>(slab + (w * BM + row) * BN + 4 * c4);
            x += e_bias[v];
            if (O.accumulate) x += e_prev[v];
            *reinterpret_cast<cg_f32x4*>(O.C + (long)grow * O.ldc + col0 + 4 * c4) = x;
        }
    } else if constexpr (EPI == CG_EPI_LSTM) {
        // LSTM gate epilogue (rfn_cell.hip lstm_fwd_k, same formulas): tile column g * U + u = gate g of unit tn * U + u
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int idx = tid + p * T;
            const int row = idx / U, u = idx - row * U, grow = row0 + row;
            if (grow >= M) continue;
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
            G[unit] = ig;
            G[R + unit] = fg;
            G[2 * R + unit] = og;
            G[3 * R + unit] = gg;
            const float c = fg * g_cprev[p] + ig * gg;
            O.c_next[(long)grow * O.ldcn + unit] = c;
            float hv = og * tanhf(c);
            if (a.drop_p > 0.f) {
                const float uu = rfn_philox_uniform(a.seed, O.drop_offset, (uint64_t)((long)grow * R + unit));
                hv = (uu >= a.drop_p) ? hv * (1.0f / (1.0f - a.drop_p)) : 0.f;
            }
            O.h_next[(long)grow * O.ldh + unit] = hv;
        }
    } else {
        // The product is the recurrent part of d h of the cell call that produced `gates` (the next one the backward
        // sweep processes): finish that gradient and run its LSTM backward here (rfn_cell.hip lstm_bwd_k, same formulas).
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
            if (O.C) O.C[(long)grow * O.ldc + unit] = dhv;   // total d h of that call (kept for the caller's bookkeeping)
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
            G[unit] = d_i * ig * (1.0f - ig);
            G[R + unit] = d_f * fg * (1.0f - fg);
            G[2 * R + unit] = d_o * og * (1.0f - og);
            G[3 * R + unit] = d_g * (1.0f - gg * gg);
            O.dc_prev[(long)grow * O.lddcp + unit] = dc * fg;
        }
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------
struct CgDevState {
    bool set[16] = {};
};
template <int BM, int BK, int WK, bool BKF, int EPI>
static int cg_launch(CgArgs& a, int blocks, int max_iters, int force_slots, hipStream_t st) {
    auto k = cell_gemm_k<BM, BK, WK, BKF, EPI>;
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
    // Ring depth 3 (two K steps in flight).  Measured on MI355X (tools/bench_cellgemm.py --slots 2,3,4,6): 2 and 3 slots tie,
    // deeper rings LOSE 10-40 % -- the kernel is bound by the rate of the L2 -> LDS path at 8 flops per operand byte, not by
    // its latency, and what helps is more co-resident blocks per CU (each with its own barrier), i.e. LESS LDS per block.
    (void)blocks;
    int slots = 3;
    if (force_slots > 0) slots = force_slots;   // tools only
    if (slots > CG_MAX_SLOTS) slots = CG_MAX_SLOTS;
    if (slots > max_iters + 1) slots = max_iters + 1;
    if (slots < 2) slots = 2;
    a.slots = slots;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * (BM / 32) * WK), slots * slot, st, a);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}

// Tile variants: 1 = 64 rows, K step 32, 2 K-waves; 2 = 64 rows, K step 64, 4 K-waves; 3 = 32 rows, K step 64, 4 K-waves.
// variant 0: the K step / K-wave count -- which fix the k order of every output element -- are chosen FROM THE K COUNTS
// ONLY, so a row's arithmetic never depends on the batch it sits in (variants 2 and 3 give bit-identical results).
template <bool BKF, int EPI>
static int cg_dispatch(CgArgs& a, const rfn_cell_out* outs, int variant, hipStream_t st) {
    const int force_slots = (variant >> 4) & 15;   // tools: bits 4-7 of `variant` force the ring depth
    variant &= 15;
    bool k64 = true;
    int max_iters = 0;
    for (int o = 0; o < a.nout; ++o) {
        int it = 0;
        for (int s = 0; s < a.out[o].nseg; ++s) {
            k64 = k64 && (a.seg[a.out[o].seg0 + s].K % 64 == 0);
            it += a.seg[a.out[o].seg0 + s].K;
        }
        max_iters = it > max_iters ? it : max_iters;
    }
    // 32-row tiles whenever the K step of 64 applies: measured fastest at every per-step shape of the path (B = 64 ... 640),
    // because they put two to three independent blocks on a CU (profiles/r03_cellgemm.md)
    if (variant == 0) variant = k64 ? 3 : 1;
    if ((variant == 2 || variant == 3) && !k64) return RFN_ERR_SHAPE;
    const int bm = (variant == 3) ? 32 : 64;
    max_iters /= (variant == 1) ? 32 : 64;
    a.tiles_m = rfn_cdiv(a.M, bm);
    int t0 = 0;
    for (int o = 0; o < a.nout; ++o) {
        a.out[o].tiles_n = a.out[o].N / CG_BN;
        a.out[o].tile0 = t0;
        t0 += a.tiles_m * a.out[o].tiles_n;
    }
    (void)outs;
    switch (variant) {
        case 1: return cg_launch<64, 32, 2, BKF, EPI>(a, t0, max_iters, force_slots, st);
        case 2: return cg_launch<64, 64, 4, BKF, EPI>(a, t0, max_iters, force_slots, st);
        case 3: return cg_launch<32, 64, 4, BKF, EPI>(a, t0, max_iters, force_slots, st);
        default: return RFN_ERR_SHAPE;
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

extern "C" int rfn_cell_gemm(int M, int nout, const rfn_cell_out* outs, int R, float drop_p, uint64_t seed, int variant,
                             void* stream) {
    if (!rfn_cell_gemm_supported(M, nout, outs, R)) return RFN_ERR_UNSUPPORTED;
    if (drop_p < 0.f || drop_p >= 1.f) return RFN_ERR_SHAPE;
    CgArgs a;
    memset(&a, 0, sizeof(a));
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
        for (int s = 0; s < t.nseg; ++s) {
            CgSeg& g = a.seg[ns++];
            g.A = t.seg[s].A; g.B = t.seg[s].B; g.bias = t.seg[s].bias;
            g.lda = t.seg[s].lda; g.ldb = t.seg[s].ldb; g.K = t.seg[s].K;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    const bool bkf = outs[0].seg[0].b_kfast != 0;
    if (outs[0].epilogue == RFN_CELL_EPI_LSTM) return cg_dispatch<true, CG_EPI_LSTM>(a, outs, variant, st);
    if (outs[0].epilogue == RFN_CELL_EPI_LSTM_BWD) return cg_dispatch<false, CG_EPI_LSTM_BWD>(a, outs, variant, st);
    if (bkf) return cg_dispatch<true, CG_EPI_STORE>(a, outs, variant, st);
    return cg_dispatch<false, CG_EPI_STORE>(a, outs, variant, st);
}
