// Exact-fp32 grouped / K-segmented GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces every nn.Linear / mm / addmm on the recurrent-fusion path (SURVEY.md 2.2 K0-K2, K5, K6,
// K8, K9, K11) and their backward matmuls.  The dominant instance is the hoisted attention feature
// projection att_2_att_h (misc/AttentionModelCore.py:32-34): (B*L x D) . (D x T1*A) per encoder,
// 91.5 % of the path's FLOPs, plus its weight gradient in backward.
//
// Numerics: every output element is a chain of fp32 fma's (MFMA f32 is bit-for-bit an fmaf chain,
// one rounding per product), i.e. true fp32 -- required for bit-exact greedy token ids.
//
// Structure (MI355X-first, not a CUDA tiling):
//   * 256 threads = 4 waves (2x2); block tile BMxBNx32, wave tile (BM/2)x(BN/2) built from 32x32
//     MFMA tiles, accumulators stay in the unified VGPR/AGPR file.
//   * two staging paths behind one entry point, bit-identical results (same k order per output element):
//       - LDS-DMA tile (gemm_tile_dma; interior 128x128 tiles): global_load_lds_dwordx4 into a ring of LDS slots,
//         one barrier per K step with the next slot in flight across it, unpadded images with the bank-conflict fix
//         as an XOR swizzle on the SOURCE address, and EVERY vector-memory instruction addressed as a wave-uniform
//         SGPR base + a 32-bit lane offset (with 64-bit per-lane addresses the issue of those instructions cost 10 %
//         of the matrix pipe: profiles/r02_pmc_gemm.md);
//       - register-staged tile (gemm_tile; ragged shapes, scalar-aligned operands, bias-gradient riders, 64x64
//         tiles): operands whose reduction index is contiguous staged [row][k] with a 36-float row (conflict-free
//         ds_read_b128), operands whose OUTPUT index is contiguous (the transposed operands of the backward GEMMs)
//         staged [k][row] and read with conflict-free ds_read_b32 -- no transposition pass anywhere.
//   * 1-D grid with a bijective XCD remap + 8-row bands so the blocks sharing an A row-panel and a
//     B column-panel run on one XCD's L2 at the same time; the last, partly filled round of a big launch runs as
//     half-height tiles (as quarter tiles when it is at most a quarter full: the projections of B <= 128 shards).
//   * options travel with the call (rfn_gemm_f32_opt flags); no environment variable is read.
#include <string.h>

#include "rfn_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GEMM_THREADS 256
// ---- tuning / ablation knobs (tools/gemm_bench.hip builds variants; the product uses the defaults) ----
#ifndef GEMM_MIN_WAVES
#define GEMM_MIN_WAVES 2   /* __launch_bounds__ waves per SIMD */
#endif
#ifndef GEMM_XCD_REMAP
#define GEMM_XCD_REMAP 1
#endif
#ifndef GEMM_SMALL_BK
#define GEMM_SMALL_BK 32     /* K step of the 64x64 tile */
#endif
#ifndef GEMM_SMALL_STAGES
#define GEMM_SMALL_STAGES 2
#endif
#ifndef GEMM_MEDIUM_MIN_FLOPS
#define GEMM_MEDIUM_MIN_FLOPS 6e9 /* per launch; below this the 64x64 tile + split-K stays */
#endif
#ifndef GEMM_MEDIUM_MAX
#define GEMM_MEDIUM_MAX 1023  /* below two full rounds of 128x128 tiles (512 slots each) the K split is priced by the cost model; above, unsplit */
#endif
#ifndef GEMM_SPLIT_MIN_ITERS
#define GEMM_SPLIT_MIN_ITERS 4 /* K iterations every split block keeps at least */
#endif
#ifndef GEMM_SPLIT_TARGET
#define GEMM_SPLIT_TARGET 768 /* blocks a split-K launch aims for */
#endif
#ifndef GEMM_ONE_WAVE
#define GEMM_ONE_WAVE 0
#endif
#ifndef GEMM_ONE_WAVE_STAGES
#define GEMM_ONE_WAVE_STAGES 1
#endif
#ifndef GEMM_BAND_ROWS
#define GEMM_BAND_ROWS 8    /* row tiles per band of the block order (B-operand panels are re-fetched once per band) */
#endif
#ifndef GEMM_FAST_PATH
#define GEMM_FAST_PATH 1   /* unchecked pointer-increment staging for interior single-segment problems */
#endif
#ifndef GEMM_SETPRIO
#define GEMM_SETPRIO 0     /* 1: s_setprio(1) around each K step's MFMA cluster; 2: static per-block priority */
#endif
#ifndef GEMM_FRAG_PIPE
#define GEMM_FRAG_PIPE 0   /* explicit register double-buffering of the LDS fragments */
#endif
#ifndef GEMM_BIG_BM
#define GEMM_BIG_BM 128
#endif
#ifndef GEMM_BIG_BN
#define GEMM_BIG_BN 128
#endif
#ifndef GEMM_BIG_BK
#define GEMM_BIG_BK 32     /* K step of the 128x128 tile */
#endif
#ifndef GEMM_XX_STAGES
#define GEMM_XX_STAGES 2   /* LDS stages of the big tile for the other layouts */
#endif
#ifndef GEMM_NT_STAGES
#define GEMM_NT_STAGES 1   /* LDS stages of the big NT tile: measured 131 TF single-buffered (3 blocks/CU) vs 125 */
#endif
#ifndef GEMM_TAIL_HALF
#define GEMM_TAIL_HALF 1   /* half-height tiles for the last, partly filled round of a big NT launch */
#endif
#ifndef GEMM_TAIL_QUARTER
#define GEMM_TAIL_QUARTER 1 /* quarter tiles when four per tail tile still fit one round (round 5) */
#endif
#ifndef GEMM_TAIL_MIN_ROUNDS
#define GEMM_TAIL_MIN_ROUNDS 2 /* full rounds in front of a tail round (4 until round 4: a B = 32 shard's projection is 3.06 rounds) */
#endif
#ifndef GEMM_DMA
#define GEMM_DMA 1         /* big interior tiles are staged by LDS-DMA (global_load_lds_dwordx4) into a ring of slots */
#endif
#ifndef GEMM_SMALL_DMA_SLOTS
#define GEMM_SMALL_DMA_SLOTS 3   /* > 0: interior 64 x 64 tiles take the LDS-DMA ring too, with this many slots (48 KB per block;
                                   0 / 4 / 3: 65.93-66.09 / 65.93-66.05 / 65.75-65.86 ms per step) */
#endif
#ifndef GEMM_DMA_MIN_WAVES
#define GEMM_DMA_MIN_WAVES 3   /* __launch_bounds__ waves per SIMD of the LDS-DMA kernels */
#endif
#ifndef GEMM_DMA_ABLATE
#define GEMM_DMA_ABLATE 0      /* diagnostics only (wrong results): 1 no DMA in the loop, 2 no barrier, 4 no LDS reads */
#endif
#ifndef GEMM_DMA_BK
#define GEMM_DMA_BK 32     /* K step of the LDS-DMA tile, both operands k-contiguous (projection): 32 or 16 */
#endif
#ifndef GEMM_DMA_BK_XX
#define GEMM_DMA_BK_XX 16  /* K step of the other layouts (weight gradient, dX): 16 x 3 slots measured 1 % over 32 x 2 */
#endif
#ifndef GEMM_DMA_SLOTS_NT
#define GEMM_DMA_SLOTS_NT 2  /* ring slots, both operands k-contiguous (projection) */
#endif
#ifndef GEMM_DMA_SLOTS_XX
#define GEMM_DMA_SLOTS_XX 3  /* ring slots, the other layouts */
#endif
#ifndef GEMM_ABLATE
#define GEMM_ABLATE 0      /* 1: skip global loads after the first tile, 2: skip the epilogue stores */
#endif

struct GemmArgs {
    int M, N, ngroups, accumulate;
    int tiles_m, tiles_n;
    int splitk;   // > 1: each tile's K iterations are cut into `splitk` ranges, raw partial tiles go to `part`
    int ws_mib;   // host side only: size of the split-K scratch in MiB
    unsigned flags;        // host side only: RFN_GEMM_OPT_* bits of the call
    int tail_main_blocks;   // > 0: blocks past this id process HALF-height tiles (see rfn_gemm_kernel)
    int tail_idx_main;      // first per-XCD tile index of the tail round
    int tail_parts;         // 2: half-height tiles (BM/2 x BN); 4: quarter tiles (BM/2 x BN/2), when four per tile still fit one round
    float* part;  // [ngroups][splitk][M][N]
    int* tickets; // non-NULL: one zeroed counter per output tile; the last K range to arrive finishes the tile in-kernel
    int n_tickets;
    rfn_gemm_lstm lstm;   // lstm.c_next != NULL: C is a gate buffer (M, 4R) and the split-K reduce ends in the LSTM update
    rfn_gemm_problem g[RFN_GEMM_MAXGROUP];
};

// ---- split-K finished inside the GEMM launch ------------------------------------------------------------------------
// Every K-range block has written its raw partial tile (and partial bias-gradient sums).  One agent-scope release per
// block, one ticket per block on the tile's counter; the block that draws the last ticket makes one agent-scope acquire
// and adds the slabs IN K-RANGE ORDER -- the order rfn_gemm_reduce_k uses, whoever arrives last -- plus bias / previous C.
// The counter is left at zero for the next launch.  (MI355X guide, 'Projection GEMM at M = 256' item 2: plain slab
// stores -> every wave's vmcnt(0) -> barrier -> lane 0 release fence -> vmcnt(0) -> relaxed agent fetch_add; reducer:
// lane 0 acquire fence -> vmcnt(0) -> barrier -> plain loads.  Correct for any placement of the blocks on XCDs / CUs.)
template <int BM, int BN, bool VEC, int THREADS>
__device__ __forceinline__ void gemm_finish_splitk(const GemmArgs& args, const rfn_gemm_problem& P, const int grp,
                                                   const int tile_id, const int row0, const int col0,
                                                   const bool colsum_tile, float* flag_lds) {
    const int tid = threadIdx.x;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int t = __hip_atomic_fetch_add(args.tickets + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *reinterpret_cast<volatile int*>(flag_lds) = t;
    }
    __syncthreads();
    const int ticket = *reinterpret_cast<volatile int*>(flag_lds);
    if (ticket != args.splitk - 1) return;
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(args.tickets + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int M = args.M, N = args.N, splitk = args.splitk;
    const long MN = (long)M * N;
    const float* part = args.part + (long)grp * splitk * MN;
    if constexpr (VEC) {   // N % 4 == 0, 16-B aligned C / bias are not guaranteed: only the slabs are read 16 B wide
        constexpr int C4 = BN / 4;
        for (int idx = tid; idx < BM * C4; idx += THREADS) {
            const int r = idx / C4, c4 = idx - r * C4, row = row0 + r, col = col0 + 4 * c4;
            if (row >= M || col >= N) continue;
            const float* p = part + (long)row * N + col;
            if (col + 4 <= N && (N & 3) == 0) {
                f32x4 sum = {0.f, 0.f, 0.f, 0.f};
                for (int k = 0; k < splitk; ++k) sum += *reinterpret_cast<const f32x4*>(p + k * MN);
                float* c = P.C + (long)row * P.ldc + col;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = sum[e];
                    for (int sg = 0; sg < P.nseg; ++sg)
                        if (P.seg[sg].bias) v += P.seg[sg].bias[col + e];
                    c[e] = args.accumulate ? c[e] + v : v;
                }
            } else {
                for (int e = 0; e < 4 && col + e < N; ++e) {
                    float v = 0.f;
                    for (int k = 0; k < splitk; ++k) v += p[k * MN + e];
                    for (int sg = 0; sg < P.nseg; ++sg)
                        if (P.seg[sg].bias) v += P.seg[sg].bias[col + e];
                    float* c = P.C + (long)row * P.ldc + col + e;
                    *c = args.accumulate ? *c + v : v;
                }
            }
        }
    } else {
        for (int idx = tid; idx < BM * BN; idx += THREADS) {
            const int r = idx / BN, cc = idx - r * BN, row = row0 + r, col = col0 + cc;
            if (row >= M || col >= N) continue;
            const float* p = part + (long)row * N + col;
            float v = 0.f;
            for (int k = 0; k < splitk; ++k) v += p[k * MN];
            for (int sg = 0; sg < P.nseg; ++sg)
                if (P.seg[sg].bias) v += P.seg[sg].bias[col];
            float* c = P.C + (long)row * P.ldc + col;
            *c = args.accumulate ? *c + v : v;
        }
    }
    if (colsum_tile && P.a_colsum) {   // the bias-gradient rider of this row tile: partial column sums of every K range, in order
        for (int r = tid; r < BM; r += THREADS) {
            const int row = row0 + r;
            if (row >= M) continue;
            const float* cs = args.part + (long)args.ngroups * splitk * MN + (long)grp * splitk * M + row;
            float v = 0.f;
            for (int k = 0; k < splitk; ++k) v += cs[(long)k * M];
            float* o = P.a_colsum + row;
            *o = args.accumulate ? *o + v : v;
        }
    }
}


// ---- staging of one ROWS x BK operand tile ---------------------------------------------------
// [row][k] tiles have BK+4 floats per row: 36 and 68 both put the 16 rows of a ds_read_b128 lane group on 16
// distinct 16-B bank slots (row*36 mod 64 and row*68 mod 64 = 4*row mod 64 are all different for 16 rows).
template <int ROWS, bool KFAST, bool VEC, int BK, int THREADS = GEMM_THREADS>
struct Stage {
    static constexpr int NV = ROWS * BK / 4 / THREADS;  // float4 per thread
    static constexpr int LDK = BK + 4;                       // [row][k] row length
    static constexpr int LDR = ROWS + 4;                     // [k][row] row length
    static constexpr int LDS_FLOATS = KFAST ? ROWS * LDK : BK * LDR;
    static constexpr int KQ = BK / 4;                        // float4 per [row][k] row
    static constexpr int RPASS = THREADS / KQ;          // rows covered per pass (kfast, vec)
    static constexpr int RQ = ROWS / 4;                      // float4 per [k][row] row
    static constexpr int KPASS = THREADS / RQ;          // k rows covered per pass (rowfast, vec)
    static constexpr int RPASS_S = THREADS / BK;        // rows per pass (kfast, scalar)
    static constexpr int KPASS_S = THREADS / ROWS;      // k rows per pass (rowfast, scalar)
    f32x4 v[NV];

    // rows [row0, row0+ROWS) x k [k0, k0+BK) of an operand with `nrows` valid rows and K valid k.
    __device__ __forceinline__ void load(const float* __restrict__ base, long ld, int row0,
                                         int nrows, int k0, int K, int tid) {
        if constexpr (VEC && KFAST) {
            const int k = k0 + 4 * (tid % KQ);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int r = row0 + tid / KQ + RPASS * j;
                f32x4 x = {0.f, 0.f, 0.f, 0.f};
                if (r < nrows && k < K) x = *reinterpret_cast<const f32x4*>(base + (long)r * ld + k);
                v[j] = x;
            }
        } else if constexpr (VEC && !KFAST) {
            const int r = row0 + 4 * (tid % RQ);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int k = k0 + tid / RQ + KPASS * j;
                f32x4 x = {0.f, 0.f, 0.f, 0.f};
                if (r < nrows && k < K) x = *reinterpret_cast<const f32x4*>(base + (long)k * ld + r);
                v[j] = x;
            }
        } else if constexpr (KFAST) {
            const int k = k0 + tid % BK;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = row0 + tid / BK + RPASS_S * (4 * j + e);
                    v[j][e] = (r < nrows && k < K) ? base[(long)r * ld + k] : 0.f;
                }
            }
        } else {
            const int r = row0 + tid % ROWS;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = k0 + tid / ROWS + KPASS_S * (4 * j + e);
                    v[j][e] = (r < nrows && k < K) ? base[(long)k * ld + r] : 0.f;
                }
            }
        }
    }

    // interior tiles of single-segment problems: per-thread source pointers are computed once and advanced by
    // a constant per K step -- no bounds checks, no exec-masked branches, no 64-bit multiplies in the loop
    __device__ __forceinline__ void init_ptrs(const float* (&p)[NV], const float* __restrict__ base, long ld,
                                              int row0, int tid) const {
        if constexpr (VEC) {   // the fast path exists for float4 staging only; never called otherwise
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                if constexpr (KFAST) p[j] = base + (long)(row0 + tid / KQ + RPASS * j) * ld + 4 * (tid % KQ);
                else p[j] = base + (long)(tid / RQ + KPASS * j) * ld + row0 + 4 * (tid % RQ);
            }
        }
    }
    __device__ __forceinline__ void load_fast(const float* (&p)[NV], long step) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            v[j] = *reinterpret_cast<const f32x4*>(p[j]);
            p[j] += step;
        }
    }

    __device__ __forceinline__ void store(float* __restrict__ lds, int tid) const {
        if constexpr (VEC && KFAST) {
#pragma unroll
            for (int j = 0; j < NV; ++j)
                *reinterpret_cast<f32x4*>(lds + (tid / KQ + RPASS * j) * LDK + 4 * (tid % KQ)) = v[j];
        } else if constexpr (VEC && !KFAST) {
#pragma unroll
            for (int j = 0; j < NV; ++j)
                *reinterpret_cast<f32x4*>(lds + (tid / RQ + KPASS * j) * LDR + 4 * (tid % RQ)) = v[j];
        } else if constexpr (KFAST) {
#pragma unroll
            for (int j = 0; j < NV; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) lds[(tid / BK + RPASS_S * (4 * j + e)) * LDK + tid % BK] = v[j][e];
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    lds[(tid / ROWS + KPASS_S * (4 * j + e)) * LDR + tid % ROWS] = v[j][e];
        }
    }

    // partial column sums of the tile just loaded (only meaningful for [k][row] operands): with float4
    // staging a thread owns rows 4*(tid % RQ)..+3, with scalar staging the single row tid % ROWS
    __device__ __forceinline__ void add_rowsum(float (&ps)[4]) const {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if constexpr (VEC) {
#pragma unroll
                for (int e = 0; e < 4; ++e) ps[e] += v[j][e];
            } else {
                ps[0] += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
            }
        }
    }

    // the 4 k-values (k = 8q + 4h + c, c = 0..3) of tile row `row` for this lane's k-half h
    __device__ __forceinline__ static f32x4 frag(const float* __restrict__ lds, int row, int q, int h) {
        if constexpr (KFAST) {
            return *reinterpret_cast<const f32x4*>(lds + row * LDK + 8 * q + 4 * h);
        } else {
            f32x4 x;
            const float* p = lds + (8 * q + 4 * h) * LDR + row;
            x[0] = p[0];
            x[1] = p[LDR];
            x[2] = p[2 * LDR];
            x[3] = p[3 * LDR];
            return x;
        }
    }
};

// STAGES = 2: double-buffered LDS, one barrier per K step (2 blocks/CU at 128x128).
// STAGES = 1: single buffer, two barriers per K step, half the LDS -> 3 blocks/CU cover each other's stalls.
// FAST: every tile is interior (M % BM == N % BN == 0 and every segment's K % BK == 0).
template <int BM, int BN, bool AK, bool BKF, bool VEC, int STAGES, int BK, bool FAST, int THREADS = GEMM_THREADS>
__device__ __forceinline__ void gemm_tile(const GemmArgs& args, const int grp, const int tn, const int ks,
                                          const int row0, const int col0) {
    constexpr int WGM = (THREADS == 256) ? 2 : 1;  // waves along M / N: 2x2 (256 threads) or one wave per block
    constexpr int WGN = WGM;
    constexpr int MT = BM / (32 * WGM);  // 32x32 MFMA tiles per wave along M
    constexpr int NT = BN / (32 * WGN);
    using StA = Stage<BM, AK, VEC, BK, THREADS>;
    using StB = Stage<BN, BKF, VEC, BK, THREADS>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // LDS: [A buf0 | A buf1 | B buf0 | B buf1]; pointers are computed, not tabulated
    float* const sA0 = smem;
    float* const sB0 = smem + STAGES * StA::LDS_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, h = lane >> 5;

    const int splitk = args.splitk;
    const rfn_gemm_problem& P = args.g[grp];
    const int M = args.M, N = args.N;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    StA stA;
    StB stB;

    // flattened (segment, k0) iteration space so the prefetch runs across segment boundaries
    int total_iters = 0;
    for (int s = 0; s < P.nseg; ++s) total_iters += (P.seg[s].K + BK - 1) / BK;

    // bias gradient riding on the weight-gradient GEMM: column sums of the [k][row] A operand, taken from
    // the staging registers by the blocks of the first column tile
    const bool do_colsum = !AK && (tn == 0) && (P.a_colsum != nullptr);
    float ps[4] = {0.f, 0.f, 0.f, 0.f};

    // split-K: this block owns iterations [it_begin, it_end) of the flattened (segment, k) space
    int seg = 0, k0 = 0;
    if (splitk > 1) {
        const int per = (total_iters + splitk - 1) / splitk;
        int it_begin = ks * per;
        const int it_end = min(total_iters, it_begin + per);
        total_iters = max(0, it_end - it_begin);
        while (seg < P.nseg) {  // locate (seg, k0) of it_begin
            const int n = (P.seg[seg].K + BK - 1) / BK;
            if (it_begin < n) break;
            it_begin -= n;
            ++seg;
        }
        k0 = it_begin * BK;
    }
    // current segment's operands live in registers; the kernarg table is re-read only when the segment changes
    const float* segA = nullptr;
    const float* segB = nullptr;
    long seg_lda = 0, seg_ldb = 0;
    int segK = 0;
    auto fetch_seg = [&]() {
        if (seg < P.nseg) {
            segA = P.seg[seg].A;
            segB = P.seg[seg].B;
            seg_lda = P.seg[seg].lda;
            seg_ldb = P.seg[seg].ldb;
            segK = P.seg[seg].K;
        }
    };
    fetch_seg();
    const float* pa[StA::NV];
    const float* pb[StB::NV];
    long stepA = 0, stepB = 0;
    auto setup_fast = [&]() {  // per-thread source pointers of the tile at (seg, k0)
        stA.init_ptrs(pa, segA + (AK ? (long)k0 : (long)k0 * seg_lda), seg_lda, row0, tid);
        stB.init_ptrs(pb, segB + (BKF ? (long)k0 : (long)k0 * seg_ldb), seg_ldb, col0, tid);
        stepA = AK ? BK : (long)BK * seg_lda;
        stepB = BKF ? BK : (long)BK * seg_ldb;
    };
    if constexpr (FAST) {
        if (seg < P.nseg) setup_fast();
    }
    auto issue_load = [&]() {
        if constexpr (FAST) {
            stA.load_fast(pa, stepA);
            stB.load_fast(pb, stepB);
        } else {
            stA.load(segA, seg_lda, row0, M, k0, segK, tid);
            stB.load(segB, seg_ldb, col0, N, k0, segK, tid);
        }
        if constexpr (!AK) {
            if (do_colsum) stA.add_rowsum(ps);
        }
        k0 += BK;
        if (k0 >= segK) {  // next K segment (rare): refresh the register-resident descriptor
            k0 = 0;
            ++seg;
            fetch_seg();
            if constexpr (FAST) {
                if (seg < P.nseg) setup_fast();
            }
        }
    };

    if (total_iters > 0) {
        issue_load();
        stA.store(sA0, tid);
        stB.store(sB0, tid);
    }
    __syncthreads();

#if GEMM_SETPRIO == 2
    {   // co-resident blocks get different static priorities so that their MFMA clusters do not phase-lock
        const int pr = __builtin_amdgcn_readfirstlane((int)(blockIdx.x % 3));
        if (pr == 1) __builtin_amdgcn_s_setprio(1);
        else if (pr == 2) __builtin_amdgcn_s_setprio(2);
    }
#endif
    for (int it = 0; it < total_iters; ++it) {
        const int cur = (STAGES == 2) ? (it & 1) : 0;
        const bool more = (it + 1 < total_iters);
#if GEMM_ABLATE == 1
        if (more && it == 0) issue_load();
#else
        if (more) issue_load();  // global loads in flight during the MFMAs below
#endif

        const float* a_l = sA0 + cur * StA::LDS_FLOATS;
        const float* b_l = sB0 + cur * StB::LDS_FLOATS;
#if GEMM_FRAG_PIPE
        // fragments of k-group q+1 are requested before the MFMAs of group q are issued, so their LDS
        // latency hides under a full 16-MFMA group instead of the last two MFMAs
        f32x4 af[2][MT], bf[2][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[0][i] = StA::frag(a_l, wm * (BM / WGM) + i * 32 + l31, 0, h);
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[0][j] = StB::frag(b_l, wn * (BN / WGN) + j * 32 + l31, 0, h);
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            const int cb = q & 1, nb = cb ^ 1;
            if (q + 1 < BK / 8) {
#pragma unroll
                for (int i = 0; i < MT; ++i) af[nb][i] = StA::frag(a_l, wm * (BM / WGM) + i * 32 + l31, q + 1, h);
#pragma unroll
                for (int j = 0; j < NT; ++j) bf[nb][j] = StB::frag(b_l, wn * (BN / WGN) + j * 32 + l31, q + 1, h);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cb][i][c], bf[cb][j][c], acc[i][j], 0, 0, 0);
        }
#else
#if GEMM_SETPRIO == 1
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            f32x4 af[MT], bf[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = StA::frag(a_l, wm * (BM / WGM) + i * 32 + l31, q, h);
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j] = StB::frag(b_l, wn * (BN / WGN) + j * 32 + l31, q, h);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][c], bf[j][c], acc[i][j], 0, 0, 0);
        }
#if GEMM_SETPRIO == 1
        __builtin_amdgcn_s_setprio(0);
#endif
#endif

        if constexpr (STAGES == 2) {
            if (more) {
                stA.store(sA0 + (cur ^ 1) * StA::LDS_FLOATS, tid);
                stB.store(sB0 + (cur ^ 1) * StB::LDS_FLOATS, tid);
            }
            __syncthreads();
        } else {
            __syncthreads();  // every wave has finished reading the tile
            if (more) {
                stA.store(sA0, tid);
                stB.store(sB0, tid);
            }
            __syncthreads();
        }
    }

    if constexpr (!AK) {
        if (do_colsum) {  // block-uniform; the K loop's last barrier has retired every LDS read
            constexpr int PARTS = VEC ? THREADS / (BM / 4) : THREADS / BM;
            if constexpr (VEC) {
#pragma unroll
                for (int e = 0; e < 4; ++e) smem[(tid / (BM / 4)) * BM + 4 * (tid % (BM / 4)) + e] = ps[e];
            } else {
                smem[(tid / BM) * BM + tid % BM] = ps[0];
            }
            __syncthreads();
            if (tid < BM && row0 + tid < M) {
                float t = 0.f;
#pragma unroll
                for (int p2 = 0; p2 < PARTS; ++p2) t += smem[p2 * BM + tid];
                if (splitk > 1) {  // partial column sum of this K range: [ngroups][splitk][M] behind the partial tiles
                    args.part[(long)args.ngroups * splitk * M * N + ((long)grp * splitk + ks) * M + row0 + tid] = t;
                } else {
                    float* o = P.a_colsum + row0 + tid;
                    *o = args.accumulate ? *o + t : t;
                }
            }
        }
    }

    if (splitk > 1) {  // raw partial tile; rfn_gemm_reduce_k adds the bias / previous C in a fixed order
        float* part = args.part + ((long)grp * splitk + ks) * (long)M * N;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = col0 + wn * (BN / WGN) + j * 32 + l31;
            if (col >= N) continue;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + wm * (BM / WGM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row < M) part[(long)row * N + col] = acc[i][j][r];
                }
        }
        if (args.tickets)
            gemm_finish_splitk<BM, BN, VEC, THREADS>(args, P, grp, (grp * args.tiles_m + row0 / BM) * args.tiles_n + tn, row0, col0,
                                                     !AK && tn == 0, smem);
        return;
    }

    // ---- epilogue: bias, optional accumulate, bounds-checked store --------------------------
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = col0 + wn * (BN / WGN) + j * 32 + l31;
        if (col >= N) continue;
        float bsum = 0.f;
        for (int s = 0; s < P.nseg; ++s)
            if (P.seg[s].bias) bsum += P.seg[s].bias[col];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + wm * (BM / WGM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
#if GEMM_ABLATE == 2
                if (row < M && acc[i][j][r] == 123.456f) {
#else
                if (row < M) {
#endif
                    float* c = P.C + (long)row * P.ldc + col;
                    float val = acc[i][j][r] + bsum;
                    if (args.accumulate) val += *c;
                    *c = val;
                }
            }
        }
    }
}

// ---- LDS-DMA staging ----------------------------------------------------------------------------------------------
// global_load_lds_dwordx4 writes 64 lanes x 16 B = 1 KiB of LDS per wave-instruction at a wave-uniform base (lane-
// linear, no per-lane scatter), so the tile images are unpadded and the bank-conflict fix goes on the SOURCE address:
//   [row][k] operands: rows of BK floats; 16-B chunk c of row r is stored at chunk position c ^ swz(r), where swz
//     spreads the 16 rows of each ds_read_b128 lane group over all 64 banks (BK = 32: 128-B rows, swz = (r>>1)&7;
//     BK = 16: 64-B rows, swz = (r>>2)&3).  Each lane fetches the chunk its LDS position must hold, every 128-B
//     (64-B) row segment is still read whole by 8 (4) neighbouring lanes.
//   [k][row] operands: k-rows of ROWS floats, linear; ds_read_b32 of 32 consecutive rows is conflict-free as it is.
// No staging registers, no ds_write pass, and the data of the next slot(s) stays in flight across the barrier.
typedef __attribute__((address_space(3))) void rfn_lds_void;
typedef const __attribute__((address_space(1))) void rfn_gbl_void;

template <int ROWS, bool KFAST, int BK>
struct Dma {
    static constexpr int FLOATS = ROWS * BK;
    static constexpr int NI = ROWS * BK / 256 / 4;   // 1-KiB wave-instructions per wave (4 waves per block)
    static constexpr int CPR = BK / 4;               // 16-B chunks per [row][k] row
    static constexpr int RPI = 64 / CPR;             // rows per wave-instruction   ([row][k])
    static constexpr int RQ = ROWS / 4;              // 16-B chunks per [k][row] k-row
    static constexpr int KPI = 64 / RQ;              // k-rows per wave-instruction ([k][row])
    static_assert(NI >= 1 && ROWS * BK % 1024 == 0 && (KFAST || RQ <= 64), "tile must be whole 1-KiB pieces per wave");

    __device__ __forceinline__ static int swz(int row) { return BK == 32 ? ((row >> 1) & 7) : ((row >> 2) & 3); }

    // Source addressing: a wave-uniform base pointer (the tile's first k, advanced by a SCALAR add per K step) plus
    // this lane's 32-bit byte offset, fixed for the whole K loop (one per wave-instruction).  The DMA instruction then
    // takes its address as SGPR pair + one VGPR: half the address-register traffic of a 64-bit per-lane pointer and
    // no per-lane pointer arithmetic in the loop (the host checks that the operand spans < 4 GiB).
    // nrows: valid rows of a [row][k] operand.  Rows of an edge tile that lie past it fetch the last valid row instead
    // (their products are computed and never stored), so ragged row / column counts need no masked loads.
    __device__ __forceinline__ static void init_offs(uint32_t (&o)[NI], long ld, int row0, int wave, int lane, int nrows) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = wave * NI + j;
            if constexpr (KFAST) {
                const int row = n * RPI + lane / CPR;
                int gr = row0 + row;
                gr = gr < nrows ? gr : nrows - 1;
                o[j] = (uint32_t)(((long)gr * ld + 4 * ((lane % CPR) ^ swz(row))) * 4);
            } else {
                o[j] = (uint32_t)(((long)(n * KPI + lane / RQ) * ld + row0 + 4 * (lane % RQ)) * 4);
            }
        }
    }
    __device__ __forceinline__ static void issue(const char* base, const uint32_t (&o)[NI], float* stage, int wave) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
            __builtin_amdgcn_global_load_lds((rfn_gbl_void*)(base + o[j]), (rfn_lds_void*)(stage + (wave * NI + j) * 256), 16,
                                             0, 0);
    }
    // the 4 k-values (k = 8q + 4h + c) of tile row `row`; sw = swz(row)
    __device__ __forceinline__ static f32x4 frag(const float* __restrict__ lds, int row, int q, int h, int sw) {
        if constexpr (KFAST) {
            return *reinterpret_cast<const f32x4*>(lds + row * BK + 4 * ((2 * q + h) ^ sw));
        } else {
            f32x4 x;
            const float* p = lds + (8 * q + 4 * h) * ROWS + row;
            x[0] = p[0];
            x[1] = p[ROWS];
            x[2] = p[2 * ROWS];
            x[3] = p[3 * ROWS];
            return x;
        }
    }
};

template <int N>
__device__ __forceinline__ void rfn_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One BM x BN output tile of an interior single- or multi-segment problem (M % BM == N % BN == K_s % BK == 0, float4
// aligned), operands staged by LDS-DMA into a ring of SLOTS slots:
//   iteration it:  wait until this wave's pieces of tile `it` have landed (the younger SLOTS-2 tiles stay in flight)
//                  -> barrier (every wave's pieces landed; every wave has finished reading slot (it-1) % SLOTS)
//                  -> issue tile it+SLOTS-1 into that freed slot -> MFMAs on slot it % SLOTS.
// One barrier per K step; the k order of every output element is the same as in the register-staged kernel.
template <int BM, int BN, bool AK, bool BKF, int BK, int SLOTS>
__device__ __forceinline__ void gemm_tile_dma(const GemmArgs& args, const int grp, const int ks, const int row0,
                                              const int col0) {
    constexpr int MT = BM / 64, NT = BN / 64;   // 2x2 waves, 32x32 MFMA tiles per wave
    using DA = Dma<BM, AK, BK>;
    using DB = Dma<BN, BKF, BK>;
    constexpr int SLOT_FLOATS = DA::FLOATS + DB::FLOATS;
    constexpr int NIW = DA::NI + DB::NI;        // DMA instructions per wave per tile
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int swa = DA::swz(l31), swb = DB::swz(l31);   // tile rows are l31 + multiples of 32: same swizzle term

    const int splitk = args.splitk;
    const rfn_gemm_problem& P = args.g[grp];

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int total_iters = 0;
    for (int s = 0; s < P.nseg; ++s) total_iters += P.seg[s].K / BK;
    int seg = 0, k0 = 0;
    if (splitk > 1) {
        const int per = (total_iters + splitk - 1) / splitk;
        int it_begin = ks * per;
        const int it_end = min(total_iters, it_begin + per);
        total_iters = max(0, it_end - it_begin);
        while (seg < P.nseg) {
            const int n = P.seg[seg].K / BK;
            if (it_begin < n) break;
            it_begin -= n;
            ++seg;
        }
        k0 = it_begin * BK;
    }
    uint32_t oa[DA::NI], ob[DB::NI];     // per-lane byte offsets inside the current segment's operands
    const char* baseA = nullptr;         // wave-uniform: first byte of the k0 column (row) of the segment's operands
    const char* baseB = nullptr;
    long stepA = 0, stepB = 0;           // bytes per K step
    int segK = 0;
    auto setup = [&]() {
        if (seg < P.nseg) {
            const rfn_gemm_seg& sg = P.seg[seg];
            segK = sg.K;
            DA::init_offs(oa, sg.lda, row0, wave, lane, args.M);
            DB::init_offs(ob, sg.ldb, col0, wave, lane, args.N);
            stepA = (AK ? (long)BK : (long)BK * sg.lda) * 4;
            stepB = (BKF ? (long)BK : (long)BK * sg.ldb) * 4;
            baseA = (const char*)sg.A + (long)(k0 / BK) * stepA;
            baseB = (const char*)sg.B + (long)(k0 / BK) * stepB;
#if GEMM_DMA_ABLATE & 8
            stepA = stepB = 0;   // every K step re-reads the first tile: same instruction stream, cache-resident data
#endif
        }
    };
    setup();
    auto issue = [&](int slot) {
        float* st = smem + slot * SLOT_FLOATS;
        DA::issue(baseA, oa, st, wave);
        DB::issue(baseB, ob, st + DA::FLOATS, wave);
        baseA += stepA;
        baseB += stepB;
        k0 += BK;
        if (k0 >= segK) {   // next K segment (rare)
            k0 = 0;
            ++seg;
            setup();
        }
    };

    int issued = 0;
#pragma unroll
    for (int s = 0; s < SLOTS - 1; ++s)
        if (s < total_iters) {
            issue(s);
            ++issued;
        }
    int cur = 0, fill = SLOTS - 1;   // slot of tile `it`, slot tile it+SLOTS-1 goes to
    for (int it = 0; it < total_iters; ++it) {
        const int younger = issued - it - 1;   // tiles issued after tile `it`
        if (SLOTS >= 4 && younger >= 2) rfn_wait_vmcnt<(SLOTS >= 4 ? 2 : 0) * NIW>();
        else if (SLOTS >= 3 && younger >= 1) rfn_wait_vmcnt<(SLOTS >= 3 ? 1 : 0) * NIW>();
        else rfn_wait_vmcnt<0>();
#if !(GEMM_DMA_ABLATE & 2)
        __builtin_amdgcn_s_barrier();
#endif
        if (issued < total_iters) {
#if GEMM_DMA_ABLATE & 1
            k0 += BK;
#else
            issue(fill);
#endif
            ++issued;
        }
        const float* a_l = smem + cur * SLOT_FLOATS;
        const float* b_l = a_l + DA::FLOATS;
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            f32x4 af[MT], bf[NT];
#if GEMM_DMA_ABLATE & 4
#pragma unroll
            for (int i = 0; i < MT; ++i) { af[i] = f32x4{1.f + q, 2.f, 3.f, 4.f + i}; asm volatile("" : "+v"(af[i])); }
#pragma unroll
            for (int j = 0; j < NT; ++j) { bf[j] = f32x4{1.f, 2.f + q, 3.f + j, 4.f}; asm volatile("" : "+v"(bf[j])); }
#else
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = DA::frag(a_l, wm * (BM / 2) + i * 32 + l31, q, h, swa);
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j] = DB::frag(b_l, wn * (BN / 2) + j * 32 + l31, q, h, swb);
#endif
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][c], bf[j][c], acc[i][j], 0, 0, 0);
        }
        cur = (cur + 1 == SLOTS) ? 0 : cur + 1;
        fill = (fill + 1 == SLOTS) ? 0 : fill + 1;
    }

    // ---- epilogue.  Addresses are a wave-uniform tile base (SGPR pair) + a 32-bit lane offset: no 64-bit per-lane
    // address arithmetic, and the stores take the cheap saddr form (the host checks that a tile spans < 4 GiB).
    const int M = args.M, N = args.N;
    const long ldo = (splitk > 1) ? (long)N : P.ldc;
    float* const obase = (splitk > 1) ? args.part + ((long)grp * splitk + ks) * (long)M * N + (long)row0 * N + col0
                                      : P.C + (long)row0 * P.ldc + col0;
    char* const tile = (char*)obase;
    const uint32_t ld4 = (uint32_t)ldo * 4u;
    const uint32_t lane_off = (uint32_t)(wm * (BM / 2) + 4 * h) * ld4 + (uint32_t)(wn * (BN / 2) + l31) * 4u;
    const bool raw = splitk > 1;   // raw partial tile; rfn_gemm_reduce_k adds the bias / previous C in a fixed order
    const bool accumulate = !raw && args.accumulate;
    if (row0 + BM > M || col0 + BN > N) {
        // edge tile of a ragged problem (both operands [row][k]: the out-of-range rows were clamped at the source): the
        // same values, bounds-checked stores
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = col0 + wn * (BN / 2) + j * 32 + l31;
            if (col >= N) continue;
            float bsum = 0.f;
            if (!raw)
                for (int s = 0; s < P.nseg; ++s)
                    if (P.seg[s].bias) bsum += P.seg[s].bias[col];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row >= M) continue;
                    float* c = raw ? args.part + ((long)grp * splitk + ks) * (long)M * N + (long)row * N + col
                                   : P.C + (long)row * P.ldc + col;
                    const float v = acc[i][j][r] + bsum;
                    *c = accumulate ? v + *c : v;
                }
        }
    } else {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        float bsum = 0.f;
        if (!raw) {
            const int col = col0 + wn * (BN / 2) + j * 32 + l31;
            for (int s = 0; s < P.nseg; ++s)
                if (P.seg[s].bias) bsum += P.seg[s].bias[col];
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const uint32_t sub = lane_off + (uint32_t)(i * 32) * ld4 + (uint32_t)(j * 32) * 4u;
            if (accumulate) {
                float prev[16];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    prev[r] = *reinterpret_cast<const float*>(tile + (sub + (uint32_t)((r & 3) + 8 * (r >> 2)) * ld4));
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    *reinterpret_cast<float*>(tile + (sub + (uint32_t)((r & 3) + 8 * (r >> 2)) * ld4)) =
                        (acc[i][j][r] + bsum) + prev[r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    *reinterpret_cast<float*>(tile + (sub + (uint32_t)((r & 3) + 8 * (r >> 2)) * ld4)) = acc[i][j][r] + bsum;
            }
        }
    }
    }
    if (raw && args.tickets)
        gemm_finish_splitk<BM, BN, true, GEMM_THREADS>(args, P, grp, (grp * args.tiles_m + row0 / BM) * args.tiles_n + col0 / BN,
                                                       row0, col0, false, smem);
}

// ---- block -> (group, tile_m, tile_n): bijective XCD remap, then 8-row bands --------------------------------------
// TAIL (big NT launches whose tile count is not a multiple of the resident slots): the blocks of the last, partly
// filled round each take HALF a tile (BM/2 rows), so that round lasts half as long on twice as many CUs -- a QUARTER
// (BM/2 x BN/2) when four blocks per tile still fit the round (args.tail_parts; B = 32 shard: 0.81 against 0.84 ms per
// projection launch).  A tile's shape does not enter the k order of an output element: same bits.  Consecutive
// block ids land on consecutive XCDs, so the tail round is "the last tile indices of every XCD", two blocks per tile.
template <int BM, int BN, bool AK, bool BKF, bool VEC, int STAGES, int BK, bool FAST, int THREADS = GEMM_THREADS,
          bool TAIL = false, int DMA = 0>
__global__ __launch_bounds__(THREADS, (THREADS == 64) ? 1 : (DMA ? GEMM_DMA_MIN_WAVES : (TAIL ? 3 : GEMM_MIN_WAVES))) void rfn_gemm_kernel(
    const GemmArgs args) {   // TAIL: 3 waves/SIMD asked for explicitly (the two-body kernel schedules better under it)
    // DMA > 0: LDS-DMA staging into a ring of DMA slots (gemm_tile_dma); STAGES is then unused
    const int NC = args.ngroups * args.tiles_n;
    const int splitk = args.splitk;
    const int nblk = NC * args.tiles_m * splitk;     // whole tiles (x K ranges)
    int lid, half = -1;
    {
        const int bid = blockIdx.x;
#if GEMM_XCD_REMAP
        const int q = nblk >> 3, r = nblk & 7;
        if (TAIL && bid >= args.tail_main_blocks) {   // r == 0, splitk == 1 (host-checked)
            const int j = bid - args.tail_main_blocks, jdx = j >> 3;
            if (args.tail_parts == 4) {
                lid = (j & 7) * q + args.tail_idx_main + (jdx >> 2);
                half = 2 + (jdx & 3);     // 2 ... 5: quarter (row half, column half) = ((half - 2) >> 1, (half - 2) & 1)
            } else {
                lid = (j & 7) * q + args.tail_idx_main + (jdx >> 1);
                half = jdx & 1;
            }
        } else {
            const int xcd = bid & 7, idx = bid >> 3;
            lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        }
#else
        lid = bid;
#endif
    }
    const int ks = lid % splitk;  // the K ranges of one tile run next to each other
    lid /= splitk;
    const int per_band = GEMM_BAND_ROWS * NC;
    const int band = lid / per_band;
    const int rem = lid - band * per_band;
    const int band_rows = min(GEMM_BAND_ROWS, args.tiles_m - band * GEMM_BAND_ROWS);
    const int vcol = rem / band_rows;
    const int tm = band * GEMM_BAND_ROWS + (rem - vcol * band_rows);
    const int grp = vcol / args.tiles_n;
    const int tn = vcol - grp * args.tiles_n;
    if constexpr (TAIL) {
        if (half >= 2) {   // quarter tiles: the same fma chain per output element (a tile's shape does not enter the k order)
            const int qr = (half - 2) >> 1, qc = (half - 2) & 1;
            if constexpr (DMA > 0)
                gemm_tile_dma<BM / 2, BN / 2, AK, BKF, BK, DMA>(args, grp, ks, tm * BM + qr * (BM / 2), tn * BN + qc * (BN / 2));
            else
                gemm_tile<BM / 2, BN / 2, AK, BKF, VEC, STAGES, BK, FAST, THREADS>(args, grp, tn, ks, tm * BM + qr * (BM / 2),
                                                                                  tn * BN + qc * (BN / 2));
            return;
        }
        if (half >= 0) {
            if constexpr (DMA > 0)
                gemm_tile_dma<BM / 2, BN, AK, BKF, BK, DMA>(args, grp, ks, tm * BM + half * (BM / 2), tn * BN);
            else
                gemm_tile<BM / 2, BN, AK, BKF, VEC, STAGES, BK, FAST, THREADS>(args, grp, tn, ks,
                                                                              tm * BM + half * (BM / 2), tn * BN);
            return;
        }
    }
    if constexpr (DMA > 0)
        gemm_tile_dma<BM, BN, AK, BKF, BK, DMA>(args, grp, ks, tm * BM, tn * BN);
    else
        gemm_tile<BM, BN, AK, BKF, VEC, STAGES, BK, FAST, THREADS>(args, grp, tn, ks, tm * BM, tn * BN);
}

// C = sum_ks part[g][ks] + sum_s bias_s (+ C): fixed summation order, one thread per output element.  The blocks
// past the C range finish the a_colsum rider the same way (partial column sums of every K range, in order).
// K ranges are read four at a time (the loads of a group are independent and in flight together; the adds keep the
// k order) and, when rows are whole float4s, four columns per thread: the launch is latency-bound, not bandwidth-bound.
template <bool VEC>
__global__ __launch_bounds__(256) void rfn_gemm_reduce_k(const GemmArgs args) {
    constexpr int W = VEC ? 4 : 1;
    typedef float vec_t __attribute__((ext_vector_type(W)));
    const long MN = (long)args.M * args.N;
    const int grp = blockIdx.y;
    const rfn_gemm_problem& P = args.g[grp];
    const int splitk = args.splitk;
    const int c_blocks = (int)((MN / W + 255) / 256);
    if ((int)blockIdx.x >= c_blocks) {
        const int row = ((int)blockIdx.x - c_blocks) * 256 + threadIdx.x;
        if (row >= args.M || !P.a_colsum) return;
        const float* cs = args.part + (long)args.ngroups * splitk * MN + (long)grp * splitk * args.M + row;
        float s = 0.f;
        for (int k = 0; k < splitk; ++k) s += cs[(long)k * args.M];
        float* o = P.a_colsum + row;
        *o = args.accumulate ? *o + s : s;
        return;
    }
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * W;
    if (i >= MN) return;
    const int row = (int)(i / args.N), col = (int)(i - (long)row * args.N);
    const float* part = args.part + (long)grp * splitk * MN + i;
    float* c = P.C + (long)row * P.ldc + col;
    vec_t prev;
    if (args.accumulate) prev = *reinterpret_cast<const vec_t*>(c);
    vec_t s;
#pragma unroll
    for (int e = 0; e < W; ++e) s[e] = 0.f;
    for (int k0 = 0; k0 < splitk; k0 += 4) {
        vec_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u < splitk ? k0 + u : splitk - 1;
            v[u] = *reinterpret_cast<const vec_t*>(part + k * MN);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k0 + u < splitk) s += v[u];
    }
    for (int sg = 0; sg < P.nseg; ++sg)
        if (P.seg[sg].bias) s += *reinterpret_cast<const vec_t*>(P.seg[sg].bias + col);
    *reinterpret_cast<vec_t*>(c) = args.accumulate ? prev + s : s;
}

// The same fixed-order reduce for a gate GEMM (N = 4R, gate chunks [in | forget | out | g]) followed by the LSTM update of
// rfn_cell.hip lstm_fwd_k in the same thread: one thread per (row, unit) adds the K-range partials of its four gates in
// order, the biases, applies the gate math and writes the activations back into C, c_next and the (dropout-masked)
// h_next.  Group g = cell g of a stage-I step: its state pointers advance by the gs_* strides, its dropout stream is
// offset + g.  (misc/RecurrentFusionModel.py:53-73)
__global__ __launch_bounds__(256) void rfn_gemm_reduce_lstm_k(const GemmArgs args) {
    const int R = args.N / 4, M = args.M;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)M * R) return;
    const int grp = blockIdx.y;
    const rfn_gemm_problem& P = args.g[grp];
    const rfn_gemm_lstm& L = args.lstm;
    const int row = (int)(idx / R), j = (int)(idx - (long)row * R);
    const long MN = (long)M * args.N;
    const float* part = args.part + (long)grp * args.splitk * MN + (long)row * args.N + j;
    float pre[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float s = 0.f;
        for (int k0 = 0; k0 < args.splitk; k0 += 4) {   // four K ranges' loads in flight, added in k order
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u < args.splitk ? k0 + u : args.splitk - 1;
                v[u] = part[k * MN + g * R];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (k0 + u < args.splitk) s += v[u];
        }
        for (int sg = 0; sg < P.nseg; ++sg)
            if (P.seg[sg].bias) s += P.seg[sg].bias[g * R + j];
        pre[g] = s;
    }
    float* G = P.C + (long)row * P.ldc;
    const float ig = rfn_sigmoid(pre[0]), fg = rfn_sigmoid(pre[1]), og = rfn_sigmoid(pre[2]);
    const float gg = tanhf(pre[3]);
    G[j] = ig;
    G[R + j] = fg;
    G[2 * R + j] = og;
    G[3 * R + j] = gg;
    const float c = fg * L.c_prev[grp * L.gs_cprev + (long)row * L.ldcp + j] + ig * gg;
    L.c_next[grp * L.gs_cnext + (long)row * L.ldcn + j] = c;
    float hv = og * tanhf(c);
    if (L.drop_p > 0.f) {
        const float u = rfn_philox_uniform(L.seed, L.offset + (uint64_t)grp, (uint64_t)idx);
        hv = (u >= L.drop_p) ? hv * (1.0f / (1.0f - L.drop_p)) : 0.f;
    }
    L.h_next[grp * L.gs_h + (long)row * L.ldh + j] = hv;
}

// Per-device launch state of one kernel instantiation: the dynamic-LDS opt-in has been made and the number of blocks
// a CU hosts is known.  Indexed by device ordinal, so a host that drives several GPUs from one process gets each
// device's own answer; the entries are write-once (a race repeats the same calls and stores the same values).
struct KernelDevState {
    bool attr_set[16] = {};
    int blocks_per_cu[16] = {};
};
template <typename K>
static int prepare_kernel(K kernel, KernelDevState& st, size_t lds, int threads, int* blocks_per_cu) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return RFN_ERR_LAUNCH;
    const int slot = dev & 15;
    if (!st.attr_set[slot]) {
        if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return RFN_ERR_LAUNCH;
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, threads, lds) != hipSuccess || occ < 1) occ = 1;
        st.blocks_per_cu[slot] = occ;
        st.attr_set[slot] = true;
    }
    if (blocks_per_cu) *blocks_per_cu = st.blocks_per_cu[slot];
    return RFN_OK;
}
static int device_cus() {
    static int cus[16] = {};
    int dev = 0;
    hipGetDevice(&dev);
    int& c = cus[dev & 15];
    if (c == 0) {
        int v = 256;
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
        c = v;
    }
    return c;
}

template <int BM, int BN, bool AK, bool BKF, bool VEC, int STAGES, int BK, bool FAST = false, int THREADS = GEMM_THREADS,
          int DMA = 0>
static int launch_cfg(const GemmArgs& a_in, hipStream_t st) {
    GemmArgs a = a_in;
    if (a.splitk <= 1 || (long)a.ngroups * a.tiles_m * a.tiles_n > a.n_tickets || a.lstm.c_next) a.tickets = nullptr;   // separate reduce launch
    using StA = Stage<BM, AK, VEC, BK, THREADS>;
    using StB = Stage<BN, BKF, VEC, BK, THREADS>;
    size_t lds = DMA > 0 ? (size_t)DMA * (BM + BN) * BK * sizeof(float)
                         : STAGES * (StA::LDS_FLOATS + StB::LDS_FLOATS) * sizeof(float);
    const int nblk = a.ngroups * a.tiles_m * a.tiles_n * a.splitk;
    bool launched = false;
#if GEMM_TAIL_HALF && GEMM_XCD_REMAP
    // Tile quantisation: with S resident blocks per XCD a launch of 8*q tiles runs ceil(q/S) rounds and the last
    // one is only (q mod S)/S full.  When that fraction is at most a half (and there are enough rounds for it to
    // be a tail at all), the last round's tiles are processed as two half-height tiles each.
    if constexpr (FAST && AK && BKF && VEC && THREADS == 256 && BM == 128 && (DMA > 0 || STAGES == 1)) {
        // The two-body kernel is also the better-scheduled one of the register-staged path (156 VGPRs = three waves
        // per SIMD, the single-body instantiation needs 170 = two), so every unsplit launch of this layout goes
        // through it; launches that do not qualify for a tail simply have no tail blocks.
        auto kt = rfn_gemm_kernel<BM, BN, AK, BKF, VEC, STAGES, BK, FAST, THREADS, true, DMA>;
        static KernelDevState ks_t;
        int occ = 1;
        RFN_TRY(prepare_kernel(kt, ks_t, lds, THREADS, &occ));
        const int slots_per_xcd = (device_cus() / 8) * occ;
        const int q = nblk / 8, tail = slots_per_xcd > 0 ? q % slots_per_xcd : 0;
        const bool has_tail = a.splitk == 1 && nblk % 8 == 0 && slots_per_xcd > 0 && q / slots_per_xcd >= GEMM_TAIL_MIN_ROUNDS &&
                              tail > 0 && 2 * tail <= slots_per_xcd;
        GemmArgs t = a;
        t.tail_idx_main = has_tail ? q - tail : 0;
        t.tail_main_blocks = has_tail ? 8 * t.tail_idx_main : 0x7fffffff;
        // a tail of at most a quarter of a round (the projections of the B <= 128 shards): quarter tiles, a round of a
        // quarter of the work on four times the CUs
        t.tail_parts = (GEMM_TAIL_QUARTER && has_tail && 4 * tail <= slots_per_xcd) ? 4 : 2;
        hipLaunchKernelGGL(kt, dim3(has_tail ? t.tail_main_blocks + 8 * t.tail_parts * tail : nblk), dim3(THREADS), lds, st, t);
        RFN_CHECK_LAUNCH();
        launched = true;
    }
#endif
    if (!launched) {
        auto k = rfn_gemm_kernel<BM, BN, AK, BKF, VEC, STAGES, BK, FAST, THREADS, false, DMA>;
        static KernelDevState ks;
        RFN_TRY(prepare_kernel(k, ks, lds, THREADS, nullptr));
        hipLaunchKernelGGL(k, dim3(nblk), dim3(THREADS), lds, st, a);
        RFN_CHECK_LAUNCH();
    }
    if (a.splitk > 1 && !a.tickets) {
        bool colsum = false;
        for (int g = 0; g < a.ngroups; ++g) colsum = colsum || (a.g[g].a_colsum != nullptr);
        const int cs_blocks = (colsum && !AK) ? rfn_cdiv(a.M, 256) : 0;
        if (a.lstm.c_next)
            hipLaunchKernelGGL(rfn_gemm_reduce_lstm_k, dim3(rfn_cdiv((long)a.M * (a.N / 4), 256), a.ngroups), dim3(256), 0, st, a);
        else {
            bool v4 = (a.N % 4 == 0);   // whole float4s per row: four columns per thread
            for (int g = 0; g < a.ngroups && v4; ++g) {
                v4 = v4 && rfn_aligned16(a.g[g].C) && (a.g[g].ldc % 4 == 0);
                for (int sg = 0; sg < a.g[g].nseg; ++sg) v4 = v4 && (!a.g[g].seg[sg].bias || rfn_aligned16(a.g[g].seg[sg].bias));
            }
            if (v4)
                hipLaunchKernelGGL(rfn_gemm_reduce_k<true>, dim3(rfn_cdiv((long)a.M * a.N / 4, 256) + cs_blocks, a.ngroups),
                                   dim3(256), 0, st, a);
            else
                hipLaunchKernelGGL(rfn_gemm_reduce_k<false>, dim3(rfn_cdiv((long)a.M * a.N, 256) + cs_blocks, a.ngroups),
                                   dim3(256), 0, st, a);
        }
        RFN_CHECK_LAUNCH();
    }
    return RFN_OK;
}

template <bool AK, bool BKF, bool VEC>
static int launch_tile(GemmArgs& a, hipStream_t st) {
    // big tile when it still fills the chip (>= 2 blocks per CU), else 64x64 for the skinny
    // per-step GEMMs (M = batch) so that more CUs get a tile.
    const long big = (long)rfn_cdiv(a.M, 128) * rfn_cdiv(a.N, 128) * a.ngroups;
    bool colsum = false;
    int iters32 = 0;
    double flops = 0;
    for (int g = 0; g < a.ngroups; ++g) colsum = colsum || (a.g[g].a_colsum != nullptr);
    for (int s = 0; s < a.g[0].nseg; ++s) {
        iters32 += rfn_cdiv(a.g[0].seg[s].K, GEMM_BIG_BK);
        flops += 2.0 * a.M * a.N * a.g[0].seg[s].K * a.ngroups;
    }
    // Medium problems (the heavier per-step GEMMs: M = batch, a few GF): the 128x128 tile is ~1.5x more efficient
    // than 64x64 but yields too few tiles, so cut K across blocks to reach ~2 blocks per CU.
    int big_split = 1;
    bool big_unsplit = false;
    const long medium_max = GEMM_MEDIUM_MAX;
    if (big <= medium_max && a.part && big >= 16 && flops >= GEMM_MEDIUM_MIN_FLOPS) {
        // K split from a cost model of the launch (tools/split_probe.py, profiles/r03_split_probe.md).  Blocks are handed
        // out in rounds of 512 (two per CU, 64 per XCD); a round costs its blocks' K steps at ~3.9 us per 32-deep step,
        // however full it is -- except a last round of <= 256 blocks (one per CU: a lone block has the matrix pipe to
        // itself, ~0.6 of the time).  A split whose blocks spill just past a whole round (136 tiles x 4 = 544) pays for
        // a nearly empty round: the old "smallest split reaching 512 blocks" rule did exactly that for the logit dX
        // product (517 -> 430 us at s = 3), and an unsplit 384-tile weight gradient ran 1.9 ms instead of 1.45.
        const long cap = (long)a.ws_mib * (1 << 18) / (((long)a.M * a.N + a.M) * a.ngroups);
        const double part_us = (double)a.M * a.N * a.ngroups * 4.0 / 4.0e6;   // one partial tile set read back by the reduce
        long want = 1;
        double best = 1e30;
        for (long s = 1; s <= 16 && s <= cap && (s == 1 || s <= iters32 / 8); ++s) {
            const long nb = big * s, full = nb / 512, rem = nb % 512;
            double t = 3.9 * ((double)full + (rem == 0 ? 0.0 : rem > 256 ? 1.0 : 0.6)) * (double)((iters32 + s - 1) / s);
            if (s > 1) t += 6.0 + s * part_us;
            if (t < best * 0.97) { best = t; want = s; }   // a deeper split has to buy 3 %
        }
        const long forced = (a.flags >> 8) & 31;   // RFN_GEMM_OPT_FORCE_SPLIT(n): tools/split_probe.py
        if (forced >= 1 && forced <= cap && forced <= iters32) want = forced;
        big_split = (want >= 2) ? (int)want : 1;
        // unsplit by choice (a split was allowed and priced higher): still the big tile; unsplit because no split is
        // allowed (short K, no workspace): the 64 x 64 path below with its own K cut, as before
        big_unsplit = want == 1 && ((cap >= 2 && iters32 / 8 >= 2) || forced == 1);
    }
    if (big >= 384 || big_split > 1 || big_unsplit) {
        a.splitk = big_split;
        a.tiles_m = rfn_cdiv(a.M, GEMM_BIG_BM);
        a.tiles_n = rfn_cdiv(a.N, GEMM_BIG_BN);
        // RFN_GEMM_OPT_LDS_LEAN (set by data-parallel hosts): big tiles that leave most of each CU's LDS free, so that
        // RCCL's kernels can co-reside with the long one-round weight-gradient GEMMs instead of waiting them out.
        const bool lean = (a.flags & RFN_GEMM_OPT_LDS_LEAN) != 0;
        constexpr int ST = (AK && BKF) ? GEMM_NT_STAGES : GEMM_XX_STAGES;
#if GEMM_FAST_PATH
        if constexpr (VEC) {
            bool kdiv = true;
            for (int g = 0; g < a.ngroups; ++g)
                for (int s = 0; s < a.g[g].nseg; ++s)
                    kdiv = kdiv && a.g[g].seg[s].K > 0 && (a.g[g].seg[s].K % GEMM_BIG_BK == 0);
            const bool fast = kdiv && (a.M % GEMM_BIG_BM == 0) && (a.N % GEMM_BIG_BN == 0);
            // the LDS-DMA kernel also takes ragged row / column counts when both operands are [row][k] (edge tiles clamp
            // their source rows and bounds-check their stores): the vocabulary (9488) needs no remainder launch
            const bool dma_ok = fast || (kdiv && AK && BKF);
#if GEMM_ONE_WAVE
            if (fast && (a.M % 64 == 0) && (a.N % 64 == 0)) {   // experiment: barrier-free single-wave 64x64 blocks
                a.tiles_m = a.M / 64;
                a.tiles_n = a.N / 64;
                return launch_cfg<64, 64, AK, BKF, true, GEMM_ONE_WAVE_STAGES, GEMM_BIG_BK, true, 64>(a, st);
            }
#endif
#if GEMM_DMA
            // interior tiles without a bias-gradient rider: LDS-DMA staging (no staging registers, no ds_write pass,
            // one barrier per K step, the next slot in flight across it)
            bool span32 = true;   // the LDS-DMA kernel addresses each operand as base + 32-bit byte offset
            for (int g = 0; g < a.ngroups; ++g)
                for (int s = 0; s < a.g[g].nseg; ++s) {
                    const rfn_gemm_seg& sg = a.g[g].seg[s];
                    const double ea = AK ? (double)a.M * sg.lda : (double)sg.K * sg.lda;
                    const double eb = BKF ? (double)a.N * sg.ldb : (double)sg.K * sg.ldb;
                    span32 = span32 && ea * 4 < 4.0e9 && eb * 4 < 4.0e9 && sg.lda >= 0 && sg.ldb >= 0;
                }
            for (int g = 0; g < a.ngroups; ++g)   // ... and one output tile as tile base + 32-bit byte offset
                span32 = span32 && a.g[g].ldc >= 0 && (double)a.g[g].ldc * 4 * GEMM_BIG_BM < 4.0e9 && (double)a.N * 4 * GEMM_BIG_BM < 4.0e9;
            if (dma_ok && span32 && !colsum && !(a.flags & RFN_GEMM_OPT_NO_DMA)) {
                constexpr int SL = (AK && BKF) ? GEMM_DMA_SLOTS_NT : GEMM_DMA_SLOTS_XX;
                constexpr int DBK = (AK && BKF) ? GEMM_DMA_BK : GEMM_DMA_BK_XX;
                // lean: 16-deep K steps, two slots = 32 KB of LDS per block (64 KB per CU at the two blocks per CU of a
                // one-round weight-gradient launch) instead of 64 KB per block; same speed (profiles/r02_pmc_gemm.md)
                if (lean) return launch_cfg<GEMM_BIG_BM, GEMM_BIG_BN, AK, BKF, true, 1, 16, true, GEMM_THREADS, 2>(a, st);
                return launch_cfg<GEMM_BIG_BM, GEMM_BIG_BN, AK, BKF, true, 1, DBK, true, GEMM_THREADS, SL>(a, st);
            }
#endif
            if (fast) {
                if (lean) return launch_cfg<GEMM_BIG_BM, GEMM_BIG_BN, AK, BKF, true, 1, GEMM_BIG_BK, true>(a, st);
                return launch_cfg<GEMM_BIG_BM, GEMM_BIG_BN, AK, BKF, true, ST, GEMM_BIG_BK, true>(a, st);
            }
        }
#endif
        if (lean) return launch_cfg<GEMM_BIG_BM, GEMM_BIG_BN, AK, BKF, VEC, 1, GEMM_BIG_BK>(a, st);
        return launch_cfg<GEMM_BIG_BM, GEMM_BIG_BN, AK, BKF, VEC, ST, GEMM_BIG_BK>(a, st);
    }
    a.tiles_m = rfn_cdiv(a.M, 64);
    a.tiles_n = rfn_cdiv(a.N, 64);
    // Skinny problems (M = batch) give too few tiles for 256 CUs and long serial MFMA chains: cut K across
    // blocks when a workspace is available.  Target ~3 blocks per CU, at least 4 K-iterations per block.
    if (a.part) {
        const long tiles = (long)a.tiles_m * a.tiles_n * a.ngroups;
        int iters = 0;
        for (int s = 0; s < a.g[0].nseg; ++s) iters += rfn_cdiv(a.g[0].seg[s].K, GEMM_SMALL_BK);
        long want = tiles > 0 ? GEMM_SPLIT_TARGET / tiles : 1;
        if (want > iters / GEMM_SPLIT_MIN_ITERS) want = iters / GEMM_SPLIT_MIN_ITERS;
        if (want > 16) want = 16;
        const long cap = (long)(a.ws_mib) * (1 << 18) / (((long)a.M * a.N + a.M) * a.ngroups);  // floats
        if (want > cap) want = cap;
        a.splitk = (want >= 2) ? (int)want : 1;
    }
#if GEMM_DMA && GEMM_SMALL_DMA_SLOTS
    // Interior skinny problems: the same LDS-DMA ring as the big tile, 64 x 64 x 32 per slot (16 KB).  A block of a split
    // launch runs 4-8 K steps, each of which used to wait out a full global-load latency (two register-staged buffers);
    // with GEMM_SMALL_DMA_SLOTS slots the whole K range of the block is in flight after the first wait.  Bit-identical
    // (same k order per output element).
    if constexpr (VEC) {
        bool ok = ((AK && BKF) || ((a.M % 64 == 0) && (a.N % 64 == 0))) && !colsum && !(a.flags & RFN_GEMM_OPT_NO_DMA);
        for (int g = 0; g < a.ngroups && ok; ++g) {
            ok = ok && a.g[g].ldc >= 0 && (double)a.g[g].ldc * 4 * 64 < 4.0e9;
            for (int s = 0; s < a.g[g].nseg; ++s) {
                const rfn_gemm_seg& sg = a.g[g].seg[s];
                const double ea = AK ? (double)a.M * sg.lda : (double)sg.K * sg.lda;
                const double eb = BKF ? (double)a.N * sg.ldb : (double)sg.K * sg.ldb;
                ok = ok && sg.K > 0 && (sg.K % 32 == 0) && ea * 4 < 4.0e9 && eb * 4 < 4.0e9 && sg.lda >= 0 && sg.ldb >= 0;
            }
        }
        if (ok) return launch_cfg<64, 64, AK, BKF, true, 1, 32, true, GEMM_THREADS, GEMM_SMALL_DMA_SLOTS>(a, st);
    }
#endif
    return launch_cfg<64, 64, AK, BKF, VEC, GEMM_SMALL_STAGES, GEMM_SMALL_BK>(a, st);
}

extern "C" int rfn_gemm_f32(int M, int N, int ngroups, const rfn_gemm_problem* problems, int accumulate,
                            void* stream) {
    return rfn_gemm_f32_opt(M, N, ngroups, problems, accumulate, nullptr, 0, 0u, stream);
}

extern "C" int rfn_gemm_f32_ws(int M, int N, int ngroups, const rfn_gemm_problem* problems, int accumulate,
                               void* ws, size_t ws_bytes, void* stream) {
    return rfn_gemm_f32_opt(M, N, ngroups, problems, accumulate, ws, ws_bytes, 0u, stream);
}

extern "C" int rfn_gemm_f32_opt(int M, int N, int ngroups, const rfn_gemm_problem* problems, int accumulate,
                                void* ws, size_t ws_bytes, unsigned flags, void* stream) {
    return rfn_gemm_f32_tk(M, N, ngroups, problems, accumulate, ws, ws_bytes, flags, nullptr, 0, stream);
}

static int gemm_entry(int M, int N, int ngroups, const rfn_gemm_problem* problems, int accumulate, void* ws,
                      size_t ws_bytes, unsigned flags, int32_t* tickets, int n_tickets, const rfn_gemm_lstm* lstm,
                      void* stream);

extern "C" int rfn_gemm_f32_tk(int M, int N, int ngroups, const rfn_gemm_problem* problems, int accumulate, void* ws,
                               size_t ws_bytes, unsigned flags, int32_t* tickets, int n_tickets, void* stream) {
    return gemm_entry(M, N, ngroups, problems, accumulate, ws, ws_bytes, flags, tickets, n_tickets, nullptr, stream);
}

// Gate GEMM + LSTM update: gates[M, 4R] = sum_s A_s W_s^T + b (rfn_gemm_f32 semantics, no accumulate), then the update of
// rfn_lstm_fwd_grouped on it.  When the product is cut along K the update rides on the fixed-order reduce (one launch
// less, the gate pre-activations never travel to HBM and back); otherwise it is the separate element-wise launch.
extern "C" int rfn_gemm_f32_lstm(int M, int R, int ngroups, const rfn_gemm_problem* problems, void* ws, size_t ws_bytes,
                                 unsigned flags, const rfn_gemm_lstm* lstm, void* stream) {
    if (!lstm || !lstm->c_prev || !lstm->c_next || !lstm->h_next || R < 1) return RFN_ERR_ARG;
    if (lstm->drop_p < 0.f || lstm->drop_p >= 1.f) return RFN_ERR_SHAPE;
    return gemm_entry(M, 4 * R, ngroups, problems, 0, ws, ws_bytes, flags, nullptr, 0, lstm, stream);
}

static int gemm_entry(int M, int N, int ngroups, const rfn_gemm_problem* problems, int accumulate, void* ws,
                      size_t ws_bytes, unsigned flags, int32_t* tickets, int n_tickets, const rfn_gemm_lstm* lstm,
                      void* stream) {
    if (M <= 0 || N <= 0) return RFN_OK;
    if (ngroups < 1 || ngroups > RFN_GEMM_MAXGROUP || !problems) return RFN_ERR_SHAPE;
    GemmArgs a;
    a.M = M;
    a.N = N;
    a.ngroups = ngroups;
    a.accumulate = accumulate;
    a.splitk = 1;
    a.tail_main_blocks = 0;
    a.tail_idx_main = 0;
    a.tail_parts = 2;
    a.flags = flags;
    a.tickets = (tickets && n_tickets > 0) ? (int*)tickets : nullptr;
    a.n_tickets = a.tickets ? n_tickets : 0;
    memset(&a.lstm, 0, sizeof(a.lstm));
    if (lstm) a.lstm = *lstm;
    a.part = (ws && ws_bytes >= (1u << 20) && rfn_aligned16(ws)) ? (float*)ws : nullptr;
    a.ws_mib = (int)(ws_bytes >> 20);
    const int ak = problems[0].seg[0].a_kfast, bk = problems[0].seg[0].b_kfast;
    bool vec = true;
    for (int g = 0; g < ngroups; ++g) {
        const rfn_gemm_problem& p = problems[g];
        if (p.nseg < 1 || p.nseg > RFN_GEMM_MAXSEG || !p.C) return RFN_ERR_SHAPE;
        for (int s = 0; s < p.nseg; ++s) {
            const rfn_gemm_seg& sg = p.seg[s];
            if (!sg.A || !sg.B || sg.K < 0) return RFN_ERR_ARG;
            if ((sg.a_kfast != 0) != (ak != 0) || (sg.b_kfast != 0) != (bk != 0)) return RFN_ERR_SHAPE;
            // float4 staging needs 16-B aligned rows and whole float4s along the contiguous index
            const bool a_ok = rfn_aligned16(sg.A) && (sg.lda % 4 == 0) && (ak ? sg.K % 4 == 0 : M % 4 == 0);
            const bool b_ok = rfn_aligned16(sg.B) && (sg.ldb % 4 == 0) && (bk ? sg.K % 4 == 0 : N % 4 == 0);
            vec = vec && a_ok && b_ok;
        }
        a.g[g] = p;
    }
    hipStream_t st = (hipStream_t)stream;
    int rc;
#define RFN_DISPATCH(AKV, BKV) rc = vec ? launch_tile<AKV, BKV, true>(a, st) : launch_tile<AKV, BKV, false>(a, st)
    if (ak && bk) { RFN_DISPATCH(true, true); }
    else if (ak && !bk) { RFN_DISPATCH(true, false); }
    else if (!ak && bk) { RFN_DISPATCH(false, true); }
    else { RFN_DISPATCH(false, false); }
#undef RFN_DISPATCH
    if (rc != RFN_OK || !lstm || a.splitk > 1) return rc;     // launch_tile records its split in a.splitk
    // unsplit product: the gate pre-activations are in C, the update is the element-wise launch
    return rfn_lstm_fwd_grouped(problems[0].C, problems[0].ldc, lstm->c_prev, lstm->ldcp, lstm->c_next, lstm->ldcn, lstm->h_next,
                                lstm->ldh, M, N / 4, 0, lstm->drop_p, lstm->seed, lstm->offset, ngroups,
                                ngroups > 1 ? (int64_t)(problems[1].C - problems[0].C) : 0, lstm->gs_cprev, lstm->gs_cnext,
                                lstm->gs_h, stream);
}
