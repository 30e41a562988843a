// f32 GEMM on the bf16 matrix cores: every f32 operand is held as THREE bf16 planes (x = x0 + x1 + x2 exactly for
// 2^-109 <= |x| < 2^128 and to 2^-133 absolute below that: 3 x 8 significant bits + 2 sign bits cover the 24-bit
// significand, bf16 has f32's exponent range) and C = A . B^T is formed from the six
// products whose weight is >= 2^-16 of the leading one,
//     a0.b0 + (a0.b1 + a1.b0) + (a0.b2 + a1.b1 + a2.b0),
// with v_mfma_f32_16x16x32_bf16 accumulating in f32.  The three dropped products are below 2^-24 relative,
// i.e. below the rounding of an f32 accumulation step.  Measured against f64 (tools/split_numerics_probe.hip, K = 512 ...
// 50176, normal / all-positive / wide-range data) the result is at least as close as the exact-f32 MFMA chain of
// rfn_gemm.hip (0.75-0.85x its rms error): the matrix cores run the bf16 shapes 16x faster than the f32 shape, so six
// products still leave 2.7x the f32 rate.
//
// Plane image of a logical operand Y[rows][K] (row = output index, K = reduction index), written by rfn_x3_split from an
// f32 matrix in either orientation: 1-KiB pieces in the lane order of the MFMA operand,
//     piece(kc, rb, p) at byte (((kc * nrb) + rb) * 3 + p) * 1024, lane l at + 16 * l:
//         plane p of Y[rb * RB + l % RB][kc * KC + 8 * (l / RB) + 0..7]        (RB x KC = 16 x 32, the product build's
//         v_mfma_f32_16x16x32_bf16; 32 x 16 with -DX3_SHAPE=32 for the A/B runs of tools/x3_gemm_bench.hip)
// rows are padded to a multiple of 256 and K to a multiple of KC with zeros, so the GEMM loads need no bounds checks.
// One wave-instruction of LDS-DMA (global_load_lds_dwordx4, 64 lanes x 16 B) moves one piece as one contiguous KiB of
// global memory into one contiguous KiB of LDS, which one ds_read_b128 per lane then reads back conflict-free (lane-
// linear 1 KiB): no swizzle, no staging registers, no address arithmetic per lane beyond lane * 16.
//
// Operands that are reduction-index-major in memory (both operands of a weight gradient) are not transposed into that
// image: they keep their orientation as K-SLOW plane images (rfn_x3_split_ks, or written directly by the producing
// kernel) and rfn_x3_gemm_ks forms the MFMA operands with the transposing LDS read -- see x3_tile<.., KS = true>.
#include "rfn_common.h"

typedef float x3_f32x16 __attribute__((ext_vector_type(16)));
typedef float x3_f32x4 __attribute__((ext_vector_type(4)));
typedef short x3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void x3_lds_void;
typedef const __attribute__((address_space(1))) void x3_gbl_void;

#define X3_ROW_PAD 256
#define X3_MAX_GROUPS 64

template <int SHAPE>
struct X3Shape {
    static constexpr int RB = SHAPE;                 // rows per piece
    static constexpr int KC = SHAPE == 32 ? 16 : 32; // reduction indices per piece
    static constexpr int ACC = SHAPE == 32 ? 16 : 4; // accumulator registers per lane and MFMA tile
};

// (x3_split: rfn_common.h)
struct X3SplitArgs {
    const float* src[X3_MAX_GROUPS];   // group g fills rows [g * rows, (g + 1) * rows) of the image
    long ld;
    int rows, K, kfast;
    int nrb, nkc;       // row blocks written per group (the last group also zero-fills the pad rows), pieces of K
    int nrb_img;        // row blocks per kc in the whole image
    int nrb_group;      // row-block distance between groups
    char* img;
};

// one wave per (kc, rb): reads 8 reduction indices of one row per lane, writes the three 1-KiB pieces
//   kfast = 1: src[row * ld + k]   kfast = 0: src[k * ld + row]
template <int SHAPE>
__global__ __launch_bounds__(256) void x3_split_k(const X3SplitArgs a) {
    using S = X3Shape<SHAPE>;
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.y;
    const float* __restrict__ src = a.src[g];
    const int nrb = a.nrb, nkc = a.nkc, rows = a.rows, K = a.K;
    const long ld = a.ld;
    const long piece = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (piece >= (long)nrb * nkc) return;
    // consecutive waves take consecutive kc of one row block when the source is k-fast (they share cache lines),
    // consecutive row blocks of one kc when it is row-fast
    int kc, rb;
    if (a.kfast) {
        rb = (int)(piece / nkc);
        kc = (int)(piece % nkc);
    } else {
        kc = (int)(piece / nrb);
        rb = (int)(piece % nrb);
    }
    const int row = rb * S::RB + lane % S::RB;
    const int k0 = kc * S::KC + 8 * (lane / S::RB);
    float x[8];
    if (a.kfast) {
        const float* p = src + (long)row * ld + k0;
        if (row < rows && k0 + 8 <= K && ((ld & 3) == 0) && ((((uintptr_t)src) & 15u) == 0)) {
            const x3_f32x4 u = *reinterpret_cast<const x3_f32x4*>(p);
            const x3_f32x4 v = *reinterpret_cast<const x3_f32x4*>(p + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                x[j] = u[j];
                x[4 + j] = v[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = (row < rows && k0 + j < K) ? p[j] : 0.f;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (row < rows && k0 + j < K) ? src[(long)(k0 + j) * ld + row] : 0.f;
    }
    unsigned q[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x3_split(x[j], q[0][j], q[1][j], q[2][j]);
    char* dst = a.img + (((long)kc * a.nrb_img + (long)g * a.nrb_group + rb) * 3) * 1024 + lane * 16;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        uint4 w;
        w.x = q[p][0] | (q[p][1] << 16);
        w.y = q[p][2] | (q[p][3] << 16);
        w.z = q[p][4] | (q[p][5] << 16);
        w.w = q[p][6] | (q[p][7] << 16);
        *reinterpret_cast<uint4*>(dst + p * 1024) = w;
    }
}

static inline long x3_rows_pad(long rows) { return (rows + X3_ROW_PAD - 1) / X3_ROW_PAD * X3_ROW_PAD; }
static inline long x3_k_pad(long K) { return (K + 31) / 32 * 32; }

extern "C" size_t rfn_x3_image_bytes(int rows, int K) {
    if (rows < 1 || K < 1) return 0;
    return (size_t)x3_rows_pad(rows) * (size_t)x3_k_pad(K) * 6;
}

#ifndef X3_SHAPE
#define X3_SHAPE 16   /* 16: v_mfma_f32_16x16x32_bf16 (pieces of 16 rows x 32 k); 32: v_mfma_f32_32x32x16_bf16 (32 x 16) */
#endif

// Image of the logical operand Y[ngroups * rows][K] whose row block g is the f32 matrix srcs_host[g] (rows x K, leading
// dimension ld; k_fast = 1: element (row, k) at src[row * ld + k], k_fast = 0: at src[k * ld + row]).  With more than
// one group, rows must be a multiple of 32.  One launch; the pad rows and pad columns of the image are zero-filled.
extern "C" int rfn_x3_split(const float* const* srcs_host, int ngroups, int64_t ld, int rows, int K, int k_fast, void* image,
                            void* stream) {
    if (!srcs_host || !image || rows < 1 || K < 1 || ngroups < 1 || ngroups > X3_MAX_GROUPS) return RFN_ERR_ARG;
    if (ngroups > 1 && rows % 32) return RFN_ERR_SHAPE;
    using S = X3Shape<X3_SHAPE>;
    X3SplitArgs a;
    for (int g = 0; g < ngroups; ++g) {
        if (!srcs_host[g]) return RFN_ERR_ARG;
        a.src[g] = srcs_host[g];
    }
    const long total = (long)ngroups * rows;
    a.ld = ld;
    a.rows = rows;
    a.K = K;
    a.kfast = k_fast;
    a.nkc = (int)(x3_k_pad(K) / S::KC);
    a.nrb_img = (int)(x3_rows_pad(total) / S::RB);
    a.nrb_group = rows / S::RB;
    // every group writes its own ceil(rows / RB) row blocks; the pad rows behind the last group get a second, small launch
    a.nrb = (rows + S::RB - 1) / S::RB;
    a.img = (char*)image;
    hipLaunchKernelGGL((x3_split_k<X3_SHAPE>), dim3((unsigned)(((long)a.nrb * a.nkc + 3) / 4), ngroups), dim3(256), 0,
                       (hipStream_t)stream, a);
    RFN_CHECK_LAUNCH();
    const int pad_rb = a.nrb_img - (int)((total + S::RB - 1) / S::RB);
    if (pad_rb > 0) {   // zero rows: a "group" of 0 valid rows placed behind the last one
        X3SplitArgs z = a;
        z.src[0] = srcs_host[0];
        z.rows = 0;
        z.nrb = pad_rb;
        z.nrb_group = 0;
        z.img = (char*)image + (long)(a.nrb_img - pad_rb) * 3 * 1024;
        hipLaunchKernelGGL((x3_split_k<X3_SHAPE>), dim3((unsigned)(((long)z.nrb * z.nkc + 3) / 4), 1), dim3(256), 0,
                           (hipStream_t)stream, z);
        RFN_CHECK_LAUNCH();
    }
    return RFN_OK;
}

// ---- the GEMM ----------------------------------------------------------------------------------------------------------
struct X3Args {
    const char* A;   // plane image of the M-side operand  [M][K]
    const char* B;   // plane image of the N-side operand  [N][K]
    int nrbA, nrbB;  // row blocks per kc in each image (fragment-order images)
    int mpA, mpB;    // padded row counts = plane-row pitch in elements (k-slow images, x3_tile<.., KS = true>)
    int M, N;        // logical output size (stores are bounds-checked)
    int nkc;         // K_pad / KC
    int splitk;      // > 1: blockIdx.z cuts nkc; raw partial tiles go to part[ks][M][N]
    float* part;
    int gm, gn, ngn; // output groups: gm rows x gn columns each, C[(m / gm) * ngn + n / gn], leading dimension ldc
    long ldc;
    int accumulate;
    int tiles_m, tiles_n;
    int main_tiles;  // tiles [0, main_tiles) are whole; each later tile is done by four blocks, a quarter each
    float* C[X3_MAX_GROUPS];
    const float* bias[X3_MAX_GROUPS];   // per group: bias[g][column inside the group] or nullptr
};

template <int N>
__device__ __forceinline__ void x3_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int SHAPE>
using x3_acc_t = typename std::conditional<SHAPE == 32, x3_f32x16, x3_f32x4>::type;

template <int SHAPE>
__device__ __forceinline__ void x3_mfma(const x3_bf16x8& a, const x3_bf16x8& b, x3_acc_t<SHAPE>& c) {
    if constexpr (SHAPE == 32) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    else c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// One BM x BN output tile at (row0, col0) over the piece rows [kc0, kc0 + iters) of K, by WGM x WGN waves; operands staged
// by LDS-DMA, one piece row of K per step (32 reduction indices for the 16 x 16 shape, 16 for the 32 x 32 shape).  The two
// shapes have their own main loops (below); setup, accumulators and epilogue are shared.
template <int SHAPE, int BM, int BN, int WGM, int WGN, int SLOTS, bool KS = false>
__device__ __forceinline__ void x3_tile(const X3Args& args, const int row0, const int col0, const int kc0, const int iters,
                                        const int ks) {
    using S = X3Shape<SHAPE>;
    constexpr int NW = WGM * WGN;
    constexpr int RB = S::RB;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int MI = WM / RB, NI = WN / RB;
    constexpr int PA = (BM / RB) * 3, PB = (BN / RB) * 3;   // pieces per kc
    constexpr int PT = PA + PB;                             // pieces per step
    constexpr int PPW = PT / NW;                            // per wave
    static_assert(PT % NW == 0 && MI >= 1 && NI >= 1, "pieces must divide over the waves");
    constexpr int SLOT_BYTES = PT * 1024;
    extern __shared__ __attribute__((aligned(16))) char x3_smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;

    const unsigned lane16 = lane * 16;

    x3_acc_t<SHAPE> acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < S::ACC; ++r) acc[i][j][r] = 0.f;

    if constexpr (SHAPE == 32) {
        // Symmetric ring of SLOTS slots of both operands:
        //   iteration it:  wait until this wave's pieces of slot `it` have landed (younger slots stay in flight) -> barrier
        //                  (every wave's pieces landed; every wave is done reading slot it-1) -> issue slot it+SLOTS-1 ->
        //                  fragment reads + MFMAs on slot it.
        // this wave's pieces of a slot: piece n -> (operand, piece inside the tile rows); wave-uniform addresses
        const char* src[PPW];
        long step[PPW];
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int n = wave * PPW + j;
            const int rem = n;
            if (rem < PA) {
                src[j] = args.A + (((long)kc0 * args.nrbA + row0 / RB) * 3 + rem) * 1024;
                step[j] = (long)args.nrbA * 3072;
            } else {
                src[j] = args.B + (((long)kc0 * args.nrbB + col0 / RB) * 3 + (rem - PA)) * 1024;
                step[j] = (long)args.nrbB * 3072;
            }
        }
        auto issue = [&](int slot) {
            char* st = x3_smem + slot * SLOT_BYTES + wave * PPW * 1024;
#pragma unroll
            for (int j = 0; j < PPW; ++j) {
                __builtin_amdgcn_global_load_lds((x3_gbl_void*)(src[j] + lane16), (x3_lds_void*)(st + j * 1024), 16, 0, 0);
                src[j] += step[j];
            }
        };

        int issued = 0;
#pragma unroll
        for (int s = 0; s < SLOTS - 1; ++s)
            if (s < iters) {
                issue(s);
                ++issued;
            }
        int cur = 0, fill = SLOTS - 1;
        for (int it = 0; it < iters; ++it) {
            const int younger = issued - it - 1;
            if (SLOTS >= 4 && younger >= 2) x3_wait_vmcnt<(SLOTS >= 4 ? 2 : 0) * PPW>();
            else if (SLOTS >= 3 && younger >= 1) x3_wait_vmcnt<(SLOTS >= 3 ? 1 : 0) * PPW>();
            else x3_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (issued < iters) {
                issue(fill);
                ++issued;
            }
            const char* sl = x3_smem + cur * SLOT_BYTES + lane16;
                {
                const char* a_l = sl + (wm * MI * 3) * 1024;
                const char* b_l = sl + (PA + wn * NI * 3) * 1024;
                x3_bf16x8 b[NI][3];
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int p = 0; p < 3; ++p) b[j][p] = *reinterpret_cast<const x3_bf16x8*>(b_l + (j * 3 + p) * 1024);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    x3_bf16x8 a[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const x3_bf16x8*>(a_l + (i * 3 + p) * 1024);
                    // smallest products first
#pragma unroll
                    for (int j = 0; j < NI; ++j) x3_mfma<SHAPE>(a[2], b[j][0], acc[i][j]);
#pragma unroll
                    for (int j = 0; j < NI; ++j) x3_mfma<SHAPE>(a[1], b[j][1], acc[i][j]);
#pragma unroll
                    for (int j = 0; j < NI; ++j) x3_mfma<SHAPE>(a[0], b[j][2], acc[i][j]);
#pragma unroll
                    for (int j = 0; j < NI; ++j) x3_mfma<SHAPE>(a[1], b[j][0], acc[i][j]);
#pragma unroll
                    for (int j = 0; j < NI; ++j) x3_mfma<SHAPE>(a[0], b[j][1], acc[i][j]);
#pragma unroll
                    for (int j = 0; j < NI; ++j) x3_mfma<SHAPE>(a[0], b[j][0], acc[i][j]);
                }
            }
            cur = (cur + 1 == SLOTS) ? 0 : cur + 1;
            fill = (fill + 1 == SLOTS) ? 0 : fill + 1;
        }
    } else {
        // 16x16x32 shape: a K step is 32 deep, so a slot of both operands is 96 KB and does not fit twice.  Instead the step's
        // B fragments go to registers first (12 per wave for a 64-wide wave tile), which frees the single B buffer for the
        // next step's DMA while the row blocks of A stream from one of two A buffers:
        //   step s:  wait for this wave's pieces of A_s, B_s -> barrier 1 (both landed everywhere; every wave is done with
        //            A_(s-1)) -> issue A_(s+1) -> B_s fragments to registers -> barrier 2 (nobody reads B_s from LDS any
        //            more) -> issue B_(s+1) -> per 16-row block: 3 fragment reads, NI x 6 MFMAs.
        constexpr int PAW = PA / NW, PBW = PB / NW;
        static_assert(PA % NW == 0 && PB % NW == 0, "each operand's pieces must divide over the waves");
        char* const bufA = x3_smem;                   // two A stages
        char* const bufB = x3_smem + 2 * PA * 1024;   // one B stage
        // k-slow images (KS): element (k, plane, m) at ((k * 3 + plane) * Mp + m) * 2 bytes.  A stage in LDS is, per plane, 32
        // k-rows of the tile's BM (BN) columns, row-major; a fragment is read with ds_read_b64_tr_b16 (4 k-rows x 16 columns
        // per 16-lane group, delivered column-major = 8 consecutive k per lane after two reads).  The 8 rows a 32-lane half
        // reads are RS = 2 BM bytes apart, i.e. on the same banks, so 16-B chunk c of row r is stored at chunk position
        // c ^ swz(r), swz(r) = 2 ((r & 3) | ((r >> 3) & 1) << 2): the 8 rows x 2 chunks of a half then cover 16 distinct
        // 16-B bank slots.  The LDS-DMA writes lane-linearly, so the swizzle goes on the SOURCE address.
        constexpr int RSA = 2 * BM, RSB = 2 * BN;   // stage row pitch in LDS, bytes
        auto ks_swz = [](int r) { return 2 * ((r & 3) | (((r >> 3) & 1) << 2)); };
        const char* gA;
        const char* gB;
        long sA, sB;
        unsigned offA[PAW], offB[PBW];   // KS: this lane's source offset per piece (fixed for the whole loop)
        if constexpr (KS) {
            gA = args.A + ((long)kc0 * 32 * 3 * args.mpA + row0) * 2;
            gB = args.B + ((long)kc0 * 32 * 3 * args.mpB + col0) * 2;
            sA = (long)32 * 3 * args.mpA * 2;
            sB = (long)32 * 3 * args.mpB * 2;
#pragma unroll
            for (int j = 0; j < PAW; ++j) {
                const int byte = (wave * PAW + j) * 1024 + lane * 16, p = byte / (32 * RSA), rem = byte % (32 * RSA);
                const int r = rem / RSA, c = (rem % RSA) / 16;
                offA[j] = (unsigned)(((long)(r * 3 + p) * args.mpA) * 2 + ((c ^ ks_swz(r)) * 16));
            }
#pragma unroll
            for (int j = 0; j < PBW; ++j) {
                const int byte = (wave * PBW + j) * 1024 + lane * 16, p = byte / (32 * RSB), rem = byte % (32 * RSB);
                const int r = rem / RSB, c = (rem % RSB) / 16;
                offB[j] = (unsigned)(((long)(r * 3 + p) * args.mpB) * 2 + ((c ^ ks_swz(r)) * 16));
            }
        } else {
            gA = args.A + (((long)kc0 * args.nrbA + row0 / RB) * 3 + wave * PAW) * 1024;
            gB = args.B + (((long)kc0 * args.nrbB + col0 / RB) * 3 + wave * PBW) * 1024;
            sA = (long)args.nrbA * 3072;
            sB = (long)args.nrbB * 3072;
#pragma unroll
            for (int j = 0; j < PAW; ++j) offA[j] = j * 1024 + lane16;
#pragma unroll
            for (int j = 0; j < PBW; ++j) offB[j] = j * 1024 + lane16;
        }
        auto issue_a = [&](int buf) {
            char* st = bufA + (buf * PA + wave * PAW) * 1024;
#pragma unroll
            for (int j = 0; j < PAW; ++j)
                __builtin_amdgcn_global_load_lds((x3_gbl_void*)(gA + offA[j]), (x3_lds_void*)(st + j * 1024), 16, 0, 0);
            gA += sA;
        };
        auto issue_b = [&]() {
            char* st = bufB + wave * PBW * 1024;
#pragma unroll
            for (int j = 0; j < PBW; ++j)
                __builtin_amdgcn_global_load_lds((x3_gbl_void*)(gB + offB[j]), (x3_lds_void*)(st + j * 1024), 16, 0, 0);
            gB += sB;
        };
        // fragment of 16-row block `blk` (of this wave's rows), plane p, from a stage at `base`
        unsigned fa_off[MI], fb_off[NI];   // KS: this lane's address inside a stage per block (plane 0, first read)
        if constexpr (KS) {
            const int g = lane >> 4, jj = lane & 15, r1 = 8 * g + (jj >> 2);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int c = (wm * WM + 16 * i) / 8 + ((jj & 3) >> 1);
                fa_off[i] = (unsigned)(r1 * RSA + ((c ^ ks_swz(r1)) * 16) + 8 * (jj & 1));
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int c = (wn * WN + 16 * j) / 8 + ((jj & 3) >> 1);
                fb_off[j] = (unsigned)(r1 * RSB + ((c ^ ks_swz(r1)) * 16) + 8 * (jj & 1));
            }
        }
        // KS fragments are read by inline asm: behind a pending LDS-DMA the compiler puts s_waitcnt vmcnt(0) in front of every
        // ds_read_b64_tr_b16 it emits itself (it cannot see that the DMA targets another buffer), which drains the prefetch
        // of the next step in every iteration.  The asm reads are invisible to that pass AND to its lgkmcnt bookkeeping:
        // every use of their results sits behind an explicit s_waitcnt that names the registers (x3_lgkm_fence).
        typedef unsigned long long x3_u64;
        typedef x3_u64 x3_u64x2 __attribute__((ext_vector_type(2)));
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)x3_smem;
#define X3_TR(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
        auto frag3_ks = [&](x3_u64x2 (&f)[3], unsigned addr, auto rs) {   // the three planes of one 16-row block
            constexpr int RS = decltype(rs)::value;
            X3_TR(f[0][0], addr, 0);
            X3_TR(f[0][1], addr, 4 * RS);
            X3_TR(f[1][0], addr, 32 * RS);
            X3_TR(f[1][1], addr, 32 * RS + 4 * RS);
            X3_TR(f[2][0], addr, 64 * RS);
            X3_TR(f[2][1], addr, 64 * RS + 4 * RS);
        };
        auto frag_a = [&](const char* base, int i, int p) -> x3_bf16x8 {
            return *reinterpret_cast<const x3_bf16x8*>(base + lane16 + ((wm * MI + i) * 3 + p) * 1024);
        };
        auto frag_b = [&](const char* base, int j, int p) -> x3_bf16x8 {
            return *reinterpret_cast<const x3_bf16x8*>(base + lane16 + ((wn * NI + j) * 3 + p) * 1024);
        };
        if (iters > 0) {
            issue_a(0);
            issue_b();
        }
        for (int st = 0; st < iters; ++st) {
            x3_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            const bool more = st + 1 < iters;
            if (more) issue_a((st + 1) & 1);
            if constexpr (KS) {
                x3_u64x2 bq[NI][3], aq[2][3];
                const unsigned bB = lds0 + 2 * PA * 1024, bA = lds0 + (st & 1) * PA * 1024;
#pragma unroll
                for (int j = 0; j < NI; ++j) frag3_ks(bq[j], bB + fb_off[j], std::integral_constant<int, RSB>{});
                frag3_ks(aq[0], bA + fa_off[0], std::integral_constant<int, RSA>{});   // A_s landed with barrier 1
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's B fragments are in registers
                // (volatile asm statements keep their order; routing each register through one pins its first use behind the wait)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int p = 0; p < 3; ++p) asm volatile("" : "+v"(bq[j][p]));
#pragma unroll
                for (int p = 0; p < 3; ++p) asm volatile("" : "+v"(aq[0][p]));
                __builtin_amdgcn_s_barrier();
                if (more) issue_b();
                x3_bf16x8 b[NI][3];
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int p = 0; p < 3; ++p) b[j][p] = __builtin_bit_cast(x3_bf16x8, bq[j][p]);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    // the next row block's fragments travel while this one multiplies
                    if (i + 1 < MI) frag3_ks(aq[(i + 1) & 1], bA + fa_off[i + 1], std::integral_constant<int, RSA>{});
                    x3_bf16x8 a[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) a[p] = __builtin_bit_cast(x3_bf16x8, aq[i & 1][p]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < NI; ++j) x3_mfma<SHAPE>(a[2], b[j][0], acc[i][j]);
#pragma unroll
                    for (int j = 0; j < NI; ++j) x3_mfma<SHAPE>(a[1], b[j][1], acc[i][j]);
#pragma unroll
                    for (int j = 0; j < NI; ++j) x3_mfma<SHAPE>(a[0], b[j][2], acc[i][j]);
#pragma unroll
                    for (int j = 0; j < NI; ++j) x3_mfma<SHAPE>(a[1], b[j][0], acc[i][j]);
#pragma unroll
                    for (int j = 0; j < NI; ++j) x3_mfma<SHAPE>(a[0], b[j][1], acc[i][j]);
#pragma unroll
                    for (int j = 0; j < NI; ++j) x3_mfma<SHAPE>(a[0], b[j][0], acc[i][j]);
                    __builtin_amdgcn_sched_barrier(0);
                    // the reads issued above are complete before the next row block (or the next step's barrier) uses them
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int p = 0; p < 3; ++p) asm volatile("" : "+v"(aq[(i + 1) & 1][p]));
                }
                continue;
            }
            x3_bf16x8 b[NI][3];
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) b[j][p] = frag_b(bufB, j, p);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's B fragments are in registers
            __builtin_amdgcn_s_barrier();
            if (more) issue_b();
            const char* a_s = bufA + (st & 1) * PA * 1024;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                x3_bf16x8 a[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) a[p] = frag_a(a_s, i, p);
#pragma unroll
                for (int j = 0; j < NI; ++j) x3_mfma<SHAPE>(a[2], b[j][0], acc[i][j]);
#pragma unroll
                for (int j = 0; j < NI; ++j) x3_mfma<SHAPE>(a[1], b[j][1], acc[i][j]);
#pragma unroll
                for (int j = 0; j < NI; ++j) x3_mfma<SHAPE>(a[0], b[j][2], acc[i][j]);
#pragma unroll
                for (int j = 0; j < NI; ++j) x3_mfma<SHAPE>(a[1], b[j][0], acc[i][j]);
#pragma unroll
                for (int j = 0; j < NI; ++j) x3_mfma<SHAPE>(a[0], b[j][1], acc[i][j]);
#pragma unroll
                for (int j = 0; j < NI; ++j) x3_mfma<SHAPE>(a[0], b[j][0], acc[i][j]);
            }
        }
    }
    __builtin_amdgcn_s_barrier();   // every wave is past its last fragment read: the ring's LDS is free for the epilogue

    // ---- epilogue ----
#ifdef X3_ABLATE_STORE   /* diagnostic builds only: keep the accumulators alive, store nothing */
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
#endif
    // One wave tile lies inside one output group (the host checks gm % BM == gn % BN == 0), so the group, its base pointer
    // and its bias row are wave-uniform: addresses are that base (SGPR pair) + a 32-bit lane offset, and a wave tile that
    // lies wholly inside M x N stores without per-element checks.
    const int M = args.M, N = args.N;
    const int row_w = row0 + wm * WM, col_w = col0 + wn * WN;
    if (row_w >= M || col_w >= N) return;
    const bool raw = args.splitk > 1;
    char* cbase;
    const float* bias_w = nullptr;
    unsigned ld4;
    if (raw) {
        cbase = (char*)(args.part + ((long)ks * M + row_w) * N + col_w);
        ld4 = (unsigned)N * 4u;
    } else {
        const int gi = row_w / args.gm, gj = col_w / args.gn, g = gi * args.ngn + gj;
        cbase = (char*)(args.C[g] + (long)(row_w - gi * args.gm) * args.ldc + (col_w - gj * args.gn));
        if (args.bias[g]) bias_w = args.bias[g] + (col_w - gj * args.gn);
        ld4 = (unsigned)args.ldc * 4u;
    }
    const int lr = lane % RB;                                      // column inside an MFMA tile
    const int lrow = SHAPE == 32 ? 4 * (lane >> 5) : 4 * (lane >> 4);   // first row of this lane's register group
    const unsigned lane_off = (unsigned)lrow * ld4 + (unsigned)lr * 4u;
    const bool accumulate = !raw && args.accumulate;
    const bool full = row_w + WM <= M && col_w + WN <= N;
    if (full && !accumulate && ((((uintptr_t)cbase) | ld4) & 15u) == 0) {
        // whole wave tile, plain store: RB rows at a time go through this wave's corner of the (now free) LDS so that the
        // global stores are 16 B per lane, whole rows of the wave tile per instruction -- a quarter of the store
        // instructions of the register layout (4 B per lane), which is what the epilogue is bound by at one block per CU
        constexpr int EPW = WN + 4;                 // padded row, floats
        constexpr int LPR = WN / 4, RPI = 64 / LPR;  // lanes per row, rows per store instruction
        // rows staged at a time: a whole MFMA tile row (RB) if 8 waves x RB x EPW floats fit the ring, else half of it
        constexpr int RING_BYTES = SHAPE == 32 ? SLOTS * SLOT_BYTES : (2 * PA + PB) * 1024;
        constexpr int EPR = (NW * RB * EPW * 4 <= RING_BYTES) ? RB : RB / 2;
        static_assert(NW * EPR * EPW * 4 <= RING_BYTES && (SHAPE == 32 || EPR == RB), "epilogue staging must fit the ring");
        float* ep = reinterpret_cast<float*>(x3_smem) + wave * (EPR * EPW);
        const int erow = lane / LPR, ecol = 4 * (lane % LPR);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int hb = 0; hb < RB / EPR; ++hb) {   // 32 x 32 tiles: registers 8 hb .. 8 hb + 7 hold rows 16 hb .. 16 hb + 15
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const float bsum = bias_w ? bias_w[j * RB + lr] : 0.f;
#pragma unroll
                    for (int r = 0; r < S::ACC; ++r) {
                        const int rr = SHAPE == 32 ? (r & 3) + 8 * (r >> 2) : r;
                        if (rr / EPR != hb) continue;
                        ep[(lrow + rr - hb * EPR) * EPW + j * RB + lr] = acc[i][j][r] + bsum;
                    }
                }
                asm volatile("" ::: "memory");   // LDS operations of one wave execute in order: the reads see the writes
#pragma unroll
                for (int it = 0; it < EPR / RPI; ++it) {
                    const int row = it * RPI + erow;
                    const x3_f32x4 v = *reinterpret_cast<const x3_f32x4*>(ep + row * EPW + ecol);
                    *reinterpret_cast<x3_f32x4*>(cbase + ((unsigned)(i * RB + hb * EPR + row) * ld4 + (unsigned)ecol * 4u)) = v;
                }
                asm volatile("" ::: "memory");
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const bool col_ok = full || col_w + j * RB + lr < N;
        const float bsum = (bias_w && col_ok) ? bias_w[j * RB + lr] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const unsigned sub = lane_off + (unsigned)(i * RB) * ld4 + (unsigned)(j * RB) * 4u;
#pragma unroll
            for (int r = 0; r < S::ACC; ++r) {
                const int rr = SHAPE == 32 ? (r & 3) + 8 * (r >> 2) : r;   // row inside the MFMA tile, less lrow
                float* o = reinterpret_cast<float*>(cbase + (sub + (unsigned)rr * ld4));
                if (full || (col_ok && row_w + i * RB + lrow + rr < M)) {
                    float v = acc[i][j][r] + bsum;
                    if (accumulate) v += *o;
                    *o = v;
                }
            }
        }
    }
}

// Tile index -> (tm, tn).  The 32 tiles an XCD keeps in flight (one per CU) should share operand panels through its L2:
// X3_BAND_ROWS x X3_BAND_COLS = 8 x 4 consecutive indices form one block of the tile grid, so a K step of those 32 tiles
// fetches 8 + 4 panels instead of the 2 + 16 of a row-major walk over 16 column tiles (fabric traffic of the projection
// launch 5.77 -> see profiles/r03_x3_traffic.md).  Bands of 8 tile rows, inside a band groups of 4 tile columns, row-major
// inside a group; ragged last band / last group handled.
#ifndef X3_BAND_ROWS
#define X3_BAND_ROWS 8
#endif
#ifndef X3_BAND_COLS
#define X3_BAND_COLS 4
#endif
__device__ __forceinline__ void x3_tile_of(int wg, int tiles_m, int tiles_n, int& tm, int& tn) {
    const int per_band = X3_BAND_ROWS * tiles_n;
    const int band = wg / per_band;
    int idx = wg - band * per_band;
    const int br = min(X3_BAND_ROWS, tiles_m - band * X3_BAND_ROWS);
    const int per_group = br * X3_BAND_COLS;
    int grp = idx / per_group;
    const int full = tiles_n / X3_BAND_COLS;           // whole groups of X3_BAND_COLS columns
    int gc = X3_BAND_COLS;
    if (grp >= full) {                                 // the ragged last group
        grp = full;
        gc = tiles_n - full * X3_BAND_COLS;
    }
    idx -= grp * per_group;
    const int r = idx / gc;
    tm = band * X3_BAND_ROWS + r;
    tn = grp * X3_BAND_COLS + (idx - r * gc);
}

// grid.x: main_tiles whole tiles (blocks that share an XCD, id % 8, take consecutive tile indices, x3_tile_of above),
// then 4 quarter-tile blocks per
// remaining tile: the last, partly filled round of a long launch is spread over four times as many CUs.  grid.z: K slices.
template <int SHAPE, int BM, int BN, int WGM, int WGN, int SLOTS, bool TAIL, bool KS = false>
__global__ __launch_bounds__(64 * WGM * WGN) void x3_gemm_k(const X3Args args) {
    int per = (args.nkc + args.splitk - 1) / args.splitk;
    const int ks = blockIdx.z;
    const int kc0 = ks * per;
    per = min(per, args.nkc - ks * per);
    if ((int)blockIdx.x < args.main_tiles) {
        int wg = blockIdx.x;
        const int q = args.main_tiles / 8, r = args.main_tiles % 8, xcd = wg % 8;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + wg / 8;
        int tm, tn;
        x3_tile_of(wg, args.tiles_m, args.tiles_n, tm, tn);
        x3_tile<SHAPE, BM, BN, WGM, WGN, SLOTS, KS>(args, tm * BM, tn * BN, kc0, per, ks);
    } else if constexpr (TAIL) {
        const int t = blockIdx.x - args.main_tiles;
        const int wg = args.main_tiles + t / 4, qd = t % 4;
        int tm, tn;
        x3_tile_of(wg, args.tiles_m, args.tiles_n, tm, tn);
        x3_tile<SHAPE, BM / 2, BN / 2, WGM, WGN, SLOTS, KS>(args, tm * BM + (qd >> 1) * (BM / 2), tn * BN + (qd & 1) * (BN / 2),
                                                             kc0, per, ks);
    }
}

// out = sum over the K slices in a fixed order (+ bias) (+ previous C)
__global__ __launch_bounds__(256) void x3_reduce_k(const X3Args args) {
    const long n4 = (long)args.M * args.N / 4;   // host: N % 4 == 0, gn % 4 == 0
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        x3_f32x4 s = reinterpret_cast<const x3_f32x4*>(args.part)[i];
        for (int k = 1; k < args.splitk; ++k) {
            const x3_f32x4 t = reinterpret_cast<const x3_f32x4*>(args.part + (long)k * args.M * args.N)[i];
            s += t;
        }
        const long e = i * 4;
        const int row = (int)(e / args.N), col = (int)(e % args.N);
        const int gi = row / args.gm, gj = col / args.gn, g = gi * args.ngn + gj;
        const int cin = col - gj * args.gn;
        float* o = args.C[g] + (long)(row - gi * args.gm) * args.ldc + cin;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = s[c];
            if (args.bias[g]) v += args.bias[g][cin + c];
            if (args.accumulate) v += o[c];
            o[c] = v;
        }
    }
}

#ifndef X3_BM
#define X3_BM 256
#endif
#ifndef X3_BN
#define X3_BN 256
#endif
#ifndef X3_WGM
#define X3_WGM 4
#endif
#ifndef X3_WGN
#define X3_WGN 2
#endif
#ifndef X3_SLOTS
#define X3_SLOTS 2
#endif
#ifndef X3_KS_WGM
#define X3_KS_WGM 2
#endif
#ifndef X3_KS_WGN
#define X3_KS_WGN 4
#endif
#ifndef X3_BLOCKS_PER_CU
#define X3_BLOCKS_PER_CU 1   /* resident blocks per CU of the chosen tile (LDS-bound); sizes the tail round */
#endif
#ifndef X3_TAIL
#define X3_TAIL 1   /* quarter tiles for a last round that is at most a quarter full */
#endif

extern "C" size_t rfn_x3_part_floats(int M, int N, int splitk) { return splitk > 1 ? (size_t)splitk * M * N : 0; }

// how many K slices a launch of this shape should take so that one round of blocks covers the chip (1 for long launches)
extern "C" int rfn_x3_splitk_for(int M, int N, int K) {
    using S = X3Shape<X3_SHAPE>;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const long tiles = (long)((M + X3_BM - 1) / X3_BM) * ((N + X3_BN - 1) / X3_BN);
    if (tiles >= cus) return 1;
    const int iters = (int)(x3_k_pad(K) / S::KC);
    int sk = (int)((cus + tiles / 2) / tiles);
    while (sk > 1 && iters / sk < 64) --sk;   // keep slices long enough to amortise the pipeline fill and the reduce
    return sk < 1 ? 1 : sk;
}

// C groups (+)= A . B^T on plane images.  M, N: logical output size; K: logical reduction length (the images hold it
// padded).  Output groups of gm rows x gn columns, pointer table C_host[(m / gm) * ngn + n / gn] (device pointers, host
// array), leading dimension ldc; bias_host may be NULL.  splitk > 1 needs part (splitk * M * N floats).
// KS: both operands are k-slow images (rfn_x3_split_ks) instead of fragment-order images (rfn_x3_split).
template <bool KS>
static int x3_launch(int M, int N, int K, const void* imgA, const void* imgB, int gm, int gn, float* const* C_host,
                     const float* const* bias_host, int64_t ldc, int accumulate, int splitk, float* part, void* stream) {
    using S = X3Shape<X3_SHAPE>;
    if (M < 1 || N < 1 || K < 1 || !imgA || !imgB || !C_host || gm < 1 || gn < 1) return RFN_ERR_ARG;
    X3Args a;
    a.A = (const char*)imgA;
    a.B = (const char*)imgB;
    a.nrbA = (int)(x3_rows_pad(M) / S::RB);
    a.nrbB = (int)(x3_rows_pad(N) / S::RB);
    a.mpA = (int)x3_rows_pad(M);
    a.mpB = (int)x3_rows_pad(N);
    a.M = M;
    a.N = N;
    a.nkc = (int)(x3_k_pad(K) / S::KC);
    a.splitk = splitk < 1 ? 1 : splitk;
    a.part = part;
    a.gm = gm;
    a.gn = gn;
    const int ngm = (M + gm - 1) / gm;
    a.ngn = (N + gn - 1) / gn;
    if ((long)ngm * a.ngn > X3_MAX_GROUPS) return RFN_ERR_SHAPE;
    if ((ngm > 1 && gm % X3_BM) || (a.ngn > 1 && gn % X3_BN)) return RFN_ERR_SHAPE;   // no tile straddles groups
    if (a.splitk > a.nkc) return RFN_ERR_SHAPE;
    if (a.splitk > 1 && (!part || (N & 3) || (gn & 3))) return RFN_ERR_ARG;
    if ((double)X3_BM * (double)(a.splitk > 1 ? N : ldc) * 4.0 >= 4294967296.0) return RFN_ERR_SHAPE;   // 32-bit tile offsets
    if (KS && (96.0 * a.mpA * 2 >= 4294967296.0 || 96.0 * a.mpB * 2 >= 4294967296.0)) return RFN_ERR_SHAPE;
    a.ldc = ldc;
    a.accumulate = accumulate;
    a.tiles_m = (M + X3_BM - 1) / X3_BM;
    a.tiles_n = (N + X3_BN - 1) / X3_BN;
    for (int g = 0; g < ngm * a.ngn; ++g) {
        a.C[g] = C_host[g];
        a.bias[g] = bias_host ? bias_host[g] : nullptr;
    }
    // KS keeps 24 more address registers per lane: the 2 x 4 wave arrangement (48 B-fragment registers instead of 96) fits
    constexpr int WGM = KS ? X3_KS_WGM : X3_WGM, WGN = KS ? X3_KS_WGN : X3_WGN;
    auto kern = x3_gemm_k<X3_SHAPE, X3_BM, X3_BN, WGM, WGN, X3_SLOTS, (X3_TAIL != 0), KS>;
    // the operand ring; the epilogue stages inside it.  32x32x16: SLOTS slots of both operands; 16x16x32: two A stages + one B
    constexpr int lds = X3_SHAPE == 32 ? X3_SLOTS * ((X3_BM + X3_BN) / S::RB) * 3 * 1024
                                       : (2 * (X3_BM / S::RB) + X3_BN / S::RB) * 3 * 1024;
    static bool attr_set[16] = {};   // write-once per device (one set per instantiation of this function)
    static int cus[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return RFN_ERR_LAUNCH;
    if (!attr_set[dev & 15]) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return RFN_ERR_LAUNCH;
        int v = 256;
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
        cus[dev & 15] = v;
        attr_set[dev & 15] = true;
    }
    const int tiles = a.tiles_m * a.tiles_n, slots = cus[dev & 15] * X3_BLOCKS_PER_CU;
    const int rem = tiles % slots;
    const bool tail = X3_TAIL && a.splitk == 1 && tiles / slots >= 2 && rem > 0 && 4 * rem <= slots;
    a.main_tiles = tail ? tiles - rem : tiles;
    hipLaunchKernelGGL(kern, dim3(a.main_tiles + 4 * (tiles - a.main_tiles), 1, a.splitk), dim3(64 * WGM * WGN), lds,
                       (hipStream_t)stream, a);
    RFN_CHECK_LAUNCH();
    if (a.splitk > 1) {
        hipLaunchKernelGGL(x3_reduce_k, dim3(1024), dim3(256), 0, (hipStream_t)stream, a);
        RFN_CHECK_LAUNCH();
    }
    return RFN_OK;
}

extern "C" int rfn_x3_gemm(int M, int N, int K, const void* imgA, const void* imgB, int gm, int gn, float* const* C_host,
                           const float* const* bias_host, int64_t ldc, int accumulate, int splitk, float* part, void* stream) {
    return x3_launch<false>(M, N, K, imgA, imgB, gm, gn, C_host, bias_host, ldc, accumulate, splitk, part, stream);
}

#if X3_SHAPE == 16
extern "C" int rfn_x3_gemm_ks(int M, int N, int K, const void* imgA, const void* imgB, int gm, int gn, float* const* C_host,
                              const float* const* bias_host, int64_t ldc, int accumulate, int splitk, float* part,
                              void* stream) {
    return x3_launch<true>(M, N, K, imgA, imgB, gm, gn, C_host, bias_host, ldc, accumulate, splitk, part, stream);
}
#endif

// ---- f32 [k][m] -> k-slow plane image -----------------------------------------------------------------------------------
// image element (k, plane, m) at ((k * 3 + plane) * Mp + m) * 2 bytes, Mp = total columns padded to 256, k padded to 32;
// column block g (cols columns) comes from srcs[g][k * ld + c].  One thread per 4 columns: a float4 in, three 8-byte
// plane groups out, everything coalesced; pad rows and pad columns are written as zeros.
struct X3SplitKsArgs {
    const float* src[X3_MAX_GROUPS];
    long ld;
    int K, cols, ngroups, mp, kpad;
    char* img;
};
__global__ __launch_bounds__(256) void x3_split_ks_k(const X3SplitKsArgs a) {
    const int k = blockIdx.x;      // the reduction index (B*L rows of the weight-gradient operands) rides on grid.x: no 65535 cap
    const int m = (blockIdx.y * 256 + threadIdx.x) * 4;
    if (m >= a.mp) return;
    const int g = m / a.cols, c = m - g * a.cols;      // host: cols % 4 == 0, so a group of 4 columns never straddles
    x3_f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (k < a.K && g < a.ngroups) v = *reinterpret_cast<const x3_f32x4*>(a.src[g] + (long)k * a.ld + c);
    unsigned q[3][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) x3_split(v[e], q[0][e], q[1][e], q[2][e]);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        uint2 w;
        w.x = q[p][0] | (q[p][1] << 16);
        w.y = q[p][2] | (q[p][3] << 16);
        *reinterpret_cast<uint2*>(a.img + (((long)k * 3 + p) * a.mp + m) * 2) = w;
    }
}
// k-slow image of Y[K][ngroups * cols] (K = reduction index = source row): the layout the weight-gradient GEMM reads its two
// operands in when both are stored reduction-index-major in memory (dP[(b,l)][a], att[(b,l)][d]): no transposing pass.
extern "C" int rfn_x3_split_ks(const float* const* srcs_host, int ngroups, int64_t ld, int K, int cols, void* image,
                               void* stream) {
    if (!srcs_host || !image || K < 1 || cols < 1 || ngroups < 1 || ngroups > X3_MAX_GROUPS) return RFN_ERR_ARG;
    if ((cols & 3) || (ld & 3)) return RFN_ERR_SHAPE;
    X3SplitKsArgs a;
    for (int g = 0; g < ngroups; ++g) {
        if (!srcs_host[g] || (((uintptr_t)srcs_host[g]) & 15u)) return RFN_ERR_ARG;
        a.src[g] = srcs_host[g];
    }
    a.ld = ld;
    a.K = K;
    a.cols = cols;
    a.ngroups = ngroups;
    a.mp = (int)x3_rows_pad((long)ngroups * cols);
    a.kpad = (int)x3_k_pad(K);
    a.img = (char*)image;
    if ((a.mp / 4 + 255) / 256 > 65535) return RFN_ERR_SHAPE;
    hipLaunchKernelGGL(x3_split_ks_k, dim3(a.kpad, (a.mp / 4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}
