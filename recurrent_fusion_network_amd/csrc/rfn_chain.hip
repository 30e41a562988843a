// Persistent recurrence kernels: ALL steps of a stage-II or decoder recurrence in ONE launch.
//
// Replaces, for the loops misc/RecurrentFusionModel.py:241-244 (stage II: T2 x LSTMSoftMultiAttentionFeatArrayNoInputCore.py:41-73)
// and :259-279 (decoder: 17 x misc/LSTMSoftAttentionCore.py:60-102) and for their backward sweeps, the three dependent launches
// per step and direction of rounds 3-4
//     forward :  K1 = every product of the recurrent h   ->  small attention  ->  K3 = z products + LSTM update
//     backward:  Kb1 = dh_rec and every dz               ->  attention bwd     ->  Kb2 = dh_rec += dhp . W_h + LSTM backward below
// by one launch whose blocks walk the same three phases step after step with a grid-wide barrier between phases.  A phase runs
// the SAME device bodies as the per-step launches (rfn_cellgemm_body.h, rfn_attn_small_body.h) on the SAME tiles with the same
// k order, so every result is bit-identical to the three-launch chain (tests/test_chain_gpu.py); only who waits for whom changes:
//   * a dependent launch costs its dispatch, ramp and tail (the per-step launches of the chains measure 7-13 us for 1-2 us of
//     work, profiles/r04_c2_step_launches.txt); the hand-off inside a launch costs one barrier -- measured on MI355X at the
//     block counts a step needs (tools/grid_barrier_probe.hip, profiles/r05_grid_barrier.md): 0.8-1.4 us for 8-64 blocks on one
//     counter, 1.8-2.2 us for 128-256 blocks with per-XCD counters;
//   * what blocks hand to each other inside the launch moves through write-through (sc1) stores and sc1 loads (rfn_xb.h): a
//     CU's L1 is never refreshed by other CUs' stores and the per-XCD L2s are not coherent with each other, and an agent-scope
//     release / acquire pair (L2 write-back + invalidate) per phase would cost more than the launch boundary it replaces.
// Hand-off recipe (MI355X guide, inter-workgroup visibility): every storing wave drains its stores (s_waitcnt vmcnt(0)), the
// block's barrier, ONE lane arrives on the counter with an agent-scope atomic add and polls it with relaxed agent-scope (sc1)
// loads + s_sleep, the block's barrier again, then sc1 loads of the handed-off bytes.  One block per CU (the launch asks for
// more than half a CU's LDS), grid <= CUs, so every block is resident; every spin is bounded and traps if it gives up.
//
// The per-step descriptors (the launch arguments of the three per-step kernels) are not uploaded: step s = base + s * delta
// word by word (activations sit at constant strides across steps), plus a short list of per-step overrides for words that do
// not (the stage-II weights of step t are separate parameter tensors); the host checks that the reconstruction is exact for
// every word of every step before it takes this path, and falls back to the per-step launches otherwise.
#include <string.h>

#include <vector>

#include "rfn_internal.h"

#define CH_THREADS 256
#define CH_LINE 32                 /* barrier counters sit on 128-B lines of their own (stride in uint32) */
#define CH_SPIN_LIMIT (1u << 23)
#define CH_NOV 40                  /* words of a step's descriptors that may carry per-step overrides */
#define CH_MAXSTEPS_OV 12          /* ... for chains of at most this many steps */
#define CH_FLAT_MAX_BLOCKS 96      /* one counter up to here, per-XCD counters above (profiles/r05_grid_barrier.md) */
#define CH_SLOTS 8                  /* ring slots of a tile: with K = 512 every K step of a tile is in flight at once */
#ifdef RFN_CHAIN_TIMING
#define CH_LDS_BYTES (CH_SLOTS * (32 + CG_BN) * 64 * 4 + 64)   /* + the in-tile stamps */
#else
#define CH_LDS_BYTES (CH_SLOTS * (32 + CG_BN) * 64 * 4)   /* 128 KB, > half of a CU's 160 KB: one block per CU */
#endif

struct ChainDesc {                 // the launch arguments of one step's three phases
    CgArgs g0;
    AttnSmallArgs at;
    CgArgs g2;
    int n0, nb, ng, n2;            // tiles of phase 0, (rows, encoders) of the attention, tiles of phase 2
};
static_assert(sizeof(ChainDesc) % 8 == 0, "descriptors are rebuilt 8 bytes at a time");
#define CH_NW (sizeof(ChainDesc) / 8)

// Diagnostic build only (make EXTRA=-DRFN_CHAIN_TIMING; tools/chain_timing.py): every block stamps the 100 MHz clock at the
// phase boundaries of every step into a buffer the tool registers through rfn_debug_chain_timing().  The product build has
// neither the stamps nor the symbol.
#ifdef RFN_CHAIN_TIMING
#define CH_STAMPS 16
static uint64_t* g_chain_timing = nullptr;
static size_t g_chain_timing_words = 0;
static int g_chain_pick = 0, g_chain_seen = 0, g_chain_last_G = 0, g_chain_last_steps = 0;
extern "C" void rfn_debug_chain_last(int* G, int* nsteps) {   // geometry of the launch that was stamped
    *G = g_chain_last_G;
    *nsteps = g_chain_last_steps;
}
// stamps go to `buf` for the pick-th persistent launch after this call (0 = the next one)
extern "C" void rfn_debug_chain_timing(void* buf, size_t bytes, int pick) {
    g_chain_timing = (uint64_t*)buf;
    g_chain_timing_words = bytes / 8;
    g_chain_pick = pick;
    g_chain_seen = 0;
}
#define CH_STAMP(i)                                                                                          \
    do {                                                                                                     \
        if (ca.timing && tid == 0) ca.timing[((size_t)blockIdx.x * nsteps + s) * CH_STAMPS + (i)] = wall_clock64(); \
    } while (0)
#else
#define CH_STAMP(i)
#endif

struct ChainArgs {
    int nsteps, nov, xcd, pad;
    uint32_t* bar;                 // zeroed by the host: [0] flat / census counter, then the per-XCD family (chain_barrier)
    uint64_t* timing;              // RFN_CHAIN_TIMING builds: [block][step][CH_STAMPS] clock stamps, else NULL
    int64_t base[CH_NW];
    int32_t delta[CH_NW];
    uint16_t ov_word[CH_NOV];
    int64_t ov_val[CH_MAXSTEPS_OV][CH_NOV];
};

__device__ __forceinline__ uint32_t ch_ld(const uint32_t* p) {
    return __hip_atomic_load((__attribute__((address_space(1))) uint32_t*)(uintptr_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ch_spin(const uint32_t* p, uint32_t target) {
    uint32_t spins = 0;
    while ((int32_t)(ch_ld(p) - target) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > CH_SPIN_LIMIT) __builtin_trap();   // a block of the grid is not running: fail loudly, never hang
    }
}
// Grid barrier number `gen` (1, 2, ...) in two halves, so that a block can put loads that do not depend on the other blocks
// between them.  arrive: every wave drains its stores, the block meets, ONE lane adds to the counter.  wait: that lane polls
// until every block has arrived, the block meets again; from then on every store any block issued before ITS arrive is visible
// to sc1 loads.   flat: one counter.   xcd: blocks of one XCD share a counter; its last arriver adds to the top counter, waits
// there for all XCDs and publishes the XCD's generation word, which the others of that XCD poll (an XCD-local line).
struct ChainBar {
    uint32_t* bar;
    uint32_t nblocks, xcc, n_xcd, mine;
    int xcd;
    uint32_t ticket;   // lane 0 of the block: what its add returned
};
__device__ __forceinline__ void chain_arrive(ChainBar& cb, uint32_t gen) {
    typedef __attribute__((address_space(1))) uint32_t gu32;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave: its stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t* c = cb.xcd ? cb.bar + CH_LINE * (2 + cb.xcc) : cb.bar;
        cb.ticket = __hip_atomic_fetch_add((gu32*)(uintptr_t)c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    (void)gen;
}
__device__ __forceinline__ void chain_wait(ChainBar& cb, uint32_t gen) {
    typedef __attribute__((address_space(1))) uint32_t gu32;
    if (threadIdx.x == 0) {
        if (!cb.xcd) {
            ch_spin(cb.bar, gen * cb.nblocks);
        } else {
            uint32_t* top = cb.bar + CH_LINE * 1;
            uint32_t* xg = cb.bar + CH_LINE * (10 + cb.xcc);
            if (cb.ticket + 1 == gen * cb.mine) {
                __hip_atomic_fetch_add((gu32*)(uintptr_t)top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ch_spin(top, gen * cb.n_xcd);
                __hip_atomic_store((gu32*)(uintptr_t)xg, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                ch_spin(xg, gen);
            }
        }
    }
    __syncthreads();
}

template <bool FWD>
__global__ __launch_bounds__(CH_THREADS) void chain_k(const ChainArgs ca) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ __attribute__((aligned(16))) ChainDesc Dbuf[2];   // this step's descriptors and the next one's (built under a barrier)
    __shared__ uint32_t s_census[2];
    const int tid = threadIdx.x, G = gridDim.x;
    const int nsteps = ca.nsteps;
    ChainBar cb;
    cb.bar = ca.bar;
    cb.nblocks = G; cb.xcc = 0; cb.n_xcd = 1; cb.mine = G; cb.xcd = 0; cb.ticket = 0;
    if (ca.xcd) {   // who shares an XCD with this block (placement is the dispatcher's business: count, do not assume)
        typedef __attribute__((address_space(1))) uint32_t gu32;
        const uint32_t xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) & 7u;   // HW_REG_XCC_ID
        uint32_t* census = ca.bar + CH_LINE * 18;
        if (tid == 0) __hip_atomic_fetch_add((gu32*)(uintptr_t)(census + xcc), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        chain_arrive(cb, 1);
        chain_wait(cb, 1);          // flat, on bar[0]
        if (tid == 0) {
            uint32_t nx = 0;
            for (int x = 0; x < 8; ++x) nx += ch_ld(census + x) != 0;
            s_census[0] = nx;
            s_census[1] = ch_ld(census + xcc);
        }
        __syncthreads();
        cb.xcc = xcc;
        cb.n_xcd = s_census[0];
        cb.mine = s_census[1];
        cb.xcd = 1;
    }
    auto build = [&](int s) {   // descriptors of step s: base + s * delta, then the per-step overrides
        int64_t* dw = reinterpret_cast<int64_t*>(&Dbuf[s & 1]);
        for (int i = tid; i < (int)CH_NW; i += CH_THREADS) dw[i] = ca.base[i] + (int64_t)s * ca.delta[i];
        __syncthreads();
        if (tid < ca.nov) dw[ca.ov_word[tid]] = ca.ov_val[s][tid];
        __syncthreads();
    };
    uint32_t gen = 0;
    build(0);
    for (int s = 0; s < nsteps; ++s) {
        const ChainDesc& D = Dbuf[s & 1];
        CH_STAMP(0);
        const int n0 = xb_uni(D.n0), nb = xb_uni(D.nb), n1 = nb * xb_uni(D.ng), n2 = xb_uni(D.n2);
        // Every phase: the block's FIRST task carries the wait of the barrier that ended the previous phase inside it, behind
        // the loads that do not depend on the other blocks; a block without a task in the phase just waits.
        bool waited = (s == 0);                        // nothing precedes phase 0 of step 0
        auto wait_prev = [&]() {
            if (!waited) chain_wait(cb, gen);
            waited = true;
        };
        CH_STAMP(1);
        // ---- phase 0: the products of the recurrent state ---------------------------------------------------------------
        for (int vb = blockIdx.x; vb < n0; vb += G) {
            if (vb != (int)blockIdx.x) __syncthreads();   // the ring of this block's previous tile is free (arrive did it for the first)
            cg_tile<32, 64, 4, FWD, CG_EPI_STORE, true, true, false>(D.g0, vb, smem, wait_prev);
        }
        wait_prev();
#ifdef RFN_CHAIN_TIMING
        if (ca.timing && tid == 0)
            for (int i = 0; i < 4; ++i) ca.timing[((size_t)blockIdx.x * nsteps + s) * CH_STAMPS + 8 + i] = reinterpret_cast<uint64_t*>(smem + 8 * (32 + CG_BN) * 64)[i];
#endif
        CH_STAMP(2);
        chain_arrive(cb, ++gen);
        waited = false;
        CH_STAMP(3);
        // ---- phase 1: the small attention of every (row, encoder) ----------------------------------------------------------
        for (int vb = blockIdx.x; vb < n1; vb += G) {
            if (vb != (int)blockIdx.x) __syncthreads();
            const int g = vb / nb, b = vb - g * nb;
            if constexpr (FWD) {
                if (attn_small_fwd_xp_ok(D.at, g)) {
                    attn_small_fwd_body_xp(D.at, b, g, smem, wait_prev);
                } else {
                    wait_prev();
                    attn_small_fwd_body<true>(D.at, b, g, smem);
                }
            } else {
                wait_prev();
                attn_small_bwd_body<true, true>(D.at, b, g, smem);
            }
        }
        wait_prev();
        CH_STAMP(4);
        chain_arrive(cb, ++gen);
        waited = false;
        CH_STAMP(5);
        // ---- phase 2: the products of the contexts + the LSTM update (forward) / of d hproj + the LSTM backward below ----------
        for (int vb = blockIdx.x; vb < n2; vb += G) {
            if (vb != (int)blockIdx.x) __syncthreads();
            cg_tile<32, 64, 4, FWD, FWD ? CG_EPI_LSTM : CG_EPI_LSTM_BWD, true, true, false>(D.g2, vb, smem, wait_prev);
        }
        wait_prev();
#ifdef RFN_CHAIN_TIMING
        if (ca.timing && tid == 0)
            for (int i = 0; i < 4; ++i) ca.timing[((size_t)blockIdx.x * nsteps + s) * CH_STAMPS + 12 + i] = reinterpret_cast<uint64_t*>(smem + 8 * (32 + CG_BN) * 64)[i];
#endif
        CH_STAMP(6);
        if (s + 1 < nsteps) {
            chain_arrive(cb, ++gen);
            build(s + 1);              // under the barrier: nothing in it depends on the other blocks
        }
        CH_STAMP(7);
    }
}

// The barrier counters are zeroed by a KERNEL of the same stream, not by hipMemsetAsync: replayed from a captured hipGraph the
// memset node left the persistent kernel looking at the previous replay's counts (a per-XCD "last arriver" test never came
// true and the launch ran into its spin limit), the kernel boundary does not.
__global__ __launch_bounds__(CH_THREADS) void chain_zero_k(uint32_t* bar) {
    for (int i = threadIdx.x; i < RFN_CHAIN_BAR_WORDS; i += CH_THREADS) bar[i] = 0u;
}

// ---- host ----------------------------------------------------------------------------------------------------------------
static int chain_as_launches(const ChainStep* steps, int nsteps, void* stream) {
    for (int s = 0; s < nsteps; ++s) {
        RFN_TRY(rfn_cg_launch(steps[s].g0, stream));
        RFN_TRY(rfn_attn_small_launch(steps[s].at, stream));
        RFN_TRY(rfn_cg_launch(steps[s].g2, stream));
    }
    return RFN_OK;
}

struct ChainDev {
    bool set[16] = {};
    int cus[16] = {};
};

// Whether the chain can run as one persistent launch, and its arguments if so.
static bool chain_plan(const ChainStep* steps, int nsteps, ChainArgs& ca, bool& fwd, int& most) {
    if (nsteps < 2) return false;
    fwd = steps[0].g0.bkf;
    most = 0;
    std::vector<ChainDesc> D((size_t)nsteps);
    for (int s = 0; s < nsteps; ++s) {
        ChainStep st = steps[s];
        // the bodies the kernel is built from: 32-row tiles, K step 64, 4 K-waves; vectorised attention backward.  A launch the
        // host would put on 16-row tiles (few tiles) computes the same bits on 32-row ones.
        if (st.g0.variant >= 4 && rfn_cg_replan32(&st.g0) != RFN_OK) return false;
        if (st.g2.variant >= 4 && rfn_cg_replan32(&st.g2) != RFN_OK) return false;
        if (st.g0.variant != 3 || st.g2.variant != 3) return false;
        if (st.g0.epi != CG_EPI_STORE || st.g0.bkf != fwd || st.g2.bkf != fwd) return false;
        if (st.g2.epi != (fwd ? CG_EPI_LSTM : CG_EPI_LSTM_BWD)) return false;
        if ((st.at.backward != 0) == fwd || (!fwd && !st.at.vec)) return false;
        if (st.at.lds > 3 * (32 + CG_BN) * 64 * sizeof(float)) return false;    // the attention scratch shares the GEMM ring
        if (st.g0.a.slots > CH_SLOTS || st.g2.a.slots > CH_SLOTS) return false;
        // 16-B sc1 accesses go through buffer descriptors with 32-bit byte offsets
        const AttnSmallArgs& a = st.at.a;
        if (!fwd && ((double)a.L * a.xsl * 4 >= 2.0e9 || (double)a.L * a.dpsl * 4 >= 2.0e9)) return false;
        for (int o = 0; o < st.g0.a.nout; ++o)
            if ((double)st.g0.a.M * st.g0.a.out[o].ldc * 4 >= 2.0e9) return false;
        memset(&D[s], 0, sizeof(ChainDesc));
        D[s].g0 = st.g0.a;
        D[s].at = st.at.a;
        D[s].g2 = st.g2.a;
        D[s].g0.slots = D[s].g2.slots = CH_SLOTS;      // one block per CU: the ring takes what the launch form leaves to co-residents
        D[s].n0 = st.g0.blocks;
        D[s].nb = st.at.B;
        D[s].ng = st.at.ngroups;
        D[s].n2 = st.g2.blocks;
        const int m = std::max(std::max(D[s].n0, D[s].n2), D[s].nb * D[s].ng);
        most = std::max(most, m);
    }
    memset(&ca, 0, sizeof(ca));
    ca.nsteps = nsteps;
    const int64_t* w0 = reinterpret_cast<const int64_t*>(&D[0]);
    const int64_t* w1 = reinterpret_cast<const int64_t*>(&D[1]);
    int nov = 0;
    for (size_t i = 0; i < CH_NW; ++i) {
        const int64_t dl = (int64_t)((uint64_t)w1[i] - (uint64_t)w0[i]);
        bool linear = dl >= INT32_MIN && dl <= INT32_MAX;
        for (int s = 2; s < nsteps && linear; ++s)
            linear = reinterpret_cast<const int64_t*>(&D[s])[i] == (int64_t)((uint64_t)w0[i] + (uint64_t)s * (uint64_t)dl);
        ca.base[i] = w0[i];
        if (linear) {
            ca.delta[i] = (int32_t)dl;
            continue;
        }
        if (nov >= CH_NOV || nsteps > CH_MAXSTEPS_OV) return false;
        ca.delta[i] = 0;
        ca.ov_word[nov] = (uint16_t)i;
        for (int s = 0; s < nsteps; ++s) ca.ov_val[s][nov] = reinterpret_cast<const int64_t*>(&D[s])[i];
        ++nov;
    }
    ca.nov = nov;
    return true;
}

int rfn_chain_run(const ChainStep* steps, int nsteps, int persist, uint32_t* bar, void* stream) {
    if (nsteps < 1) return RFN_OK;
    if (!steps) return RFN_ERR_ARG;
    static ChainDev dev_state;
    ChainArgs ca;
    bool fwd = true;
    int most = 0;
    if (!persist || !bar || !chain_plan(steps, nsteps, ca, fwd, most)) return chain_as_launches(steps, nsteps, stream);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return RFN_ERR_LAUNCH;
    const int di = dev & 15;
    if (!dev_state.set[di]) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) return RFN_ERR_LAUNCH;
        if (hipFuncSetAttribute((const void*)chain_k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void*)chain_k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS_BYTES) != hipSuccess)
            return RFN_ERR_LAUNCH;
        // every block must be resident at once: one per CU by construction (LDS), so the grid is bounded by the CU count --
        // provided the kernel fits a CU at all
        int per_cu_f = 0, per_cu_b = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_f, chain_k<true>, CH_THREADS, CH_LDS_BYTES) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_b, chain_k<false>, CH_THREADS, CH_LDS_BYTES) != hipSuccess)
            return RFN_ERR_LAUNCH;
        dev_state.cus[di] = (per_cu_f >= 1 && per_cu_b >= 1) ? cus : 0;
        dev_state.set[di] = true;
    }
    if (dev_state.cus[di] < 8) return chain_as_launches(steps, nsteps, stream);
    const int G = std::min(most, dev_state.cus[di]);
    ca.xcd = G > CH_FLAT_MAX_BLOCKS ? 1 : 0;
    ca.bar = bar;
#ifdef RFN_CHAIN_TIMING
    ca.timing = (g_chain_timing && g_chain_seen++ == g_chain_pick && (size_t)G * nsteps * CH_STAMPS <= g_chain_timing_words) ? g_chain_timing : nullptr;
    if (ca.timing) {
        g_chain_last_G = G;
        g_chain_last_steps = nsteps;
    }
#endif
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(chain_zero_k, dim3(1), dim3(CH_THREADS), 0, st, bar);
    if (fwd) hipLaunchKernelGGL(chain_k<true>, dim3(G), dim3(CH_THREADS), CH_LDS_BYTES, st, ca);
    else hipLaunchKernelGGL(chain_k<false>, dim3(G), dim3(CH_THREADS), CH_LDS_BYTES, st, ca);
    RFN_CHECK_LAUNCH();
    return RFN_OK;
}
