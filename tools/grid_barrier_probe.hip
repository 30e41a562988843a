// Price of a grid-wide hand-off INSIDE one launch on MI355X, at the block counts one step of the stage-II / decoder
// recurrences needs (8 ... 256 co-resident blocks, one per CU), next to the dependent kernel boundary it would replace.
// VERDICT r04 item 2, step A.  hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_probe.hip -o /tmp/grid_barrier_probe
//
// Barrier forms (all: one monotonic counter family, lane 0 of every block arrives and polls with relaxed agent-scope
// (sc1) loads + s_sleep, every spin bounded -- a block that gives up raises `fail` and the run is reported as failed):
//   flat      one counter
//   xcd       one counter per XCD (blocks b, b+8, ... share an XCD; the id is read from HW_REG_XCC_ID), the last
//             arriver of an XCD adds to the top counter, polls it and publishes the XCD's generation word
// Payload forms (what a recurrence step hands over: every block writes 2 KB of a (blocks x 2 KB) vector, every block
// reads ALL of it after the barrier -- the h / z all-gather of a cell step; every word is checked):
//   none      barrier only
//   fence     plain stores, lane-0 agent release before the arrive, agent acquire after the poll, plain loads
//   sc1       write-through (sc1) stores drained by every wave before the block barrier, no fence, sc1 loads
// Baseline: the same number of dependent launches of a kernel that does the same 2 KB write / all-read, eager and
// replayed from a hipGraph.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

enum { PAY_NONE = 0, PAY_FENCE = 1, PAY_SC1 = 2 };
enum { BAR_FLAT = 0, BAR_XCD = 1 };
#define SPIN_LIMIT (1u << 20)
#define RING 16                 /* step buffers of the payload (the product writes every step to fresh addresses) */
#define LINE 32                 /* counters sit on 128-B lines of their own (index stride in unsigned) */

struct BarMem {
    unsigned* cnt;       // [0]: flat / top counter; [LINE * (1 + x)]: XCD x arrive counter; [LINE * (9 + x)]: XCD x generation
    unsigned* census;    // [x]: blocks on XCD x (filled by the kernel's first phase)
    int* fail;
};

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ bool spin_until(const unsigned* p, unsigned target, int* fail) {
    unsigned spins = 0;
    while ((int)(ld_sc1(p) - target) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_LIMIT) {
            *fail = 1;
            return false;
        }
    }
    return true;
}

// gen = 1, 2, ...: the number of this barrier.  All threads call it.
template <int BAR, bool FENCE>
__device__ __forceinline__ bool grid_barrier(const BarMem& m, unsigned gen, unsigned nblocks, unsigned xcc, unsigned n_xcd,
                                             unsigned my_xcd_blocks) {
    __shared__ int s_ok;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave: own stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        bool ok = true;
        if (FENCE) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (BAR == BAR_FLAT) {
            __hip_atomic_fetch_add(m.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = spin_until(m.cnt, gen * nblocks, m.fail);
        } else {
            unsigned* xc = m.cnt + LINE * (1 + xcc);
            unsigned* xg = m.cnt + LINE * (9 + xcc);
            const unsigned t = __hip_atomic_fetch_add(xc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t + 1 == gen * my_xcd_blocks) {   // last of this XCD
                __hip_atomic_fetch_add(m.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = spin_until(m.cnt, gen * n_xcd, m.fail);
                __hip_atomic_store(xg, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                ok = spin_until(xg, gen, m.fail);
            }
        }
        if (FENCE) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0;
}

__device__ __forceinline__ unsigned word_of(unsigned it, unsigned idx) { return (it * 2654435761u) ^ (idx * 40503u + 17u); }

// the 2 KB a block publishes in step `it` (256 threads x 8 B) and the check of all blocks' slices
template <bool SC1>
__device__ __forceinline__ void publish(unsigned* buf, unsigned it, unsigned nblocks) {
    unsigned* slot = buf + (size_t)(it % RING) * nblocks * 512;
    const unsigned idx = blockIdx.x * 512 + threadIdx.x * 2;
    u2 v = {word_of(it, idx), word_of(it, idx + 1)};
    if (SC1) {
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(slot, 0, 0x7fffffff, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b64(v, r, idx * 4, 0, 16);
    } else {
        *reinterpret_cast<u2*>(slot + idx) = v;
    }
}
template <bool SC1>
__device__ __forceinline__ unsigned consume(const unsigned* buf, unsigned it, unsigned nblocks) {
    const unsigned* slot = buf + (size_t)(it % RING) * nblocks * 512;
    unsigned bad = 0;
    const unsigned n4 = nblocks * 128;   // 16-B words
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(slot), 0, 0x7fffffff, 0x00020000);
    for (unsigned i0 = threadIdx.x; i0 < n4; i0 += 4 * 256) {
        u4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned i = i0 + j * 256 < n4 ? i0 + j * 256 : n4 - 1;
            if (SC1) v[j] = __builtin_amdgcn_raw_buffer_load_b128(r, i * 16, 0, 16);
            else v[j] = *reinterpret_cast<const u4*>(slot + i * 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned i = i0 + j * 256 < n4 ? i0 + j * 256 : n4 - 1;
#pragma unroll
            for (int e = 0; e < 4; ++e) bad += (v[j][e] != word_of(it, i * 4 + e));
        }
    }
    return bad;
}

template <int BAR, int PAY>
__global__ __launch_bounds__(256) void persistent_k(BarMem m, unsigned* buf, unsigned iters, unsigned* bad_out, unsigned gen0) {
    const unsigned nblocks = gridDim.x;
    const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) & 7u;
    __shared__ unsigned s_nx, s_mine;
    // census: who sits where (one flat barrier; its generations are gen0 + 1)
    if (threadIdx.x == 0) __hip_atomic_fetch_add(m.census + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!grid_barrier<BAR_FLAT, false>(m, 1, nblocks, 0, 0, 0)) return;
    if (threadIdx.x == 0) {
        unsigned nx = 0;
        for (int x = 0; x < 8; ++x) nx += ld_sc1(m.census + x) != 0;
        s_nx = nx;
        s_mine = ld_sc1(m.census + xcc);
    }
    __syncthreads();
    const unsigned n_xcd = s_nx, mine = s_mine;
    (void)gen0;
    unsigned bad = 0;
    // the flat counter has seen one round (census); barrier numbering of the timed loop continues from it for BAR_FLAT and
    // starts at 1 for the XCD family, whose top counter is the flat counter: keep the families apart by giving the XCD
    // form its own top word
    BarMem mm = m;
    if (BAR == BAR_XCD) mm.cnt = m.cnt + LINE * 20;
    for (unsigned it = 0; it < iters; ++it) {
        if (PAY != PAY_NONE) publish<PAY == PAY_SC1>(buf, it, nblocks);
        const unsigned gen = (BAR == BAR_FLAT) ? it + 2 : it + 1;
        if (!grid_barrier<BAR, PAY == PAY_FENCE>(mm, gen, nblocks, xcc, n_xcd, mine)) break;   // a spin gave up: leave
        if (PAY != PAY_NONE) bad += consume<PAY == PAY_SC1>(buf, it, nblocks);
    }
    if (PAY != PAY_NONE) {
        for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
        if ((threadIdx.x & 63) == 0 && bad) atomicAdd(bad_out, bad);
    }
}

// the launch-per-step form of the same exchange
__global__ __launch_bounds__(256) void step_k(unsigned* buf, unsigned it, unsigned* bad_out, int pay) {
    const unsigned nblocks = gridDim.x;
    unsigned bad = 0;
    if (pay) {
        if (it > 0) bad = consume<false>(buf, it - 1, nblocks);
        publish<false>(buf, it, nblocks);
        for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
        if ((threadIdx.x & 63) == 0 && bad) atomicAdd(bad_out, bad);
    }
}
__global__ void empty_k() {}

template <int BAR, int PAY>
static double run_persistent(int nblocks, unsigned iters, BarMem m, unsigned* buf, unsigned* bad, hipStream_t st, int reps,
                             unsigned* bad_h, int* fail_h) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double best = 1e30;
    for (int r = 0; r < reps + 1; ++r) {
        CK(hipMemsetAsync(m.cnt, 0, 64 * LINE * sizeof(unsigned), st));
        CK(hipMemsetAsync(m.census, 0, 8 * sizeof(unsigned), st));
        CK(hipMemsetAsync(m.fail, 0, sizeof(int), st));
        CK(hipMemsetAsync(bad, 0, sizeof(unsigned), st));
        empty_k<<<1, 64, 0, st>>>();
        CK(hipEventRecord(e0, st));
        persistent_k<BAR, PAY><<<nblocks, 256, 0, st>>>(m, buf, iters, bad, 0);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
        CK(hipMemcpy(bad_h, bad, sizeof(unsigned), hipMemcpyDeviceToHost));
        CK(hipMemcpy(fail_h, m.fail, sizeof(int), hipMemcpyDeviceToHost));
        if (*fail_h) break;
    }
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return best * 1e3;   // us
}

int main(int argc, char** argv) {
    unsigned iters = 2000;
    if (argc > 1) iters = (unsigned)atoi(argv[1]);
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("# device %s, %d CUs; %u steps per launch; times in us per step (best of 3 launches), total launch minus the\n"
           "# same launch with 0 steps\n", prop.gcnArchName, prop.multiProcessorCount, iters);
    hipStream_t st;
    CK(hipStreamCreate(&st));
    BarMem m;
    CK(hipMalloc(&m.cnt, 64 * LINE * sizeof(unsigned)));
    CK(hipMalloc(&m.census, 8 * sizeof(unsigned)));
    CK(hipMalloc(&m.fail, sizeof(int)));
    unsigned *buf, *bad;
    CK(hipMalloc(&buf, (size_t)RING * 256 * 512 * sizeof(unsigned)));
    CK(hipMalloc(&bad, sizeof(unsigned)));
    CK(hipMemset(buf, 0, (size_t)RING * 256 * 512 * sizeof(unsigned)));
    const int counts[] = {8, 16, 32, 64, 128, 256};
    printf("%-7s | %-31s | %-31s | %-31s | %s\n", "blocks", "barrier only: flat / xcd", "fence payload: flat / xcd",
           "sc1 payload: flat / xcd", "launch per step: trivial eager / graph, payload eager / graph");
    for (int nb : counts) {
        if (nb > prop.multiProcessorCount) continue;
        unsigned bad_h = 0, bad_any = 0;
        int fail_h = 0, fail_any = 0;
        double t[6], z[6];
#define RUN(i, BAR, PAY)                                                                          \
        z[i] = run_persistent<BAR, PAY>(nb, 0, m, buf, bad, st, 3, &bad_h, &fail_h);              \
        t[i] = run_persistent<BAR, PAY>(nb, iters, m, buf, bad, st, 3, &bad_h, &fail_h);          \
        bad_any += bad_h;                                                                         \
        fail_any += fail_h;
        RUN(0, BAR_FLAT, PAY_NONE)
        RUN(1, BAR_XCD, PAY_NONE)
        RUN(2, BAR_FLAT, PAY_FENCE)
        RUN(3, BAR_XCD, PAY_FENCE)
        RUN(4, BAR_FLAT, PAY_SC1)
        RUN(5, BAR_XCD, PAY_SC1)
        // launch per step
        double lt[4];
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        const unsigned L = 500;
        for (int pay = 0; pay < 2; ++pay) {
            CK(hipMemsetAsync(bad, 0, sizeof(unsigned), st));
            double best = 1e30;
            for (int r = 0; r < 3; ++r) {
                CK(hipEventRecord(e0, st));
                for (unsigned it = 0; it < L; ++it) step_k<<<nb, 256, 0, st>>>(buf, it, bad, pay);
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            lt[2 * pay] = best * 1e3 / L;
            hipGraph_t g;
            hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (unsigned it = 0; it < L; ++it) step_k<<<nb, 256, 0, st>>>(buf, it, bad, pay);
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            best = 1e30;
            for (int r = 0; r < 4; ++r) {
                CK(hipEventRecord(e0, st));
                CK(hipGraphLaunch(ge, st));
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (r > 0 && ms < best) best = ms;
            }
            lt[2 * pay + 1] = best * 1e3 / L;
            CK(hipGraphExecDestroy(ge));
            CK(hipGraphDestroy(g));
            CK(hipMemcpy(&bad_h, bad, sizeof(unsigned), hipMemcpyDeviceToHost));
            bad_any += bad_h;
        }
        CK(hipEventDestroy(e0));
        CK(hipEventDestroy(e1));
        printf("%-7d | %6.2f / %6.2f %16s | %6.2f / %6.2f %16s | %6.2f / %6.2f %16s | %5.2f / %5.2f, %5.2f / %5.2f   %s%s\n", nb,
               (t[0] - z[0]) / iters, (t[1] - z[1]) / iters, "", (t[2] - z[2]) / iters, (t[3] - z[3]) / iters, "",
               (t[4] - z[4]) / iters, (t[5] - z[5]) / iters, "", lt[0], lt[1], lt[2], lt[3], bad_any ? "BAD WORDS " : "",
               fail_any ? "SPIN LIMIT HIT" : "");
        fflush(stdout);
    }
    return 0;
}
