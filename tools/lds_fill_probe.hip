// How fast can ONE block per CU pull operands from L2 / Infinity Cache into LDS -- by LDS-DMA (global_load_lds_dwordx4, what
// the cell GEMMs use) or through registers (global_load_dwordx4 + ds_write_b128)?  The few-tile per-step products of the
// recurrences fit time = fixed + bytes_per_block / 28 GB/s (profiles/r05_chain.md): is that the DMA path or the chip?
//   hipcc --offload-arch=gfx950 -O3 tools/lds_fill_probe.hip -o /tmp/lds_fill_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// every block streams `per_block` bytes (a multiple of 16 KB) of its own slice of `src`, `reps` times, into a 64 KB LDS ring
template <int MODE, int DEPTH>   // MODE 0: LDS-DMA, DEPTH KB-pieces per wave in flight; MODE 1: register loads, DEPTH x 16 B per lane in flight
__global__ __launch_bounds__(256) void fill_k(const float* __restrict__ src, size_t per_block, int reps, float* out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const char* base = (const char*)src + (size_t)blockIdx.x * per_block;
    const size_t chunks = per_block / (4 * DEPTH * 1024);      // a chunk = DEPTH KB per wave, 4 waves
    float acc = 0.f;
    for (int r = 0; r < reps; ++r) {
        for (size_t c = 0; c < chunks; ++c) {
            const char* p = base + c * (4 * DEPTH * 1024) + wave * (DEPTH * 1024) + lane * 16;
            float* dst = sm + (wave * DEPTH * 256) % (16 * 1024);
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < DEPTH; ++j)
                    __builtin_amdgcn_global_load_lds((gbl_void*)(p + j * 1024), (lds_void*)(dst + j * 256), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                f4 v[DEPTH];
#pragma unroll
                for (int j = 0; j < DEPTH; ++j) v[j] = *reinterpret_cast<const f4*>(p + j * 1024);
#pragma unroll
                for (int j = 0; j < DEPTH; ++j) *reinterpret_cast<f4*>(dst + j * 256 + lane * 4) = v[j];
            }
            acc += sm[(tid * 4 + (int)c) & 4095];
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}
template <int MODE, int DEPTH>
static void run(const char* name, const float* src, int blocks, size_t per_block, int reps, float* out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)fill_k<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    float best = 1e30f;
    for (int t = 0; t < 4; ++t) {
        CK(hipEventRecord(e0));
        fill_k<MODE, DEPTH><<<blocks, 256, 96 * 1024>>>(src, per_block, reps, out);   // 96 KB: one block per CU
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (t && ms < best) best = ms;
    }
    const double bytes = (double)per_block * reps;
    printf("  %-34s %4d blocks x %5zu KB x %4d: %7.1f GB/s per block, %6.2f TB/s chip\n", name, blocks, per_block >> 10, reps,
           bytes / best / 1e6, bytes * blocks / best / 1e9);
}
int main() {
    const size_t total = (size_t)1 << 30;
    float *src, *out;
    CK(hipMalloc(&src, total)); CK(hipMalloc(&out, 64));
    CK(hipMemset(src, 0, total));
    for (int blocks : {64, 128, 256}) {
        for (int big = 0; big < 2; ++big) {
            const size_t per_block = big ? ((size_t)2 << 20) : ((size_t)64 << 10);     // 2 MB per block (past L2) / 64 KB (L2-resident)
            const int reps = big ? 8 : 256;
            printf("%d blocks, %s:\n", blocks, big ? "2 MB per block, streamed (Infinity Cache / HBM)" : "64 KB per block, re-read (L2)");
            run<0, 4>("LDS-DMA, 4 KB per wave in flight", src, blocks, per_block, reps, out);
            run<0, 16>("LDS-DMA, 16 KB per wave in flight", src, blocks, per_block, reps, out);
            run<1, 4>("registers, 4 x 16 B per lane", src, blocks, per_block, reps, out);
            run<1, 8>("registers, 8 x 16 B per lane", src, blocks, per_block, reps, out);
            run<1, 16>("registers, 16 x 16 B per lane", src, blocks, per_block, reps, out);
        }
    }
    return 0;
}
