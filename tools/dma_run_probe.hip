// What does the LENGTH OF THE CONTIGUOUS RUN a wave asks for per operand row cost an LDS-DMA stream?  The wave-private cell GEMM
// requests, per K step and wave, 16 B x 64 lanes = 1 KB per instruction laid out as (1024 / RUN) rows x RUN bytes, the rows a
// whole weight row apart (8 KB at K = 2048): RUN = 32 B (rounds 3-5: two quarter-line runs per row), 64 B (round 6: k-group
// pairs), 128 B (a full line: what a K step of 128 with four adjacent k-groups per wave would ask for), 1024 B (contiguous:
// tools/lds_fill_probe.hip).  One block of 4 waves per CU, 4 instructions per wave and step, DEPTH steps in flight, cold data.
//   hipcc --offload-arch=gfx950 -O3 tools/dma_run_probe.hip -o /tmp/dma_run_probe && /tmp/dma_run_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

#define ROW_BYTES 8192
#define NI 4

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int RUN, int DEPTH>
__global__ __launch_bounds__(256) void run_k(const char* __restrict__ src, int steps, float* out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int CPR = RUN / 16, RPI = 64 / CPR;            // 16-B chunks per run, rows per instruction
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int rows_per_wave = NI * RPI;
    const char* base = src + ((size_t)blockIdx.x * 4 + wave) * rows_per_wave * ROW_BYTES + (size_t)(lane / CPR) * ROW_BYTES + (lane % CPR) * 16;
    float acc = 0.f;
    for (int s = 0; s < steps; ++s) {
        float* dst = sm + ((s % DEPTH) * 4 + wave) * (NI * 256);
#pragma unroll
        for (int j = 0; j < NI; ++j)
            __builtin_amdgcn_global_load_lds((gbl_void*)(base + (size_t)j * RPI * ROW_BYTES + (size_t)s * RUN), (lds_void*)(dst + j * 256), 16, 0, 0);
        wait_vm<(DEPTH - 1) * NI>();
        acc += sm[(tid + s) & 1023];
    }
    wait_vm<0>();
    if (acc == 12345.678f) out[0] = acc;
}
__global__ void flush_k(const float* p, size_t n, float* out) {
    float a = 0.f;
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a += p[i];
    if (a == 1.2345f) out[1] = a;
}
template <int RUN, int DEPTH>
static void run(const char* src, const float* junk, size_t junk_n, int blocks, float* out) {
    const int steps = ROW_BYTES / RUN;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)run_k<RUN, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, DEPTH * 16 * 1024));
    float best = 1e30f;
    for (int t = 0; t < 3; ++t) {
        flush_k<<<2048, 256>>>(junk, junk_n, out);      // evict L2 / the memory-side cache
        CK(hipEventRecord(e0));
        run_k<RUN, DEPTH><<<blocks, 256, DEPTH * 16 * 1024>>>(src, steps, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double bytes = (double)steps * 4 * NI * 1024;      // per block
    printf("  run %4d B, %d steps in flight, %3d blocks: %6.1f us for %5.0f KB per block = %6.1f GB/s per block, %5.2f TB/s chip; %5.2f us per 16-KB step\n",
           RUN, DEPTH - 1, blocks, best * 1e3, bytes / 1024, bytes / best / 1e6, bytes * blocks / best / 1e9, best * 1e3 / steps);
}
int main() {
    const size_t total = (size_t)1 << 30, junk_n = (size_t)1 << 28;
    char* src; float *junk, *out;
    CK(hipMalloc(&src, total)); CK(hipMalloc(&junk, junk_n * 4)); CK(hipMalloc(&out, 64));
    CK(hipMemset(src, 0, total)); CK(hipMemset(junk, 0, junk_n * 4));
    for (int blocks : {64, 256}) {
        run<32, 3>(src, junk, junk_n, blocks, out);
        run<64, 3>(src, junk, junk_n, blocks, out);
        run<128, 3>(src, junk, junk_n, blocks, out);
        run<256, 3>(src, junk, junk_n, blocks, out);
        run<1024, 3>(src, junk, junk_n, blocks, out);
        run<64, 6>(src, junk, junk_n, blocks, out);
        run<128, 6>(src, junk, junk_n, blocks, out);
        run<1024, 6>(src, junk, junk_n, blocks, out);
    }
    return 0;
}
