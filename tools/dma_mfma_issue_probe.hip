// What does a K step of the wave-private cell GEMM cost when the SIMD has ONE wave (which issues its three 1-KiB LDS-DMA requests
// AND its eight MFMAs, in order) against TWO waves sharing the same work (each: two requests + four MFMAs on one accumulator)?
// Synthetic loop, operands L2-resident, one block per CU.  Prints shader-clock cycles per step and SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dma_mfma_issue_probe.hip -o /tmp/dma_mfma_issue_probe && /tmp/dma_mfma_issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) void gbl_void;
typedef __attribute__((address_space(3))) void lds_void;

// WAVES_PER_SIMD 1: 256 threads, each wave: NDMA = 3 requests, 8 MFMAs on two accumulators per step.
// WAVES_PER_SIMD 2: 512 threads, each wave: NDMA = 2 requests, 4 MFMAs on one accumulator per step.
template <int WPS, int NDMA, bool DMA_ON, bool MFMA_ON>
__global__ __launch_bounds__(256 * WPS) void probe_k(const float* __restrict__ src, float* __restrict__ out, int steps,
                                                     unsigned long long* __restrict__ cyc) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int SL = 3;
    float* my = smem + wave * (SL * NDMA * 256);
    const float* p = src + ((size_t)blockIdx.x * 8 + wave) * 2048 + lane * 4;      // 8 KiB per wave, 16 MiB in all: stays in L2
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float a = (float)lane * 1e-3f, b = 1.0f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    int slot = 0;
    for (int it = 0; it < steps; ++it) {
        if constexpr (DMA_ON) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");      // the step before last has landed
            const float* q = p + (size_t)(it & 1) * 1024 * 0;
#pragma unroll
            for (int j = 0; j < NDMA; ++j)
                __builtin_amdgcn_global_load_lds((gbl_void*)(q + ((j + it) & 7) * 256), (lds_void*)(my + (slot * NDMA + j) * 256), 16, 0, 0);
        }
        if constexpr (MFMA_ON) {
            const f32x4 fa = {a, b, a + 1.f, b + 1.f};      // operands in registers: the probe prices issue, not LDS latency
            if constexpr (WPS == 1) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c], b, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c], a, acc1, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c], b, acc0, 0, 0, 0);
            }
        }
        slot = (slot + 1 == SL) ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
    out[(size_t)blockIdx.x * blockDim.x + tid] = acc0[0] + acc1[1] + acc0[2] + acc1[3];
}

template <int WPS, int NDMA, bool DMA_ON, bool MFMA_ON>
static double run(const float* src, float* out, unsigned long long* cyc, int steps) {
    const int lds = 8 * 3 * 3 * 1024;
    hipFuncSetAttribute((const void*)probe_k<WPS, NDMA, DMA_ON, MFMA_ON>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((probe_k<WPS, NDMA, DMA_ON, MFMA_ON>), dim3(256), dim3(256 * WPS), lds, 0, src, out, steps, cyc);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0;
    for (int w = 0; w < 4 * WPS; ++w) m += (double)h[w];
    return m / (4 * WPS) / steps;
}

int main() {
    float *src, *out;
    unsigned long long* cyc;
    hipMalloc(&src, (size_t)256 * 8 * 2048 * 4 + 65536);
    hipMemset(src, 0, (size_t)256 * 8 * 2048 * 4 + 65536);
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 64);
    const int steps = 512;
    printf("cycles per step (a step = 3 KiB requested + 8 x v_mfma_f32_16x16x4_f32 per SIMD; the MFMAs alone are 256 cycles)\n");
    printf("one wave per SIMD  (3 requests + 8 MFMAs each):  requests only %5.0f | MFMAs only %5.0f | both %5.0f\n",
           run<1, 3, true, false>(src, out, cyc, steps), run<1, 3, false, true>(src, out, cyc, steps), run<1, 3, true, true>(src, out, cyc, steps));
    printf("two waves per SIMD (2 requests + 4 MFMAs each):  requests only %5.0f | MFMAs only %5.0f | both %5.0f\n",
           run<2, 2, true, false>(src, out, cyc, steps), run<2, 2, false, true>(src, out, cyc, steps), run<2, 2, true, true>(src, out, cyc, steps));
    printf("two waves per SIMD (3 requests + 8 MFMAs each = two tiles' worth):  both %5.0f per step and wave\n",
           run<2, 3, true, true>(src, out, cyc, steps));
    return 0;
}
