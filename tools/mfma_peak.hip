// Diagnostic (not part of the product): sustained v_mfma_f32_32x32x2_f32 rate and in-kernel clock on this
// device, with random operands, for kernels of GEMM-like duration.  Build: hipcc --offload-arch=gfx950 -O3
// tools/mfma_peak.hip -o tools/mfma_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(const float* __restrict__ in, float* __restrict__ out, int iters,
                                                 unsigned long long* __restrict__ clk) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int gid = blockIdx.x * 256 + threadIdx.x;
    float a = in[gid & 0xffff], b = in[(gid * 7 + 3) & 0xffff];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        a += 1e-3f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[gid] = s;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = t1 - t0;
        clk[2 * blockIdx.x + 1] = r1 - r0;
    }
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 512;  // 2 blocks x 4 waves per CU
    const int iters = argc > 2 ? atoi(argv[2]) : 20000;
    float *in, *out;
    unsigned long long* clk;
    hipMalloc(&in, 65536 * 4);
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipMalloc(&clk, blocks * 16);
    float* h = (float*)malloc(65536 * 4);
    srand(1);
    for (int i = 0; i < 65536; ++i) h[i] = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    hipMemcpy(in, h, 65536 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(256), 0, 0, in, out, iters, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long hc[2];
        hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
        const double flops = (double)blocks * 4 * iters * 16.0 * 32 * 32 * 2 * 2;
        printf("blocks %d iters %d: %.3f ms  %.1f TFLOP/s  in-kernel clock %.0f MHz (cycles %llu, realtime ticks %llu)\n",
               blocks, iters, ms, flops / ms / 1e9, (double)hc[0] / (double)hc[1] * 100.0, hc[0], hc[1]);
    }
    return 0;
}
