// Diagnostic (not part of the product): does the f32-input MFMA shape change what the chip sustains?  Same FLOP per
// SIMD cycle for v_mfma_f32_32x32x2_f32 (64-cycle issue) and v_mfma_f32_16x16x4_f32 (32-cycle issue); the clock the
// chip holds under each may differ (MI355X_MICROARCH.md, DVFS give-back item 7, measured there for bf16 only).
// Random operands, 64 accumulator registers per wave in both variants, 1-3 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_probe.hip -o /tmp/mfma_shape_probe && /tmp/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, bool LDSR = false>
__global__ __launch_bounds__(256) void mfma_loop(const float* __restrict__ in, float* __restrict__ out, int iters,
                                                 unsigned long long* __restrict__ clk) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = in[(gid * 5 + i) & 0xffff];
        b[i] = in[(gid * 7 + 3 + i) & 0xffff];
    }
    // LDSR: every iteration re-reads its operands from LDS (4 x ds_read_b128 per lane and 16 / 32 MFMAs: the LDS traffic per
    // MFMA of the product GEMM's main loop), so the loop runs under the power of LDS + matrix pipe instead of the pipe alone
    __shared__ __attribute__((aligned(16))) float lds[256 * 16];
    if constexpr (LDSR) {
        for (int i = 0; i < 16; ++i) lds[threadIdx.x * 16 + i] = in[(gid * 16 + i) & 0xffff];
        __syncthreads();
    }
    auto reload = [&](int it) {
        if constexpr (LDSR) {
            const f32x4* p = reinterpret_cast<const f32x4*>(lds + ((threadIdx.x + it) & 255) * 16);
            const f32x4 u = p[0], v = p[1];
            a[0] = u[0]; a[1] = u[1]; a[2] = u[2]; a[3] = u[3];
            b[0] = v[0]; b[1] = v[1]; b[2] = v[2]; b[3] = v[3];
            const f32x4 u2 = p[2], v2 = p[3];
            a[0] += u2[0] * 1e-6f; b[0] += v2[0] * 1e-6f;
        }
    };
    float s = 0.f;
    unsigned long long t0, r0, t1, r1;
    if constexpr (SHAPE == 32) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
            reload(it);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[i], acc[i], 0, 0, 0);
            a[0] += 1e-3f;
        }
        t1 = __builtin_amdgcn_s_memtime();
        r1 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i)
            for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
            reload(it);
#pragma unroll
            for (int u = 0; u < 2; ++u)      // 2 x 16 MFMAs of 16x16x4 = the FLOP of 4 x 4 MFMAs of 32x32x2
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(i >> 2) & 3], b[i & 3], acc[i], 0, 0, 0);
            a[0] += 1e-3f;
        }
        t1 = __builtin_amdgcn_s_memtime();
        r1 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < 16; ++i)
            for (int r = 0; r < 4; ++r) s += acc[i][r];
    }
    out[gid] = s;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = t1 - t0;
        clk[2 * blockIdx.x + 1] = r1 - r0;
    }
}

int main() {
    float *in, *out;
    unsigned long long* clk;
    const int maxblocks = 1024;
    hipMalloc(&in, 65536 * 4);
    hipMalloc(&out, (size_t)maxblocks * 256 * 4);
    hipMalloc(&clk, maxblocks * 16);
    float* h = (float*)malloc(65536 * 4);
    srand(1);
    for (int i = 0; i < 65536; ++i) h[i] = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    hipMemcpy(in, h, 65536 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int round = 0; round < 3; ++round)
        for (int blocks : {256, 512, 768})
            for (int shape : {32, 16, 1032, 1016}) {
                hipEventRecord(e0);
                if (shape == 32) hipLaunchKernelGGL(mfma_loop<32>, dim3(blocks), dim3(256), 0, 0, in, out, iters, clk);
                else if (shape == 16) hipLaunchKernelGGL(mfma_loop<16>, dim3(blocks), dim3(256), 0, 0, in, out, iters, clk);
                else if (shape == 1032) hipLaunchKernelGGL((mfma_loop<32, true>), dim3(blocks), dim3(256), 0, 0, in, out, iters, clk);
                else hipLaunchKernelGGL((mfma_loop<16, true>), dim3(blocks), dim3(256), 0, 0, in, out, iters, clk);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                unsigned long long hc[2];
                hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
                const double flops = (double)blocks * 4 * iters * 16.0 * 32 * 32 * 2 * 2;
                printf("%s  %d blocks (%d waves/SIMD): %.3f ms  %.1f TFLOP/s  in-kernel clock %.0f MHz\n",
                       shape == 32 ? "32x32x2" : shape == 16 ? "16x16x4" : shape == 1032 ? "32x32x2 + LDS reads" : "16x16x4 + LDS reads", blocks, blocks / 256, ms, flops / ms / 1e9,
                       (double)hc[0] / (double)hc[1] * 100.0);
            }
    return 0;
}
