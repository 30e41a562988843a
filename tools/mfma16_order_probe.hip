// Is v_mfma_f32_16x16x4_f32 the same k-ordered fp32 fma chain per output element as v_mfma_f32_32x32x2_f32 (and as a
// scalar fmaf loop)?  If so, a 32x32 output tile of the per-step cell GEMMs can be cut into 16x16 tiles on four times as
// many CUs without changing a bit of the result.  hipcc --offload-arch=gfx950 -O3 tools/mfma16_order_probe.hip -o /tmp/mfma16
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void k16(const float* A, const float* B, float* C, int K) {   // A [16][K], B [16][K] (C = A B^T), one wave
    const int l = threadIdx.x, i = l & 15, kq = l >> 4;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * K + k0 + kq], B[i * K + k0 + kq], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[(4 * kq + r) * 16 + i] = acc[r];
}
__global__ void k32(const float* A, const float* B, float* C, int K) {   // A [32][K], B [32][K]
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    f16v acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k0 + h], B[i * K + k0 + h], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}
int main() {
    const int K = 2048;
    float *A = (float*)malloc(32 * K * 4), *B = (float*)malloc(32 * K * 4), *C16 = (float*)malloc(256 * 4), *C32 = (float*)malloc(1024 * 4);
    float *dA, *dB, *dC16, *dC32;
    hipMalloc(&dA, 32 * K * 4); hipMalloc(&dB, 32 * K * 4); hipMalloc(&dC16, 256 * 4); hipMalloc(&dC32, 1024 * 4);
    srand(5);
    for (int scale = 0; scale < 3; ++scale) {
        for (int i = 0; i < 32 * K; ++i) {
            float a = (float)rand() / RAND_MAX - 0.5f, b = (float)rand() / RAND_MAX - 0.5f;
            if (scale == 1) { a *= expf(8.f * ((float)rand() / RAND_MAX - 0.5f)); b *= expf(8.f * ((float)rand() / RAND_MAX - 0.5f)); }
            if (scale == 2) { a = fabsf(a); b = fabsf(b); }
            A[i] = a; B[i] = b;
        }
        hipMemcpy(dA, A, 32 * K * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B, 32 * K * 4, hipMemcpyHostToDevice);
        k16<<<1, 64>>>(dA, dB, dC16, K);
        k32<<<1, 64>>>(dA, dB, dC32, K);
        hipMemcpy(C16, dC16, 256 * 4, hipMemcpyDeviceToHost); hipMemcpy(C32, dC32, 1024 * 4, hipMemcpyDeviceToHost);
        int bad16 = 0, bad32 = 0, bad_x = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                float s = 0.f;
                for (int k = 0; k < K; ++k) s = fmaf(A[i * K + k], B[j * K + k], s);
                if (memcmp(&s, &C32[i * 32 + j], 4)) ++bad32;
                if (i < 16 && j < 16) {
                    if (memcmp(&s, &C16[i * 16 + j], 4)) ++bad16;
                    if (memcmp(&C16[i * 16 + j], &C32[i * 32 + j], 4)) ++bad_x;
                }
            }
        printf("data set %d, K = %d: 32x32x2 vs scalar fmaf chain: %d / 1024 differ; 16x16x4 vs scalar chain: %d / 256 differ; 16x16x4 vs 32x32x2: %d / 256 differ\n",
               scale, K, bad32, bad16, bad_x);
    }
    return 0;
}
