// Diagnostic (not part of the product): how accurate is an f32 product C = A * B^T when each f32 operand is split into
// three bf16 planes (a = a0 + a1 + a2 exactly) and the product is formed from bf16 MFMAs with f32 accumulation, compared
// with the exact-f32 MFMA chain the product uses?  Errors are measured against an f64 product of the same f32 inputs.
//   variants: f32 (v_mfma_f32_32x32x2_f32 chain), x3 (a0b0+a0b1+a1b0), x6 (+a0b2+a1b1+a2b0), x9 (all nine),
//             each with ONE accumulator or with one accumulator per magnitude class (summed small to large at the end)
//   hipcc --offload-arch=gfx950 -O3 tools/split_numerics_probe.hip -o /tmp/split_probe && /tmp/split_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned short to_bf16(float x) {   // round to nearest even (inputs are finite)
    unsigned u = __float_as_uint(x);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ inline float from_bf16(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

__device__ inline void split3(float x, unsigned short& p0, unsigned short& p1, unsigned short& p2) {
    p0 = to_bf16(x);
    float r = x - from_bf16(p0);
    p1 = to_bf16(r);
    r = r - from_bf16(p1);
    p2 = to_bf16(r);
}

// one wave per 32x32 output tile; A [M][K], B [N][K] row-major f32
// MODE 0: f32 MFMA chain; 3/6/9: number of bf16 products.  SEP: separate accumulators per class
template <int MODE, bool SEP>
__global__ __launch_bounds__(64) void probe(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                            int N, int K) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    f32x16 acc[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    if constexpr (MODE == 0) {
        for (int k = 0; k < K; k += 2)
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(size_t)(m0 + r) * K + k + h], B[(size_t)(n0 + r) * K + k + h], acc[0], 0,
                                                          0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            bf16x8 a[3], b[3];
            for (int j = 0; j < 8; ++j) {
                unsigned short p0, p1, p2;
                split3(A[(size_t)(m0 + r) * K + k + 8 * h + j], p0, p1, p2);
                a[0][j] = (short)p0; a[1][j] = (short)p1; a[2][j] = (short)p2;
                split3(B[(size_t)(n0 + r) * K + k + 8 * h + j], p0, p1, p2);
                b[0][j] = (short)p0; b[1][j] = (short)p1; b[2][j] = (short)p2;
            }
#define MM(i, j, c) acc[SEP ? (c) : 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[SEP ? (c) : 0], 0, 0, 0)
            if constexpr (MODE >= 9) { MM(2, 2, 2); MM(1, 2, 2); MM(2, 1, 2); }
            if constexpr (MODE >= 6) { MM(0, 2, 2); MM(1, 1, 2); MM(2, 0, 2); }
            MM(0, 1, 1); MM(1, 0, 1);
            MM(0, 0, 0);
#undef MM
        }
    }
    for (int j = 0; j < 16; ++j) {
        const int row = (j & 3) + 8 * (j >> 2) + 4 * h;
        C[(size_t)(m0 + row) * N + n0 + r] = SEP ? (acc[2][j] + acc[1][j]) + acc[0][j] : acc[0][j];
    }
}

static double urand() { return (rand() + 0.5) / (RAND_MAX + 1.0); }
static double nrand() { return sqrt(-2.0 * log(urand())) * cos(6.283185307179586 * urand()); }

template <int MODE, bool SEP>
static void run(const char* name, const float* dA, const float* dB, float* dC, int M, int N, int K, const std::vector<double>& ref,
                const std::vector<double>& mag) {
    hipLaunchKernelGGL((probe<MODE, SEP>), dim3(N / 32, M / 32), dim3(64), 0, 0, dA, dB, dC, N, K);
    std::vector<float> c((size_t)M * N);
    hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost);
    double emax = 0, e2 = 0, bias = 0;
    for (size_t i = 0; i < c.size(); ++i) {
        const double e = ((double)c[i] - ref[i]) / mag[i];     // relative to sum |a||b| (the forward error bound's scale)
        emax = fmax(emax, fabs(e));
        e2 += e * e;
        bias += e;
    }
    printf("    %-22s max %.3e  rms %.3e  mean %+.3e   (x 2^-24: max %.2f rms %.3f)\n", name, emax, sqrt(e2 / c.size()),
           bias / c.size(), emax * 16777216.0, sqrt(e2 / c.size()) * 16777216.0);
}

int main() {
    const int M = 64, N = 64;
    for (int K : {512, 2048, 50176})
        for (int dist = 0; dist < 3; ++dist) {
            std::vector<float> A((size_t)M * K), B((size_t)N * K);
            srand(1234 + K + dist);
            for (auto& x : A) x = dist == 0 ? (float)nrand() : dist == 1 ? (float)urand() : (float)(nrand() * exp(3.0 * nrand()));
            for (auto& x : B) x = dist == 0 ? (float)nrand() : dist == 1 ? (float)urand() : (float)(nrand() * exp(3.0 * nrand()));
            std::vector<double> ref((size_t)M * N), mag((size_t)M * N);
            for (int m = 0; m < M; ++m)
                for (int n = 0; n < N; ++n) {
                    double s = 0, t = 0;
                    for (int k = 0; k < K; ++k) {
                        const double p = (double)A[(size_t)m * K + k] * (double)B[(size_t)n * K + k];
                        s += p;
                        t += fabs(p);
                    }
                    ref[(size_t)m * N + n] = s;
                    mag[(size_t)m * N + n] = t;
                }
            float *dA, *dB, *dC;
            hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, (size_t)M * N * 4);
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            printf("K = %d, %s\n", K, dist == 0 ? "normal" : dist == 1 ? "uniform [0,1) (all positive)" : "normal x lognormal(3) (wide range)");
            run<0, false>("f32 MFMA chain", dA, dB, dC, M, N, K, ref, mag);
            run<3, false>("bf16 x3, 1 acc", dA, dB, dC, M, N, K, ref, mag);
            run<6, false>("bf16 x6, 1 acc", dA, dB, dC, M, N, K, ref, mag);
            run<6, true>("bf16 x6, 3 acc", dA, dB, dC, M, N, K, ref, mag);
            run<9, false>("bf16 x9, 1 acc", dA, dB, dC, M, N, K, ref, mag);
            run<9, true>("bf16 x9, 3 acc", dA, dB, dC, M, N, K, ref, mag);
            hipFree(dA); hipFree(dB); hipFree(dC);
        }
    return 0;
}
