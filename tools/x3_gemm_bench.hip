// Diagnostic: times the bf16-plane GEMM (rfn_gemm_x3.hip, included as source) at the headline shapes and checks it against
// an f64 reference on sampled outputs.
//   NT: hoisted projection of one encoder  (M = B*L = 50176, N = 8 x 512, K = 2048)
//   TN: its weight gradient                 (M = 8 x 512, N = 2048, K = 50176, split-K)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/x3_gemm_bench.hip -o /tmp/x3b && /tmp/x3b
#include "../recurrent_fusion_network_amd/csrc/rfn_gemm_x3.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#ifndef VARIANT
#define VARIANT "default"
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void fill_rand(float* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull;
        z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29; z *= 0x94D049BB133111EBull; z ^= z >> 32;
        const float u = (float)(z & 0xFFFFFF) / 16777216.f, v = (float)((z >> 24) & 0xFFFFFF) / 16777216.f;
        p[i] = scale * sqrtf(-2.f * logf(u + 1e-7f)) * cosf(6.2831853f * v);
    }
}
// f64 reference of sampled outputs: C[m][n] = sum_k A(m,k) B(n,k) with element accessors given by strides
__global__ void ref_k(const float* A, long a_rs, long a_ks, const float* B, long b_rs, long b_ks, int K, const int* ms, const int* ns,
                      int count, double* out, double* mag) {
    const int i = blockIdx.x;
    if (i >= count) return;
    double s = 0, t = 0;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const double p = (double)A[ms[i] * a_rs + k * a_ks] * (double)B[ns[i] * b_rs + k * b_ks];
        s += p; t += fabs(p);
    }
    __shared__ double ss[256], tt[256];
    ss[threadIdx.x] = s; tt[threadIdx.x] = t;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { ss[threadIdx.x] += ss[threadIdx.x + o]; tt[threadIdx.x] += tt[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[i] = ss[0]; mag[i] = tt[0]; }
}

int main() {
    const int BL = 50176, D = 2048, A = 512, T = 8;
    float *X, *W, *P, *dW, *bias;
    CK(hipMalloc(&X, (size_t)BL * D * 4));
    CK(hipMalloc(&W, (size_t)T * A * D * 4));
    CK(hipMalloc(&P, (size_t)T * BL * A * 4));
    CK(hipMalloc(&dW, (size_t)T * A * D * 4));
    CK(hipMalloc(&bias, (size_t)T * A * 4));
    fill_rand<<<4096, 256>>>(X, (size_t)BL * D, 1, 1.f);
    fill_rand<<<4096, 256>>>(W, (size_t)T * A * D, 2, 0.05f);
    fill_rand<<<64, 256>>>(bias, (size_t)T * A, 3, 0.1f);
    void *imgX, *imgXT, *imgW, *imgPT;
    CK(hipMalloc(&imgX, rfn_x3_image_bytes(BL, D)));
    CK(hipMalloc(&imgXT, rfn_x3_image_bytes(D, BL)));
    CK(hipMalloc(&imgW, rfn_x3_image_bytes(T * A, D)));
    CK(hipMalloc(&imgPT, rfn_x3_image_bytes(T * A, BL)));
    float* part;
    const int SK = rfn_x3_splitk_for(T * A, D, BL); printf("splitk %d\n", SK);
    CK(hipMalloc(&part, (size_t)SK * T * A * D * 4));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, double flops, double bytes, int reps, auto fn) {
        fn();
        CK(hipDeviceSynchronize());
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) fn();
        hipEventRecord(e1);
        CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        if (flops > 0) printf("%-14s %-34s %8.3f ms  %7.1f TF f32-equivalent  (%7.1f TF of bf16 MFMA)\n", VARIANT, name, ms, flops / ms * 1e-9, 6 * flops / ms * 1e-9);
        else printf("%-14s %-34s %8.3f ms  %7.1f GB/s\n", VARIANT, name, ms, bytes / ms * 1e-6);
    };
    // ---- splits
    timeit("split X  [BL][D] k-fast", 0, (double)BL * D * 10, 5, [&] { const float* sp[1] = {X}; rfn_x3_split(sp, 1, D, BL, D, 1, imgX, 0); });
    timeit("split X^T (rows = D, k = BL)", 0, (double)BL * D * 10, 5, [&] { const float* sp[1] = {X}; rfn_x3_split(sp, 1, D, D, BL, 0, imgXT, 0); });
    timeit("split W  8 x [A][D]", 0, (double)T * A * D * 10, 5, [&] {
        const float* sp[8];
        for (int t = 0; t < T; ++t) sp[t] = W + (size_t)t * A * D;
        rfn_x3_split(sp, T, D, A, D, 1, imgW, 0);
    });
    // ---- NT projection
    float* Cp[8]; const float* bp[8];
    for (int t = 0; t < T; ++t) { Cp[t] = P + (size_t)t * BL * A; bp[t] = bias + t * A; }
    const double flops = 2.0 * BL * D * A * T;
#ifdef NT_ROWS
    timeit("NT projection, fewer rows", 2.0 * NT_ROWS * D * A * T, 0, 10, [&] { rfn_x3_gemm(NT_ROWS, T * A, D, imgX, imgW, NT_ROWS, A, Cp, bp, A, 0, 1, nullptr, 0); });
#endif
    timeit("NT projection", flops, 0, 10, [&] { int rc = rfn_x3_gemm(BL, T * A, D, imgX, imgW, BL, A, Cp, bp, A, 0, 1, nullptr, 0); if (rc) { printf("rc %d\n", rc); exit(1); } });
    // check
    const int NS = 4096;
    std::vector<int> ms(NS), ns(NS);
    srand(7);
    for (int i = 0; i < NS; ++i) { ms[i] = rand() % BL; ns[i] = rand() % (T * A); }
    ms[0] = 0; ns[0] = 0; ms[1] = BL - 1; ns[1] = T * A - 1; ms[2] = 255; ns[2] = 256; ms[3] = 256; ns[3] = 255;
    int *dms, *dns; double *dref, *dmag;
    CK(hipMalloc(&dms, NS * 4)); CK(hipMalloc(&dns, NS * 4)); CK(hipMalloc(&dref, NS * 8)); CK(hipMalloc(&dmag, NS * 8));
    CK(hipMemcpy(dms, ms.data(), NS * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dns, ns.data(), NS * 4, hipMemcpyHostToDevice));
    ref_k<<<NS, 256>>>(X, D, 1, W, D, 1, D, dms, dns, NS, dref, dmag);
    std::vector<double> ref(NS), mag(NS);
    CK(hipMemcpy(ref.data(), dref, NS * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(mag.data(), dmag, NS * 8, hipMemcpyDeviceToHost));
    std::vector<float> hb((size_t)T * A);
    CK(hipMemcpy(hb.data(), bias, hb.size() * 4, hipMemcpyDeviceToHost));
    {
        double emax = 0, e2 = 0;
        for (int i = 0; i < NS; ++i) {
            const int t = ns[i] / A, a = ns[i] % A;
            float v; CK(hipMemcpy(&v, P + (size_t)t * BL * A + (size_t)ms[i] * A + a, 4, hipMemcpyDeviceToHost));
            const double e = ((double)v - (ref[i] + (double)hb[ns[i]])) / mag[i];
            emax = fmax(emax, fabs(e)); e2 += e * e;
        }
        printf("%-14s NT check: %d samples, error / sum|a||b|: max %.3e rms %.3e  (x 2^-24: %.2f / %.3f)\n", VARIANT, NS, emax, sqrt(e2 / NS), emax * 16777216., sqrt(e2 / NS) * 16777216.);
    }
    // ---- TN weight gradient: dW[t][a][d] = sum_(b,l) P[t][(b,l)][a] X[(b,l)][d]; P plays the role of the upstream gradient
    fill_rand<<<4096, 256>>>(P, (size_t)T * BL * A, 5, 0.01f);
    timeit("split dP^T (8 x rows = A, k = BL)", 0, (double)T * BL * A * 10, 5, [&] {
        const float* sp[8];
        for (int t = 0; t < T; ++t) sp[t] = P + (size_t)t * BL * A;
        rfn_x3_split(sp, T, A, A, BL, 0, imgPT, 0);
    });
    float* Cw[8];
    for (int t = 0; t < T; ++t) Cw[t] = dW + (size_t)t * A * D;
    timeit("TN weight gradient (split-K 2)", flops, 0, 10, [&] { int rc = rfn_x3_gemm(T * A, D, BL, imgPT, imgXT, A, D, Cw, nullptr, D, 0, SK, part, 0); if (rc) { printf("rc %d\n", rc); exit(1); } });
    auto tn_check = [&](const char* label) {
        for (int i = 0; i < NS; ++i) { ms[i] = rand() % (T * A); ns[i] = rand() % D; }
        ms[0] = 0; ns[0] = 0; ms[1] = T * A - 1; ns[1] = D - 1;
        CK(hipMemcpy(dms, ms.data(), NS * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dns, ns.data(), NS * 4, hipMemcpyHostToDevice));
        {
            // row m = t*A + a of the M-side operand is P[t][k][a]: not a single stride pair over t, so reference per t
            double emax = 0, e2 = 0;
            std::vector<int> mt(NS);
            for (int t = 0; t < T; ++t) {
                std::vector<int> idx;
                for (int i = 0; i < NS; ++i) if (ms[i] / A == t) idx.push_back(i);
                std::vector<int> m2(idx.size()), n2(idx.size());
                for (size_t j = 0; j < idx.size(); ++j) { m2[j] = ms[idx[j]] % A; n2[j] = ns[idx[j]]; }
                if (idx.empty()) continue;
                CK(hipMemcpy(dms, m2.data(), m2.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dns, n2.data(), n2.size() * 4, hipMemcpyHostToDevice));
                ref_k<<<(int)idx.size(), 256>>>(P + (size_t)t * BL * A, 1, A, X, 1, D, BL, dms, dns, (int)idx.size(), dref, dmag);
                CK(hipMemcpy(ref.data(), dref, idx.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(mag.data(), dmag, idx.size() * 8, hipMemcpyDeviceToHost));
                for (size_t j = 0; j < idx.size(); ++j) {
                    float v; CK(hipMemcpy(&v, dW + (size_t)t * A * D + (size_t)m2[j] * D + n2[j], 4, hipMemcpyDeviceToHost));
                    const double e = ((double)v - ref[j]) / mag[j];
                    emax = fmax(emax, fabs(e)); e2 += e * e;
                }
            }
            printf("%-14s %s: %d samples, error / sum|a||b|: max %.3e rms %.3e  (x 2^-24: %.2f / %.3f)\n", VARIANT, label, NS, emax, sqrt(e2 / NS), emax * 16777216., sqrt(e2 / NS) * 16777216.);
        }
    };
    tn_check("TN check");
#if X3_SHAPE == 16
    // ---- the same weight gradient from k-slow images (no transposing passes; fragments by ds_read_b64_tr_b16)
    void *ksX, *ksP;
    CK(hipMalloc(&ksX, rfn_x3_image_bytes(D, BL)));
    CK(hipMalloc(&ksP, rfn_x3_image_bytes(T * A, BL)));
    timeit("split X  k-slow [BL][D]", 0, (double)BL * D * 10, 5, [&] { const float* sp[1] = {X}; rfn_x3_split_ks(sp, 1, D, BL, D, ksX, 0); });
    timeit("split dP k-slow 8 x [BL][A]", 0, (double)T * BL * A * 10, 5, [&] {
        const float* sp[8];
        for (int t = 0; t < T; ++t) sp[t] = P + (size_t)t * BL * A;
        rfn_x3_split_ks(sp, T, A, BL, A, ksP, 0);
    });
    CK(hipMemset(dW, 0, (size_t)T * A * D * 4));
    timeit("TN weight gradient, k-slow images", flops, 0, 10, [&] { int rc = rfn_x3_gemm_ks(T * A, D, BL, ksP, ksX, A, D, Cw, nullptr, D, 0, SK, part, 0); if (rc) { printf("rc %d\n", rc); exit(1); } });
    tn_check("TN k-slow check");
#endif
    return 0;
}
