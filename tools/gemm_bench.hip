// Diagnostic: times the product GEMM kernel (included as source) at the headline shapes.
//   NT: hoisted att_2_att_h projection of one encoder  (M=B*L=50176, N=512, K=2048, 8 groups)
//   TN: its weight gradient                              (M=512, N=2048, K=50176, 8 groups)
// Built several times with -DGEMM_* knobs by tools/run_gemm_bench.sh; prints TFLOP/s per variant.
#include "../recurrent_fusion_network_amd/csrc/rfn_gemm.hip"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#ifndef VARIANT
#define VARIANT "default"
#endif
static float* dev_rand(size_t n, unsigned seed) {
    std::vector<float> h(n);
    srand(seed);
    for (size_t i = 0; i < n; ++i) h[i] = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    float* d;
    hipMalloc(&d, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    return d;
}
int main() {
    const int BL = 50176, D = 2048, A = 512, T = 8;
    float* X = dev_rand((size_t)BL * D, 1);
    float* W = dev_rand((size_t)T * A * D, 2);
    float* P;   hipMalloc(&P, (size_t)BL * T * A * 4);
    float* dW;  hipMalloc(&dW, (size_t)T * A * D * 4);
    const size_t ws_bytes = (size_t)160 << 20;   // split-K scratch, as the path provides one
    void* ws;   hipMalloc(&ws, ws_bytes);
    hipMemset(P, 0, (size_t)BL * T * A * 4);
    rfn_gemm_problem nt[8], tn[8];
    for (int t = 0; t < T; ++t) {
        memset(&nt[t], 0, sizeof(nt[t])); memset(&tn[t], 0, sizeof(tn[t]));
        nt[t].C = P + (size_t)t * A; nt[t].ldc = (long)T * A; nt[t].nseg = 1;
        nt[t].seg[0].A = X; nt[t].seg[0].lda = D; nt[t].seg[0].a_kfast = 1;
        nt[t].seg[0].B = W + (size_t)t * A * D; nt[t].seg[0].ldb = D; nt[t].seg[0].b_kfast = 1; nt[t].seg[0].K = D;
        tn[t].C = dW + (size_t)t * A * D; tn[t].ldc = D; tn[t].nseg = 1;
        tn[t].seg[0].A = P + (size_t)t * A; tn[t].seg[0].lda = (long)T * A; tn[t].seg[0].a_kfast = 0;
        tn[t].seg[0].B = X; tn[t].seg[0].ldb = D; tn[t].seg[0].b_kfast = 0; tn[t].seg[0].K = BL;
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flops = 2.0 * BL * D * A * T;
    for (int which = 0; which < 2; ++which) {
        float best = 1e9, sum = 0;
        const int reps = 6;
        for (int r = 0; r < reps + 1; ++r) {
            hipEventRecord(e0);
            int rc = which == 0 ? rfn_gemm_f32_ws(BL, A, T, nt, 0, ws, ws_bytes, 0) : rfn_gemm_f32_ws(A, D, T, tn, 0, ws, ws_bytes, 0);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rc) { printf("rc %d\n", rc); return 1; }
            if (r) { sum += ms; if (ms < best) best = ms; }
        }
        printf("%-28s %s: avg %.3f ms (%.1f TF)  best %.3f ms (%.1f TF)\n", VARIANT, which ? "TN dW  " : "NT proj",
               sum / reps, flops / (sum / reps) / 1e9, best, flops / best / 1e9);
    }
    return 0;
}
