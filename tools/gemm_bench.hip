// Diagnostic: times the product GEMM kernel (included as source) at the headline shapes.
//   NT: hoisted att_2_att_h projection of one encoder  (M=B*L=50176, N=512, K=2048, 8 groups)
//   TN: its weight gradient                              (M=512, N=2048, K=50176, 8 groups)
// Built several times with -DGEMM_* knobs by tools/run_gemm_bench.sh; prints TFLOP/s per variant.
#include "../recurrent_fusion_network_amd/csrc/rfn_gemm.hip"
#include "../recurrent_fusion_network_amd/csrc/rfn_cell.hip"   // rfn_gemm_f32_lstm falls back to rfn_lstm_fwd_grouped
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
        nt[t].C = P + (size_t)t * BL * A; nt[t].ldc = A; nt[t].nseg = 1;   // step-major slabs, as rfn_prefix_fwd lays them out
        nt[t].seg[0].A = X; nt[t].seg[0].lda = D; nt[t].seg[0].a_kfast = 1;
        nt[t].seg[0].B = W + (size_t)t * A * D; nt[t].seg[0].ldb = D; nt[t].seg[0].b_kfast = 1; nt[t].seg[0].K = D;
        tn[t].C = dW + (size_t)t * A * D; tn[t].ldc = D; tn[t].nseg = 1;
        tn[t].seg[0].A = P + (size_t)t * BL * A; tn[t].seg[0].lda = A; tn[t].seg[0].a_kfast = 0;
        tn[t].seg[0].B = X; tn[t].seg[0].ldb = D; tn[t].seg[0].b_kfast = 0; tn[t].seg[0].K = BL;
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flops = 2.0 * BL * D * A * T;
    // bit-equality of the LDS-DMA kernel and the register-staged kernel (same k order per output element)
    {
        std::vector<float> r0((size_t)T * A * D), r1((size_t)T * A * D), p0(1 << 22), p1(1 << 22);
        for (unsigned fl = 0; fl < 2; ++fl) {
            const unsigned flags = fl ? RFN_GEMM_OPT_NO_DMA : 0u;
            hipMemset(P, 0, (size_t)BL * T * A * 4);
            rfn_gemm_f32_opt(BL, A, T, nt, 0, ws, ws_bytes, flags, 0);
            hipMemcpy((fl ? p1 : p0).data(), P + (size_t)3 * BL * A + 12345, p0.size() * 4, hipMemcpyDeviceToHost);
            rfn_gemm_f32_opt(A, D, T, tn, 0, ws, ws_bytes, flags, 0);
            hipMemcpy((fl ? r1 : r0).data(), dW, r0.size() * 4, hipMemcpyDeviceToHost);
        }
        size_t bad_p = 0, bad_w = 0;
        for (size_t i = 0; i < p0.size(); ++i) bad_p += memcmp(&p0[i], &p1[i], 4) != 0;
        for (size_t i = 0; i < r0.size(); ++i) bad_w += memcmp(&r0[i], &r1[i], 4) != 0;
        printf("%-28s DMA vs register-staged: %zu / %zu projection values differ, %zu / %zu dW values differ (p[7]=%g w[7]=%g)\n", VARIANT,
               bad_p, p0.size(), bad_w, r0.size(), p0[7], r0[7]);
    }
    for (int which = 0; which < 4; ++which) {
        const unsigned flags = (which & 2) ? RFN_GEMM_OPT_NO_DMA : 0u;
        float best = 1e9, sum = 0;
        const int reps = 6;
        for (int r = 0; r < reps + 1; ++r) {
            hipEventRecord(e0);
            int rc = (which & 1) == 0 ? rfn_gemm_f32_opt(BL, A, T, nt, 0, ws, ws_bytes, flags, 0)
                                      : rfn_gemm_f32_opt(A, D, T, tn, 0, ws, ws_bytes, flags, 0);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rc) { printf("rc %d\n", rc); return 1; }
            if (r) { sum += ms; if (ms < best) best = ms; }
        }
        printf("%-28s %s %s: avg %.3f ms (%.1f TF)  best %.3f ms (%.1f TF)\n", VARIANT, (which & 1) ? "TN dW  " : "NT proj",
               flags ? "reg-staged" : "LDS-DMA   ", sum / reps, flops / (sum / reps) / 1e9, best, flops / best / 1e9);
    }
    return 0;
}
