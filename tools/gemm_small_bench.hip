// Diagnostic: the skinny per-step GEMMs of the recurrence (M = batch = 256), timed back to back with cold-ish
// weights (a different weight matrix per call, as in the path).  Variants via -DGEMM_SMALL_* knobs.
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
    float* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d;
}
struct Case { const char* name; int M, N, K, nseg, ngroups, ak, bk; };
int main() {
    const Case cases[] = {
        {"dec dx   256x512  K2048 NN g2", 256, 512, 2048, 1, 2, 1, 0},
        {"st2 dx   256x512  K2048 NN g5", 256, 512, 2048, 1, 5, 1, 0},
        {"st1 dH   256x2048 K2048x4 NN ", 256, 2048, 2048, 4, 1, 1, 0},
        {"st1 dz   256x2048 K2048 NN g4", 256, 2048, 2048, 1, 4, 1, 0},
        {"hproj    256x512  K512  NT g4", 256, 512, 512, 1, 4, 1, 1},
        {"st1 gate 256x2048 K2048x2 NT g4", 256, 2048, 2048, 2, 4, 1, 1},
        {"dec gate 256x2048 K512x2 NT  ", 256, 2048, 512, 2, 1, 1, 1},
        {"st2 gate 256x2048 K512x5 NT  ", 256, 2048, 512, 5, 1, 1, 1},
    };
    const int NW = 8;   // rotate over 8 weight copies so weights are not L2-hot
    float* ws; hipMalloc(&ws, 48u << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (const Case& c : cases) {
        std::vector<float*> A, B, C;
        for (int i = 0; i < NW * c.ngroups * c.nseg; ++i) {
            A.push_back(dev_rand((size_t)c.M * c.K, 1 + i));
            B.push_back(dev_rand((size_t)c.N * c.K, 100 + i));
        }
        for (int i = 0; i < NW * c.ngroups; ++i) { float* p; hipMalloc(&p, (size_t)c.M * c.N * 4); C.push_back(p); }
        auto run = [&](int w) {
            rfn_gemm_problem pr[8];
            for (int g = 0; g < c.ngroups; ++g) {
                memset(&pr[g], 0, sizeof(pr[g]));
                pr[g].C = C[w * c.ngroups + g]; pr[g].ldc = c.N; pr[g].nseg = c.nseg;
                for (int s = 0; s < c.nseg; ++s) {
                    rfn_gemm_seg& sg = pr[g].seg[s];
                    const int idx = (w * c.ngroups + g) * c.nseg + s;
                    sg.A = A[idx]; sg.lda = c.ak ? c.K : c.M; sg.a_kfast = c.ak;
                    sg.B = B[idx]; sg.ldb = c.bk ? c.K : c.N; sg.b_kfast = c.bk; sg.K = c.K;
                }
            }
            return rfn_gemm_f32_ws(c.M, c.N, c.ngroups, pr, 0, ws, 48u << 20, 0);
        };
        for (int w = 0; w < NW; ++w) run(w);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        const int reps = 4;
        for (int r = 0; r < reps; ++r) for (int w = 0; w < NW; ++w) run(w);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / (reps * NW);
        const double gf = 2.0 * c.M * c.N * c.K * c.nseg * c.ngroups / 1e9;
        printf("%-22s %-34s %7.1f us  (%5.1f GF -> %5.1f TF)\n", VARIANT, c.name, us, gf, gf / us * 1e-3 * 1e3);
        for (float* p : A) hipFree(p);
        for (float* p : B) hipFree(p);
        for (float* p : C) hipFree(p);
    }
    return 0;
}
