// Where does a K step of the cell GEMM's loop go?  The launch kernel (product source included, -DCG_LOOP_STAMPS) on the decoder's
// backward product Kb1 at B = 64 ([dh_rec | dz] = dgates (64 x 2048) . [W_hh | W_z] (2048 x 512 each)) and on the forward K3 shape,
// one launch each per tile variant; block 0 stamps the shader clock inside every K step.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCG_LOOP_STAMPS -I recurrent_fusion_network_amd/csrc tools/cg_loop_probe.hip -o /tmp/cg_loop_probe
#include "../recurrent_fusion_network_amd/csrc/rfn_cellgemm.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
static float* dev_rand(size_t n, unsigned seed) {
    std::vector<float> h(n);
    srand(seed);
    for (size_t i = 0; i < n; ++i) h[i] = (rand() / (float)RAND_MAX) * 0.2f - 0.1f;
    float* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d;
}
int main() {
    const int M = 64, R = 512;
    float* dg = dev_rand((size_t)M * 4 * R, 1);
    float* W[2] = {dev_rand((size_t)4 * R * R, 2), dev_rand((size_t)4 * R * R, 3)};
    float* C[2] = {dev_rand((size_t)M * R, 4), dev_rand((size_t)M * R, 5)};
    float* z = dev_rand((size_t)M * R, 6);
    float* Wz = dev_rand((size_t)4 * R * R, 7);
    float* g = dev_rand((size_t)M * 4 * R, 8);
    for (int shape = 0; shape < 2; ++shape) {
        rfn_cell_out outs[2];
        memset(outs, 0, sizeof(outs));
        int nout;
        if (shape == 0) {       // Kb1: two outputs, K = 2048, weights [k][n]
            nout = 2;
            for (int o = 0; o < 2; ++o) {
                outs[o].C = C[o]; outs[o].ldc = R; outs[o].N = R; outs[o].nseg = 1; outs[o].epilogue = RFN_CELL_EPI_STORE;
                outs[o].seg[0].A = dg; outs[o].seg[0].lda = 4 * R; outs[o].seg[0].a_kfast = 1;
                outs[o].seg[0].B = W[o]; outs[o].seg[0].ldb = R; outs[o].seg[0].b_kfast = 0; outs[o].seg[0].K = 4 * R;
            }
        } else {                // K1-like forward store: g (64 x 2048) = z . Wz^T, K = 512, weights [n][k]
            nout = 1;
            outs[0].C = g; outs[0].ldc = 4 * R; outs[0].N = 4 * R; outs[0].nseg = 1; outs[0].epilogue = RFN_CELL_EPI_STORE;
            outs[0].seg[0].A = z; outs[0].seg[0].lda = R; outs[0].seg[0].a_kfast = 1;
            outs[0].seg[0].B = Wz; outs[0].seg[0].ldb = R; outs[0].seg[0].b_kfast = 1; outs[0].seg[0].K = R;
        }
        for (int variant : {6, 8}) {     // wave-private slots: the steady-state loop stamps its top and the end of its counted wait
            for (int rep = 0; rep < 3; ++rep) {
                if (rfn_cell_gemm(M, nout, outs, R, 0.f, 0, variant, 0) != 0) { printf("variant %d refused\n", variant); break; }
                hipDeviceSynchronize();
            }
            unsigned long long st[64 * 8];
            hipMemcpyFromSymbol(st, HIP_SYMBOL(g_cg_loop_stamps), sizeof(st));
            const int T = (shape == 0 ? 2048 : 512) / 64 - CG_WP_SLOTS;     // steady-state steps
            double wait = 0, total = 0;
            for (int it = 0; it < T && it < 64; ++it) {
                wait += (double)(st[it * 8 + 1] - st[it * 8]);
                if (it + 1 < T) total += (double)(st[(it + 1) * 8] - st[it * 8]);
            }
            printf("%s variant %d: %2d steady K steps, cycles per step: counted wait %6.0f | step total %6.0f\n",
                   shape == 0 ? "Kb1 (K = 2048, [k][n] weights)" : "forward store (K = 512, [n][k] weights)", variant, T, wait / T,
                   total / (T - 1));
        }
        for (int variant : {3, 4, 5}) {
            for (int rep = 0; rep < 3; ++rep) {
                if (rfn_cell_gemm(M, nout, outs, R, 0.f, 0, variant, 0) != 0) { printf("variant %d refused\n", variant); break; }
                hipDeviceSynchronize();
            }
            unsigned long long st[64 * 8];
            hipMemcpyFromSymbol(st, HIP_SYMBOL(g_cg_loop_stamps), sizeof(st));
            const int T = (shape == 0 ? 2048 : 512) / (variant == 5 ? 128 : 64);
            double seg[5] = {0, 0, 0, 0, 0};
            for (int it = 0; it < T && it < 64; ++it) {
                for (int k = 0; k < 4; ++k) seg[k] += (double)(st[it * 8 + k + 1] - st[it * 8 + k]);
                if (it + 1 < T) seg[4] += (double)(st[(it + 1) * 8] - st[it * 8 + 4]);
            }
            printf("%s variant %d: %2d K steps, cycles per step: wait own pieces %6.0f | block barrier %6.0f | request next %6.0f | "
                   "fragment reads + MFMA issue %6.0f | loop back %5.0f | step total %6.0f  (first-to-last stamp %.0f cycles)\n",
                   shape == 0 ? "Kb1 (K = 2048, [k][n] weights)" : "forward store (K = 512, [n][k] weights)", variant, T, seg[0] / T, seg[1] / T,
                   seg[2] / T, seg[3] / T, seg[4] / (T - 1), (seg[0] + seg[1] + seg[2] + seg[3]) / T + seg[4] / (T - 1),
                   (double)(st[(T - 1) * 8 + 4] - st[0]));
        }
    }
    return 0;
}
