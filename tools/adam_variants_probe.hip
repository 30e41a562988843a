// Access-pattern variants of the clamp + Adam stream (4 arrays in, 3 out, 28 B per parameter) on one 390 M-parameter bucket:
// which of them moves the bytes fastest on MI355X.  Same element arithmetic in every variant (csrc/rfn_misc.hip adam_elem).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/adam_variants_probe.hip -o /tmp/adam_variants && /tmp/adam_variants
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void adam_elem(float& pv, float gv, float& mv, float& vv, float lr_over_bc1, float beta1, float beta2,
                                          float eps, float inv_sqrt_bc2, float wd, float clip, float gscale) {
#pragma clang fp contract(off)
    gv *= gscale;
    gv = fminf(fmaxf(gv, -clip), clip);
    gv = __builtin_fmaf(wd, pv, gv);
    mv = __builtin_fmaf(beta1, mv, (1.0f - beta1) * gv);
    vv = __builtin_fmaf(beta2, vv, ((1.0f - beta2) * gv) * gv);
    pv = pv - (lr_over_bc1 * mv) / __builtin_fmaf(sqrtf(vv), inv_sqrt_bc2, eps);
}
__device__ __forceinline__ void adam4(f32x4& pv, const f32x4& gv, f32x4& mv, f32x4& vv) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float pe = pv[e], me = mv[e], ve = vv[e];
        adam_elem(pe, gv[e], me, ve, 5e-4f, 0.9f, 0.999f, 1e-8f, 1.0f, 1e-5f, 1.0f, 1.0f);
        pv[e] = pe;
        mv[e] = me;
        vv[e] = ve;
    }
}

// MODE 0: the product's pattern (grid-stride, one float4 per array per thread and trip)
// MODE 1: two trips' loads issued together, then the six stores
// MODE 2: four trips' loads together
// MODE 3: MODE 0 with nontemporal loads and stores
// MODE 4: MODE 1 with nontemporal loads and stores
// MODE 5: block-contiguous: a block owns a contiguous run of the arrays (UNR float4 per thread per trip, back to back)
template <int MODE>
__global__ __launch_bounds__(256) void adam_var_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                  float* __restrict__ v, long n4) {
    f32x4* p4 = reinterpret_cast<f32x4*>(p);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
    f32x4* m4 = reinterpret_cast<f32x4*>(m);
    f32x4* v4 = reinterpret_cast<f32x4*>(v);
    const long stride = (long)gridDim.x * 256, i0 = (long)blockIdx.x * 256 + threadIdx.x;
    constexpr bool NT = (MODE == 3 || MODE == 4 || MODE == 6 || MODE == 9);
    auto ld = [&](const f32x4* a, long i) -> f32x4 { return NT ? __builtin_nontemporal_load(a + i) : a[i]; };
    auto st = [&](f32x4* a, long i, const f32x4& x) {
        if (NT) __builtin_nontemporal_store(x, a + i);
        else a[i] = x;
    };
    if constexpr (MODE == 0 || MODE == 3) {
        for (long i = i0; i < n4; i += stride) {
            f32x4 pv = ld(p4, i), mv = ld(m4, i), vv = ld(v4, i);
            const f32x4 gv = ld(g4, i);
            adam4(pv, gv, mv, vv);
            st(m4, i, mv);
            st(v4, i, vv);
            st(p4, i, pv);
        }
    } else if constexpr (MODE == 1 || MODE == 2 || MODE == 4) {
        constexpr int U = (MODE == 2) ? 4 : 2;
        for (long i = i0; i < n4; i += U * stride) {
            f32x4 pv[U], mv[U], vv[U], gv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long j = i + u * stride < n4 ? i + u * stride : i;
                pv[u] = ld(p4, j);
                mv[u] = ld(m4, j);
                vv[u] = ld(v4, j);
                gv[u] = ld(g4, j);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) adam4(pv[u], gv[u], mv[u], vv[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long j = i + u * stride;
                if (j < n4) {
                    st(m4, j, mv[u]);
                    st(v4, j, vv[u]);
                    st(p4, j, pv[u]);
                }
            }
        }
    } else {
        constexpr int U = (MODE == 7 || MODE == 9) ? 8 : (MODE == 8 ? 2 : 4);
        const long per_block = (n4 + gridDim.x - 1) / gridDim.x;
        const long b0 = (long)blockIdx.x * per_block, b1 = b0 + per_block < n4 ? b0 + per_block : n4;
        for (long i = b0 + threadIdx.x; i < b1; i += U * 256) {
            f32x4 pv[U], mv[U], vv[U], gv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long j = i + u * 256 < b1 ? i + u * 256 : i;
                pv[u] = ld(p4, j);
                mv[u] = ld(m4, j);
                vv[u] = ld(v4, j);
                gv[u] = ld(g4, j);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) adam4(pv[u], gv[u], mv[u], vv[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long j = i + u * 256;
                if (j < b1) {
                    st(m4, j, mv[u]);
                    st(v4, j, vv[u]);
                    st(p4, j, pv[u]);
                }
            }
        }
    }
}

template <int MODE>
static float run(float* p, float* g, float* m, float* v, long n, int blocks, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(adam_var_k<MODE>, dim3(blocks), dim3(256), 0, 0, p, g, m, v, n / 4);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(adam_var_k<MODE>, dim3(blocks), dim3(256), 0, 0, p, g, m, v, n / 4);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    const long n = 390L * 1000 * 1000;
    float *p, *g, *m, *v;
    hipMalloc(&p, n * 4);
    hipMalloc(&g, n * 4);
    hipMalloc(&m, n * 4);
    hipMalloc(&v, n * 4);
    hipMemset(p, 0, n * 4);
    hipMemset(g, 0, n * 4);
    hipMemset(m, 0, n * 4);
    hipMemset(v, 0, n * 4);
    const char* names[10] = {"product pattern (grid-stride, one float4 per array per thread and trip)", "two trips' loads together",
                             "four trips' loads together", "= 0 nontemporal", "= 1 nontemporal",
                             "block-contiguous runs, 4 float4 per thread and trip", "= 5 nontemporal", "= 5 with 8 float4", "= 5 with 2 float4",
                             "= 7 nontemporal"};
    for (int round = 0; round < 2; ++round)
        for (int blocks : {512, 1024, 1536, 2048, 2560, 3072, 4096, 8192}) {
            float t[10];
            t[0] = run<0>(p, g, m, v, n, blocks, 5);
            t[3] = run<3>(p, g, m, v, n, blocks, 5);
            t[5] = run<5>(p, g, m, v, n, blocks, 5);
            t[6] = run<6>(p, g, m, v, n, blocks, 5);
            t[7] = run<7>(p, g, m, v, n, blocks, 5);
            t[8] = run<8>(p, g, m, v, n, blocks, 5);
            t[9] = run<9>(p, g, m, v, n, blocks, 5);
            printf("blocks %5d:", blocks);
            for (int k : {0, 3, 5, 6, 7, 8, 9}) printf("  [%d] %.3f ms %.2f TB/s", k, t[k], 28.0 * n / t[k] / 1e9);
            printf("\n");
        }
    for (int k = 0; k < 10; ++k) printf("  [%d] %s\n", k, names[k]);
    return 0;
}
