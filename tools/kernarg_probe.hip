// Does a 16 KB by-value kernel argument launch on gfx950 / ROCm 7.2 -- directly, and as a kernel node of a captured hipGraph
// that is replayed?  (the descriptor tables of the persistent recurrence kernels, csrc/rfn_chain.hip, travel as one 13.7 KB
// kernel argument; bench.py --graph / graphed.GraphedTrainStep replay them from a graph)
#include <hip/hip_runtime.h>
#include <stdio.h>
struct Big { long w[2000]; };
__global__ void k(Big b, long* out) { out[threadIdx.x] = b.w[threadIdx.x * 7 % 2000] + b.w[1999]; }
int main() {
    Big b;
    for (int i = 0; i < 2000; ++i) b.w[i] = i;
    long* o;
    hipMalloc(&o, 64 * 8);
    hipStream_t st;
    hipStreamCreate(&st);
    k<<<1, 64, 0, st>>>(b, o);
    printf("direct launch: %s\n", hipGetErrorString(hipGetLastError()));
    hipStreamSynchronize(st);
    long h[64];
    hipMemcpy(h, o, 512, hipMemcpyDeviceToHost);
    printf("direct: out[1] = %ld (want 2006), out[63] = %ld (want 2440)\n", h[1], h[63]);
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    hipMemsetAsync(o, 0, 512, st);
    k<<<1, 64, 0, st>>>(b, o);
    hipStreamEndCapture(st, &g);
    for (int i = 0; i < 2000; ++i) b.w[i] = -1;     // the node must hold its own copy of the argument
    printf("instantiate: %s\n", hipGetErrorString(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0)));
    for (int r = 0; r < 3; ++r) {
        hipMemset(o, 0xff, 512);
        printf("replay %d: %s", r, hipGetErrorString(hipGraphLaunch(ge, st)));
        printf(", sync: %s", hipGetErrorString(hipStreamSynchronize(st)));
        hipMemcpy(h, o, 512, hipMemcpyDeviceToHost);
        printf(", out[1] = %ld (want 2006), out[63] = %ld (want 2440)\n", h[1], h[63]);
    }
    return 0;
}
