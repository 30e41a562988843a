// Does a 16 KB by-value kernel argument launch on gfx950 / ROCm 7.2?  (descriptor tables of the persistent recurrence kernels)
#include <hip/hip_runtime.h>
#include <stdio.h>
struct Big { long w[2000]; };
__global__ void k(Big b, long* out) { out[threadIdx.x] = b.w[threadIdx.x * 7 % 2000] + b.w[1999]; }
int main() { Big b; for (int i = 0; i < 2000; ++i) b.w[i] = i; long* o; hipMalloc(&o, 64 * 8); k<<<1, 64>>>(b, o); printf("%s\n", hipGetErrorString(hipGetLastError())); hipDeviceSynchronize(); long h[64]; hipMemcpy(h, o, 512, hipMemcpyDeviceToHost); printf("%ld %ld\n", h[1], h[63]); }
