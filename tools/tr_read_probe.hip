// Diagnostic: what does ds_read_b64_tr_b16 deliver?  LDS holds T[row][col] = 100 * row + col (16-bit, 64 columns per row);
// every 16-lane group g reads the 4 x 16 block at rows 4g .. 4g+3, columns 16 .. 31: lane 4q + p supplies the address of
// T[4g + q][16 + 4p].  Expected (cdna_hip_programming.md T10): lane i of the group receives column 16 + i of the 4 rows.
//   hipcc --offload-arch=gfx950 -O3 tools/tr_read_probe.hip -o /tmp/trp && /tmp/trp
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short T[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) T[i] = (short)(100 * (i / 64) + (i % 64));
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, j = l & 15, q = j >> 2, p = j & 3;
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(&T[(4 * g + q) * 64 + 16 + 4 * p]));
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = r[e];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    k<<<1, 64>>>(d);
    short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, i = l & 15;
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) {
            printf(" %4d", h[l * 4 + e]);
            bad += h[l * 4 + e] != 100 * (4 * g + e) + 16 + i;
        }
        printf("%s", (l & 3) == 3 ? "\n" : "   ");
    }
    printf("%s\n", bad ? "MISMATCH with the expected mapping" : "mapping as expected: lane i of a group gets column i, element e = row e of the block");
    return 0;
}
