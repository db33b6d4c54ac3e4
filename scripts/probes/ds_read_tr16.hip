// Probe: ds_read_b64_tr_b16 (gfx950) through __builtin_amdgcn_ds_read_tr16_b64_v4i16.  Image [16 rows][32 cols] of 16-bit values
// img[r][c] = 100 r + c; lane 16 g + 4 q + p supplies the address of (row r0 + q, col c0 + 4 p) of its group's 4 x 16 block.
// Expected: lane 16 g + i receives { img[r0 + e][c0 + i] : e = 0..3 } -- column i of the block, a free transpose.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short img[16 * 32];
    for (int i = threadIdx.x; i < 512; i += 64) img[i] = (unsigned short)(100 * (i / 32) + (i % 32));
    __syncthreads();
    const int lane = threadIdx.x;
    const int g = lane >> 4, idx = lane & 15, q = idx >> 2, p = idx & 3;
    const int c0 = 16 * (g & 1), r0 = 8 * (g >> 1);
    unsigned short* a = img + (r0 + q) * 32 + c0 + 4 * p;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (unsigned short)v[e];
}
int main() {
    unsigned short* d; unsigned short h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, i = lane & 15, c0 = 16 * (g & 1), r0 = 8 * (g >> 1);
        for (int e = 0; e < 4; ++e) if (h[lane * 4 + e] != 100 * (r0 + e) + c0 + i) ++bad;
    }
    printf("lane 0: %d %d %d %d | lane 17: %d %d %d %d | lane 37: %d %d %d %d | mismatches vs expected column layout: %d / 256\n",
           h[0], h[1], h[2], h[3], h[68], h[69], h[70], h[71], h[148], h[149], h[150], h[151], bad);
    return bad != 0;
}
