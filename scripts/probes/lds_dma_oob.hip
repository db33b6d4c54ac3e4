// Probe (not product): what does an LDS-DMA buffer load (`buffer_load_dwordx4 ... offen lds`) write into LDS for a lane whose
// offset is out of the descriptor's range -- zeros, or nothing (stale LDS bytes)?  The bf16 conv kernel's padding relies on it.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/lds_dma_oob.hip -o /tmp/lds_dma_oob && /tmp/lds_dma_oob
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* g, float* out, int n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* f = (float*)smem;
    for (int i = threadIdx.x; i < 512; i += 64) f[i] = -7.f;                       // poison
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, n * 4, 0x00020000);
    unsigned voff = threadIdx.x * 16;
    if (threadIdx.x & 1) voff = 0x80000000u;                                       // odd lanes: out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)smem, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = f[i];
}
int main() {
    float h[256], *g, *o, ho[256];
    for (int i = 0; i < 256; ++i) h[i] = 1.f + i;
    hipMalloc(&g, sizeof(h)); hipMalloc(&o, sizeof(h));
    hipMemcpy(g, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, g, o, 256);
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    int zeros = 0, stale = 0, good = 0, other = 0;
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e) {
            const float v = ho[l * 4 + e];
            if (l & 1) { if (v == 0.f) ++zeros; else if (v == -7.f) ++stale; else ++other; }
            else { if (v == h[l * 4 + e]) ++good; else ++other; }
        }
    printf("LDS-DMA OOB probe: in-range lanes correct %d/128; out-of-range lanes: zeros %d, stale %d, other %d (of 128)\n", good, zeros, stale, other);
    return 0;
}
