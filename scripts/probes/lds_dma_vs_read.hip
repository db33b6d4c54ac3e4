// Probe (not product): do LDS-DMA writes (`buffer_load_dwordx4 ... lds`, L2-resident source) and ds_read_b128 fragment reads share the LDS
// cycle budget of a CU, and what does an MFMA stream get beside them?  One 512-thread block per CU; NL loader waves stream 1-KiB
// pieces into a ring, NR reader waves stream conflict-free ds_read_b128, NM waves issue dependent-free MFMAs.  Rates per CU.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/lds_dma_vs_read.hip -o /tmp/lds_dma_vs_read && /tmp/lds_dma_vs_read
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void lds_void;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// roles by wave id: [0, nl) loaders, [nl, nl + nr) readers, [nl + nr, nl + nr + nm) MFMA waves; the rest exit
// every wave runs until `ticks` of the 100-MHz clock have passed (checked every 16 iterations) and reports its iteration count
__device__ __forceinline__ unsigned long long now100() { return __builtin_amdgcn_s_memrealtime(); }
__global__ __launch_bounds__(512, 2) void k(const float* src, unsigned src_bytes, float* sink, unsigned* counts, int ticks, int nl, int nr, int nm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long t_end = now100() + (unsigned long long)ticks;
    unsigned n = 0;
    if (wid < nl) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, src_bytes, 0x00020000);
        char* ring = smem + wid * 8192;                     // 8 pieces per loader
        unsigned off = (blockIdx.x * 65536u + wid * 8192u) % (src_bytes - 65536u);
        do {
            for (int it = 0; it < 16; ++it) {
#pragma unroll
                for (int pz = 0; pz < 8; ++pz)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(ring + pz * 1024), 16, off + pz * 1024 + lane * 16, 0, 0, 0);
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                off += 65536u;
                if (off >= src_bytes - 65536u) off -= src_bytes - 65536u;
            }
            n += 16;
        } while (now100() < t_end);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (wid < nl + nr) {
        const char* base = smem + 65536 + (wid - nl) * 4096 + lane * 16;      // 64 lanes x 16 B contiguous: conflict-free
        u32x4 acc = {0u, 0u, 0u, 0u};
        do {
            for (int it = 0; it < 16; ++it) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    u32x4 v;
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)base), "n"(0));
                    asm volatile("s_waitcnt lgkmcnt(4)");
                    acc ^= v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)");
            }
            n += 16;
        } while (now100() < t_end);
        if (acc[0] == 0x12345678u) sink[threadIdx.x] = 1.f;
    } else if (wid < nl + nr + nm) {
        f32x16 c0, c1;
        for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)lane; b[e] = (__bf16)1.f; }
        do {
            for (int it = 0; it < 16; ++it) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                }
            }
            n += 16;
        } while (now100() < t_end);
        if (c0[0] + c1[0] == 12345.f) sink[threadIdx.x] = 2.f;
    }
    if (lane == 0) counts[blockIdx.x * 8 + wid] = n;
}

int main() {
    const unsigned src_bytes = 2u << 20;                    // 2 MiB: L2-resident on every XCD
    float *src, *sink;
    unsigned* counts;
    hipMalloc(&src, src_bytes); hipMalloc(&sink, 4096); hipMalloc(&counts, 8 * 1024 * 4);
    hipMemset(src, 0, src_bytes);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, ticks = 30000;          // 300 us
    static unsigned hc[8 * 1024];
    const int cfgs[][3] = {{4, 0, 0}, {8, 0, 0}, {2, 0, 0}, {0, 4, 0}, {0, 8, 0}, {0, 0, 4}, {0, 0, 8}, {4, 4, 0}, {4, 0, 4}, {0, 4, 4}, {2, 2, 4}, {2, 4, 2}, {1, 3, 4}};
    printf("%-22s %12s %12s %12s   (300 us per run, every wave counts its own iterations)\n", "waves (ld, rd, mfma)", "DMA GB/s/CU", "read GB/s/CU", "MFMA TF/CU");
    for (auto& c : cfgs) {
        const int nl = c[0], nr = c[1], nm = c[2];
        hipLaunchKernelGGL(k, dim3(cus), dim3(512), 96 * 1024, 0, src, src_bytes, sink, counts, 2000, nl, nr, nm);
        hipDeviceSynchronize();
        hipMemset(counts, 0, sizeof(hc));
        hipLaunchKernelGGL(k, dim3(cus), dim3(512), 96 * 1024, 0, src, src_bytes, sink, counts, ticks, nl, nr, nm);
        hipDeviceSynchronize();
        hipMemcpy(hc, counts, sizeof(hc), hipMemcpyDeviceToHost);
        double il = 0, ir = 0, im = 0;
        for (int b = 0; b < cus; ++b)
            for (int w = 0; w < 8; ++w) {
                const double v = hc[b * 8 + w];
                if (w < nl) il += v; else if (w < nl + nr) ir += v; else if (w < nl + nr + nm) im += v;
            }
        const double s = ticks / 100e6;
        printf("(%d, %d, %d)%14s %12.1f %12.1f %12.2f\n", nl, nr, nm, "", il / cus * 8192.0 / s / 1e9, ir / cus * 8192.0 / s / 1e9, im / cus * 8.0 * 32768 / s / 1e12);
    }
    return 0;
}
