// Probe (not product): throughput of v_mfma_f32_32x32x2_f32 when W waves share a SIMD, each issuing a chain of DEPENDENT MFMAs
// (one accumulator, what a 64x64 block tile gives each wave) or alternating between 2 / 4 independent accumulators; no memory traffic.
// Answers: is the ~85 % plateau of the conv kernels' MFMA pipe a property of multi-wave dependent chains?
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_f32_waves.hip -o scripts/probes/bin/mfma_f32_waves
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, int VALU>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    extern __shared__ char pad[];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    float v = a;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u % NACC], 0, 0, 0);
            if (VALU) {                       // VALU instructions between the MFMAs (address math in the real kernel)
#pragma unroll
                for (int q = 0; q < VALU; ++q) v = v * 1.0001f + b;
            }
        }
    }
    float s = v;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) pad[0] = 0;
}
template <int NACC, int VALU>
double run(int blocks_per_cu, int iters) {
    float* out; hipMalloc(&out, 4096);
    const int lds = 160 * 1024 / blocks_per_cu - 512;           // LDS footprint fixes the residency
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<NACC, VALU>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL((k<NACC, VALU>), dim3(grid), dim3(256), lds, 0, out, iters, 1.f, 2.f);
    hipDeviceSynchronize();
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<NACC, VALU>), dim3(grid), dim3(256), lds, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<NACC, VALU>), dim3(grid), dim3(256), lds, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    const double flop = (double)reps * grid * 4 /*waves*/ * iters * 16.0 * (2.0 * 32 * 32 * 2);
    return flop / (ms * 1e-3) / 1e12;
}
int main() {
    printf("v_mfma_f32_32x32x2_f32, W waves per SIMD (= blocks per CU), TFLOP/s (peak 157.3 at 2.4 GHz)\n");
    printf("waves/SIMD | 1 acc (dependent chain) | 2 acc | 4 acc | 1 acc + 2 VALU per MFMA | 1 acc + 6 VALU per MFMA\n");
    for (int w = 1; w <= 8; ++w) {
        const int it = 3000 / w;
        printf("%d | %.1f | %.1f | %.1f | %.1f | %.1f\n", w, run<1, 0>(w, it), run<2, 0>(w, it), run<4, 0>(w, it), run<1, 2>(w, it), run<1, 6>(w, it));
    }
    return 0;
}
