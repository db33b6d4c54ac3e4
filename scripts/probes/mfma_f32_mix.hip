// Probe (not product): what does each kind of instruction cost the matrix pipe when it sits between DEPENDENT
// v_mfma_f32_32x32x2_f32 (6 waves per SIMD, the conv kernels' regime)?  Per 16 MFMAs one wave issues: nothing | 24 VALU |
// 6 ds_read_b128 | 4 ds_write_b128 | 4 buffer_load_dwordx4 | 12 SALU | 2 s_barrier | the conv loop's whole mix.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_f32_mix.hip -o scripts/probes/bin/mfma_f32_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* src, int iters, float a0, float b0) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f, v = a;
    f32x4 q = {a, b, a, b}, ld = q;
    f32x4 tq[4] = {q, q, q, q};
    float tb[16];
    for (int i = 0; i < 16; ++i) tb[i] = b;
    int sc = blockIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1 << 20, 0x00020000);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            if (MODE == 1 || MODE == 7) { v = v * 1.0001f + b; if (u & 1) v = v * 1.0002f + a; }                 // 24 VALU per 16 MFMAs
            if ((MODE == 2 || MODE == 7) && u < 6) { q += *reinterpret_cast<f32x4*>(lds + ((threadIdx.x * 4 + u * 1024) & 4095)); }
            // LDS reads whose results feed MFMA operands 4+ MFMAs later (software-pipelined like the conv loop: no VALU, no early wait)
            if (MODE == 8 && u < 4) tq[u] = *reinterpret_cast<f32x4*>(lds + ((threadIdx.x * 4 + u * 1024) & 4095));
            if (MODE == 9 && u < 8) tq[u & 3] += *reinterpret_cast<f32x4*>(lds + ((threadIdx.x * 4 + u * 1024) & 4095));
            if (MODE == 10 && u < 12) { if (u < 4) tq[u] = *reinterpret_cast<f32x4*>(lds + ((threadIdx.x * 4 + u * 1024) & 4095));
                                        else { tb[(u - 4) * 2] = lds[(threadIdx.x + u * 64) & 4095]; tb[(u - 4) * 2 + 1] = lds[(threadIdx.x + u * 64 + 2048) & 4095]; } }
            if ((MODE == 8 || MODE == 10) && u >= 8) a = tq[(u - 8) >> 1][(u - 8) & 3];
            if (MODE == 10 && u >= 8) b = tb[(u - 8) * 2];
            if ((MODE == 3 || MODE == 7) && u >= 12) { *reinterpret_cast<f32x4*>(lds + ((threadIdx.x * 4 + u * 1024) & 4095)) = ld; }
            if ((MODE == 4 || MODE == 7) && u < 4) { ld = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (threadIdx.x * 16 + u * 4096 + (it & 15) * 16384) & 0xfffff, 0, 0)); }
            if ((MODE == 5 || MODE == 7) && u < 12) { sc = sc * 3 + u; asm volatile("" : "+s"(sc)); }
            if ((MODE == 6 || MODE == 7) && (u == 11 || u == 15)) __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    for (int i = 0; i < 4; ++i) q += tq[i];
    for (int i = 0; i < 16; ++i) v += tb[i];
    float s = v + q[0] + q[1] + q[2] + q[3] + ld[0] + (float)sc;
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
template <int MODE>
double run(int blocks_per_cu, int iters, float* src) {
    float* out; hipMalloc(&out, 4096);
    const int lds = 160 * 1024 / blocks_per_cu - 512;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * blocks_per_cu;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), lds, 0, out, src, iters, 1.f, 2.f);
    hipEventRecord(e0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), lds, 0, out, src, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return (double)reps * grid * 4 * iters * 16.0 * (2.0 * 32 * 32 * 2) / (ms * 1e-3) / 1e12;
}
int main() {
    float* src; hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20);
    const char* names[12] = {"MFMA only", "+24 VALU", "+6 ds_read_b128", "+4 ds_write_b128", "+4 buffer_load_dwordx4", "+12 SALU", "+2 s_barrier", "all of them (the conv loop's mix)", "+4 ds_read_b128 feeding MFMA operands", "+8 ds_read_b128 (+8 VALU adds)", "+4 ds_read_b128 + 16 ds_read_b32 feeding MFMA operands", "(unused)"};
    printf("dependent v_mfma_f32_32x32x2_f32 chains, per 16 MFMAs of a wave; TFLOP/s (157.3 = peak at 2.4 GHz)\n| mix | 4 waves/SIMD | 6 waves/SIMD |\n|---|---|---|\n");
    double r4[12], r6[12];
    r4[0] = run<0>(4, 600, src); r6[0] = run<0>(6, 400, src);
    r4[1] = run<1>(4, 600, src); r6[1] = run<1>(6, 400, src);
    r4[2] = run<2>(4, 600, src); r6[2] = run<2>(6, 400, src);
    r4[3] = run<3>(4, 600, src); r6[3] = run<3>(6, 400, src);
    r4[4] = run<4>(4, 600, src); r6[4] = run<4>(6, 400, src);
    r4[5] = run<5>(4, 600, src); r6[5] = run<5>(6, 400, src);
    r4[6] = run<6>(4, 600, src); r6[6] = run<6>(6, 400, src);
    r4[7] = run<7>(4, 600, src); r6[7] = run<7>(6, 400, src);
    r4[8] = run<8>(4, 600, src); r6[8] = run<8>(6, 400, src);
    r4[9] = run<9>(4, 600, src); r6[9] = run<9>(6, 400, src);
    r4[10] = run<10>(4, 600, src); r6[10] = run<10>(6, 400, src);
    r4[11] = 0; r6[11] = 0;
    for (int m = 0; m < 12; ++m) printf("| %s | %.1f | %.1f |\n", names[m], r4[m], r6[m]);
    return 0;
}
