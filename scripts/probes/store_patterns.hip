// Probe (not product): what does a 16-byte-per-lane global store / load cost when a wave's lanes address 64 different rows (the
// swapped-operand epilogue of igemm16rw: lane = output row, half-waves on adjacent 16-byte chunks) against 8 lanes per 128-byte row
// (row-coalesced, what an LDS transposition of the accumulators would give)?  One 512-thread block per CU, every wave walks its own
// 64-row x 128-byte slabs of a [rows][256 B] bf16 tensor (N = 128): 8 instructions per slab either way.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/store_patterns.hip -o /tmp/store_patterns && /tmp/store_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int PATTERN, bool LOAD>
__global__ __launch_bounds__(512) void k(u32x4* buf, long rows, int slabs_per_wave, unsigned* sink) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const long wave = (long)blockIdx.x * 8 + wid, nwaves = (long)gridDim.x * 8;
    u32x4 acc = {0u, 0u, 0u, 0u};
    const u32x4 val = {(unsigned)lane, 1u, 2u, 3u};
    for (int s = 0; s < slabs_per_wave; ++s) {
        // slab: 64 rows x 128 B = the wave's 64-column half of 64 rows of the 256-byte-row tensor
        const long slab = wave + (long)s * nwaves;
        const long row0 = (slab >> 1) * 64 % rows;
        const int half = (int)(slab & 1);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            long row; int chunk;                       // 16-byte chunk 0..7 of the slab's 128-byte row segment
            if (PATTERN == 0) { row = row0 + (p >> 2) * 32 + li; chunk = (p & 3) * 2 + lh; }
            else { row = row0 + p * 8 + (lane >> 3); chunk = lane & 7; }
            u32x4* q = buf + (row * 256 + half * 128 + chunk * 16) / 16;
            if (LOAD) acc ^= *q; else *q = val;
        }
    }
    if (LOAD && acc[0] == 0x12345u) sink[threadIdx.x] = acc[1];
}

template <int PATTERN, bool LOAD>
static void run(u32x4* buf, long rows, unsigned* sink, int cus, const char* name) {
    const int spw = 400;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<PATTERN, LOAD>), dim3(cus), dim3(512), 0, 0, buf, rows, spw, sink);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<PATTERN, LOAD>), dim3(cus), dim3(512), 0, 0, buf, rows, spw, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double bytes = (double)cus * 8 * spw * 8192;
    printf("%-44s %8.3f ms  %7.1f GB/s  %6.1f cycles per instruction and CU at 2.4 GHz\n", name, ms, bytes / ms / 1e6, ms * 1e-3 * 2.4e9 / (8.0 * spw * 8));
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const long rows = 1 << 21;                      // 512 MiB tensor
    u32x4* buf; unsigned* sink;
    hipMalloc(&buf, rows * 256); hipMalloc(&sink, 4096);
    hipMemset(buf, 0, rows * 256);
    for (int div = 1; div <= 8; div *= 8) {         // every CU busy (the HBM share bounds both), then an eighth of them (the CU side shows)
        printf("-- %d blocks (one per CU)\n", cus / div);
        run<0, false>(buf, rows, sink, cus / div, "store, lane = row (64 rows per instruction)");
        run<1, false>(buf, rows, sink, cus / div, "store, 8 lanes per 128-byte row");
        run<0, true>(buf, rows, sink, cus / div, "load,  lane = row (64 rows per instruction)");
        run<1, true>(buf, rows, sink, cus / div, "load,  8 lanes per 128-byte row");
    }
    return 0;
}
