// pw16.hip -- streaming pointwise (1x1, stride 1) convolution on bf16 tensors for gfx950, with the batch-norm passes of the
// graph nets' conv -> BN -> ReLU chains (nets/resnet.py:47-61, nets/resnext.py:34-67) folded into its loader and its epilogue.
//
//   OUT[M, N] = T(A)[M, K] * W[N, K]^T        M = n * h * w pixels (10^4 .. 10^5 rows), K, N = 64 .. 512 channels
//
// Why a separate kernel: these products are HBM-bound by an order of magnitude (K = 128, N = 256: 85 FLOP per byte against the
// 312 of the bf16 ridge), but the tile kernels of igemm16.hip -- built for 3x3 layers with K = 9 * cin -- see them as one or two
// K-steps between a prologue and an epilogue and run them at 1.4 - 2.9 TB/s (profiles/r3_resnext50_*).  Here the layer is a
// STREAM of 32-row tiles through waves that never synchronise with each other:
//   * the whole filter slice W[NB][K] (<= 64 KB) is put into LDS once per block and stays there: no per-tile filter traffic,
//     no barrier in the steady state;
//   * a wave fetches its next tile's rows (32 x K bf16, whole rows: 128 .. 512 contiguous bytes) into registers while it
//     multiplies the current one -- 4 .. 16 KB in flight per wave, 64 KB per CU, which is what an HBM-latency stream needs;
//   * the loader may TRANSFORM what it fetched before it becomes the MFMA operand, and write the transformed rows back
//     (SIDE) for the filter-gradient kernel: PRO_FWD y = relu(scale[k] * z + shift[k]) -- the normalise pass of the batch norm
//     in front of the conv -- and PRO_BWD dz = A[k] * g + B[k] * z + C0[k] -- the apply pass of the batch norm BEHIND the conv in
//     the backward walk.  The bn_apply / bn_bwd_apply launches and their extra read of the tensor disappear;
//   * the epilogue leaves the statistics of the rows it stores (EPI_STATS: n, mean, M2 per column from shifted sums -- the
//     forward statistics pass of the NEXT batch norm) or applies the ReLU mask of the batch norm below and leaves its two backward
//     sums (EPI_BN), exactly as igemm_dev.h's BNM epilogue does; one partial row per BLOCK (the waves' sums meet in LDS, Chan's
//     merge in wave order), so the finalize kernels see <= 512 rows;
//   * results leave through a per-wave LDS stage as whole 128 / 256-byte row pieces (16 bytes per lane).
// Arithmetic: bf16 operands, fp32 accumulate (v_mfma_f32_32x32x16_bf16); every stored value rounded once, to nearest even.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "pw16.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pk2(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ float lo16(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float hi16(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ float rbf(float a) {      // the value a bf16 store of `a` leaves in memory
    return __builtin_bit_cast(float, (unsigned)__builtin_bit_cast(unsigned short, (__bf16)a) << 16);
}
__device__ __forceinline__ void unpack8(const u32x4& h, float (&d)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { d[2 * e] = lo16(h[e]); d[2 * e + 1] = hi16(h[e]); }
}
__device__ __forceinline__ u32x4 pack8(const float (&d)[8]) {
    return u32x4{pk2(d[0], d[1]), pk2(d[2], d[3]), pk2(d[4], d[5]), pk2(d[6], d[7])};
}
__device__ __forceinline__ void chan_merge(float& n, float& mean, float& m2, float nb, float meanb, float m2b) {
    if (nb == 0.f) return;
    if (n == 0.f) { n = nb; mean = meanb; m2 = m2b; return; }
    const float tot = n + nb, d = meanb - mean;
    mean += d * (nb / tot);
    m2 += m2b + d * d * (n * nb / tot);
    n = tot;
}

constexpr unsigned OOB = 0x80000000u;      // beyond every resource's 2 GiB: the load returns zeros, the store is dropped

// K: reduction length (channels of A); NB: output columns per block; NW: waves per block
template <int K, int NB, int NW, int PRO, int EPI>
__global__ __launch_bounds__(64 * NW) void pw16_kernel(const Pw16Params p) {
    constexpr int CPR = K / 8;                      // 16-byte chunks per operand row
    constexpr int NLD = K / 16;                     // 16-byte pieces a lane fetches per 32-row tile
    constexpr int RPI = 64 / CPR;                   // rows one fetch instruction covers
    constexpr int JB = NB / 32;                     // 32-column accumulator blocks
    constexpr int NTH = 64 * NW;
    constexpr int ABYTES = 32 * K * 2;
    constexpr int STG = ABYTES > 8192 ? ABYTES : 8192;      // per-wave stage: the tile's A rows, later 8 KB of its results
    static_assert(K == 64 || K == 128 || K == 256, "row of 8 / 16 / 32 chunks");
    static_assert(NB % 64 == 0 && NB <= 256, "column tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Wsh = smem;                                           // [NB][K] bf16, chunk c of row r at slot swz(r, c)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    char* stg = smem + NB * K * 2 + wv * STG;

    auto swz = [](int row, int chunk) -> int { return CPR >= 16 ? (chunk ^ (row & 15)) : (chunk ^ ((row >> 1) & 7)); };

    // block -> (row block, column tile): the column tiles of one row block sit on ONE XCD (ids 8 apart), so that the rows both of
    // them read leave HBM once
    int rb, ct;
    {
        const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
        const int per = (p.nrb + 7) / 8;                        // row blocks per XCD (the grid is per * 8 * nct blocks)
        ct = j % p.nct;
        rb = (j / p.nct) + xcd * per;
    }
    const int n0 = ct * NB;
    const bool live = rb < p.nrb;


    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.A), 0, (unsigned)((long)p.M * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(PRO == PW_PRO_BWD ? p.A2 : p.A), 0,
                                                                         (unsigned)((long)p.M * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc(p.OUT, 0, (unsigned)((long)p.M * p.N * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc(p.SIDE ? p.SIDE : p.OUT, 0, (unsigned)((long)p.M * K * 2), 0x00020000);

    // loader geometry: piece i of a tile is chunk `lch` of row i * RPI + lrow
    const int lrow = lane / CPR, lch = lane - lrow * CPR;
    float pc0[8], pc1[8], pc2[8];
    if constexpr (PRO != PW_PRO_NONE) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            pc0[e] = p.c0[lch * 8 + e];
            pc1[e] = p.c1[lch * 8 + e];
            pc2[e] = PRO == PW_PRO_BWD ? p.c2[lch * 8 + e] : 0.f;
        }
    }
    const int ntile = (p.M + 31) >> 5;
    const int tstep = p.nrb * NW;
    int t = rb * NW + wv;
    u32x4 areg[NLD], breg[PRO == PW_PRO_BWD ? NLD : 1];
    auto fetch = [&](int tt) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const long row = (long)tt * 32 + i * RPI + lrow;
            const unsigned off = row < p.M ? (unsigned)(row * (K * 2) + lch * 16) : OOB;
            areg[i] = __builtin_amdgcn_raw_buffer_load_b128(rA, off, 0, 0);
            if constexpr (PRO == PW_PRO_BWD) breg[i] = __builtin_amdgcn_raw_buffer_load_b128(rA2, off, 0, 0);
        }
    };
    if (live && t < ntile) fetch(t);                             // the first tile's rows are on their way before anything else

    // ---- the filter slice -> LDS, once -----------------------------------------------------------------------------------------
    for (int idx = tid; idx < NB * CPR; idx += NTH) {
        const int row = idx / CPR, ch = idx - row * CPR;
        const u32x4 v = *reinterpret_cast<const u32x4*>(p.W + (long)(n0 + row) * K + ch * 8);
        *reinterpret_cast<u32x4*>(Wsh + row * (K * 2) + (swz(row, ch) << 4)) = v;
    }
    __syncthreads();

    // running sums of this wave (EPI_STATS: per accumulator block -- lane = column: n, shifted sum, shifted sum of squares, shift;
    // EPI_BN: per 64-column pass and lane -- 8 columns x (sum g, sum g xhat))
    float st_n = 0.f, st_s[JB], st_q[JB], st_sh[JB];
    float bn_g[EPI == PW_EPI_BN ? NB / 64 : 1][8], bn_x[EPI == PW_EPI_BN ? NB / 64 : 1][8];
#pragma unroll
    for (int j = 0; j < JB; ++j) { st_s[j] = 0.f; st_q[j] = 0.f; st_sh[j] = 0.f; }
    if constexpr (EPI == PW_EPI_BN) {
#pragma unroll
        for (int q = 0; q < NB / 64; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) { bn_g[q][e] = 0.f; bn_x[q][e] = 0.f; }
    }

    for (; live && t < ntile; t += tstep) {
        // ---- this tile's rows: (transform,) side store, LDS image ----------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int row = i * RPI + lrow;
            u32x4 v = areg[i];
            if constexpr (PRO == PW_PRO_FWD) {
                float x[8];
                unpack8(v, x);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = fmaxf(__builtin_fmaf(x[e], pc0[e], pc1[e]), 0.f);
                v = pack8(x);
            } else if constexpr (PRO == PW_PRO_BWD) {
                float g[8], z[8];
                unpack8(v, g);
                unpack8(breg[i], z);
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] = pc0[e] * g[e] + pc1[e] * z[e] + pc2[e];
                v = pack8(g);
            }
            if constexpr (PRO != PW_PRO_NONE) {
                const long grow = (long)t * 32 + row;
                if (p.SIDE && ct == 0)
                    __builtin_amdgcn_raw_buffer_store_b128(v, rS, grow < p.M ? (unsigned)(grow * (K * 2) + lch * 16) : OOB, 0, 0);
            }
            *reinterpret_cast<u32x4*>(stg + row * (K * 2) + (swz(row, lch) << 4)) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // lanes read each other's rows back as MFMA operands
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (t + tstep < ntile) fetch(t + tstep);                 // the next tile's rows travel while this one multiplies

        // ---- 32 x NB product ---------------------------------------------------------------------------------------------------------
        f32x16 acc[JB];
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        // (fragment address: row li, chunk 2 ks + lh, swizzled -- the row term is the same for A and every W block, the W blocks are
        // immediate offsets; one K-step's fragments at a time: left alone, the scheduler hoists all K / 16 x NB / 32 reads)
        const char* fa = stg + li * (K * 2);
        const char* fb = Wsh + li * (K * 2);
#pragma unroll
        for (int ks = 0; ks < K / 16; ++ks) {
            const int so = swz(li, 2 * ks + lh) << 4;
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(fa + so);
#pragma unroll
            for (int j = 0; j < JB; ++j) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(fb + j * 32 * (K * 2) + so);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const int rows_here = min(32, p.M - t * 32);             // valid rows of this tile
        __builtin_amdgcn_sched_barrier(0);

        if constexpr (EPI != PW_EPI_BN) {
            // ---- statistics from the accumulators (lane = column; rows (r & 3) + 8 (r >> 2) + 4 lh), of the ROUNDED values --------------
            if constexpr (EPI == PW_EPI_STATS) {
#pragma unroll
                for (int j = 0; j < JB; ++j) {
                    if (st_n == 0.f) st_sh[j] = rbf(acc[j][0]);      // (lh = 1 lanes: row 4 -- any stored value of the column will do)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const bool in = (r & 3) + 8 * (r >> 2) + 4 * lh < rows_here;
                        const float d = rbf(acc[j][r]) - st_sh[j];
                        st_s[j] += in ? d : 0.f;
                        st_q[j] += in ? d * d : 0.f;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                int cnt = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) cnt += ((r & 3) + 8 * (r >> 2) + 4 * lh < rows_here) ? 1 : 0;
                st_n += (float)cnt;
            }
            // ---- results: bf16 through the wave's stage, 128 columns at a time, out as 256-byte row pieces ----------------------------
#pragma unroll
            for (int hf = 0; hf < (NB + 127) / 128; ++hf) {
                constexpr int HC = NB >= 128 ? 128 : 64;            // columns per pass
                constexpr int HCH = HC / 8;                         // 16-byte chunks per staged row
#pragma unroll
                for (int jj = 0; jj < HC / 32; ++jj) {
                    const int j = hf * (HC / 32) + jj;
                    const int col = jj * 32 + li;
                    // element (row, col) -> row * HC * 2 + (((col >> 3) ^ (row & (HCH - 1))) << 4) + (col & 7) * 2 with row = (r & 3) +
                    // 8 (r >> 2) + 4 lh: the lane's part once, the register's part an XOR constant and an immediate offset
                    const int lanepart = 4 * lh * (HC * 2) + ((((col >> 3) ^ ((4 * lh) & (HCH - 1))) << 4) | ((col & 7) << 1));
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        constexpr int dummy = 0; (void)dummy;
                        const int rrow = (r & 3) + 8 * (r >> 2);
                        *reinterpret_cast<unsigned short*>(stg + rrow * (HC * 2) + (lanepart ^ ((rrow & (HCH - 1)) << 4))) =
                            __builtin_bit_cast(unsigned short, (__bf16)acc[j][r]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                constexpr int RPP = 64 / HCH;                       // rows per store instruction
                const int rr = lane / HCH, rc = lane - rr * HCH;
#pragma unroll
                for (int it = 0; it < 32 / RPP; ++it) {
                    const int row = it * RPP + rr;
                    u32x4 v = *reinterpret_cast<const u32x4*>(stg + row * (HC * 2) + ((rc ^ (row & (HCH - 1))) << 4));
                    const long grow = (long)t * 32 + row;
                    const long o = grow * p.N + n0 + hf * HC + rc * 8;
                    if constexpr (EPI == PW_EPI_PLAIN) {
                        if (p.ADD && grow < p.M) {
                            float x[8], a8[8];
                            unpack8(v, x);      // (PLAIN + addin rounds twice -- not used by the nets: the BN epilogue below is)
                            unpack8(*reinterpret_cast<const u32x4*>(p.ADD + o), a8);
#pragma unroll
                            for (int e = 0; e < 8; ++e) x[e] += a8[e];
                            v = pack8(x);
                        }
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(v, rO, grow < p.M ? (unsigned)(o * 2) : OOB, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        } else {
            // ---- data gradient landing on a BN (+ ReLU) output: fp32 through the stage, 64 columns at a time; the row-coalesced pass
            // adds the other consumer's gradient, rounds, masks, sums and stores ----------------------------------------------------------
            const int rr = lane >> 3, rc = lane & 7;                // row of eight, 8-column chunk
#pragma unroll
            for (int q = 0; q < NB / 64; ++q) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = 2 * q + jj, col = jj * 32 + li;
                    const int lanepart = 4 * lh * 256 + ((((col >> 2) ^ (4 * lh)) << 4) | ((col & 3) << 2));
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rrow = (r & 3) + 8 * (r >> 2);
                        *reinterpret_cast<float*>(stg + rrow * 256 + (lanepart ^ ((rrow & 15) << 4))) = acc[j][r];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                int c0 = n0 + q * 64 + rc * 8;
                asm volatile("" : "+v"(c0));                       // per tile: hoisted out of the tile loop the 32 coefficients stay live throughout
                float mu[8], rs[8], sc[8], sh[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    mu[e] = p.mu[c0 + e]; rs[e] = p.rs[c0 + e];
                    sc[e] = p.sc ? p.sc[c0 + e] : 0.f; sh[e] = p.sc ? p.sh[c0 + e] : 0.f;
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = it * 8 + rr;
                    const long grow = (long)t * 32 + row;
                    const bool in = grow < p.M;
                    const long o = (in ? grow : 0) * p.N + c0;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(stg + row * 256 + (((2 * rc) ^ (row & 15)) << 4));
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(stg + row * 256 + (((2 * rc + 1) ^ (row & 15)) << 4));
                    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    float zm[8], zx[8];
                    if (p.ADD) {
                        float a8[8];
                        unpack8(*reinterpret_cast<const u32x4*>(p.ADD + o), a8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += a8[e];
                    }
                    unpack8(*reinterpret_cast<const u32x4*>(p.Zm + o), zm);
                    if (p.Zx) unpack8(*reinterpret_cast<const u32x4*>(p.Zx + o), zx);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float g = rbf(v[e]);
                        const float zb = p.Zx ? zx[e] : zm[e];
                        const bool on = in && (p.sc ? __builtin_fmaf(zm[e], sc[e], sh[e]) > 0.f : (p.Zx ? zm[e] > 0.f : true));
                        g = on ? g : 0.f;
                        v[e] = g;
                        bn_g[q][e] += g;
                        bn_x[q][e] += g * ((zb - mu[e]) * rs[e]);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(pack8(v), rO, in ? (unsigned)(o * 2) : OOB, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }

    // ---- one partial row per block: the waves' sums meet in LDS (merge order = wave order) ---------------------------------------------
    if constexpr (EPI == PW_EPI_STATS) {
        __syncthreads();                                            // every wave is past its last stage access
        float* red = reinterpret_cast<float*>(smem + NB * K * 2);   // [NW][3][NB] floats <= NW * 3 KB: inside the stages
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            float n = st_n, mean = 0.f, m2 = 0.f;
            if (n > 0.f) { mean = st_sh[j] + st_s[j] / n; m2 = fmaxf(st_q[j] - st_s[j] * st_s[j] / n, 0.f); }
            const float nb = __shfl_xor(n, 32), mb = __shfl_xor(mean, 32), qb = __shfl_xor(m2, 32);
            if (lh == 0) {
                chan_merge(n, mean, m2, nb, mb, qb);
                float* rp = red + (wv * 3) * NB + j * 32 + li;
                rp[0] = n; rp[NB] = mean; rp[2 * NB] = m2;
            }
        }
        __syncthreads();
        if (live)
            for (int c = tid; c < NB; c += NTH) {
                float n = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) chan_merge(n, mean, m2, red[(w * 3) * NB + c], red[(w * 3 + 1) * NB + c], red[(w * 3 + 2) * NB + c]);
                float* pp = p.part + (long)rb * 3 * p.N + n0 + c;
                pp[0] = n; pp[p.N] = mean; pp[2 * (long)p.N] = m2;
            }
    } else if constexpr (EPI == PW_EPI_BN) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem + NB * K * 2);   // [NW][2][NB]
#pragma unroll
        for (int q = 0; q < NB / 64; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float a = bn_g[q][e], b = bn_x[q][e];
                a += __shfl_xor(a, 8); b += __shfl_xor(b, 8);
                a += __shfl_xor(a, 16); b += __shfl_xor(b, 16);
                a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
                if (lane < 8) {
                    const int c = q * 64 + lane * 8 + e;
                    red[(wv * 2) * NB + c] = a;
                    red[(wv * 2 + 1) * NB + c] = b;
                }
            }
        __syncthreads();
        if (live)
            for (int c = tid; c < NB; c += NTH) {
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) { a += red[(w * 2) * NB + c]; b += red[(w * 2 + 1) * NB + c]; }
                p.part[(long)rb * p.N + n0 + c] = a;
                p.pgx[(long)rb * p.N + n0 + c] = b;
            }
    }
}

template <int K, int NB, int NW, int PRO, int EPI>
hipError_t launch_one(const Pw16Params& p, hipStream_t st) {
    constexpr int ABYTES = 32 * K * 2;
    constexpr int STG = ABYTES > 8192 ? ABYTES : 8192;
    const size_t lds = (size_t)NB * K * 2 + (size_t)NW * STG;
    auto kern = pw16_kernel<K, NB, NW, PRO, EPI>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int per = (p.nrb + 7) / 8;
    hipLaunchKernelGGL(kern, dim3(per * 8 * p.nct), dim3(64 * NW), lds, st, p);
    return hipGetLastError();
}

template <int K, int NB, int NW>
hipError_t launch_modes(const Pw16Params& p, int pro, int epi, hipStream_t st) {
    if (epi == PW_EPI_STATS) {
        if (pro == PW_PRO_NONE) return launch_one<K, NB, NW, PW_PRO_NONE, PW_EPI_STATS>(p, st);
        if (pro == PW_PRO_FWD) return launch_one<K, NB, NW, PW_PRO_FWD, PW_EPI_STATS>(p, st);
    } else if (epi == PW_EPI_BN) {
        if (pro == PW_PRO_NONE) return launch_one<K, NB, NW, PW_PRO_NONE, PW_EPI_BN>(p, st);
        if (pro == PW_PRO_BWD) return launch_one<K, NB, NW, PW_PRO_BWD, PW_EPI_BN>(p, st);
    } else {
        if (pro == PW_PRO_NONE) return launch_one<K, NB, NW, PW_PRO_NONE, PW_EPI_PLAIN>(p, st);
        if (pro == PW_PRO_FWD) return launch_one<K, NB, NW, PW_PRO_FWD, PW_EPI_PLAIN>(p, st);
        if (pro == PW_PRO_BWD) return launch_one<K, NB, NW, PW_PRO_BWD, PW_EPI_PLAIN>(p, st);
    }
    return hipErrorInvalidValue;
}

// column tile / waves per K: the filter slice NB x K x 2 B <= 64 KB, the wave stages beside it inside 160 KB
inline int nb_for(int K, int N, int epi) {
    // what fits 256 registers at two waves per SIMD (K = 64 / 128: eight waves per block) without spilling -- the statistics form
    // with 128 accumulators only at K = 64, the BN form (16 running sums per 64 columns and lane, four coefficient vectors) with 32;
    // K = 256 runs four waves per block with the whole register file
    int nb = K == 64 ? 256 : 128;
    if (epi == PW_EPI_BN && K != 256) nb = 64;
    while (nb > 64 && N % nb) nb >>= 1;
    return nb;
}

}  // namespace

bool pw16_plan(long M, int K, int N, int epi, Pw16Params* p) {
    static const bool off = getenv("FTE_PW16") && atoi(getenv("FTE_PW16")) == 0;      // A/B hook
    if (off || (K != 64 && K != 128 && K != 256) || N < 64 || N % 64 || M < 32 || M * (long)(K > N ? K : N) * 2 >= ((long)1 << 31)) return false;
    // K = 256: the 128-column slice (64 KB) leaves room for four 16 KB wave stages -- ONE wave per SIMD, every phase of a tile exposed.  A
    // 64-column slice admits eight (32 + 128 KB): 28x28 256->128 at 128 images 34.4 -> 27.9 us, 256->256 44.0 -> 38.0 (the rows are read by
    // two column tiles of one XCD instead of one).  The BN-backward forms keep four waves: with eight they spill.  FTE_PW16_K256=0: as before.
    static const bool k256_wide = getenv("FTE_PW16_K256") && atoi(getenv("FTE_PW16_K256")) == 0;
    int nb = nb_for(K, N, epi);
    const bool eight256 = K == 256 && !k256_wide && epi != PW_EPI_BN;
    if (eight256) nb = 64;
    if (N % nb) return false;
    const int nw = (K == 256 && !eight256) ? 4 : 8;
    p->nct = N / nb;
    const long tiles = (M + 31) / 32;
    long nrb = (tiles + nw - 1) / nw;                 // at least one tile per wave ...
    // ONE block per CU, its waves walking several tiles each with the next tile's rows in flight: measured on MI355X at 128 images
    // (kernel + finalize, us; 128 / 192 / 256 / 384 / 512 blocks): 28x28 64->256 39.8 / 35.2 / 31.2 / 38.7 / 39.2, 128->256 39.1 / 32.6 /
    // 29.9 / 36.1 / 38.9, 256->256 63.5 / 50.8 / 44.9 / 56.0 / 54.0 -- a second round of blocks starts cold, fewer blocks leave CUs idle
    static const long want_env = getenv("FTE_PW16_BLOCKS") ? atol(getenv("FTE_PW16_BLOCKS")) : 256;      // tuning hook
    const long want = want_env / p->nct > 0 ? want_env / p->nct : 1;      // ... and about two blocks' worth of row blocks per CU over all column tiles
    if (nrb > want) nrb = want;
    if (nrb > 512) nrb = 512;                         // partial rows the finalize kernels take in one pass
    p->nrb = (int)nrb;
    p->M = (int)M; p->K = K; p->N = N;
    return true;
}

// waves per block of the instantiation pw16_launch picks (the launch records spell the symbol with it)
int pw16_waves(const Pw16Params& p, int pro, int epi) {
    if (p.K != 256) return 8;
    static const bool k256_wide = getenv("FTE_PW16_K256") && atoi(getenv("FTE_PW16_K256")) == 0;
    return (p.N / p.nct == 64 && !k256_wide && pro != PW_PRO_BWD && epi != PW_EPI_BN) ? 8 : 4;
}

hipError_t pw16_launch(const Pw16Params& p, int pro, int epi, hipStream_t st) {
    const int nb = p.N / p.nct;
    if (p.K == 64) {
        if (nb == 256) return launch_modes<64, 256, 8>(p, pro, epi, st);
        if (nb == 128) return launch_modes<64, 128, 8>(p, pro, epi, st);
        return launch_modes<64, 64, 8>(p, pro, epi, st);
    }
    if (p.K == 128) {
        if (nb == 256) return launch_modes<128, 256, 8>(p, pro, epi, st);
        if (nb == 128) return launch_modes<128, 128, 8>(p, pro, epi, st);
        return launch_modes<128, 64, 8>(p, pro, epi, st);
    }
    if (p.K == 256) {
        static const bool k256_wide = getenv("FTE_PW16_K256") && atoi(getenv("FTE_PW16_K256")) == 0;
        if (nb == 128) return launch_modes<256, 128, 4>(p, pro, epi, st);
        if (!k256_wide && pro != PW_PRO_BWD && epi != PW_EPI_BN) return launch_modes<256, 64, 8>(p, pro, epi, st);
        return launch_modes<256, 64, 4>(p, pro, epi, st);
    }
    return hipErrorInvalidValue;
}
