// kernels.hip -- the HBM-bound kernels of the training step for gfx950:
// first-layer direct conv (+wgrad), ordered reductions, softmax-CE / A-softmax /
// center / triplet heads, flat-arena momentum and Adam.  Wave = 64 lanes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <math.h>

#include "kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// block-wide reductions for 256-thread blocks (4 waves); result valid in every thread
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ float block_max(float v, float* sh) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

// ------------------------------------------------------------------------------------
// First conv (nets/sphere.py:57): Cin in {1,3}, Cout = 64, 3x3, TF-SAME; bound by the 2 x [N,Ho,Wo,64] fp32 writes (z, y).
// out[pixel][co] = sum_k patch(pixel)[k] * W[k][co] with K = 9*Cin <= 27 on v_mfma_f32_32x32x2_f32: 32 pixels on the rows,
// k pairs on the reduction axis, two 32-column halves.  Lane (li, lh) gathers x[patch position 2s + lh] of pixel li for the
// 14 k-steps (14 loads per lane and 32 pixels; a thread-per-(pixels x channel quad) loop re-loaded every x value from 16
// lanes and ran at 31 % of HBM speed) and keeps its 28 weight values W[2s + lh][li], W[2s + lh][32 + li] in registers.
// A wave walks 32-pixel groups of whole output rows.
// ------------------------------------------------------------------------------------
typedef float f32x16f __attribute__((ext_vector_type(16)));
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void conv_first_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ alpha, float* __restrict__ z, float* __restrict__ y,
    unsigned short* __restrict__ z16, unsigned short* __restrict__ y16,      // bf16 storage: z / y rounded once, where they are written
    int n, int h, int wd, int ho, int wo, int stride, int pt, int pl) {
    constexpr int NH = COUT / 32, K = 9 * CIN, KS = (K + 1) / 2;      // NH accumulator blocks of 32 output channels
    static_assert(COUT == 32 || COUT == 64, "cout 32 (the 24-wide ShuffleNet stem, padded) or 64");
    const int lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    float wb[NH][KS];
    int kr[KS], kq[KS], kc[KS];                 // (row tap, column tap, channel) of this lane's k index in every k-step
    bool kv[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int k = 2 * s + lh;
        kv[s] = k < K;
        const int kk = kv[s] ? k : 0;
        const int tap = kk / CIN;
        kc[s] = kk - tap * CIN; kr[s] = tap / 3; kq[s] = tap - 3 * (tap / 3);
#pragma unroll
        for (int e = 0; e < NH; ++e) wb[e][s] = kv[s] ? w[kk * COUT + NH * li + e] : 0.f;
    }
    // column li of accumulator block e is output channel NH * li + e: a lane owns NH ADJACENT channels of a pixel, so its results leave
    // as one 8-byte (fp32) or 4-byte (bf16) store per tensor instead of NH stores 128 bytes apart
    float bb[NH], aa[NH];
#pragma unroll
    for (int e = 0; e < NH; ++e) { bb[e] = bias ? bias[NH * li + e] : 0.f; aa[e] = alpha ? alpha[NH * li + e] : 1.f; }
    const int gpr = (wo + 31) / 32;                                  // 32-pixel groups per output row
    const long ngrp = (long)n * ho * gpr;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    for (long grp = wave; grp < ngrp; grp += nwaves) {
        const int g = (int)(grp % gpr);
        const long row = grp / gpr;
        const int oh = (int)(row % ho), img = (int)(row / ho);
        const int ow = g * 32 + li;
        const bool pok = ow < wo;
        f32x16f acc[NH];
#pragma unroll
        for (int e = 0; e < NH; ++e)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[e][i] = 0.f;
        float av[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int ih = oh * stride + kr[s] - pt, iw = ow * stride + kq[s] - pl;
            const bool ok = pok && kv[s] && ih >= 0 && ih < h && iw >= 0 && iw < wd;
            const float v = x[((long)(img * h + (ok ? ih : 0)) * wd + (ok ? iw : 0)) * CIN + kc[s]];
            av[s] = ok ? v : 0.f;
        }
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int e = 0; e < NH; ++e) acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], wb[e][s], acc[e], 0, 0, 0);
        // C layout: column (channel) = lane & 31, row (pixel) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
        const long obase = ((long)(img * ho + oh) * wo + g * 32) * COUT;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int pr = (i & 3) + 8 * (i >> 2) + 4 * lh;
            if (g * 32 + pr < wo) {
                const long o = obase + (long)pr * COUT + NH * li;
                float zv[NH], yv[NH];
#pragma unroll
                for (int e = 0; e < NH; ++e) {
                    zv[e] = acc[e][i] + bb[e];
                    yv[e] = alpha ? (zv[e] > 0.f ? zv[e] : aa[e] * zv[e]) : zv[e];
                }
                if constexpr (NH == 2) {
                    if (z) *reinterpret_cast<float2*>(z + o) = make_float2(zv[0], zv[1]);
                    if (y) *reinterpret_cast<float2*>(y + o) = make_float2(yv[0], yv[1]);
                    if (z16) *reinterpret_cast<unsigned*>(z16 + o) = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)zv[0]) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)zv[1]) << 16);
                    if (y16) *reinterpret_cast<unsigned*>(y16 + o) = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)yv[0]) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)yv[1]) << 16);
                } else {
                    if (z) z[o] = zv[0];
                    if (y) y[o] = yv[0];
                    if (z16) z16[o] = __builtin_bit_cast(unsigned short, (__bf16)zv[0]);
                    if (y16) y16[o] = __builtin_bit_cast(unsigned short, (__bf16)yv[0]);
                }
            }
        }
    }
}

// dW[k = (r,s,c)][co] = sum_pixels patch(p)[k] * dz[p][co] is a [K <= 27] x [64] x [pixels] product: two
// v_mfma_f32_32x32x2_f32 per PAIR of pixels (k index on the rows, the two pixels on the reduction axis), operands straight
// from global memory: lane (li, lh) gathers the one input element x[patch position li] of pixel p + lh (a few cache lines
// per wave: 3 rows x 9 floats) and the two dz values dz[p + lh][li], dz[p + lh][32 + li].  3 loads per pixel pair instead
// of the 28 loads per pixel of a lane-per-output-channel loop (which ran at 10 % of HBM speed).  A wave owns whole output
// rows; the 4 waves of a block are summed through LDS and each block writes one ordered partial.
typedef float f32x16k __attribute__((ext_vector_type(16)));
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void conv_first_wgrad_kernel(
    const float* __restrict__ x, const float* __restrict__ dz, const unsigned short* __restrict__ dz16, float* __restrict__ part,
    int n, int h, int wd, int ho, int wo, int stride, int pt, int pl, long rows_per_block) {
    constexpr int NH = COUT / 32, K = 9 * CIN;
    __shared__ float red[4][32][COUT];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const bool kok = li < K;
    const int tap = kok ? li / CIN : 0, ch = kok ? li - tap * CIN : 0;
    const int r = tap / 3, sx = tap - r * 3;
    const long nrows = (long)n * ho;                       // output rows (img, oh)
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(nrows, r0 + rows_per_block);
    f32x16k acc[NH];
#pragma unroll
    for (int e = 0; e < NH; ++e)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[e][i] = 0.f;
    // A wave's work is a sequence of TRIPS (4 pixel pairs of one output row); the loads of trip t + 1 are issued before the MFMAs of
    // trip t (issued and awaited inside one trip, the 12 loads cost a round trip of latency per 512 MFMA cycles: the kernel ran at
    // 1.2 TB/s on five waves per SIMD).  Same pixel pairs in the same order: results are bit-identical.
    constexpr int UP = 4;                                   // pixel pairs per trip: 12 independent loads
    const int tpr = (wo + 2 * UP - 1) / (2 * UP);           // trips per output row
    const long myrows = r1 > r0 + wv ? (r1 - r0 - wv + 3) / 4 : 0;
    const long ntrips = myrows * tpr;
    auto fetch = [&](long t, float (&a)[UP], float (&bv)[NH][UP]) {
        const long rk = t / tpr;
        const int ow0 = (int)(t - rk * tpr) * 2 * UP;
        const long row = r0 + wv + 4 * rk;
        const int img = (int)(row / ho), oh = (int)(row - (long)img * ho);
        const int ih = oh * stride + r - pt;
        const bool rok = kok && ih >= 0 && ih < h;
        const float* xrow = x + ((long)(img * h + (rok ? ih : 0)) * wd) * CIN + ch;
        const float* drow = dz ? dz + row * wo * COUT : nullptr;
        const unsigned short* drow16 = dz16 ? dz16 + row * wo * COUT : nullptr;
#pragma unroll
        for (int u = 0; u < UP; ++u) {
            const int ow = ow0 + 2 * u + lh;
            const bool pok = ow < wo;
            const int iw = ow * stride + sx - pl;
            const bool ok = rok && pok && iw >= 0 && iw < wd;
            const float av = xrow[(long)(ok ? iw : 0) * CIN];
            a[u] = ok ? av : 0.f;
            const long doff = (long)(pok ? ow : 0) * COUT + NH * li;      // column li of block e = channel NH * li + e (one load per lane and pixel)
            if constexpr (NH == 2) {
                float v0, v1;
                if (drow) { const float2 tt = *reinterpret_cast<const float2*>(drow + doff); v0 = tt.x; v1 = tt.y; }
                else { const unsigned tt = *reinterpret_cast<const unsigned*>(drow16 + doff); v0 = __builtin_bit_cast(float, tt << 16); v1 = __builtin_bit_cast(float, tt & 0xffff0000u); }
                bv[0][u] = pok ? v0 : 0.f; bv[1][u] = pok ? v1 : 0.f;
            } else {
                const float v = drow ? drow[doff] : __builtin_bit_cast(float, (unsigned)drow16[doff] << 16);
                bv[0][u] = pok ? v : 0.f;
            }
        }
    };
    float a0[UP], b0[NH][UP], a1[UP], b1[NH][UP];
    if (ntrips > 0) fetch(0, a0, b0);
    for (long t = 0; t < ntrips; t += 2) {
        if (t + 1 < ntrips) fetch(t + 1, a1, b1);
#pragma unroll
        for (int u = 0; u < UP; ++u)
#pragma unroll
            for (int e = 0; e < NH; ++e) acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b0[e][u], acc[e], 0, 0, 0);
        if (t + 1 < ntrips) {
            if (t + 2 < ntrips) fetch(t + 2, a0, b0);
#pragma unroll
            for (int u = 0; u < UP; ++u)
#pragma unroll
                for (int e = 0; e < NH; ++e) acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b1[e][u], acc[e], 0, 0, 0);
        }
    }
    // C layout: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = (i & 3) + 8 * (i >> 2) + 4 * lh;
#pragma unroll
        for (int e = 0; e < NH; ++e) red[wv][k][NH * li + e] = acc[e][i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K * COUT; i += 256) {
        const int k = i / COUT, c = i % COUT;
        part[(long)blockIdx.x * K * COUT + i] = (red[0][k][c] + red[1][k][c]) + (red[2][k][c] + red[3][k][c]);
    }
}

// ------------------------------------------------------------------------------------
// Ordered (deterministic) column sums of a [rows, cols] matrix:
//   out[j] = scale * sum_r in[r, j]  (+ bias[j % bmod])
// grid = (ceil(cols/64), row_splits); block = 64 columns x 4 row lanes.  With row_splits > 1 the
// kernel writes partial[rs, cols] and is run a second time over those partials (fixed order).
// ------------------------------------------------------------------------------------
// blockIdx.z = 1 reduces a second matrix of the same shape (in2 -> out2): dalpha and dbias partials in one launch
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          const float* __restrict__ bias, int bmod, long rows,
                                                          long cols, long rows_per_split, float scale,
                                                          const float* __restrict__ in2, float* __restrict__ out2, int act) {
    __shared__ float sh[4][64];
    if (blockIdx.z == 1) { in = in2; out = out2; }
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const long j = (long)blockIdx.x * 64 + c;
    const long r0 = (long)blockIdx.y * rows_per_split;
    const long r1 = min(rows, r0 + rows_per_split);
    float s0 = 0.f, s1 = 0.f;
    if (j < cols) {
        long r = r0 + rl;
        for (; r + 28 < r1; r += 32) {              // eight loads in flight (a row lane's rows are a dependent chain of round trips otherwise)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = in[(r + 4 * u) * cols + j];
            s0 += (v[0] + v[2]) + (v[4] + v[6]);
            s1 += (v[1] + v[3]) + (v[5] + v[7]);
        }
        for (; r + 4 < r1; r += 8) {
            s0 += in[r * cols + j];
            s1 += in[(r + 4) * cols + j];
        }
        if (r < r1) s0 += in[r * cols + j];
    }
    sh[rl][c] = s0 + s1;
    __syncthreads();
    if (rl == 0 && j < cols) {
        float v = ((sh[0][c] + sh[1][c]) + (sh[2][c] + sh[3][c])) * scale;
        if (bias) v += bias[j % bmod];
        if (act == 1) v = fmaxf(v, 0.f);                         // (act_fwd_kernel's two activations, for fte_gemm_nn_act)
        else if (act == 2) v = 1.f / (1.f + expf(-v));
        out[(long)blockIdx.y * cols + j] = v;
    }
}

// Split-K slabs: out[j] = sum_r in[r, j] for a few (8..256) rows of a long (>= 4096, multiple of 4) column range -- the filter-gradient
// partials.  reduce_rows_kernel above reads them with 4-byte loads, two in flight per thread, one thread column per 64: 30-220 us per
// filter gradient of the bf16-storage step (13 % of it).  Here a thread owns FOUR columns (16-byte loads), its row lane takes every
// RL-th row with eight loads in flight, and the RL row lanes of a block meet in LDS in a fixed order.  QB = column quads per block:
// 64 (a wave reads 1 KiB of a row) for long rows, 16 (16 row lanes) when the columns alone would not fill the chip.
template <int QB>
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, int rows, long quads) {
    constexpr int RL = 256 / QB;
    __shared__ f32x4 sh[RL][QB];
    const int cq = threadIdx.x % QB, rl = threadIdx.x / QB;
    const long q = (long)blockIdx.x * QB + cq;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (q < quads) {
        const f32x4* src = in + q;
        int r = rl;
        for (; r + 7 * RL < rows; r += 8 * RL) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(long)(r + u * RL) * quads];
            s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; r < rows; r += RL) s += src[(long)r * quads];
    }
    sh[rl][cq] = s;
    __syncthreads();
    if (rl == 0 && q < quads) {
        f32x4 t = sh[0][cq];
#pragma unroll
        for (int w = 1; w < RL; ++w) t += sh[w][cq];
        out[q] = t;
    }
}

// Tall and narrow partial matrices (the dalpha / dbias partials a data gradient leaves: 1.5-50 k rows of 64-512 columns): 16-byte loads,
// QB column quads x 256 / QB row lanes per block, eight loads in flight per lane, the lanes of a block meet in LDS in a fixed order; blockIdx.y
// takes a row range (the split partials go through the same kernel once more), blockIdx.z = 1 the second matrix.  reduce_rows_kernel read the
// same data with 4-byte loads, one column per thread: 29 us per data gradient of the bf16-storage step at 512 images (0.44 TB/s).
template <int QB>
__global__ __launch_bounds__(256) void reduce_rows_q_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, long rows, long quads,
                                                            long rows_per_split, const f32x4* __restrict__ in2, f32x4* __restrict__ out2) {
    constexpr int RL = 256 / QB;
    __shared__ f32x4 sh[RL][QB];
    if (blockIdx.z == 1) { in = in2; out = out2; }
    const int cq = threadIdx.x % QB, rl = threadIdx.x / QB;
    const long q = (long)blockIdx.x * QB + cq;
    const long r0 = (long)blockIdx.y * rows_per_split, r1 = min(rows, r0 + rows_per_split);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (q < quads) {
        const f32x4* src = in + q;
        long r = r0 + rl;
        for (; r + 7 * RL < r1; r += 8 * RL) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(r + u * RL) * quads];
            s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; r < r1; r += RL) s += src[r * quads];
    }
    sh[rl][cq] = s;
    __syncthreads();
    if (rl == 0 && q < quads) {
        f32x4 t = sh[0][cq];
#pragma unroll
        for (int w = 1; w < RL; ++w) t += sh[w][cq];
        out[(long)blockIdx.y * quads + q] = t;
    }
}

// two-stage scalar reductions (sum / sum of squares): 1024 block partials, then one block
template <bool SQ>
__global__ __launch_bounds__(256) void partial_sum_kernel(const float* __restrict__ a, long n, float* __restrict__ part) {
    __shared__ float sh[4];
    float s = 0.f;
    const long n4 = n >> 2;
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = a4[i];
        s += SQ ? (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]) : (v[0] + v[1]) + (v[2] + v[3]);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = a[(n4 << 2) + threadIdx.x];
        s += SQ ? v * v : v;
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void final_sum_kernel(const float* __restrict__ part, int np, float scale, float* out) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < np; i += 256) s += part[i];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) out[0] = s * scale;
}

// ------------------------------------------------------------------------------------
// Softmax cross-entropy, forward + gradient in one launch; one block per row.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                                         float* __restrict__ loss_rows, float* __restrict__ dlogits,
                                                         int c, int ld, float gscale) {
    __shared__ float sh[4];
    const int row = blockIdx.x;
    const float* z = logits + (long)row * ld;
    float* d = dlogits + (long)row * ld;
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < c; j += 256) mx = fmaxf(mx, z[j]);
    mx = block_max(mx, sh);
    float s = 0.f;
    for (int j = threadIdx.x; j < c; j += 256) s += expf(z[j] - mx);
    s = block_sum(s, sh);
    // a label outside [0, c) (list / num_classes mismatch) must not become an out-of-bounds read: the row's loss is NaN
    // (TF's sparse_softmax_cross_entropy gives NaN loss rows on GPU), so the caller's non-finite-loss check trips
    const int yl = labels[row];
    const bool bad = (unsigned)yl >= (unsigned)c;
    const int y = bad ? 0 : yl;
    const float inv = 1.f / s;
    for (int j = threadIdx.x; j < ld; j += 256) {
        float g = 0.f;
        if (j < c) g = (expf(z[j] - mx) * inv - (j == y ? 1.f : 0.f)) * gscale;
        d[j] = bad ? NAN : g;
    }
    if (threadIdx.x == 0) loss_rows[row] = bad ? NAN : logf(s) - (z[y] - mx);
}

// The same loss with the row held in registers (c <= 256 * NPT): one read of the logits instead of three passes -- 512 rows x
// 10575 classes took 36 us (15 % of HBM speed) walking each 42-KB row three times with two blocks per CU.  Same arithmetic in the
// same order per thread (strided ownership j = t + 256*i), same block reductions: results are bit-identical to the generic kernel.
template <int NPT>
__global__ __launch_bounds__(256) void softmax_ce_reg_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                                             float* __restrict__ loss_rows, float* __restrict__ dlogits,
                                                             int c, int ld, float gscale) {
    __shared__ float sh[4];
    const int row = blockIdx.x;
    const float* z = logits + (long)row * ld;
    float* d = dlogits + (long)row * ld;
    float v[NPT];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int j = threadIdx.x + 256 * i;
        v[i] = j < c ? z[j] : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
    mx = block_max(mx, sh);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int j = threadIdx.x + 256 * i;
        v[i] = expf(v[i] - mx);
        if (j < c) s += v[i];
    }
    s = block_sum(s, sh);
    const int yl = labels[row];
    const bool bad = (unsigned)yl >= (unsigned)c;
    const int y = bad ? 0 : yl;
    const float inv = 1.f / s;
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int j = threadIdx.x + 256 * i;
        if (j < ld) {
            const float g = j < c ? (v[i] * inv - (j == y ? 1.f : 0.f)) * gscale : 0.f;
            d[j] = bad ? NAN : g;
        }
    }
    if (threadIdx.x == 0) loss_rows[row] = bad ? NAN : logf(s) - (z[y] - mx);
}

// Focal loss (loss.py:18-27): F_i = gamma * (1 - p_y)^alpha * CE_i.  With q = p_y, dF/dz_j = coef * (p_j - [j = y]),
// coef = gamma * ((1-q)^alpha - alpha * q * (1-q)^(alpha-1) * log q): the softmax-CE gradient rescaled per row.
__global__ __launch_bounds__(256) void focal_loss_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                                         float* __restrict__ loss_rows, float* __restrict__ dlogits,
                                                         int c, int ld, float gamma, float alpha, float gscale) {
    __shared__ float sh[4];
    const int row = blockIdx.x;
    const float* z = logits + (long)row * ld;
    float* d = dlogits + (long)row * ld;
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < c; j += 256) mx = fmaxf(mx, z[j]);
    mx = block_max(mx, sh);
    float s = 0.f;
    for (int j = threadIdx.x; j < c; j += 256) s += expf(z[j] - mx);
    s = block_sum(s, sh);
    const int yl = labels[row];
    const bool bad = (unsigned)yl >= (unsigned)c;              // out-of-range label: NaN row, no out-of-bounds access
    const int y = bad ? 0 : yl;
    const float inv = 1.f / s;
    const float logq = bad ? NAN : (z[y] - mx) - logf(s);      // log p_y <= 0
    const float q = expf(logq);
    const float omq = fmaxf(1.f - q, 0.f);
    const float pw = powf(omq, alpha);                          // (1-q)^alpha
    const float pw1 = alpha == 1.f ? 1.f : powf(omq, alpha - 1.f);
    const float coef = gamma * (pw - alpha * q * pw1 * logq) * gscale;
    for (int j = threadIdx.x; j < ld; j += 256) {
        float g = 0.f;
        if (j < c) g = (expf(z[j] - mx) * inv - (j == y ? 1.f : 0.f)) * coef;
        d[j] = g;
    }
    if (threadIdx.x == 0) loss_rows[row] = -gamma * pw * logq;
}

// ------------------------------------------------------------------------------------
// A-softmax (m = 4).  One block per row; s = raw x.W row.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void asoftmax_kernel(const float* __restrict__ s, const float* __restrict__ xn,
                                                       const float* __restrict__ wn, const int32_t* __restrict__ labels,
                                                       float lam, float* __restrict__ f, float* __restrict__ loss_rows,
                                                       float* __restrict__ G, float* __restrict__ rowcoef,
                                                       int c, int ld, float gscale) {
    __shared__ float sh[4];
    const int row = blockIdx.x;
    const float* sr = s + (long)row * ld;
    float* gr = G + (long)row * ld;
    const int yl = labels[row];
    const bool bad = (unsigned)yl >= (unsigned)c;              // out-of-range label: NaN row, no out-of-bounds access
    const int y = bad ? 0 : yl;
    const float xnorm = bad ? NAN : xn[row];
    // target logit
    const float cy = sr[y] / (xnorm * wn[y]);
    const int k = (cy <= 0.70710678118654752f) + (cy <= 0.f) + (cy <= -0.70710678118654752f);
    const float sign = (k & 1) ? -1.f : 1.f;
    const float c2 = cy * cy;
    const float psi = sign * (8.f * c2 * c2 - 8.f * c2 + 1.f) - 2.f * k;
    const float dpsi = sign * (32.f * c2 * cy - 16.f * cy);
    const float phi = lam * cy + psi, dphi = lam + dpsi;
    const float fy = xnorm * phi / (1.f + lam);
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < c; j += 256) mx = fmaxf(mx, j == y ? fy : sr[j] / wn[j]);
    mx = block_max(mx, sh);
    float se = 0.f;
    for (int j = threadIdx.x; j < c; j += 256) se += expf((j == y ? fy : sr[j] / wn[j]) - mx);
    se = block_sum(se, sh);
    const float inv = 1.f / se;
    for (int j = threadIdx.x; j < ld; j += 256) {
        float g = 0.f, fv = 0.f;
        if (j < c) {
            const float w_ = wn[j];
            fv = j == y ? fy : sr[j] / w_;
            const float sm = expf(fv - mx) * inv;
            if (j == y) {
                const float gy = (sm - 1.f) * gscale;
                g = gy * dphi / ((1.f + lam) * w_);
                rowcoef[row] = gy * (phi - dphi * cy) / ((1.f + lam) * xnorm);
            } else {
                g = sm * gscale / w_;
            }
        }
        gr[j] = g;
        if (f) f[(long)row * ld + j] = fv;
    }
    if (threadIdx.x == 0) loss_rows[row] = logf(se) - (fy - mx);
}

// block = 64 columns x 4 row lanes, 4 independent row pairs per trip, lanes of a column summed in a FIXED order (deterministic):
// one thread per column walking all rows serially took 130 us for the 512 x 10575 classifier (42 blocks on 256 CUs)
__global__ __launch_bounds__(256) void asoftmax_colcoef_kernel(const float* __restrict__ G, const float* __restrict__ s,
                                                               const float* __restrict__ wn, float* __restrict__ cc,
                                                               int n, int c, int ld) {
    __shared__ float sh[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + cl;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (j < c) {
        int i = rl;
        for (; i + 12 < n; i += 16) {
            const long o0 = (long)i * ld + j, o1 = (long)(i + 4) * ld + j, o2 = (long)(i + 8) * ld + j, o3 = (long)(i + 12) * ld + j;
            a0 += G[o0] * s[o0]; a1 += G[o1] * s[o1]; a2 += G[o2] * s[o2]; a3 += G[o3] * s[o3];
        }
        for (; i < n; i += 4) { const long o = (long)i * ld + j; a0 += G[o] * s[o]; }
    }
    sh[rl][cl] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (rl == 0 && j < ld) {
        float a = 0.f;
        if (j < c) {
            a = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
            const float w_ = wn[j];
            a = -a / (w_ * w_);
        }
        cc[j] = a;
    }
}

__global__ __launch_bounds__(256) void row_norms_kernel(const float* __restrict__ a, float* __restrict__ out, int cols, int ld) {
    __shared__ float sh[4];
    const float* r = a + (long)blockIdx.x * ld;
    float s = 0.f;
    for (int j = threadIdx.x; j < cols; j += 256) s += r[j] * r[j];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) out[blockIdx.x] = sqrtf(s);
}
// block = 64 columns x 4 row lanes, 4 independent loads per trip (one thread per column walking all rows serially took
// 121 us for the 512 x 10575 classifier)
__global__ __launch_bounds__(256) void col_norms_kernel(const float* __restrict__ a, float* __restrict__ out, int rows, int cols, int ld) {
    __shared__ float sh[4][64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + c;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (j < cols) {
        int i = rl;
        for (; i + 12 < rows; i += 16) {
            const float v0 = a[(long)i * ld + j], v1 = a[(long)(i + 4) * ld + j], v2 = a[(long)(i + 8) * ld + j], v3 = a[(long)(i + 12) * ld + j];
            s0 += v0 * v0; s1 += v1 * v1; s2 += v2 * v2; s3 += v3 * v3;
        }
        for (; i < rows; i += 4) { const float v = a[(long)i * ld + j]; s0 += v * v; }
    }
    sh[rl][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rl == 0 && j < cols) out[j] = sqrtf((sh[0][c] + sh[1][c]) + (sh[2][c] + sh[3][c]));
}
__global__ __launch_bounds__(256) void add_scaled_kernel(float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ rc, const float* __restrict__ cc,
                                                         int rows, int cols, int ld) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)rows * cols) return;
    const int i = (int)(idx / cols), j = (int)(idx % cols);
    const long o = (long)i * ld + j;
    float v = a[o];
    const float bv = b[o];
    if (rc) v += rc[i] * bv;
    if (cc) v += cc[j] * bv;
    a[o] = v;
}

// ------------------------------------------------------------------------------------
// center loss (loss.py:29-45): one block per sample
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void center_loss_kernel(const float* __restrict__ feat, const int32_t* __restrict__ labels,
                                                          const float* __restrict__ centers, float* __restrict__ diff,
                                                          float* __restrict__ loss_rows, float* __restrict__ dfeat,
                                                          int d, int num_classes, float gscale) {
    __shared__ float sh[4];
    const int i = blockIdx.x, y = labels[i];
    if ((unsigned)y >= (unsigned)num_classes) {      // a label outside the table: NaN loss / gradient row, zero update -- never an out-of-bounds access
        for (int j = threadIdx.x; j < d; j += 256) { dfeat[(long)i * d + j] = NAN; diff[(long)i * d + j] = 0.f; }
        if (threadIdx.x == 0) loss_rows[i] = NAN;
        return;
    }
    float s = 0.f;
    for (int j = threadIdx.x; j < d; j += 256) {
        const float df = feat[(long)i * d + j] - centers[(long)y * d + j];
        s += df * df;
        dfeat[(long)i * d + j] = 2.f * df * gscale;
        diff[(long)i * d + j] = df;
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) loss_rows[i] = s;
}
// scatter_sub after every gather has been served from the OLD centers (loss.py:37-39):
// centers[y] -= (1-alpha)*(c_y - f)  ==  += (1-alpha)*(f - c_y); duplicate labels accumulate.
// Deterministic (no float atomics): the block of the FIRST sample carrying a label owns that row of the table and sums the rows
// of all samples with the same label in sample order; the other blocks of that label leave.  Replicas that apply the same
// (labels, diff) list therefore produce bit-identical tables (the opt-in reconciliation of data_parallel.py), and a repeated
// step repeats bit for bit.  Labels are staged in LDS (n <= CENTER_LDS_LABELS), else read from global memory.
// (dynamic LDS: n ints when n <= CENTER_LDS_LABELS -- 512 bytes for a 128-image shard, not a fixed 32 KiB per block -- and the
// "does an earlier sample own this row" scan is shared by the block's threads instead of run serially by each of them)
constexpr int CENTER_LDS_LABELS = 8192;
__global__ __launch_bounds__(256) void center_update_kernel(const float* __restrict__ diff, const int32_t* __restrict__ labels,
                                                            float* __restrict__ centers, int n, int d, int num_classes, float alpha) {
    extern __shared__ int lab[];
    const int i = blockIdx.x, y = labels[i];
    if ((unsigned)y >= (unsigned)num_classes) return;
    const bool staged = n <= CENTER_LDS_LABELS;
    if (staged) {
        for (int k = threadIdx.x; k < n; k += 256) lab[k] = labels[k];
        __syncthreads();
    }
    const int* L = staged ? lab : labels;
    int dup = 0;
    for (int k = threadIdx.x; k < i; k += 256) dup |= (L[k] == y) ? 1 : 0;
    if (__syncthreads_or(dup)) return;               // an earlier sample owns this row (block-uniform)
    for (int j0 = threadIdx.x; j0 < d; j0 += 256 * 4) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = i; k < n; ++k) {
            if (L[k] != y) continue;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * 256;
                if (j < d) acc[u] += diff[(long)k * d + j];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u * 256;
            if (j < d) centers[(long)y * d + j] += (1.f - alpha) * acc[u];
        }
    }
}

// ------------------------------------------------------------------------------------
// batch-hard triplet (loss.py:47-78).  Gram form: D_ij^2 = |f_i|^2 + |f_j|^2 - 2 f_i.f_j
// is NOT used for the distances themselves (cancellation would cost the fp32 tolerance):
// one block per (i) computes D_i: directly from differences, N x N x D is tiny (N <= 256/GPU).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void triplet_dist_kernel(const float* __restrict__ feat, float* __restrict__ dist, int n, int d) {
    __shared__ float sh[4];
    const int i = blockIdx.x / n, j = blockIdx.x % n;
    float s = 0.f;
    for (int k = threadIdx.x; k < d; k += 256) {
        const float df = feat[(long)i * d + k] - feat[(long)j * d + k];
        s += df * df;
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) dist[(long)i * n + j] = sqrtf(s + 1e-12f);
}
// one wave-sized block per anchor: hardest positive (max) / hardest negative (min), loss, and the
// coefficient matrix coef[i][j] such that dfeat_a = sum_b (coef[a][b] + coef[b][a]) (f_a - f_b)
__global__ __launch_bounds__(64) void triplet_mine_kernel(const float* __restrict__ dist, const int32_t* __restrict__ labels,
                                                          float margin, bool soft, float lw, float* __restrict__ loss_rows,
                                                          float* __restrict__ coef, int n) {
    const int i = blockIdx.x, lane = threadIdx.x;
    const int yi = labels[i];
    float bp = -1.f, bn = INFINITY;
    int ip = -1, in_ = -1;
    for (int j = lane; j < n; j += 64) {
        const bool same = labels[j] == yi;
        const float dv = dist[(long)i * n + j];
        const float pv = (same && j != i) ? dv : 0.f;            // dists*pos_mask
        const float nv = same ? 1e6f : dv;                        // dists*neg_mask + 1e6*intra (0*d + 1e6)
        if (pv > bp) { bp = pv; ip = j; }
        if (nv < bn) { bn = nv; in_ = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {       // argmax / argmin with first-index tie-break
        const float op = __shfl_xor(bp, o); const int oip = __shfl_xor(ip, o);
        if (op > bp || (op == bp && oip >= 0 && (ip < 0 || oip < ip))) { bp = op; ip = oip; }
        const float on = __shfl_xor(bn, o); const int oin = __shfl_xor(in_, o);
        if (on < bn || (on == bn && oin >= 0 && (in_ < 0 || oin < in_))) { bn = on; in_ = oin; }
    }
    for (int j = lane; j < n; j += 64) coef[(long)i * n + j] = 0.f;
    __syncthreads();
    if (lane == 0) {
        const float v = bp - bn;
        float loss, dv;
        if (soft) {
            loss = v > 20.f ? v : log1pf(expf(v));
            dv = 1.f / (1.f + expf(-v));
        } else {
            loss = fmaxf(0.f, v + margin);
            dv = v + margin > 0.f ? 1.f : 0.f;
        }
        loss_rows[i] = loss;
        dv *= lw;
        const bool pos_real = ip >= 0 && ip != i && labels[ip] == yi;
        const bool neg_real = in_ >= 0 && labels[in_] != yi;
        if (pos_real) coef[(long)i * n + ip] += dv / dist[(long)i * n + ip];
        if (neg_real) coef[(long)i * n + in_] -= dv / dist[(long)i * n + in_];
    }
}
__global__ __launch_bounds__(256) void triplet_grad_kernel(const float* __restrict__ feat, const float* __restrict__ coef,
                                                           float* __restrict__ dfeat, int n, int d) {
    // the anchor's coefficient row (its own choices + the anchors that chose it) goes to LDS once -- every thread walked the two global
    // rows itself before: 2 n dependent-latency loads per thread, 128 us for 128 x 2048 -- and is almost all zeros (<= 2 entries per
    // anchor): the walk below reads LDS and touches feat only for the non-zero ones, in the same order b = 0 .. n - 1
    extern __shared__ float cfs[];
    const int a = blockIdx.x;
    for (int b = threadIdx.x; b < n; b += 256) cfs[b] = coef[(long)a * n + b] + coef[(long)b * n + a];
    __syncthreads();
    for (int k = threadIdx.x; k < d; k += 256) {
        float g = 0.f;
        const float fa = feat[(long)a * d + k];
        for (int b = 0; b < n; ++b) {
            const float cf = cfs[b];
            if (cf != 0.f) g += cf * (fa - feat[(long)b * d + k]);
        }
        dfeat[(long)a * d + k] = g;
    }
}

// ------------------------------------------------------------------------------------
// Optimizers on the flat arena
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void momentum_kernel(float* __restrict__ w, float* __restrict__ acc, const float* __restrict__ g,
                                                       long n, float lr, float mom, float wd, float gs) {
    const long n4 = n >> 2;
    f32x4* w4 = reinterpret_cast<f32x4*>(w);
    f32x4* a4 = reinterpret_cast<f32x4*>(acc);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 wv = w4[i], av = a4[i];
        const f32x4 gv = g4[i];
        av = mom * av + (gs * gv + wd * wv);
        wv = wv - lr * av;
        a4[i] = av;
        w4[i] = wv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = (n4 << 2) + threadIdx.x;
        const float av = mom * acc[i] + (gs * g[i] + wd * w[i]);
        acc[i] = av;
        w[i] = w[i] - lr * av;
    }
}
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                                   const float* __restrict__ g, long n, float lr_t, float b1, float b2,
                                                   float eps, float wd, float gs) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gv = gs * g[i] + wd * w[i];
        const float mv = b1 * m[i] + (1.f - b1) * gv;
        const float vv = b2 * v[i] + (1.f - b2) * gv * gv;
        m[i] = mv;
        v[i] = vv;
        w[i] = w[i] - lr_t * mv / (sqrtf(vv) + eps);
    }
}

inline int grid_for(long n, int per) { long b = (n + per - 1) / per; return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); }

}  // namespace

// ---------------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------------
hipError_t k_conv_first_fwd(const float* x, const float* w, const float* bias, const float* alpha, float* z, float* y,
                            unsigned short* z16, unsigned short* y16,
                            int n, int h, int wd, int cin, int cout, int ho, int wo, int stride, int pt, int pl, hipStream_t st) {
    const long ngrp = (long)n * ho * ((wo + 31) / 32);
    long nb = (ngrp + 3) / 4;
    if (nb > 4096) nb = 4096;
    const int blocks = (int)nb;
#define FTE_CF(CI_, CO_) hipLaunchKernelGGL((conv_first_fwd_kernel<CI_, CO_>), dim3(blocks), dim3(256), 0, st, x, w, bias, alpha, z, y, z16, y16, n, h, wd, ho, wo, stride, pt, pl)
    if (cin == 1 && cout == 64) FTE_CF(1, 64);
    else if (cin == 3 && cout == 64) FTE_CF(3, 64);
    else if (cin == 1 && cout == 32) FTE_CF(1, 32);
    else if (cin == 3 && cout == 32) FTE_CF(3, 32);
    else return hipErrorInvalidValue;
#undef FTE_CF
    return hipGetLastError();
}
int k_conv_first_wgrad_blocks(long npix) { long b = (npix + 511) / 512; return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b)); }
hipError_t k_conv_first_wgrad(const float* x, const float* dz, const unsigned short* dz16, float* part, int n, int h, int wd, int cin, int cout, int ho, int wo,
                              int stride, int pt, int pl, int blocks, hipStream_t st) {
    const long nrows = (long)n * ho;
    const long rpb = (nrows + blocks - 1) / blocks;           // blocks that get no rows write zero partials
#define FTE_CW(CI_, CO_) hipLaunchKernelGGL((conv_first_wgrad_kernel<CI_, CO_>), dim3(blocks), dim3(256), 0, st, x, dz, dz16, part, n, h, wd, ho, wo, stride, pt, pl, rpb)
    if (cin == 1 && cout == 64) FTE_CW(1, 64);
    else if (cin == 3 && cout == 64) FTE_CW(3, 64);
    else if (cin == 1 && cout == 32) FTE_CW(1, 32);
    else if (cin == 3 && cout == 32) FTE_CW(3, 32);
    else return hipErrorInvalidValue;
#undef FTE_CW
    return hipGetLastError();
}
// scratch: REDUCE_SCRATCH_FLOATS floats, only touched when the matrix is tall and narrow
hipError_t k_reduce_rows2(const float* in, float* out, const float* in2, float* out2, const float* bias, int bmod, long rows, long cols,
                          int fold, float scale, float* scratch, hipStream_t st, int act) {
    // [rows, fold, cols/fold] is the same memory as [rows*fold, cols/fold]: folding is a reshape
    rows *= fold;
    cols /= fold;
    static const bool slabs_off = getenv("FTE_REDUCE_SLABS") && atoi(getenv("FTE_REDUCE_SLABS")) == 0;      // A/B hook
    if (!slabs_off && !act && !in2 && !bias && scale == 1.f && rows >= 2 && rows < (1 << 20) && cols >= 4096 && cols % 4 == 0 &&
        (reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) % 16 == 0) {
        const long quads = cols / 4;
        if (quads / 64 >= 1024 || rows < 32)
            hipLaunchKernelGGL(reduce_slabs_kernel<64>, dim3((unsigned)((quads + 63) / 64)), dim3(256), 0, st, reinterpret_cast<const f32x4*>(in),
                               reinterpret_cast<f32x4*>(out), (int)rows, quads);
        else
            hipLaunchKernelGGL(reduce_slabs_kernel<16>, dim3((unsigned)((quads + 15) / 16)), dim3(256), 0, st, reinterpret_cast<const f32x4*>(in),
                               reinterpret_cast<f32x4*>(out), (int)rows, quads);
        return hipGetLastError();
    }
    const unsigned nz = in2 ? 2 : 1;
    static const bool rq_off = getenv("FTE_REDUCE_ROWS_Q") && atoi(getenv("FTE_REDUCE_ROWS_Q")) == 0;      // A/B hook
    // (rows < 256: ONE launch, a row lane walks <= 64 rows -- the partial rows of a small shard's 128-row tiles: two 4-us passes before)
    if (!rq_off && !act && scratch && !bias && scale == 1.f && rows >= 32 && cols % 4 == 0 && cols <= 1024 &&
        (reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(in2) | reinterpret_cast<uintptr_t>(out2) |
         reinterpret_cast<uintptr_t>(scratch)) % 16 == 0) {
        const long quads = cols / 4;
        const int QB = quads >= 64 ? 64 : 16, RL = 256 / QB;
        const long qb = (quads + QB - 1) / QB;
        long rs = 1024 / (qb * nz);                            // ~4 blocks per CU in the first pass ...
        if (rs > rows / (8L * RL)) rs = rows / (8L * RL);      // ... each with at least one round of eight loads per lane
        if (rs * cols * nz > REDUCE_SCRATCH_FLOATS) rs = REDUCE_SCRATCH_FLOATS / (cols * nz);
        if (rs < 1 || rows < 256) rs = 1;
        const long rps = (rows + rs - 1) / rs;
        rs = (rows + rps - 1) / rps;
        f32x4* s1 = reinterpret_cast<f32x4*>(scratch);
        f32x4* s2 = s1 + rs * quads;
#define FTE_RQ(QB_, GY_, IN_, OUT_, ROWS_, RPS_, IN2_, OUT2_) hipLaunchKernelGGL((reduce_rows_q_kernel<QB_>), dim3((unsigned)qb, (unsigned)(GY_), nz), dim3(256), 0, st, \
            reinterpret_cast<const f32x4*>(IN_), reinterpret_cast<f32x4*>(OUT_), (long)(ROWS_), quads, (long)(RPS_), reinterpret_cast<const f32x4*>(IN2_), reinterpret_cast<f32x4*>(OUT2_))
        if (rs == 1) {
            if (QB == 64) FTE_RQ(64, 1, in, out, rows, rows, in2, out2); else FTE_RQ(16, 1, in, out, rows, rows, in2, out2);
            return hipGetLastError();
        }
        if (QB == 64) { FTE_RQ(64, rs, in, s1, rows, rps, in2, s2); FTE_RQ(64, 1, s1, out, rs, rs, s2, out2); }
        else { FTE_RQ(16, rs, in, s1, rows, rps, in2, s2); FTE_RQ(16, 1, s1, out, rs, rs, s2, out2); }
#undef FTE_RQ
        return hipGetLastError();
    }
    const long cb = (cols + 63) / 64;
    long rs = 1;
    if (scratch && cb < 512 && rows >= 64) {
        // two passes: a row lane walks rows / (4 rs) rows in the first and rs / 4 in the second -- balanced at rs = sqrt(rows) (12544
        // partial rows of a 64-column layer: 1024 / cb = 784 splits left the second pass 196 dependent loads per lane, 29 us), more
        // splits only while the first pass has less than a block per CU
        rs = 1;
        while (rs * rs < rows) ++rs;
        if (rs < 256 / cb) rs = 256 / cb;
        if (rs > 1024 / cb) rs = 1024 / cb;
        if (rs > rows / 16) rs = rows / 16;
        if (rs * cols * nz > REDUCE_SCRATCH_FLOATS) rs = REDUCE_SCRATCH_FLOATS / (cols * nz);
        if (rs < 1) rs = 1;
    }
    if (rs == 1) {
        hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)cb, 1, nz), dim3(256), 0, st, in, out, bias, bmod, rows, cols, rows, scale, in2, out2, act);
        return hipGetLastError();
    }
    const long rps = (rows + rs - 1) / rs;
    rs = (rows + rps - 1) / rps;
    float* scratch2 = scratch + rs * cols;
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)cb, (unsigned)rs, nz), dim3(256), 0, st, in, scratch, (const float*)nullptr, 1, rows, cols, rps, 1.f,
                       in2, scratch2, 0);
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)cb, 1, nz), dim3(256), 0, st, (const float*)scratch, out, bias, bmod, rs, cols, rs, scale,
                       (const float*)scratch2, out2, act);
    return hipGetLastError();
}
// scratch: REDUCE_SCRATCH_FLOATS floats, only touched when the matrix is tall and narrow
hipError_t k_reduce_rows(const float* in, float* out, const float* bias, int bmod, long rows, long cols, int fold, float scale,
                         float* scratch, hipStream_t st, int act) {
    return k_reduce_rows2(in, out, nullptr, nullptr, bias, bmod, rows, cols, fold, scale, scratch, st, act);
}
hipError_t k_sum(const float* a, long n, float scale, float* out, float* ws, bool sq, hipStream_t st) {
    const int nb = grid_for(n, 256 * 4 * 8) > 1024 ? 1024 : grid_for(n, 256 * 4 * 8);
    if (sq) hipLaunchKernelGGL(partial_sum_kernel<true>, dim3(nb), dim3(256), 0, st, a, n, ws);
    else hipLaunchKernelGGL(partial_sum_kernel<false>, dim3(nb), dim3(256), 0, st, a, n, ws);
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, st, ws, nb, scale, out);
    return hipGetLastError();
}
hipError_t k_softmax_ce(const float* logits, const int32_t* labels, float* loss_rows, float* dlogits, int n, int c, int ld, float gs, hipStream_t st) {
    if (ld <= 256 * 8) hipLaunchKernelGGL(softmax_ce_reg_kernel<8>, dim3(n), dim3(256), 0, st, logits, labels, loss_rows, dlogits, c, ld, gs);
    else if (ld <= 256 * 48) hipLaunchKernelGGL(softmax_ce_reg_kernel<48>, dim3(n), dim3(256), 0, st, logits, labels, loss_rows, dlogits, c, ld, gs);
    else hipLaunchKernelGGL(softmax_ce_kernel, dim3(n), dim3(256), 0, st, logits, labels, loss_rows, dlogits, c, ld, gs);
    return hipGetLastError();
}
namespace {
__global__ __launch_bounds__(256) void to_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
        unsigned short o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = __builtin_bit_cast(unsigned short, (__bf16)v[e]);
        *reinterpret_cast<uint2*>(y + i * 4) = make_uint2(o[0] | ((unsigned)o[1] << 16), o[2] | ((unsigned)o[3] << 16));
    }
}
// w [taps][cin][cout] fp32 -> w16 (same layout) and w16t [taps][cout][cin]
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ w16,
                                                           unsigned short* __restrict__ w16t, int taps, int cin, int cout) {
    const long total = (long)taps * cin * cout;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const unsigned short v = __builtin_bit_cast(unsigned short, (__bf16)w[i]);
        if (w16) w16[i] = v;
        if (w16t) {
            const int co = (int)(i % cout);
            const long t2 = i / cout;
            const int ci = (int)(t2 % cin), tap = (int)(t2 / cin);
            w16t[((long)tap * cout + co) * cin + ci] = v;
        }
    }
}
// All filters of a net in ONE launch: table row k = {source offset in `params`, destination offset in `w16t`, taps, cin, cout,
// first destination index of this conv in the flattened walk}; thread = 4 consecutive cin of one (conv, tap, cout): the bf16
// [tap][cout][cin] pack is written in 8-byte pieces, the fp32 HWIO source is gathered through the caches.
typedef float f32x4k __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void pack_weights_table_kernel(const float* __restrict__ params, unsigned short* __restrict__ w16t,
                                                                 const int* __restrict__ table, int nconv, long total4, int transposed) {
    __shared__ int tb[64 * 6];
    for (int i = threadIdx.x; i < nconv * 6; i += 256) tb[i] = table[i];
    __syncthreads();
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        int lo = 0, hi = nconv - 1;                      // last conv whose first index (in units of 4 elements) is <= i
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if ((long)tb[mid * 6 + 5] <= i) lo = mid; else hi = mid - 1;
        }
        const int* d = tb + lo * 6;
        const int cin = d[3], cout = d[4];
        const long j = (i - d[5]) * 4;                   // destination index inside this conv: ((tap * cout + co) * cin + ci)
        if (!transposed) {                               // the HWIO pack: a plain conversion, same index on both sides
            const f32x4k v = *reinterpret_cast<const f32x4k*>(params + d[0] + j);
            *reinterpret_cast<uint2*>(w16t + d[1] + j) = make_uint2(
                __builtin_bit_cast(unsigned short, (__bf16)v[0]) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[1]) << 16),
                __builtin_bit_cast(unsigned short, (__bf16)v[2]) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[3]) << 16));
            continue;
        }
        const int ci = (int)(j % cin);
        const long t2 = j / cin;
        const int co = (int)(t2 % cout), tap = (int)(t2 / cout);
        const float* src = params + d[0] + ((long)tap * cin + ci) * cout + co;
        unsigned short o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = __builtin_bit_cast(unsigned short, (__bf16)src[(long)e * cout]);
        *reinterpret_cast<uint2*>(w16t + d[1] + j) = make_uint2(o[0] | ((unsigned)o[1] << 16), o[2] | ((unsigned)o[3] << 16));
    }
}
// The [tap][cout][cin] pack by 64 x 64 tiles through LDS (round 5).  The kernel above gathers its four source elements down a
// column of the HWIO filter (stride cout floats): every 4-byte read is a cache line of its own, the lines come back 16-32 times
// through L2 and the launch moves 4x its algorithmic bytes (290 MB for ResNet-50's 23.5 M filter elements, 61 us beside the stem;
// profiles/r5_resnet50_*).  Here a block reads a tile's 64 source rows as whole 256-byte pieces, rounds, transposes in LDS and writes
// 128-byte destination row pieces.  Same table; a block finds its (conv, tap, tile) by walking the <= 64 rows.
__global__ __launch_bounds__(256) void pack_weights_tiles_kernel(const float* __restrict__ params, unsigned short* __restrict__ w16t,
                                                                 const int* __restrict__ table, int nconv) {
    __shared__ int tb[64 * 6];
    __shared__ unsigned short tile[64][68];              // [co][ci], rows 136 bytes apart (34 words: the transposing 2-byte stores spread over the banks)
    for (int i = threadIdx.x; i < nconv * 6; i += 256) tb[i] = table[i];
    __syncthreads();
    const int tid = threadIdx.x;
    for (long t = blockIdx.x;; t += gridDim.x) {
        long acc = 0;
        int c = -1, nci = 0, nco = 0;
        for (int q = 0; q < nconv; ++q) {
            nci = (tb[q * 6 + 3] + 63) >> 6; nco = (tb[q * 6 + 4] + 63) >> 6;
            const long nt = (long)tb[q * 6 + 2] * nci * nco;
            if (t < acc + nt) { c = q; break; }
            acc += nt;
        }
        if (c < 0) break;                                 // (uniform: every thread of the block walks the same t)
        const int* d = tb + c * 6;
        const int cin = d[3], cout = d[4];
        const int local = (int)(t - acc), tap = local / (nci * nco), r2 = local - tap * (nci * nco);
        const int ci0 = (r2 / nco) << 6, co0 = (r2 % nco) << 6;
        const float* src = params + d[0] + (long)tap * cin * cout;
        f32x4k v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ci = ci0 + (tid >> 4) + 16 * u, co = co0 + 4 * (tid & 15);
            v[u] = (ci < cin && co < cout) ? *reinterpret_cast<const f32x4k*>(src + (long)ci * cout + co) : f32x4k{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[4 * (tid & 15) + e][(tid >> 4) + 16 * u] = __builtin_bit_cast(unsigned short, (__bf16)v[u][e]);
        __syncthreads();
        unsigned short* dst = w16t + d[1] + (long)tap * cout * cin;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = (tid >> 4) + 16 * u, co = co0 + rr, ci = ci0 + 4 * (tid & 15);
            if (co < cout && ci < cin) *reinterpret_cast<uint2*>(dst + (long)co * cin + ci) = *reinterpret_cast<const uint2*>(&tile[rr][4 * (tid & 15)]);
        }
        __syncthreads();
    }
}
}  // namespace
hipError_t k_pack_weights_table(const float* params, unsigned short* w16t, const int* table, int nconv, long total, int transposed, hipStream_t st) {
    const long t4 = total / 4;
    static const bool tiles_off = getenv("FTE_PACK_TILES") && atoi(getenv("FTE_PACK_TILES")) == 0;      // A/B hook: the element gather
    if (transposed && !tiles_off && nconv <= 64) {
        const long tiles = total / 4096 + 9L * nconv;     // upper bound on the tile count is all a grid-stride walk needs
        hipLaunchKernelGGL(pack_weights_tiles_kernel, dim3((unsigned)(tiles > 4096 ? 4096 : (tiles < 1 ? 1 : tiles))), dim3(256), 0, st, params, w16t, table, nconv);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(pack_weights_table_kernel, dim3((unsigned)((t4 + 255) / 256 > 8192 ? 8192 : (t4 + 255) / 256)), dim3(256), 0, st,
                       params, w16t, table, nconv, t4, transposed);
    return hipGetLastError();
}
hipError_t k_to_bf16(const float* x, unsigned short* y, long n, hipStream_t st) {
    const long n4 = n / 4;
    hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)((n4 + 255) / 256 > 8192 ? 8192 : (n4 + 255) / 256)), dim3(256), 0, st, x, y, n4);
    return hipGetLastError();
}
hipError_t k_pack_weights_bf16(const float* w, unsigned short* w16, unsigned short* w16t, int taps, int cin, int cout, hipStream_t st) {
    const long total = (long)taps * cin * cout;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256), 0, st, w, w16, w16t, taps, cin, cout);
    return hipGetLastError();
}
hipError_t k_focal_loss(const float* logits, const int32_t* labels, float* loss_rows, float* dlogits, int n, int c, int ld,
                        float gamma, float alpha, float gs, hipStream_t st) {
    hipLaunchKernelGGL(focal_loss_kernel, dim3(n), dim3(256), 0, st, logits, labels, loss_rows, dlogits, c, ld, gamma, alpha, gs);
    return hipGetLastError();
}
hipError_t k_asoftmax(const float* s, const float* xn, const float* wn, const int32_t* labels, float lam, float* f, float* loss_rows,
                      float* G, float* rowcoef, int n, int c, int ld, float gs, hipStream_t st) {
    hipLaunchKernelGGL(asoftmax_kernel, dim3(n), dim3(256), 0, st, s, xn, wn, labels, lam, f, loss_rows, G, rowcoef, c, ld, gs);
    return hipGetLastError();
}
hipError_t k_asoftmax_colcoef(const float* G, const float* s, const float* wn, float* cc, int n, int c, int ld, hipStream_t st) {
    hipLaunchKernelGGL(asoftmax_colcoef_kernel, dim3((ld + 63) / 64), dim3(256), 0, st, G, s, wn, cc, n, c, ld);
    return hipGetLastError();
}
hipError_t k_row_norms(const float* a, float* out, int rows, int cols, int ld, hipStream_t st) {
    hipLaunchKernelGGL(row_norms_kernel, dim3(rows), dim3(256), 0, st, a, out, cols, ld);
    return hipGetLastError();
}
hipError_t k_col_norms(const float* a, float* out, int rows, int cols, int ld, hipStream_t st) {
    hipLaunchKernelGGL(col_norms_kernel, dim3((cols + 63) / 64), dim3(256), 0, st, a, out, rows, cols, ld);
    return hipGetLastError();
}
namespace {
// y[n][h][w'][c] = x[n][h][w - 1 - w'][c]: tf.reverse(images, axis=[2]) of the eval path (nets/sphere.py:97-101), 16 bytes per lane
__global__ __launch_bounds__(256) void flip_w_kernel(const float* __restrict__ x, float* __restrict__ y, long rows, int w, int c4) {
    const long total = rows * w * c4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long pix = i / c4;
        const int q = (int)(i - pix * c4);
        const long row = pix / w;
        const int xo = (int)(pix - row * w);
        *reinterpret_cast<f32x4*>(y + i * 4) = *reinterpret_cast<const f32x4*>(x + ((row * w + (w - 1 - xo)) * c4 + q) * 4);
    }
}
__global__ __launch_bounds__(256) void flip_w1_kernel(const float* __restrict__ x, float* __restrict__ y, long rows, int w, int c) {
    const long total = rows * w * c;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long pix = i / c;
        const int q = (int)(i - pix * c);
        const long row = pix / w;
        const int xo = (int)(pix - row * w);
        y[i] = x[(row * w + (w - 1 - xo)) * c + q];
    }
}
// out = a * x + b * y  (the mean of the two embeddings of the eval path: a = b = 1/2)
__global__ __launch_bounds__(256) void axpby_kernel(float a, const float* __restrict__ x, float b, const float* __restrict__ y, float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = a * x[i] + b * y[i];
}
}  // namespace
hipError_t k_flip_w(const float* x, float* y, long rows, int w, int c, hipStream_t st) {
    const long total = rows * w * c;
    const int nb = (int)((total / ((c % 4) ? 1 : 4) + 255) / 256 > 4096 ? 4096 : (total / ((c % 4) ? 1 : 4) + 255) / 256);
    if (c % 4 == 0) hipLaunchKernelGGL(flip_w_kernel, dim3(nb > 0 ? nb : 1), dim3(256), 0, st, x, y, rows, w, c / 4);
    else hipLaunchKernelGGL(flip_w1_kernel, dim3(nb > 0 ? nb : 1), dim3(256), 0, st, x, y, rows, w, c);
    return hipGetLastError();
}
hipError_t k_axpby(float a, const float* x, float b, const float* y, float* out, long n, hipStream_t st) {
    const int nb = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(axpby_kernel, dim3(nb > 0 ? nb : 1), dim3(256), 0, st, a, x, b, y, out, n);
    return hipGetLastError();
}
hipError_t k_add_scaled(float* a, const float* b, const float* rc, const float* cc, int rows, int cols, int ld, hipStream_t st) {
    const long tot = (long)rows * cols;
    hipLaunchKernelGGL(add_scaled_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, a, b, rc, cc, rows, cols, ld);
    return hipGetLastError();
}
hipError_t k_center_loss(const float* feat, const int32_t* labels, float* centers, float* loss_rows, float* dfeat,
                         int n, int d, int num_classes, float alpha, float gs, float* ws, hipStream_t st) {
    hipLaunchKernelGGL(center_loss_kernel, dim3(n), dim3(256), 0, st, feat, labels, centers, ws, loss_rows, dfeat, d, num_classes, gs);
    if (alpha != 1.f)      // alpha == 1: loss and gradient only (graph construction, or the update is applied later by k_center_update)
        hipLaunchKernelGGL(center_update_kernel, dim3(n), dim3(256), n <= CENTER_LDS_LABELS ? (size_t)n * sizeof(int) : 0, st, ws, labels, centers, n, d, num_classes, alpha);
    return hipGetLastError();
}
hipError_t k_center_update(const float* diff, const int32_t* labels, float* centers, int n, int d, int num_classes, float alpha, hipStream_t st) {
    hipLaunchKernelGGL(center_update_kernel, dim3(n), dim3(256), n <= CENTER_LDS_LABELS ? (size_t)n * sizeof(int) : 0, st, diff, labels, centers, n, d, num_classes, alpha);
    return hipGetLastError();
}
hipError_t k_triplet(const float* feat, const int32_t* labels, float margin, bool soft, float lw, float* loss_rows, float* dfeat,
                     int n, int d, float* ws, hipStream_t st) {
    if (n > 8192) return hipErrorInvalidValue;              // triplet_grad_kernel keeps a coefficient row of n floats in LDS
    float* dist = ws;
    float* coef = ws + (long)n * n;
    hipLaunchKernelGGL(triplet_dist_kernel, dim3(n * n), dim3(256), 0, st, feat, dist, n, d);
    hipLaunchKernelGGL(triplet_mine_kernel, dim3(n), dim3(64), 0, st, dist, labels, margin, soft, lw, loss_rows, coef, n);
    hipLaunchKernelGGL(triplet_grad_kernel, dim3(n), dim3(256), (size_t)n * sizeof(float), st, feat, coef, dfeat, n, d);
    return hipGetLastError();
}
// ---------------------------------------------------------------------------------------------------
// Small dense products in ONE launch (round 5): out[m, n] = act(a[m, k] * op(w) + bias) [* (mask > 0)] for the squeeze-excitation
// gate's layers (nets/shufflenet_v2.py:79-85: [images, c] x [c, c / 2] and back; m = the shard's images).  Through fte_gemm_* these
// were a split-K launch of the tile kernel plus a reduction launch, 12-15 us per layer whatever its size -- four of them per SE
// block and step on the critical path.  Here a block of four waves owns one 32 x 32 output tile: wave v multiplies the tile's rows by
// the v-th quarter of K straight from global memory (no LDS staging: every operand element is used once per block), the four
// partial tiles meet in LDS and are added in wave order (fixed: bit-identical run to run), bias / activation / mask in the same pass.
// BF: operands rounded to bf16 (v_mfma_f32_32x32x16_bf16; the precision of every other product in the bf16 modes); else fp32
// (v_mfma_f32_32x32x2_f32).  TW: w is [n][k] (out = a * w^T: the data gradient through a layer).  The k index inside a K-step is
// permuted (lane half lh takes k0 + 8 lh .. + 7, resp. k0 + 4 lh .. + 3) identically for both operands: a sum's order, nothing else.
// ---------------------------------------------------------------------------------------------------
namespace {
typedef float f32x16d __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8d __attribute__((ext_vector_type(8)));
template <bool BF, bool TW, int NW>
__global__ __launch_bounds__(64 * NW) void dense_small_kernel(const float* __restrict__ a, const float* __restrict__ w, const float* __restrict__ bias,
                                                          const float* __restrict__ mask, float* __restrict__ out, int m, int n, int k, int act) {
    __shared__ float part[NW][32][32];                        // (16 waves: the whole 64 KB)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
    const int col0 = blockIdx.x * 32, row0 = blockIdx.y * 32;
    const int kw = k / NW, kb = wv * kw;                       // this wave's K range (whole 64-deep trips: the launcher checks)
    const int arow = min(row0 + li, m - 1);                    // (rows past m are computed and dropped)
    const float* pa = a + (long)arow * k;
    const float* pw = TW ? w + (long)(col0 + li) * k : w + col0 + li;
    f32x16d acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if constexpr (BF) {
        auto cvt = [](const f32x4 lo, const f32x4 hi) -> bf16x8d {
            bf16x8d v;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = (__bf16)lo[e]; v[4 + e] = (__bf16)hi[e]; }
            return v;
        };
        for (int k0 = kb; k0 < kb + kw; k0 += 64) {            // four K-steps' loads in flight
            f32x4 al[4], ah[4], bl[4], bh[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kk = k0 + 16 * u + 8 * lh;
                al[u] = *reinterpret_cast<const f32x4*>(pa + kk); ah[u] = *reinterpret_cast<const f32x4*>(pa + kk + 4);
                if constexpr (TW) { bl[u] = *reinterpret_cast<const f32x4*>(pw + kk); bh[u] = *reinterpret_cast<const f32x4*>(pw + kk + 4); }
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bl[u][e] = pw[(long)(kk + e) * n]; bh[u][e] = pw[(long)(kk + 4 + e) * n]; }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cvt(al[u], ah[u]), cvt(bl[u], bh[u]), acc, 0, 0, 0);
        }
    } else {
        for (int k0 = kb; k0 < kb + kw; k0 += 32) {            // four 8-deep steps in flight, four MFMAs each
            f32x4 av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kk = k0 + 8 * u + 4 * lh;
                av[u] = *reinterpret_cast<const f32x4*>(pa + kk);
                if constexpr (TW) bv[u] = *reinterpret_cast<const f32x4*>(pw + kk);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) bv[u][e] = pw[(long)(kk + e) * n];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][e], bv[u][e], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wv][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = acc[r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16 / NW; ++i) {
        const int idx = threadIdx.x + 64 * NW * i, r = idx >> 5, c = idx & 31;
        if (row0 + r >= m) continue;
        float v = part[0][r][c];
#pragma unroll
        for (int wq = 1; wq < NW; ++wq) v += part[wq][r][c];      // wave order
        if (bias) v += bias[col0 + c];
        if (act == 1) v = fmaxf(v, 0.f);
        else if (act == 2) v = 1.f / (1.f + expf(-v));
        const long o = (long)(row0 + r) * n + col0 + c;
        if (mask) v = mask[o] > 0.f ? v : 0.f;
        out[o] = v;
    }
}
}  // namespace
hipError_t k_dense_small(const float* a, const float* w, const float* bias, const float* mask, float* out, int m, int n, int k,
                         bool trans_w, int act, bool bf16, hipStream_t st) {
    if (m <= 0 || n <= 0 || n % 32 || k <= 0 || k % 128) return hipErrorInvalidValue;
    const dim3 grid(n / 32, (m + 31) / 32);
    // waves per tile: each takes k / NW in whole 64-deep trips (bf16) / 32-deep (fp32).  A wave's loop is a chain of ~1.5-us trips
    // (every trip's loads depend on nothing but are issued after the previous trip's MFMAs), and these launches are 16-64 blocks on
    // 256 CUs -- the chain IS the kernel: 128 x 128 <- 2048 ran 21 us with four waves (8 trips), 17 with eight (4), and runs one or
    // two trips with sixteen.  As many waves as give whole trips, up to 16.
    static const int max_nw = getenv("FTE_DENSE_SMALL_NW") ? atoi(getenv("FTE_DENSE_SMALL_NW")) : 16;      // tuning hook
    const int nw = (k % 1024 == 0 && max_nw >= 16) ? 16 : (k % 512 == 0 && max_nw >= 8) ? 8 : (k % 256 == 0 ? 4 : 2);
#define FTE_DS(BF_, TW_) do { if (nw == 16) hipLaunchKernelGGL((dense_small_kernel<BF_, TW_, 16>), grid, dim3(1024), 0, st, a, w, bias, mask, out, m, n, k, act); \
                              else if (nw == 8) hipLaunchKernelGGL((dense_small_kernel<BF_, TW_, 8>), grid, dim3(512), 0, st, a, w, bias, mask, out, m, n, k, act); \
                              else if (nw == 4) hipLaunchKernelGGL((dense_small_kernel<BF_, TW_, 4>), grid, dim3(256), 0, st, a, w, bias, mask, out, m, n, k, act); \
                              else hipLaunchKernelGGL((dense_small_kernel<BF_, TW_, 2>), grid, dim3(128), 0, st, a, w, bias, mask, out, m, n, k, act); } while (0)
    if (bf16) { if (trans_w) FTE_DS(true, true); else FTE_DS(true, false); }
    else { if (trans_w) FTE_DS(false, true); else FTE_DS(false, false); }
#undef FTE_DS
    return hipGetLastError();
}
hipError_t k_momentum(float* w, float* acc, const float* g, long n, float lr, float mom, float wd, float gs, hipStream_t st) {
    hipLaunchKernelGGL(momentum_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, st, w, acc, g, n, lr, mom, wd, gs);
    return hipGetLastError();
}
hipError_t k_adam(float* w, float* m, float* v, const float* g, long n, float lr_t, float b1, float b2, float eps, float wd, float gs, hipStream_t st) {
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, w, m, v, g, n, lr_t, b1, b2, eps, wd, gs);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// input transform of the loader (data.py:206-223 of the reference: convert_image_dtype + resize_images + random_crop +
// random_flip_left_right + (x - 0.5) / 0.5) on DECODED uint8 images.  One slot per image: a 64-byte header of int32
// {mode, h0, w0, y0, x0, flip} and then the h0 x w0 x C bytes of the decoded image (mode 0) or the finished float32 crop
// (mode 1: an image that did not fit its slot was transformed by the worker).  TF-1.x ResizeBilinear, align_corners = False:
// in = out * (n_in / n_out), low = floor(in), high = min(low + 1, n_in - 1); the two neighbours of a row are blended along x,
// then the two rows along y, every operation rounded to float32 on its own (no contraction): the results are the BITS
// tf_face_toolbox_amd/_decode_worker.py computes on the host.  One thread per output pixel.
template <int C>
__global__ __launch_bounds__(256) void preprocess_u8_kernel(const unsigned char* __restrict__ slots, float* __restrict__ out, long slot_stride,
                                                             int in_h, int in_w, int crop_h, int crop_w) {
#pragma clang fp contract(off)     // every product and sum below rounds on its own, as numpy's do (hipcc's default would fuse them into FMAs;
    // the __fmul_rn / __fadd_rn intrinsics are inline operators that carry the translation unit's contraction flag with them)
    const int img = blockIdx.y;
    const unsigned char* slot = slots + (long)img * slot_stride;
    const int* hd = reinterpret_cast<const int*>(slot);
    const int mode = hd[0], h0 = hd[1], w0 = hd[2], y0 = hd[3], x0 = hd[4], flip = hd[5];
    const int px = blockIdx.x * 256 + threadIdx.x;
    if (px >= crop_h * crop_w) return;
    float* o = out + ((long)img * crop_h * crop_w + px) * C;
    if (mode == 1) {
        const float* f = reinterpret_cast<const float*>(slot + 64) + (long)px * C;
#pragma unroll
        for (int c = 0; c < C; ++c) o[c] = f[c];
        return;
    }
    const int r = px / crop_w, cc = px - r * crop_w;
    const int col = flip ? crop_w - 1 - cc : cc;
    const float s = (float)(1.0 / 255.0);
    const float py = (float)(y0 + r) * ((float)h0 / (float)in_h);
    const float pxs = (float)(x0 + col) * ((float)w0 / (float)in_w);
    const int ylo = (int)floorf(py), xlo = (int)floorf(pxs);
    const int yhi = min(ylo + 1, h0 - 1), xhi = min(xlo + 1, w0 - 1);
    const float yw = py - (float)ylo, xw = pxs - (float)xlo;
    const unsigned char* img0 = slot + 64;
    const unsigned char* tl = img0 + ((long)ylo * w0 + xlo) * C;
    const unsigned char* tr = img0 + ((long)ylo * w0 + xhi) * C;
    const unsigned char* bl = img0 + ((long)yhi * w0 + xlo) * C;
    const unsigned char* br = img0 + ((long)yhi * w0 + xhi) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float a = (float)tl[c] * s, b = (float)tr[c] * s;
        const float d = (float)bl[c] * s, e = (float)br[c] * s;
        const float top = (b - a) * xw + a;
        const float bot = (e - d) * xw + d;
        const float v = (bot - top) * yw + top;
        o[c] = (v - 0.5f) / 0.5f;
    }
}

hipError_t k_preprocess_u8(const unsigned char* slots, float* out, int n, long slot_stride, int channels, int in_h, int in_w,
                           int crop_h, int crop_w, hipStream_t st) {
    const dim3 grid((crop_h * crop_w + 255) / 256, n);
    if (channels == 3) preprocess_u8_kernel<3><<<grid, 256, 0, st>>>(slots, out, slot_stride, in_h, in_w, crop_h, crop_w);
    else preprocess_u8_kernel<1><<<grid, 256, 0, st>>>(slots, out, slot_stride, in_h, in_w, crop_h, crop_w);
    return hipGetLastError();
}
