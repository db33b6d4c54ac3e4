// layers.hip -- HBM-bound layer kernels of the BN / pooling nets (nets/resnet.py, nets/resnext.py,
// nets/shufflenet_v2.py) for gfx950: batch-norm statistics / apply / backward, ReLU backward,
// 3x3-s2 max-pool, global average pool, dropout, first-layer im2col.  Tensors are NHWC fp32, so the
// channel index is the fastest one: per-channel vectors are read once per thread and rows stream
// through 16-byte coalesced accesses.  All reductions are ordered (two-level, no float atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>
#include "layers.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// ---------------------------------------------------------------------------------------------------
// Per-channel statistics of x[rows, C].  grid = (ceil(C/64), splits), block = 64 channels x 4 row lanes.
// Each thread keeps shifted sums (shift = its first element) -> (n, mean, M2); lanes / blocks are merged
// with Chan's parallel-variance formula, so no E[x^2]-E[x]^2 cancellation in fp32.
// partial layout: part[(split*3 + {0:n,1:mean,2:M2}) * C + c]
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void chan_merge(float& n, float& mean, float& m2, float nb, float meanb, float m2b) {
    if (nb == 0.f) return;
    if (n == 0.f) { n = nb; mean = meanb; m2 = m2b; return; }
    const float tot = n + nb, d = meanb - mean;
    mean += d * (nb / tot);
    m2 += m2b + d * d * (n * nb / tot);
    n = tot;
}

// ---------------------------------------------------------------------------------------------------
// bf16 STORAGE of the BN nets (fte.h, "bf16 STORAGE": the *_s16 layer entry points).  A tensor is either fp32 or bf16 in HBM; the
// kernels below take the choice as template flags -- ZH: the pre-activation side (z, dz), AH: the activation side (y, the
// shortcut, dy, the masked gradient) -- and move 4 channels per lane either way (16 or 8 bytes).  Arithmetic is fp32; a value
// is rounded to nearest even once, where it is stored.  Pointers stay `float*` in the signatures; with the flag set they
// address bf16 elements.
// ---------------------------------------------------------------------------------------------------
typedef unsigned u32x2_l __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4_s __attribute__((ext_vector_type(4)));
template <bool H>
__device__ __forceinline__ f32x4 ldq(const float* p, long off) {            // 4 consecutive elements at element offset `off`
    if constexpr (H) {
        const u32x2_l u = *reinterpret_cast<const u32x2_l*>(reinterpret_cast<const unsigned short*>(p) + off);
        return f32x4{__builtin_bit_cast(float, u[0] << 16), __builtin_bit_cast(float, u[0] & 0xffff0000u),
                     __builtin_bit_cast(float, u[1] << 16), __builtin_bit_cast(float, u[1] & 0xffff0000u)};
    } else {
        return *reinterpret_cast<const f32x4*>(p + off);
    }
}
template <bool H>
__device__ __forceinline__ void stq(float* p, long off, const f32x4 v) {
    if constexpr (H) {
        *reinterpret_cast<u32x2_l*>(reinterpret_cast<unsigned short*>(p) + off) = __builtin_bit_cast(u32x2_l, __builtin_convertvector(v, bf16x4_s));
    } else {
        *reinterpret_cast<f32x4*>(p + off) = v;
    }
}
template <bool H>
__device__ __forceinline__ void st1(float* p, long off, float v) {
    if constexpr (H) reinterpret_cast<unsigned short*>(p)[off] = __builtin_bit_cast(unsigned short, (__bf16)v);
    else p[off] = v;
}

// ---------------------------------------------------------------------------------------------------
// In-launch finalize of the split reductions (BN statistics, BN backward sums): instead of a second, 7-8 us launch that merges
// the row splits (112 of ShuffleNet-v2's 688 launches per step were those), the blocks of a channel column hand their partials
// over inside the launch.  Two levels so that no block reads more than a few tens of KB: the splits of a column form groups of
// GROUP consecutive splits; the LAST block of a group to arrive merges the group's partials (in split order, whoever arrives
// last: deterministic) into a group partial; the last GROUP-merger of the column merges those and finalizes.  Hand-off =
// the write-through form of the programming guide's Guideline 16 (R1): every partial word is stored sc1 (relaxed agent-scope
// atomic store: straight to memory, no release fence -- a release fence in each of ~2000 blocks, tried first, made the step
// 12 % SLOWER than the separate launch) -> every wave s_waitcnt vmcnt(0) -> barrier -> one lane: relaxed agent-scope fetch_add
// on the arrival counter; the block that draws the last ticket reads EVERY partial word with sc1 loads (relaxed agent-scope
// atomic loads: past its L1, which may hold stale lines of a reused workspace).  Placement-independent (any XCD / CU).
// Counters come from a per-device ring that is zero when a launch starts; the last arriver puts its counter back to zero.
// ---------------------------------------------------------------------------------------------------
constexpr int TAIL_GROUP = 16;
struct SplitTail {
    unsigned* cnt;      // [columns][1 + groups] arrival counters of THIS launch; NULL: no in-launch finalize
    float* part2;       // group partials
    int splits, groups;
};
__device__ __forceinline__ bool last_arriver(unsigned* cnt, unsigned expected, volatile unsigned* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                        // every wave's partial stores have left; `flag`'s LDS is free
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = t + 1 == expected;
        if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the ring slot's next user finds zero
        *flag = last ? 1u : 0u;
    }
    __syncthreads();
    const bool last = *flag != 0u;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // no instruction: keeps the compiler from hoisting the partial loads
    return last;
}
// a word handed to another block inside the launch: write-through store / L1-bypassing load
__device__ __forceinline__ void st_wt(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_wt(const float* p) { return __hip_atomic_load(const_cast<float*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// (n, mean, M2) partials [s][3][C] of `cnt` splits merged as plain sums around the first one's mean (see bn_finalize_kernel)
__device__ __forceinline__ void stat_merge(const float* p, int cnt, int C, int ch, float& N, float& mean, float& M2) {
    const float m0 = ld_wt(p + C + ch);
    float n = 0.f, d1 = 0.f, q = 0.f;
#pragma unroll 4
    for (int s = 0; s < cnt; ++s) {
        const float* pp = p + (long)s * 3 * C;
        const float nb = ld_wt(pp + ch), d = ld_wt(pp + C + ch) - m0;
        n += nb;
        d1 += nb * d;
        q += ld_wt(pp + 2 * C + ch) + nb * d * d;
    }
    N = n;
    mean = n > 0.f ? m0 + d1 / n : m0;
    M2 = n > 0.f ? fmaxf(q - d1 * d1 / n, 0.f) : 0.f;
}
struct BnFwdOut {
    const float* gamma; const float* beta; float eps, decay;
    float* mean; float* rstd; float* scale; float* shift; float* mov_mean; float* mov_var;
};
__device__ __forceinline__ void bn_finalize_channel(const BnFwdOut& o, int ch, float N, float mean, float m2) {
    const float var = m2 / N;
    const float rstd = 1.f / sqrtf(var + o.eps);
    o.mean[ch] = mean;
    o.rstd[ch] = rstd;
    const float sc = o.gamma[ch] * rstd;
    o.scale[ch] = sc;
    o.shift[ch] = o.beta[ch] - mean * sc;
    if (o.mov_mean) {
        o.mov_mean[ch] = o.decay * o.mov_mean[ch] + (1.f - o.decay) * mean;
        o.mov_var[ch] = o.decay * o.mov_var[ch] + (1.f - o.decay) * (m2 / fmaxf(N - 1.f, 1.f));
    }
}
struct BnBwdOut {
    float count; const float* gamma; const float* mean; const float* rstd; float* dgamma; float* dbeta; float* coef;
};
__device__ __forceinline__ void bn_bwd_finalize_channel(const BnBwdOut& o, int C, int ch, float sg, float sgx) {
    o.dbeta[ch] = sg;
    o.dgamma[ch] = sgx;
    const float gr = o.gamma[ch] * o.rstd[ch];
    const float b = -gr * o.rstd[ch] * sgx / o.count;
    o.coef[ch] = gr;
    o.coef[C + ch] = b;
    o.coef[2 * C + ch] = -gr * sg / o.count - b * o.mean[ch];
}

// i = ((img*H + b)*W + a)*C4 + c4  ->  (c4, a, b, img).  Every tensor here has fewer than 2^32 16-byte units, and those indices
// are decomposed with 32-bit unsigned arithmetic: a 64-bit division by a run-time value is ~100 instructions, and the
// grid-stride loops below did up to six of them per 16 bytes of output (`small` is uniform: total <= 0xffffffff).
__device__ __forceinline__ void unflat4(long i, bool small, int C4, int W, int H, int& c4, int& a, int& b, int& img) {
    if (small) {
        unsigned u = (unsigned)i;
        c4 = (int)(u % (unsigned)C4); u /= (unsigned)C4;
        a = (int)(u % (unsigned)W); u /= (unsigned)W;
        b = (int)(u % (unsigned)H);
        img = (int)(u / (unsigned)H);
    } else {
        c4 = (int)(i % C4); long t = i / C4;
        a = (int)(t % W); t /= W;
        b = (int)(t % H);
        img = (int)(t / H);
    }
}
__device__ __forceinline__ void unflat2(long i, bool small, int Q, int& k, long& row) {
    if (small) { const unsigned u = (unsigned)i; k = (int)(u % (unsigned)Q); row = (long)(u / (unsigned)Q); }
    else { k = (int)(i % Q); row = i / Q; }
}

// scale*z + shift as ONE fused multiply-add per element: the backward pass recomputes the ReLU mask from z with this same
// expression (the normalised activation need not be kept, or even stored: bn_gather), so both must round identically
__device__ __forceinline__ f32x4 bn_affine(const f32x4 z, const f32x4 sc, const f32x4 sf) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(z[e], sc[e], sf[e]);
    return v;
}

__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                       long rows, int C, long rows_per_split) {
    __shared__ float sh[3][4][64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int ch = blockIdx.x * 64 + c;
    const long r0 = (long)blockIdx.y * rows_per_split, r1 = min(rows, r0 + rows_per_split);
    float n = 0.f, s = 0.f, ss = 0.f, shift = 0.f;
    if (ch < C) {
        long r = r0 + rl;
        if (r < r1) shift = x[r * C + ch];
        for (; r < r1; r += 4) {
            const float v = x[r * C + ch] - shift;
            s += v;
            ss += v * v;
            n += 1.f;
        }
    }
    float mean = 0.f, m2 = 0.f;
    if (n > 0.f) { mean = shift + s / n; m2 = ss - s * s / n; }
    sh[0][rl][c] = n; sh[1][rl][c] = mean; sh[2][rl][c] = m2;
    __syncthreads();
    if (rl == 0 && ch < C) {
#pragma unroll
        for (int l = 1; l < 4; ++l) chan_merge(n, mean, m2, sh[0][l][c], sh[1][l][c], sh[2][l][c]);
        float* pp = part + (long)blockIdx.y * 3 * C;
        pp[ch] = n; pp[C + ch] = mean; pp[2 * C + ch] = m2 < 0.f ? 0.f : m2;
    }
}

// float4 version: Q channel quads per row segment (Q = min(C/4, 64)), RL = 256/Q row lanes; a wave reads
// 1 KiB (C >= 256) or several whole rows per instruction, two rows in flight per lane.
template <int Q, bool ZH = false>
__global__ __launch_bounds__(256) void bn_stats_v4_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                          long rows, int C, long rows_per_split, SplitTail tail, BnFwdOut fin) {
    constexpr int RL = 256 / Q;
    __shared__ f32x4 sh[3][RL][Q];
    const int q = threadIdx.x % Q, rl = threadIdx.x / Q;
    const int ch = (blockIdx.x * Q + q) * 4;
    const long r0 = (long)blockIdx.y * rows_per_split, r1 = min(rows, r0 + rows_per_split);
    f32x4 n = {0.f, 0.f, 0.f, 0.f}, s = n, ss = n, shift = n;
    const bool ok = ch < C;
    if (ok) {
        long r = r0 + rl;
        if (r < r1) shift = ldq<ZH>(x, r * C + ch);
        for (; r + RL < r1; r += 2 * RL) {
            const f32x4 a = ldq<ZH>(x, r * C + ch) - shift;
            const f32x4 b = ldq<ZH>(x, (r + RL) * C + ch) - shift;
            s += a + b;
            ss += a * a + b * b;
            n += 2.f;
        }
        if (r < r1) {
            const f32x4 a = ldq<ZH>(x, r * C + ch) - shift;
            s += a;
            ss += a * a;
            n += 1.f;
        }
    }
    f32x4 mean = {0.f, 0.f, 0.f, 0.f}, m2 = mean;
    if (n[0] > 0.f) { mean = shift + s / n; m2 = ss - s * s / n; }
    sh[0][rl][q] = n; sh[1][rl][q] = mean; sh[2][rl][q] = m2;
    __syncthreads();
    if (rl == 0 && ok) {
        float* pp = part + (long)blockIdx.y * 3 * C;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float nn = n[e], mm = mean[e], m22 = m2[e];
            for (int l = 1; l < RL; ++l) chan_merge(nn, mm, m22, sh[0][l][q][e], sh[1][l][q][e], sh[2][l][q][e]);
            m22 = m22 < 0.f ? 0.f : m22;
            if (tail.cnt) { st_wt(pp + ch + e, nn); st_wt(pp + C + ch + e, mm); st_wt(pp + 2 * C + ch + e, m22); }
            else { pp[ch + e] = nn; pp[C + ch + e] = mm; pp[2 * C + ch + e] = m22; }
        }
    }
    if (!tail.cnt) return;                                   // the splits are merged by bn_finalize_kernel
    // ---- in-launch finalize: thread t of the column's last block owns channel col*4Q + t
    volatile unsigned* flag = reinterpret_cast<volatile unsigned*>(&sh[0][0][0]);
    unsigned* cnt = tail.cnt + blockIdx.x * (tail.groups + 1);
    const int g = blockIdx.y / TAIL_GROUP, s0 = g * TAIL_GROUP;
    const int gcount = min(TAIL_GROUP, tail.splits - s0);
    if (!last_arriver(cnt + 1 + g, (unsigned)gcount, flag)) return;
    const int fch = blockIdx.x * Q * 4 + threadIdx.x;
    const bool fok = threadIdx.x < Q * 4 && fch < C;
    float fN = 0.f, fmean = 0.f, fM2 = 0.f;
    if (fok) stat_merge(part + (long)s0 * 3 * C, gcount, C, fch, fN, fmean, fM2);
    if (tail.groups > 1) {
        if (fok) {
            float* p2 = tail.part2 + (long)g * 3 * C;
            st_wt(p2 + fch, fN); st_wt(p2 + C + fch, fmean); st_wt(p2 + 2 * C + fch, fM2);
        }
        if (!last_arriver(cnt, (unsigned)tail.groups, flag)) return;
        if (fok) stat_merge(tail.part2, tail.groups, C, fch, fN, fmean, fM2);
    }
    if (fok) bn_finalize_channel(fin, fch, fN, fmean, fM2);
}

template <int Q, bool ZH = false, bool AH = false>
__global__ __launch_bounds__(256) void bn_bwd_reduce_v4_kernel(const float* __restrict__ dy, const float* __restrict__ ymask,
                                                               const float* __restrict__ z, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, const float* __restrict__ zsc,
                                                               const float* __restrict__ zsf, float* __restrict__ gout,
                                                               float* __restrict__ part, long rows, int C, long rows_per_split,
                                                               SplitTail tail, BnBwdOut fin) {
    constexpr int RL = 256 / Q;
    __shared__ f32x4 sh[2][RL][Q];
    const int q = threadIdx.x % Q, rl = threadIdx.x / Q;
    const int ch = (blockIdx.x * Q + q) * 4;
    const long r0 = (long)blockIdx.y * rows_per_split, r1 = min(rows, r0 + rows_per_split);
    f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sgx = sg;
    const bool ok = ch < C;
    if (ok) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + ch), rs = *reinterpret_cast<const f32x4*>(rstd + ch);
        f32x4 sc = mu, sf = mu;
        if (zsc) { sc = *reinterpret_cast<const f32x4*>(zsc + ch); sf = *reinterpret_cast<const f32x4*>(zsf + ch); }
        auto one = [&](long r) {
            f32x4 g = ldq<AH>(dy, r * C + ch);
            const f32x4 zz = ldq<ZH>(z, r * C + ch);
            if (zsc) {                                   // ReLU mask recomputed from z (the output is not read)
                const f32x4 m = bn_affine(zz, sc, sf);
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = m[e] > 0.f ? g[e] : 0.f;
            } else if (ymask) {
                const f32x4 m = ldq<AH>(ymask, r * C + ch);
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = m[e] > 0.f ? g[e] : 0.f;
            }
            if (gout) stq<AH>(gout, r * C + ch, g);      // the masked gradient as a by-product (residual BN)
            sg += g;
            sgx += g * ((zz - mu) * rs);
        };
        long r = r0 + rl;
        // four rows in flight per lane (two were: 16 KB in flight per CU at 2 blocks of 256 threads -- the pass ran at 2.5-3.5 TB/s on
        // the 25-50 MB tensors of a 128-image shard)
        for (; r + 3 * RL < r1; r += 4 * RL) {
            f32x4 g4[4], z4[4], m4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                g4[u] = ldq<AH>(dy, (r + u * RL) * C + ch);
                z4[u] = ldq<ZH>(z, (r + u * RL) * C + ch);
                if (!zsc && ymask) m4[u] = ldq<AH>(ymask, (r + u * RL) * C + ch);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                f32x4 g = g4[u];
                if (zsc) {
                    const f32x4 m = bn_affine(z4[u], sc, sf);
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[e] = m[e] > 0.f ? g[e] : 0.f;
                } else if (ymask) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[e] = m4[u][e] > 0.f ? g[e] : 0.f;
                }
                if (gout) stq<AH>(gout, (r + u * RL) * C + ch, g);
                sg += g;
                sgx += g * ((z4[u] - mu) * rs);
            }
        }
        for (; r < r1; r += RL) one(r);
    }
    sh[0][rl][q] = sg; sh[1][rl][q] = sgx;
    __syncthreads();
    if (rl == 0 && ok) {
        float* pp = part + (long)blockIdx.y * 2 * C;
        f32x4 a = sg, b = sgx;
        for (int l = 1; l < RL; ++l) { a += sh[0][l][q]; b += sh[1][l][q]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (tail.cnt) { st_wt(pp + ch + e, a[e]); st_wt(pp + C + ch + e, b[e]); }
            else { pp[ch + e] = a[e]; pp[C + ch + e] = b[e]; }
        }
    }
    if (!tail.cnt) return;                                   // the splits are summed by bn_bwd_finalize_kernel
    volatile unsigned* flag = reinterpret_cast<volatile unsigned*>(&sh[0][0][0]);
    unsigned* cnt = tail.cnt + blockIdx.x * (tail.groups + 1);
    const int g = blockIdx.y / TAIL_GROUP, s0 = g * TAIL_GROUP;
    const int gcount = min(TAIL_GROUP, tail.splits - s0);
    if (!last_arriver(cnt + 1 + g, (unsigned)gcount, flag)) return;
    const int fch = blockIdx.x * Q * 4 + threadIdx.x;
    const bool fok = threadIdx.x < Q * 4 && fch < C;
    float a = 0.f, b = 0.f;
    if (fok) {
#pragma unroll 4
        for (int s = 0; s < gcount; ++s) { const float* pp = part + (long)(s0 + s) * 2 * C; a += ld_wt(pp + fch); b += ld_wt(pp + C + fch); }
    }
    if (tail.groups > 1) {
        if (fok) { float* p2 = tail.part2 + (long)g * 2 * C; st_wt(p2 + fch, a); st_wt(p2 + C + fch, b); }
        if (!last_arriver(cnt, (unsigned)tail.groups, flag)) return;
        if (fok) {
            a = 0.f; b = 0.f;
#pragma unroll 4
            for (int s = 0; s < tail.groups; ++s) { const float* pp = tail.part2 + (long)s * 2 * C; a += ld_wt(pp + fch); b += ld_wt(pp + C + fch); }
        }
    }
    if (fok) bn_bwd_finalize_channel(fin, C, fch, a, b);
}

// merge the splits; scale = gamma*rstd, shift = beta - mean*scale; moving statistics (decay, unbiased var).
// Block = 16 channels x 16 split lanes.  The split partials (n_i, mean_i, M2_i) are combined as plain sums around the first
// split's mean m0:  N = sum n_i,  D = sum n_i*(mean_i - m0),  Q = sum [M2_i + n_i*(mean_i - m0)^2]  ->  mean = m0 + D/N,
// M2 = Q - D^2/N.  The split means differ from m0 by ~sigma/sqrt(n_i), so the subtraction cancels nothing that matters, every
// load is independent (4 splits = 12 loads in flight per lane and trip) and there is no chain of dependent divisions: the
// pairwise Chan merges this replaces, one after the other, took 10.7 us -- longer than the statistics pass itself.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int splits, int C,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, float decay, float* __restrict__ mean_out,
                                                          float* __restrict__ rstd_out, float* __restrict__ scale,
                                                          float* __restrict__ shift, float* __restrict__ mov_mean,
                                                          float* __restrict__ mov_var) {
    __shared__ float sh[3][16][16];
    const int cl = threadIdx.x & 15, lane = threadIdx.x >> 4;
    const int ch = blockIdx.x * 16 + cl;
    const bool ok = ch < C;
    const float m0 = ok ? part[C + ch] : 0.f;
    float n = 0.f, d1 = 0.f, q = 0.f;
    if (ok)
        for (int s0 = lane; s0 < splits; s0 += 64) {
            float nb[4], mb[4], qb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = s0 + 16 * u;
                const bool in = s < splits;
                const float* pp = part + (long)(in ? s : 0) * 3 * C;
                nb[u] = in ? pp[ch] : 0.f; mb[u] = pp[C + ch]; qb[u] = in ? pp[2 * C + ch] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float d = mb[u] - m0;
                n += nb[u];
                d1 += nb[u] * d;
                q += qb[u] + nb[u] * d * d;
            }
        }
    sh[0][lane][cl] = n; sh[1][lane][cl] = d1; sh[2][lane][cl] = q;
    __syncthreads();
    if (lane != 0 || !ok) return;
    float N = 0.f, D = 0.f, Q = 0.f;
#pragma unroll
    for (int l = 0; l < 16; ++l) { N += sh[0][l][cl]; D += sh[1][l][cl]; Q += sh[2][l][cl]; }
    const float mean = m0 + D / N;
    const float m2 = fmaxf(Q - D * D / N, 0.f);
    const float var = m2 / N;
    const float rstd = 1.f / sqrtf(var + eps);
    mean_out[ch] = mean;
    rstd_out[ch] = rstd;
    const float sc = gamma[ch] * rstd;
    scale[ch] = sc;
    shift[ch] = beta[ch] - mean * sc;
    if (mov_mean) {
        mov_mean[ch] = decay * mov_mean[ch] + (1.f - decay) * mean;
        mov_var[ch] = decay * mov_var[ch] + (1.f - decay) * (m2 / fmaxf(N - 1.f, 1.f));
    }
}

// The same merge for MANY partial rows (the conv epilogues of the "BN fusion" path leave one per tile row: up to M / 64 of them):
// block = ONE channel quad x 256 split lanes, 16-byte loads, four rows in flight per lane; wave shuffles, then the four waves
// through LDS.  Same sums around the first split's mean as above.
__device__ __forceinline__ f32x4 wave_sum4(f32x4 v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x = v[e];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) x += __shfl_xor(x, m);
        v[e] = x;
    }
    return v;
}
__global__ __launch_bounds__(256) void bn_finalize_wide_kernel(const float* __restrict__ part, int splits, int C,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float eps, float decay, float* __restrict__ mean_out,
                                                               float* __restrict__ rstd_out, float* __restrict__ scale,
                                                               float* __restrict__ shift, float* __restrict__ mov_mean,
                                                               float* __restrict__ mov_var) {
    __shared__ f32x4 sh[3][4];
    const int ch = blockIdx.x * 4, lane = threadIdx.x;
    const f32x4 m0 = *reinterpret_cast<const f32x4*>(part + C + ch);
    f32x4 n = {0.f, 0.f, 0.f, 0.f}, d1 = n, q = n;
    for (int s0 = lane; s0 < splits; s0 += 1024) {
        f32x4 nb[4], mb[4], qb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = s0 + 256 * u;
            const bool in = s < splits;
            const float* pp = part + (long)(in ? s : 0) * 3 * C + ch;
            nb[u] = *reinterpret_cast<const f32x4*>(pp); mb[u] = *reinterpret_cast<const f32x4*>(pp + C); qb[u] = *reinterpret_cast<const f32x4*>(pp + 2 * C);
            if (!in) { nb[u] = f32x4{0.f, 0.f, 0.f, 0.f}; qb[u] = nb[u]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f32x4 d = mb[u] - m0;
            n += nb[u];
            d1 += nb[u] * d;
            q += qb[u] + nb[u] * d * d;
        }
    }
    n = wave_sum4(n); d1 = wave_sum4(d1); q = wave_sum4(q);
    if ((lane & 63) == 0) { sh[0][lane >> 6] = n; sh[1][lane >> 6] = d1; sh[2][lane >> 6] = q; }
    __syncthreads();
    if (lane >= 4) return;
    float N = 0.f, D = 0.f, Q = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { N += sh[0][w][lane]; D += sh[1][w][lane]; Q += sh[2][w][lane]; }
    const int c = ch + lane;
    const float mean = m0[lane] + D / N;
    const float m2 = fmaxf(Q - D * D / N, 0.f);
    const float rstd = 1.f / sqrtf(m2 / N + eps);
    mean_out[c] = mean;
    rstd_out[c] = rstd;
    const float sc = gamma[c] * rstd;
    scale[c] = sc;
    shift[c] = beta[c] - mean * sc;
    if (mov_mean) {
        mov_mean[c] = decay * mov_mean[c] + (1.f - decay) * mean;
        mov_var[c] = decay * mov_var[c] + (1.f - decay) * (m2 / fmaxf(N - 1.f, 1.f));
    }
}

// inference-mode scale/shift from the moving statistics
__global__ __launch_bounds__(256) void bn_infer_coef_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ mm, const float* __restrict__ mv,
                                                            float eps, int C, float* __restrict__ scale, float* __restrict__ shift) {
    const int ch = blockIdx.x * 256 + threadIdx.x;
    if (ch >= C) return;
    const float sc = gamma[ch] / sqrtf(mv[ch] + eps);
    scale[ch] = sc;
    shift[ch] = beta[ch] - mm[ch] * sc;
}

// y = [relu]( scale[c]*z + shift[c] [+ res] )
// The launchers size the grid so that the grid stride is a multiple of C/4 (grid_for_c): a thread then stays on ONE channel
// quad for its whole walk, and scale / shift are loaded once instead of being re-derived -- with a 64-bit modulo -- for every
// 16 bytes (`inv`; any other grid still works through the per-iteration path).
template <bool ZH = false, bool AH = false>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ res,
                                                       float* __restrict__ y, long n4, int C, int relu) {
    const unsigned q = (unsigned)C >> 2;
    const long step = (long)gridDim.x * 256;
    const bool inv = step % q == 0;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    int c = (int)((unsigned)(i % q) << 2);
    f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c), sf = *reinterpret_cast<const f32x4*>(shift + c);
    if (inv) {
        // four 16-byte (8-byte) pieces in flight per thread: the walk is a latency-bound stream at the 25-50 MB of a 128-image shard
        for (; i + 3 * step < n4; i += 4 * step) {
            f32x4 zv[4], rv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                zv[u] = ldq<ZH>(z, (i + u * step) * 4);
                if (res) rv[u] = ldq<AH>(res, (i + u * step) * 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                f32x4 v = bn_affine(zv[u], sc, sf);
                if (res) v += rv[u];
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                stq<AH>(y, (i + u * step) * 4, v);
            }
        }
    }
    for (; i < n4; i += step) {
        if (!inv) {
            c = (int)((unsigned)(i % q) << 2);
            sc = *reinterpret_cast<const f32x4*>(scale + c); sf = *reinterpret_cast<const f32x4*>(shift + c);
        }
        f32x4 v = bn_affine(ldq<ZH>(z, i * 4), sc, sf);
        if (res) v += ldq<AH>(res, i * 4);
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        stq<AH>(y, i * 4, v);
    }
}

// g = dy * (y > 0)
template <bool AH = false>
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                       float* __restrict__ g, long n4) {
    const long step = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * step < n4; i += 4 * step) {              // four pieces of both inputs in flight per thread (l_relu_bwd caps the grid)
        f32x4 d[4], yy[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { d[u] = ldq<AH>(dy, (i + u * step) * 4); yy[u] = ldq<AH>(y, (i + u * step) * 4); }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int e = 0; e < 4; ++e) d[u][e] = yy[u][e] > 0.f ? d[u][e] : 0.f;
            stq<AH>(g, (i + u * step) * 4, d[u]);
        }
    }
    for (; i < n4; i += step) {
        f32x4 d = ldq<AH>(dy, i * 4);
        const f32x4 yy = ldq<AH>(y, i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = yy[e] > 0.f ? d[e] : 0.f;
        stq<AH>(g, i * 4, d);
    }
}

// partial sums for the BN backward: sum g and sum g*xhat per channel, g = dy * (ymask > 0 if given)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ ymask,
                                                            const float* __restrict__ z, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ zsc,
                                                            const float* __restrict__ zsf, float* __restrict__ gout,
                                                            float* __restrict__ part, long rows, int C, long rows_per_split) {
    __shared__ float sh[2][4][64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int ch = blockIdx.x * 64 + c;
    const long r0 = (long)blockIdx.y * rows_per_split, r1 = min(rows, r0 + rows_per_split);
    float sg = 0.f, sgx = 0.f;
    if (ch < C) {
        const float mu = mean[ch], rs = rstd[ch];
        const float sc = zsc ? zsc[ch] : 0.f, sf = zsc ? zsf[ch] : 0.f;
        for (long r = r0 + rl; r < r1; r += 4) {
            float g = dy[r * C + ch];
            const float zz = z[r * C + ch];
            if (zsc ? !(__builtin_fmaf(zz, sc, sf) > 0.f) : (ymask && !(ymask[r * C + ch] > 0.f))) g = 0.f;
            if (gout) gout[r * C + ch] = g;
            sg += g;
            sgx += g * ((zz - mu) * rs);
        }
    }
    sh[0][rl][c] = sg; sh[1][rl][c] = sgx;
    __syncthreads();
    if (rl == 0 && ch < C) {
        float* pp = part + (long)blockIdx.y * 2 * C;
        pp[ch] = (sh[0][0][c] + sh[0][1][c]) + (sh[0][2][c] + sh[0][3][c]);
        pp[C + ch] = (sh[1][0][c] + sh[1][1][c]) + (sh[1][2][c] + sh[1][3][c]);
    }
}

// dgamma, dbeta and the coefficients of dz = A*g + B*z + C0
// (pg / pgx: the two partial-sum arrays, `ld` floats apart from split to split: [s][2][C] of the reduce kernels is (part, part + C,
// 2 C); the conv epilogues of the "BN fusion" path write two separate [rows][C] arrays)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ pg, const float* __restrict__ pgx, long ld,
                                                              int splits, int C, float count,
                                                              const float* __restrict__ gamma, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float* __restrict__ coef) {
    __shared__ float sh[2][16][16];                          // 16 channels x 16 split lanes, as bn_finalize_kernel
    const int cl = threadIdx.x & 15, lane = threadIdx.x >> 4;
    const int ch = blockIdx.x * 16 + cl;
    float sg = 0.f, sgx = 0.f;
    if (ch < C) {
#pragma unroll 4
        for (int s = lane; s < splits; s += 16) {
            sg += pg[(long)s * ld + ch];
            sgx += pgx[(long)s * ld + ch];
        }
    }
    sh[0][lane][cl] = sg; sh[1][lane][cl] = sgx;
    __syncthreads();
    if (lane != 0 || ch >= C) return;
#pragma unroll
    for (int l = 1; l < 16; ++l) { sg += sh[0][l][cl]; sgx += sh[1][l][cl]; }
    dbeta[ch] = sg;
    dgamma[ch] = sgx;
    const float gr = gamma[ch] * rstd[ch];
    const float b = -gr * rstd[ch] * sgx / count;
    coef[ch] = gr;
    coef[C + ch] = b;
    coef[2 * C + ch] = -gr * sg / count - b * mean[ch];
}

// many partial rows (conv epilogues): one channel quad x 256 split lanes per block, as bn_finalize_wide_kernel
__global__ __launch_bounds__(256) void bn_bwd_finalize_wide_kernel(const float* __restrict__ pg, const float* __restrict__ pgx, long ld,
                                                                   int splits, int C, float count,
                                                                   const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd, float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta, float* __restrict__ coef) {
    __shared__ f32x4 sh[2][4];
    const int ch = blockIdx.x * 4, lane = threadIdx.x;
    f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sgx = sg;
    for (int s0 = lane; s0 < splits; s0 += 1024) {
        f32x4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = s0 + 256 * u;
            const bool in = s < splits;
            a[u] = *reinterpret_cast<const f32x4*>(pg + (long)(in ? s : 0) * ld + ch);
            b[u] = *reinterpret_cast<const f32x4*>(pgx + (long)(in ? s : 0) * ld + ch);
            if (!in) { a[u] = f32x4{0.f, 0.f, 0.f, 0.f}; b[u] = a[u]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { sg += a[u]; sgx += b[u]; }
    }
    sg = wave_sum4(sg); sgx = wave_sum4(sgx);
    if ((lane & 63) == 0) { sh[0][lane >> 6] = sg; sh[1][lane >> 6] = sgx; }
    __syncthreads();
    if (lane >= 4) return;
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { a += sh[0][w][lane]; b += sh[1][w][lane]; }
    const int c = ch + lane;
    dbeta[c] = a;
    dgamma[c] = b;
    const float gr = gamma[c] * rstd[c];
    const float bb = -gr * rstd[c] * b / count;
    coef[c] = gr;
    coef[C + c] = bb;
    coef[2 * C + c] = -gr * a / count - bb * mean[c];
}

// (round 4's one-launch cooperative BN backward -- reduce, wait for the column's blocks on device-global counters, apply -- was measured
// 2x slower per call than the three launches and 15 % slower per step (profiles/r4_notes.md 7) and relied on a bounded spin that could fall
// through with incomplete partials; it was removed in round 5)
template <bool ZH = false, bool AH = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ ymask,
                                                           const float* __restrict__ z, const float* __restrict__ coef,
                                                           const float* __restrict__ zsc, const float* __restrict__ zsf,
                                                           float* __restrict__ dz, long n4, int C) {
    const unsigned q = (unsigned)C >> 2;
    const long step = (long)gridDim.x * 256;
    const bool inv = step % q == 0;                            // one channel quad per thread (see bn_apply_kernel)
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    f32x4 A, B, C0, msc = {0.f, 0.f, 0.f, 0.f}, msf = msc;
    auto coefs = [&](long ii) {
        const int c = (int)((unsigned)(ii % q) << 2);
        A = *reinterpret_cast<const f32x4*>(coef + c); B = *reinterpret_cast<const f32x4*>(coef + C + c);
        C0 = *reinterpret_cast<const f32x4*>(coef + 2 * C + c);
        if (zsc) { msc = *reinterpret_cast<const f32x4*>(zsc + c); msf = *reinterpret_cast<const f32x4*>(zsf + c); }
    };
    coefs(i);
    if (inv) {
        for (; i + 3 * step < n4; i += 4 * step) {          // four pieces of each input in flight per thread (see bn_apply_kernel)
            f32x4 gv[4], zv[4], mv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                gv[u] = ldq<AH>(dy, (i + u * step) * 4);
                zv[u] = ldq<ZH>(z, (i + u * step) * 4);
                if (!zsc && ymask) mv[u] = ldq<AH>(ymask, (i + u * step) * 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                f32x4 g = gv[u];
                if (zsc) {
                    const f32x4 m = bn_affine(zv[u], msc, msf);
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[e] = m[e] > 0.f ? g[e] : 0.f;
                } else if (ymask) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[e] = mv[u][e] > 0.f ? g[e] : 0.f;
                }
                stq<ZH>(dz, (i + u * step) * 4, A * g + B * zv[u] + C0);
            }
        }
    }
    for (; i < n4; i += step) {
        if (!inv) coefs(i);
        f32x4 g = ldq<AH>(dy, i * 4);
        const f32x4 zz = ldq<ZH>(z, i * 4);
        if (zsc) {
            const f32x4 m = bn_affine(zz, msc, msf);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = m[e] > 0.f ? g[e] : 0.f;
        } else if (ymask) {
            const f32x4 m = ldq<AH>(ymask, i * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = m[e] > 0.f ? g[e] : 0.f;
        }
        stq<ZH>(dz, i * 4, A * g + B * zz + C0);
    }
}

// ---------------------------------------------------------------------------------------------------
// max_pool 3x3 stride 2 SAME.  idx = window position (0..8) of the FIRST maximum; backward is a gather
// over the <= 4 windows that cover an input pixel (ordered, no atomics).
// ---------------------------------------------------------------------------------------------------
template <bool AH = false>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          uint8_t* __restrict__ idx, int n, int h, int w, int c,
                                                          int ho, int wo, int pt, int pl) {
    const long total = (long)n * ho * wo * (c >> 2);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int c4, ow, oh, img;
        unflat4(i, total <= 0xffffffffL, (c >> 2), wo, ho, c4, ow, oh, img);
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int ih = oh * 2 + r - pt;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int iw = ow * 2 + s - pl;
                if (ih < 0 || ih >= h || iw < 0 || iw >= w) continue;
                const f32x4 v = ldq<AH>(x, ((long)(img * h + ih) * w + iw) * c + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (v[e] > best[e]) { best[e] = v[e]; bi[e] = r * 3 + s; }
            }
        }
        stq<AH>(y, i * 4, best);
        *reinterpret_cast<uchar4*>(idx + i * 4) = make_uchar4(bi[0], bi[1], bi[2], bi[3]);
    }
}

template <bool AH = false>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                          float* __restrict__ dx, int n, int h, int w, int c,
                                                          int ho, int wo, int pt, int pl) {
    const long total = (long)n * h * w * (c >> 2);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int c4, iw, ih, img;
        unflat4(i, total <= 0xffffffffL, (c >> 2), w, h, c4, iw, ih, img);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // windows oh with oh*2 - pt <= ih <= oh*2 - pt + 2
        for (int oh = (ih + pt - 2 + 1) >> 1; oh * 2 - pt <= ih; ++oh) {
            if (oh < 0 || oh >= ho) continue;
            const int r = ih - (oh * 2 - pt);
            for (int ow = (iw + pl - 2 + 1) >> 1; ow * 2 - pl <= iw; ++ow) {
                if (ow < 0 || ow >= wo) continue;
                const int s = iw - (ow * 2 - pl);
                const long o = (((long)(img * ho + oh) * wo + ow) * c) + c4 * 4;
                const uchar4 k = *reinterpret_cast<const uchar4*>(idx + o);
                const f32x4 d = ldq<AH>(dy, o);
                const int pos = r * 3 + s;
                if (k.x == pos) acc[0] += d[0];
                if (k.y == pos) acc[1] += d[1];
                if (k.z == pos) acc[2] += d[2];
                if (k.w == pos) acc[3] += d[3];
            }
        }
        stq<AH>(dx, i * 4, acc);
    }
}

// Even image, no leading pad (112 / 56 / 28 inputs): a thread owns a 2x2 input patch of 4 channels.  Window (a,b) covers rows
// 2a..2a+2 and columns 2b..2b+2, so the patch (2a.., 2b..) is touched by exactly the windows (a-1,b-1), (a-1,b), (a,b-1), (a,b):
// 4 window loads serve 4 input pixels (the per-pixel gather above issues 9 for them, behind data-dependent loop bounds) and are
// summed in the same order, so the result is bit-identical.
template <bool AH = false>
__global__ __launch_bounds__(256) void maxpool_bwd_even_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                               float* __restrict__ dx, int n, int h, int w, int c) {
    const int ho = h >> 1, wo = w >> 1, cq = c >> 2;
    const long total = (long)n * ho * wo * cq;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int c4, b, a, img;
        unflat4(i, total <= 0xffffffffL, cq, wo, ho, c4, b, a, img);
        f32x4 p00 = {0.f, 0.f, 0.f, 0.f}, p01 = p00, p10 = p00, p11 = p00;
        auto window = [&](int oh, int ow, uchar4& k, f32x4& d) {
            const long o = (((long)(img * ho + oh) * wo + ow) * c) + c4 * 4;
            k = *reinterpret_cast<const uchar4*>(idx + o);
            d = ldq<AH>(dy, o);
        };
        auto take = [](f32x4& acc, const uchar4& k, const f32x4& d, int pos) {
            if (k.x == pos) acc[0] += d[0];
            if (k.y == pos) acc[1] += d[1];
            if (k.z == pos) acc[2] += d[2];
            if (k.w == pos) acc[3] += d[3];
        };
        // the four windows' loads go out together (clamped coordinates, results discarded where the window does not exist); the sums
        // keep their order
        uchar4 k0, k1, k2, k3; f32x4 d0, d1, d2, d3;
        const int am = a > 0 ? a - 1 : 0, bm = b > 0 ? b - 1 : 0;
        window(am, bm, k0, d0);
        window(am, b, k1, d1);
        window(a, bm, k2, d2);
        window(a, b, k3, d3);
        if (a > 0 && b > 0) take(p00, k0, d0, 8);
        if (a > 0) { take(p00, k1, d1, 6); take(p01, k1, d1, 7); }
        if (b > 0) { take(p00, k2, d2, 2); take(p10, k2, d2, 5); }
        take(p00, k3, d3, 0); take(p01, k3, d3, 1); take(p10, k3, d3, 3); take(p11, k3, d3, 4);
        const long o = (((long)(img * h + 2 * a) * w + 2 * b) * c) + c4 * 4;
        stq<AH>(dx, o, p00);
        stq<AH>(dx, o + c, p01);
        stq<AH>(dx, o + (long)w * c, p10);
        stq<AH>(dx, o + (long)w * c + c, p11);
    }
}

// ---------------------------------------------------------------------------------------------------
// global average pool: y[n,c] = mean_hw x[n,h,w,c]
// ---------------------------------------------------------------------------------------------------
// block = one image x 16 channel quads x 16 row lanes (float4 loads, two rows in flight per lane), lanes summed through LDS in a
// fixed order.  (One thread per (image, channel) walking all hw positions serially left a 128-image SE squeeze at 128 blocks
// and 52 us for <= 51 MB.)
template <bool AH = false>
__global__ __launch_bounds__(256) void gap_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int hw, int c) {
    __shared__ f32x4 sh[16][16];
    const int q = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int ch = (blockIdx.x * 16 + q) * 4, img = blockIdx.y;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (ch < c) {
        const long px = (long)img * hw * c + ch;
        int r = rl;
        for (; r + 16 < hw; r += 32) {
            s0 += ldq<AH>(x, px + (long)r * c);
            s1 += ldq<AH>(x, px + (long)(r + 16) * c);
        }
        if (r < hw) s0 += ldq<AH>(x, px + (long)r * c);
    }
    sh[rl][q] = s0 + s1;
    __syncthreads();
    if (rl == 0 && ch < c) {
        f32x4 s = sh[0][q];
#pragma unroll
        for (int l = 1; l < 16; ++l) s += sh[l][q];
        *reinterpret_cast<f32x4*>(y + (long)img * c + ch) = s / (float)hw;
    }
}
template <bool AH = false>
__global__ __launch_bounds__(256) void gap_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long total, int hw, int c) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % c);
        const long img = i / ((long)hw * c);
        st1<AH>(dx, i, dy[img * c + ch] / hw);
    }
}

// ---------------------------------------------------------------------------------------------------
// dropout (inverted): mask[i] = U(seed, i) < keep ; y = x * mask / keep.  The generator is a 64-bit
// mix (splitmix64) of (seed, element index): stateless, reproducible for a given seed, different per rank
// when the caller folds the rank into the seed.  TF's Philox stream is not reproduced (cannot be).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t v) {
    v += 0x9E3779B97F4A7C15ull;
    v = (v ^ (v >> 30)) * 0xBF58476D1CE4E5B9ull;
    v = (v ^ (v >> 27)) * 0x94D049BB133111EBull;
    return v ^ (v >> 31);
}
__global__ __launch_bounds__(256) void dropout_fwd_kernel(const float* __restrict__ x, float* __restrict__ mask,
                                                          float* __restrict__ y, long n, float keep, uint64_t seed) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float u = (float)(splitmix64(seed * 0x100000001B3ull + (uint64_t)i) >> 40) * (1.0f / 16777216.0f);
        const float m = u < keep ? 1.f : 0.f;
        mask[i] = m;
        y[i] = x[i] * m / keep;
    }
}
__global__ __launch_bounds__(256) void scale_mask_kernel(const float* __restrict__ dy, const float* __restrict__ mask,
                                                         float* __restrict__ dx, long n, float inv_keep) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dx[i] = dy[i] * mask[i] * inv_keep;
}

// ---------------------------------------------------------------------------------------------------
// im2col of the first layer (Cin = 1 or 3, any k): cols[m, kpad], k ordered (r, s, c) like HWIO rows,
// columns >= k*k*cin are zero.  Turns the 7x7-s2 stem of nets/resnet.py:109 into a dense MFMA GEMM.
// ---------------------------------------------------------------------------------------------------
// KS / CIN > 0: compile-time filter size and channel count (the 7x7x3 stem: every index decomposition below becomes multiplies and
// shifts; with run-time divisors the bf16 form of this kernel -- 8-byte stores, twice the index work per byte -- took 204 us for
// 128 images against 123 us for the fp32 form)
template <bool H = false, int KS = 0, int CIN = 0>      // H: cols are bf16 (bf16 storage: the stem GEMM then reads half the bytes and runs on the bf16-source kernels)
__global__ __launch_bounds__(256) void im2col_first_kernel(const float* __restrict__ x, float* __restrict__ cols,
                                                           int n, int h, int w, int cin_, int ks_, int stride, int ho, int wo,
                                                           int pt, int pl, int kpad) {
    const int cin = CIN > 0 ? CIN : cin_, ks = KS > 0 ? KS : ks_;
    const unsigned k4 = (unsigned)kpad >> 2;
    const long total = (long)n * ho * wo * k4;
    const int kreal = ks * ks * cin;
    const bool small = total <= 0xffffffffL;
    auto gather = [&](long i) {
        int kc, ow, oh, img;
        if (small) {
            unsigned u = (unsigned)i;
            kc = (int)(u % k4) * 4; u /= k4;
            ow = (int)(u % (unsigned)wo); u /= (unsigned)wo;
            oh = (int)(u % (unsigned)ho);
            img = (int)(u / (unsigned)ho);
        } else {
            kc = (int)(i % k4) * 4;
            long t = i / k4;
            ow = (int)(t % wo); t /= wo;
            oh = (int)(t % ho);
            img = (int)(t / ho);
        }
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = kc + e;
            if (k < kreal) {
                const int cch = k % cin, rs = k / cin, s = rs % ks, r = rs / ks;
                const int ih = oh * stride + r - pt, iw = ow * stride + s - pl;
                if (ih >= 0 && ih < h && iw >= 0 && iw < w) v[e] = x[((long)(img * h + ih) * w + iw) * cin + cch];
            }
        }
        return v;
    };
    const long step = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * step < total; i += 4 * step) {              // four quads' gathers in flight per thread
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = gather(i + u * step);
#pragma unroll
        for (int u = 0; u < 4; ++u) stq<H>(cols, (i + u * step) * 4, v[u]);
    }
    for (; i < total; i += step) stq<H>(cols, i * 4, gather(i));
}

// the same columns from LDS: block = one output row (img, oh); its KS input rows (w * CIN floats each, contiguous) are copied in with
// coalesced loads, and a thread that KEEPS its eight columns -- (r, s, channel) decoded once -- picks them out of LDS for every PX-th
// pixel: one 16-byte store of bf16 columns per thread and pixel.  (The flat walk above gathers every column with its own 4-byte global
// load behind ~40 instructions of index arithmetic: 81 us for the 7x7x3 stem at 128 images, neither its bytes nor its stores.)
template <bool H, int KS, int CIN, int K4, int PX>
__global__ __launch_bounds__(K4 / 2 * PX) void im2col_first_rows_kernel(const float* __restrict__ x, float* __restrict__ cols,
                                                                        int h, int w, int stride, int ho, int wo, int pt, int pl) {
    extern __shared__ float xs[];                                   // [KS][w * CIN]
    constexpr int KREAL = KS * KS * CIN, K8 = K4 / 2, NT = K8 * PX;
    const int row = blockIdx.x, img = row / ho, oh = row - img * ho;
    const int wc = w * CIN;
    for (int r = 0; r < KS; ++r) {
        const int ih = oh * stride + r - pt;
        const bool ok = ih >= 0 && ih < h;
        const float* src = x + ((long)(img * h + (ok ? ih : 0)) * w) * CIN;
        for (int t = threadIdx.x; t < wc; t += NT) xs[r * wc + t] = ok ? src[t] : 0.f;
    }
    __syncthreads();
    const int ko = threadIdx.x % K8, p0 = threadIdx.x / K8;
    int loff[8], sx[8];
    bool kok[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = ko * 8 + e;
        const int cch = k % CIN, rs = k / CIN, s_ = rs % KS, r = rs / KS;
        kok[e] = k < KREAL;
        loff[e] = (kok[e] ? r : 0) * wc + cch;
        sx[e] = s_ - pl;
    }
    for (int ow = p0; ow < wo; ow += PX) {
        f32x4 v0, v1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int iw = ow * stride + sx[e];
            const bool ok = kok[e] && iw >= 0 && iw < w;
            const float t = xs[loff[e] + (ok ? iw : 0) * CIN];
            if (e < 4) v0[e] = ok ? t : 0.f; else v1[e - 4] = ok ? t : 0.f;
        }
        const long o = ((long)row * wo + ow) * (K4 * 4) + ko * 8;
        if constexpr (H) {
            typedef unsigned u32x4_i __attribute__((ext_vector_type(4)));
            const u32x2_l a = __builtin_bit_cast(u32x2_l, __builtin_convertvector(v0, bf16x4_s)), b = __builtin_bit_cast(u32x2_l, __builtin_convertvector(v1, bf16x4_s));
            *reinterpret_cast<u32x4_i*>(reinterpret_cast<unsigned short*>(cols) + o) = u32x4_i{a[0], a[1], b[0], b[1]};
        } else {
            stq<false>(cols, o, v0);
            stq<false>(cols, o + 4, v1);
        }
    }
}

inline int grid_for(long n) { long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b)); }
// grid for a walk over [rows][C] in 16-byte steps whose stride (blocks * 256 threads) is a multiple of C / 4: every thread keeps
// its channel quad (bn_apply_kernel).  blocks is rounded UP to a multiple of (C/4) / gcd(C/4, 256).
inline int grid_for_c(long n4, int C, bool wide = false) {      // wide: fp32 tensors (16 bytes per lane and piece)
    long q = C / 4, a = q, b = 256;
    while (b) { const long t = a % b; a = b; b = t; }
    const long m = q / a;                                       // blocks must be a multiple of m
    long blocks = grid_for(n4);
    // about 4 resident blocks per CU (measured at 128 images, ms per ResNeXt-50 / SE-ResNet-50 / ResNet-50 step: 512 blocks 7.53 / 9.33 / 7.13,
    // 768: 7.52 / 9.26 / 7.10, 1024: 7.56 / 9.27 / 7.08, 2048: 7.68 / 9.44 / 7.22, one piece per thread: 7.70 / 9.44 / 7.24), each thread walking >= 4 pieces with all of them in flight (bn_apply_kernel's unrolled walk):
    // one piece per thread on 6000+ blocks left 32 KB in flight per CU
    // fp32 tensors put twice the bytes in flight per lane: two blocks per CU there (ShuffleNet-v2 fp32 @256, ms per step: 256 blocks 8.08,
    // 384: 8.05, 512: 8.02, 1024: 8.11, 2048: 8.14; the bf16-storage nets: 512 -> 1024 6.73 -> 6.71 / 6.35 -> 6.32 / 8.11 -> 8.07)
    static const long cap_env = getenv("FTE_BN_APPLY_BLOCKS") ? atol(getenv("FTE_BN_APPLY_BLOCKS")) : 0;      // tuning hook
    const long cap = cap_env > 0 ? cap_env : (wide ? 512 : 1024);
    if (blocks > cap && n4 >= 4 * cap * 256) blocks = cap;
    blocks = (blocks + m - 1) / m * m;
    return (int)blocks;
}

inline int quads_per_block(int C) { return C % 4 ? 0 : (C >= 256 ? 64 : (C >= 128 ? 32 : (C >= 64 ? 16 : (C >= 32 ? 8 : 0)))); }

// arrival counters of the in-launch finalize: a per-device ring of zeroed words; a launch takes the next slot and its last
// arrivers put every counter they drew the last ticket of back to zero.  2048 slots: a slot is reused 2048 BN launches later.
constexpr int TICKET_SLOT = 512, TICKET_SLOTS = 2048, TICKET_DEVS = 16;
unsigned* g_tickets[TICKET_DEVS];
unsigned g_ticket_pos[TICKET_DEVS];
unsigned* ticket_slot(int need) {
    // OFF by default (FTE_BN_TAIL=1 turns it on).  Measured on MI355X, ShuffleNet-v2 x2 step, in-launch finalize vs the separate
    // 7-8 us finalize launches (112 of 688 launches per step): batch 512 14.30 vs 14.16 ms, batch 256 8.79 vs 8.61 ms, ResNet-50
    // 39.8 vs 39.7 ms -- the tail's own chain (drain stores, barrier, agent-scope atomic, barrier, two levels of write-through
    // loads that miss every cache by construction) costs what the kernel boundary (~2 us) plus the small kernel cost, and the
    // step is not launch-bound at these sizes.  With a release fence per block instead of write-through stores: 15.68 ms.
    static const bool off = !(getenv("FTE_BN_TAIL") && atoi(getenv("FTE_BN_TAIL")) == 1);
    int dev = 0;
    if (off || need > TICKET_SLOT || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= TICKET_DEVS) return nullptr;
    if (!g_tickets[dev]) {
        unsigned* p = nullptr;
        const size_t bytes = (size_t)TICKET_SLOT * TICKET_SLOTS * sizeof(unsigned);
        if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;                 // first BN launch on this device (once, 4 MB)
        if (hipMemset(p, 0, bytes) != hipSuccess) { (void)hipFree(p); return nullptr; }
        g_tickets[dev] = p;
    }
    const unsigned k = g_ticket_pos[dev]++ % TICKET_SLOTS;
    return g_tickets[dev] + (size_t)k * TICKET_SLOT;
}

inline void stat_split(long rows, int C, int* splits, long* rps) {
    const int Q = quads_per_block(C);
    const long cb = Q ? (C / 4 + Q - 1) / Q : (C + 63) / 64;
    const long lanes = Q ? 256 / Q : 4;
    static const long target = getenv("FTE_BN_SPLIT_BLOCKS") ? atol(getenv("FTE_BN_SPLIT_BLOCKS")) : 2048;      // tuning hook: blocks of a reduce pass
    long rs = target / cb;
    if (rs > rows / (lanes * 8)) rs = rows / (lanes * 8);
    if (rs > BN_MAX_SPLITS) rs = BN_MAX_SPLITS;
    if (rs < 1) rs = 1;
    *rps = (rows + rs - 1) / rs;
    *splits = (int)((rows + *rps - 1) / *rps);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// storage flags of the *_s16 entry points (fte.h): bit 0 = the pre-activation side (z, dz) is bf16, bit 1 = the activation side
// (y, shortcut, dy, masked gradient) is bf16.  The bf16 forms need the float4 layouts (C % 4 == 0, C >= 32).
#define FTE_ZA(ZH_, AH_, ...) do { if (ZH_ && AH_) { constexpr bool ZH = true, AH = true; __VA_ARGS__; } \
                                    else if (ZH_) { constexpr bool ZH = true, AH = false; __VA_ARGS__; } \
                                    else if (AH_) { constexpr bool ZH = false, AH = true; __VA_ARGS__; } \
                                    else { constexpr bool ZH = false, AH = false; __VA_ARGS__; } } while (0)
hipError_t l_bn_train_stats(const float* z, const float* gamma, const float* beta, long rows, int C, float eps, float decay,
                            float* mean, float* rstd, float* scale, float* shift, float* mov_mean, float* mov_var,
                            float* part, hipStream_t st, int flags) {
    int splits; long rps;
    stat_split(rows, C, &splits, &rps);
    const int Q = quads_per_block(C);
    const bool zh = flags & 1;
    if (zh && !Q) return hipErrorInvalidValue;
    SplitTail tail = {nullptr, part + (long)splits * 3 * C, splits, (splits + TAIL_GROUP - 1) / TAIL_GROUP};
    const BnFwdOut fin = {gamma, beta, eps, decay, mean, rstd, scale, shift, mov_mean, mov_var};
    if (Q) tail.cnt = ticket_slot(((C / 4 + Q - 1) / Q) * (tail.groups + 1));
#define FTE_ST(Q_) do { if (zh) hipLaunchKernelGGL((bn_stats_v4_kernel<Q_, true>), dim3((C / 4 + Q_ - 1) / Q_, splits), dim3(256), 0, st, z, part, rows, C, rps, tail, fin); \
                         else hipLaunchKernelGGL((bn_stats_v4_kernel<Q_, false>), dim3((C / 4 + Q_ - 1) / Q_, splits), dim3(256), 0, st, z, part, rows, C, rps, tail, fin); } while (0)
    switch (Q) {
        case 64: FTE_ST(64); break;
        case 32: FTE_ST(32); break;
        case 16: FTE_ST(16); break;
        case 8: FTE_ST(8); break;
        default: hipLaunchKernelGGL(bn_stats_kernel, dim3((C + 63) / 64, splits), dim3(256), 0, st, z, part, rows, C, rps);
    }
#undef FTE_ST
    if (!tail.cnt)      // scalar-channel layouts (C % 4 != 0, C < 32) and FTE_BN_TAIL unset: the splits are merged by a second launch
        return l_bn_finalize(part, splits, gamma, beta, C, eps, decay, mean, rstd, scale, shift, mov_mean, mov_var, st);
    return hipGetLastError();
}
hipError_t l_bn_infer_coef(const float* gamma, const float* beta, const float* mm, const float* mv, float eps, int C,
                           float* scale, float* shift, hipStream_t st) {
    hipLaunchKernelGGL(bn_infer_coef_kernel, dim3((C + 255) / 256), dim3(256), 0, st, gamma, beta, mm, mv, eps, C, scale, shift);
    return hipGetLastError();
}
hipError_t l_bn_apply(const float* z, const float* scale, const float* shift, const float* res, float* y, long rows, int C,
                      int relu, hipStream_t st, int flags) {
    const long n4 = rows * C / 4;
    FTE_ZA(flags & 1, flags & 2, hipLaunchKernelGGL((bn_apply_kernel<ZH, AH>), dim3(grid_for_c(n4, C, !flags)), dim3(256), 0, st, z, scale, shift, res, y, n4, C, relu));
    return hipGetLastError();
}
hipError_t l_relu_bwd(const float* dy, const float* y, float* g, long n, hipStream_t st, int flags) {
    const long n4 = n / 4;
    long blocks = grid_for(n4);
    const long cap = flags ? 1024 : 512;                     // as grid_for_c: ~4 (bf16 storage) / ~2 (fp32) blocks per CU, four pieces per thread
    if (blocks > cap && n4 >= 4 * cap * 256) blocks = cap;
    if (flags & 2) hipLaunchKernelGGL(relu_bwd_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, dy, y, g, n4);
    else hipLaunchKernelGGL(relu_bwd_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, dy, y, g, n4);
    return hipGetLastError();
}
hipError_t l_bn_bwd(const float* dy, const float* ymask, const float* z, const float* gamma, const float* mean,
                    const float* rstd, const float* zsc, const float* zsf, float* gout, float* dz, float* dgamma, float* dbeta,
                    long rows, int C, float* part, hipStream_t st, int flags) {
    int splits; long rps;
    stat_split(rows, C, &splits, &rps);
    float* coef = part + (long)splits * 2 * C;
    const int Q = quads_per_block(C);
    if (flags && !Q) return hipErrorInvalidValue;
    SplitTail tail = {nullptr, coef + 3 * (long)C, splits, (splits + TAIL_GROUP - 1) / TAIL_GROUP};
    const BnBwdOut fin = {(float)rows, gamma, mean, rstd, dgamma, dbeta, coef};
    if (Q) tail.cnt = ticket_slot(((C / 4 + Q - 1) / Q) * (tail.groups + 1));
#define FTE_BR(Q_) FTE_ZA(flags & 1, flags & 2, hipLaunchKernelGGL((bn_bwd_reduce_v4_kernel<Q_, ZH, AH>), dim3((C / 4 + Q_ - 1) / Q_, splits), dim3(256), 0, st, \
                                                                   dy, ymask, z, mean, rstd, zsc, zsf, gout, part, rows, C, rps, tail, fin))
    switch (Q) {
        case 64: FTE_BR(64); break;
        case 32: FTE_BR(32); break;
        case 16: FTE_BR(16); break;
        case 8: FTE_BR(8); break;
        default: hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((C + 63) / 64, splits), dim3(256), 0, st, dy, ymask, z, mean, rstd, zsc, zsf, gout, part, rows, C, rps);
    }
#undef FTE_BR
    if (!tail.cnt) {
        hipError_t fe = l_bn_bwd_finalize(part, part + C, 2L * C, splits, rows, C, gamma, mean, rstd, dgamma, dbeta, coef, st);
        if (fe != hipSuccess) return fe;
    }
    const long n4 = rows * C / 4;
    if (gout) FTE_ZA(flags & 1, flags & 2, hipLaunchKernelGGL((bn_bwd_apply_kernel<ZH, AH>), dim3(grid_for_c(n4, C, !flags)), dim3(256), 0, st, gout, nullptr, z, coef, nullptr, nullptr, dz, n4, C));
    else FTE_ZA(flags & 1, flags & 2, hipLaunchKernelGGL((bn_bwd_apply_kernel<ZH, AH>), dim3(grid_for_c(n4, C, !flags)), dim3(256), 0, st, dy, ymask, z, coef, zsc, zsf, dz, n4, C));
    return hipGetLastError();
}
// partial rows from which the finalize kernels take the wide form (one channel quad x 256 split lanes per block) instead of 16 channels x
// 16 split lanes: FTE_BN_FIN_WIDE tunes it
static int fin_wide_from() {
    static const int v = getenv("FTE_BN_FIN_WIDE") ? atoi(getenv("FTE_BN_FIN_WIDE")) : 512;
    return v;
}
// ---- the pieces of the "BN fusion" path (conv epilogues leave the partials; fte_conv2d_bn_fwd / fte_conv2d_dgrad_bn) ------------------
// statistics partials [splits][3][C] (n, mean, M2) -> mean, rstd, scale, shift, moving statistics
hipError_t l_bn_finalize(const float* part, int splits, const float* gamma, const float* beta, int C, float eps, float decay,
                         float* mean, float* rstd, float* scale, float* shift, float* mov_mean, float* mov_var, hipStream_t st) {
    if (splits > fin_wide_from() && C % 4 == 0)
        hipLaunchKernelGGL(bn_finalize_wide_kernel, dim3(C / 4), dim3(256), 0, st, part, splits, C, gamma, beta, eps, decay, mean, rstd, scale, shift, mov_mean, mov_var);
    else
        hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, st, part, splits, C, gamma, beta, eps, decay, mean, rstd, scale, shift, mov_mean, mov_var);
    return hipGetLastError();
}
// backward partials pg / pgx [splits][ld] (sum g, sum g * xhat) -> dgamma, dbeta, coef[3][C] of dz = A g + B z + C0
hipError_t l_bn_bwd_finalize(const float* pg, const float* pgx, long ld, int splits, long rows, int C, const float* gamma, const float* mean,
                             const float* rstd, float* dgamma, float* dbeta, float* coef, hipStream_t st) {
    if (splits > fin_wide_from() && C % 4 == 0 && ld % 4 == 0)
        hipLaunchKernelGGL(bn_bwd_finalize_wide_kernel, dim3(C / 4), dim3(256), 0, st, pg, pgx, ld, splits, C, (float)rows, gamma, mean, rstd, dgamma, dbeta, coef);
    else
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, st, pg, pgx, ld, splits, C, (float)rows, gamma, mean, rstd, dgamma, dbeta, coef);
    return hipGetLastError();
}
// dz = A g + B z + C0 with g ALREADY masked (the producing data gradient's epilogue did that)
hipError_t l_bn_bwd_apply(const float* g, const float* z, const float* coef, float* dz, long rows, int C, hipStream_t st, int flags) {
    if (C % 4 || (flags && !quads_per_block(C))) return hipErrorInvalidValue;
    const long n4 = rows * C / 4;
    FTE_ZA(flags & 1, flags & 2, hipLaunchKernelGGL((bn_bwd_apply_kernel<ZH, AH>), dim3(grid_for_c(n4, C, !flags)), dim3(256), 0, st, g, nullptr, z, coef, nullptr, nullptr, dz, n4, C));
    return hipGetLastError();
}

hipError_t l_maxpool_fwd(const float* x, float* y, uint8_t* idx, int n, int h, int w, int c, int ho, int wo, int pt, int pl, hipStream_t st, int flags) {
    const dim3 grid(grid_for((long)n * ho * wo * (c / 4)));
    if (flags & 2) hipLaunchKernelGGL(maxpool_fwd_kernel<true>, grid, dim3(256), 0, st, x, y, idx, n, h, w, c, ho, wo, pt, pl);
    else hipLaunchKernelGGL(maxpool_fwd_kernel<false>, grid, dim3(256), 0, st, x, y, idx, n, h, w, c, ho, wo, pt, pl);
    return hipGetLastError();
}
hipError_t l_maxpool_bwd(const float* dy, const uint8_t* idx, float* dx, int n, int h, int w, int c, int ho, int wo, int pt, int pl, hipStream_t st, int flags) {
    if (pt == 0 && pl == 0 && h % 2 == 0 && w % 2 == 0) {
        const dim3 grid(grid_for((long)n * ho * wo * (c / 4)));
        if (flags & 2) hipLaunchKernelGGL(maxpool_bwd_even_kernel<true>, grid, dim3(256), 0, st, dy, idx, dx, n, h, w, c);
        else hipLaunchKernelGGL(maxpool_bwd_even_kernel<false>, grid, dim3(256), 0, st, dy, idx, dx, n, h, w, c);
        return hipGetLastError();
    }
    const dim3 grid(grid_for((long)n * h * w * (c / 4)));
    if (flags & 2) hipLaunchKernelGGL(maxpool_bwd_kernel<true>, grid, dim3(256), 0, st, dy, idx, dx, n, h, w, c, ho, wo, pt, pl);
    else hipLaunchKernelGGL(maxpool_bwd_kernel<false>, grid, dim3(256), 0, st, dy, idx, dx, n, h, w, c, ho, wo, pt, pl);
    return hipGetLastError();
}
hipError_t l_gap_fwd(const float* x, float* y, int n, int hw, int c, hipStream_t st, int flags) {
    if (flags & 2) hipLaunchKernelGGL(gap_fwd_kernel<true>, dim3((c / 4 + 15) / 16, n), dim3(256), 0, st, x, y, hw, c);
    else hipLaunchKernelGGL(gap_fwd_kernel<false>, dim3((c / 4 + 15) / 16, n), dim3(256), 0, st, x, y, hw, c);
    return hipGetLastError();
}
hipError_t l_gap_bwd(const float* dy, float* dx, int n, int hw, int c, hipStream_t st, int flags) {
    const long total = (long)n * hw * c;
    if (flags & 2) hipLaunchKernelGGL(gap_bwd_kernel<true>, dim3(grid_for(total)), dim3(256), 0, st, dy, dx, total, hw, c);
    else hipLaunchKernelGGL(gap_bwd_kernel<false>, dim3(grid_for(total)), dim3(256), 0, st, dy, dx, total, hw, c);
    return hipGetLastError();
}
hipError_t l_dropout_fwd(const float* x, float* mask, float* y, long n, float keep, uint64_t seed, hipStream_t st) {
    hipLaunchKernelGGL(dropout_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, mask, y, n, keep, seed);
    return hipGetLastError();
}
hipError_t l_scale_mask(const float* dy, const float* mask, float* dx, long n, float inv_keep, hipStream_t st) {
    hipLaunchKernelGGL(scale_mask_kernel, dim3(grid_for(n)), dim3(256), 0, st, dy, mask, dx, n, inv_keep);
    return hipGetLastError();
}
hipError_t l_im2col_first(const float* x, float* cols, int n, int h, int w, int cin, int ks, int stride, int ho, int wo,
                          int pt, int pl, int kpad, hipStream_t st, int h16) {
    static const bool rows_form = !(getenv("FTE_IM2COL_ROWS") && atoi(getenv("FTE_IM2COL_ROWS")) == 0);      // A/B hook
    if (rows_form && ks == 7 && cin == 3 && kpad == 160 && w <= 1024 && (long)n * h * w * cin < (1L << 31) && (long)n * ho < (1L << 31)) {
        if (h16) hipLaunchKernelGGL((im2col_first_rows_kernel<true, 7, 3, 40, 14>), dim3((unsigned)(n * ho)), dim3(280), (size_t)7 * w * 3 * sizeof(float), st, x, cols, h, w, stride, ho, wo, pt, pl);
        else hipLaunchKernelGGL((im2col_first_rows_kernel<false, 7, 3, 40, 14>), dim3((unsigned)(n * ho)), dim3(280), (size_t)7 * w * 3 * sizeof(float), st, x, cols, h, w, stride, ho, wo, pt, pl);
        return hipGetLastError();
    }
    const dim3 grid(grid_for((long)n * ho * wo * (kpad / 4)));
#define FTE_I2C(...) do { if (h16) hipLaunchKernelGGL((im2col_first_kernel<true, __VA_ARGS__>), grid, dim3(256), 0, st, x, cols, n, h, w, cin, ks, stride, ho, wo, pt, pl, kpad); \
                          else hipLaunchKernelGGL((im2col_first_kernel<false, __VA_ARGS__>), grid, dim3(256), 0, st, x, cols, n, h, w, cin, ks, stride, ho, wo, pt, pl, kpad); } while (0)
    if (ks == 7 && cin == 3) FTE_I2C(7, 3);
    else if (ks == 7 && cin == 1) FTE_I2C(7, 1);
    else FTE_I2C(0, 0);
#undef FTE_I2C
    return hipGetLastError();
}

// ===================================================================================================
// Grouped 3x3 convolution (nets/resnext.py:41-51: tf.split into 32 groups, 32 convs, tf.concat -- here one
// kernel, no split / concat traffic).  Group width gw = C/groups is 4..32: N = gw is far too narrow for
// a 32x32 MFMA tile (12 % utilisation at gw = 4), and the layer carries ~5 % of ResNeXt's MACs, so this is
// a direct VALU kernel: a block owns one group, its 9*gw*gw weights sit in LDS, a thread produces 4 output
// channels for PX consecutive pixels (weight reads are reused PX times).
//   x [N,H,W,C], w [G][3][3][gw_in][gw_out], y [N,Ho,Wo,C]; TF-SAME; stride 1 or 2.
// ===================================================================================================
namespace {

// GPBK groups share a block so that a wave reads GPBK*GW CONTIGUOUS channels of a pixel (one group per block made every
// lane fetch 16-128 B out of a different 512-B+ pixel row, and every row was fetched by all 32 group-blocks).
// S = stride (1 or 2) is a template parameter: the dgrad index map divides by it for every tap of every pixel, and a run-time
// divisor made that ~35 instructions each (the data gradient ran 2.5x slower than the forward pass of the same layer)
template <int GW, bool DGRAD, int GPBK, int S>
__global__ __launch_bounds__(256) void gconv3x3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       float* __restrict__ y, int n, int h, int wd, int c,
                                                       int ho, int wo, int pt, int pl) {
    constexpr int stride = S;
    // DGRAD = false: y[n,ho,wo] = sum_taps x[n, oh*s + r - pt, ow*s + q - pl] * W[r][q][ic][oc]
    // DGRAD = true : x is dz [n,ho,wo,C], y is dx [n,h,wd,C]: dx[ih,iw][ic] = sum dz[(ih+pt-r)/s,(iw+pl-q)/s][oc] * W[r][q][ic][oc]
    constexpr int PX = 4, OC4 = GW / 4;
    __shared__ __attribute__((aligned(16))) float wsh[GPBK * 9 * GW * GW];
    const float* wg = w + (long)blockIdx.y * GPBK * 9 * GW * GW;
    for (int i = threadIdx.x; i < GPBK * 9 * GW * GW; i += 256) {
        if (!DGRAD) wsh[i] = wg[i];
        else {                     // store transposed: ws[g][tap][oc][ic] so that the inner loop reads a float4 over ic
            const int oc = i % GW, ic = (i / GW) % GW, tap = (i / (GW * GW)) % 9, gg = i / (9 * GW * GW);
            wsh[((gg * 9 + tap) * GW + oc) * GW + ic] = wg[i];
        }
    }
    __syncthreads();
    const int oq = threadIdx.x % OC4;                       // output channel quad inside the group
    const int gl = (threadIdx.x / OC4) % GPBK;              // group inside the block's slab
    const int g = blockIdx.y * GPBK + gl;
    const float* ws = wsh + gl * 9 * GW * GW;
    constexpr int UPB = 256 / (OC4 * GPBK);                 // pixel units per block iteration
    const int oh_ = DGRAD ? h : ho, ow_ = DGRAD ? wd : wo;   // grid the threads walk
    const int owp = (ow_ + PX - 1) / PX;
    const long nunits = (long)n * oh_ * owp;
    for (long u = (long)blockIdx.x * UPB + threadIdx.x / (OC4 * GPBK); u < nunits; u += (long)gridDim.x * UPB) {
        const int wq = (int)(u % owp);
        long t = u / owp;
        const int oy = (int)(t % oh_);
        const int img = (int)(t / oh_);
        f32x4 acc[PX];
#pragma unroll
        for (int p = 0; p < PX; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                int sy;
                bool rowok;
                if (!DGRAD) { sy = oy * stride + r - pt; rowok = sy >= 0 && sy < h; }
                else { const int num = oy + pt - r; sy = num / stride; rowok = num >= 0 && num % stride == 0 && sy < ho; }
                if (!rowok) continue;
                const float* wt = ws + (r * 3 + q) * GW * GW;
                for (int ic = 0; ic < GW; ic += 4) {
                    f32x4 xv[PX];
#pragma unroll
                    for (int p = 0; p < PX; ++p) {
                        const int ox = wq * PX + p;
                        int sx;
                        bool ok;
                        if (!DGRAD) { sx = ox * stride + q - pl; ok = sx >= 0 && sx < wd && ox < ow_; }
                        else { const int num = ox + pl - q; sx = num / stride; ok = num >= 0 && num % stride == 0 && sx < wo && ox < ow_; }
                        xv[p] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (ok) {
                            const long src = DGRAD ? ((long)(img * ho + sy) * wo + sx) : ((long)(img * h + sy) * wd + sx);
                            xv[p] = *reinterpret_cast<const f32x4*>(x + src * c + g * GW + ic);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // forward: W[ic+e][oc quad]; dgrad: the LDS image is transposed ([tap][oc][ic]), so the same address
                        // is W^T[oc = ic+e][ic quad] -- one 16-byte read either way
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(wt + (ic + e) * GW + oq * 4);
#pragma unroll
                        for (int p = 0; p < PX; ++p) acc[p] += xv[p][e] * wv;
                    }
                }
            }
        }
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            const int ox = wq * PX + p;
            if (ox < ow_) *reinterpret_cast<f32x4*>(y + ((long)(img * oh_ + oy) * ow_ + ox) * c + g * GW + oq * 4) = acc[p];
        }
    }
}

// ---- bf16-mode grouped 3x3 (stride 1) on the matrix cores -------------------------------------------------------------------
// ResNeXt's grouped conv has 4..32 channels per group: as fp32 vector code it runs at the VALU roofline (60 us for 3.7 GFLOP)
// and, in the bf16 MFMA mode of the step, became a quarter of it.  Here a 32-channel SLICE of the tensor (= 32 / gw whole
// groups, input and output channels alike) is a dense 3x3 convolution 32 -> 32 whose filter is block-diagonal: per wave a
// [32 pixels] x [32 channels] output tile = 9 taps x 2 k-steps of v_mfma_f32_32x32x16_bf16.  The zero blocks multiply for
// nothing (8x the FLOPs at gw = 4), but at 16x the fp32 MFMA rate the kernel is HBM-bound anyway.  No LDS, no barrier: a lane
// loads the 8 channels of ITS pixel and k-half for a tap straight from global memory (fp32, rounded to bf16 in registers --
// the operand rounding of the bf16 mode, RNE) and the slice's 18 filter fragments live in registers for the wave's whole walk.
typedef __bf16 bf16x8_l __attribute__((ext_vector_type(8)));
typedef float f32x16_l __attribute__((ext_vector_type(16)));
typedef float f32x2_l __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_l __attribute__((ext_vector_type(2)));
typedef short s16x4_l __attribute__((ext_vector_type(4)));
typedef short s16x8_l __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_l __attribute__((ext_vector_type(4)));

// wpk[slice][tap][col 32][k 32] (k contiguous), bf16.  fwd: col = output channel, k = input channel, tap as given;
// dgrad (dx = conv of dz with the mirrored, transposed filter): col = input channel, k = output channel, tap mirrored.
// w is [group][tap][ic][oc] fp32 (the grouped layer's own layout).
__global__ __launch_bounds__(256) void gconv_pack16_kernel(const float* __restrict__ w, unsigned short* __restrict__ wf,
                                                           unsigned short* __restrict__ wd, int c, int gw) {
    const int total = (c / 32) * 9 * 32 * 32;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int k = i & 31, col = (i >> 5) & 31, tap = (i >> 10) % 9, slice = (i >> 10) / 9;
        const bool same = (k / gw) == (col / gw);
        const int g = slice * (32 / gw) + col / gw;
        float vf = 0.f, vd = 0.f;
        if (same) {
            vf = w[(((long)g * 9 + tap) * gw + (k % gw)) * gw + (col % gw)];            // W[tap][ic = k][oc = col]
            vd = w[(((long)g * 9 + (8 - tap)) * gw + (col % gw)) * gw + (k % gw)];      // W[mirror][ic = col][oc = k]
        }
        wf[i] = __builtin_bit_cast(unsigned short, (__bf16)vf);
        wd[i] = __builtin_bit_cast(unsigned short, (__bf16)vd);
    }
}

// "BN fusion" of the grouped 3x3 (bf16 storage only).  BNF = 1 (forward): every wave keeps shifted sums of the values it STORES
// (lane = channel, shift = the first value the lane sees) over its whole walk and leaves one (n, mean, M2) partial row per wave --
// bn_finalize's layout, part[((block * 4 + wave) * 3 + k) * c + channel].  BNF = 2 (data gradient landing on a BN + ReLU output):
// g = stored(dx) where fma(zbn, sc, sh) > 0 (sc == NULL: everywhere), else 0; g is what is stored, and the wave leaves
// sum g -> pg[row][channel], sum g * (zbn - mu) * rs -> pgx[row][channel] for bn_bwd_finalize.
struct GconvBn {
    float* part;                       // BNF 1: statistics partials; BNF 2: pg
    float* pgx;                        // BNF 2
    const unsigned short* zbn;         // BNF 2: the BN layer's input z (bf16), same shape as the gradient written
    const float* mu; const float* rs; const float* sc; const float* sh;
    // PRO (window kernel, forward): x is the PRE-normalisation tensor of the batch norm in front of the layer; the loader applies
    // y = relu(isc[c] * x + ish[c]) -- bn_apply's expression, rounded to bf16 as its store would -- and writes the tile's own pixels
    // back to yside (the filter gradient's operand; nothing else reads y): the bn_apply launch in front of the layer disappears
    const float* isc; const float* ish; unsigned short* yside;
};
struct GconvAcc { float a, b, c, shift; };      // BNF 1: n, sum d, sum d^2, shift;  BNF 2: sum g, sum g xhat
// BNF 2: this lane's 16 z values of a tile (pixel rows (i & 3) + 8 (i >> 2) + 4 lh, its channel), requested at the TOP of the tile's
// iteration so that they land under its matrix work (fetched where they are used, every tile stalled on 16 two-byte loads: 65 vs
// 28 us for the 28x28x128 data gradient)
template <int BNF>
__device__ __forceinline__ void gconv_bn_prefetch(const GconvBn& bn, unsigned short (&zr)[16], long tile, long npix, int c, int ch, int lh) {
    if constexpr (BNF == 2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const long pr = tile * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
            zr[i] = pr < npix ? bn.zbn[pr * c + ch] : (unsigned short)0;
        }
    }
}
template <int BNF>
__device__ __forceinline__ void gconv_bn_tile(const GconvBn& bn, GconvAcc& st, f32x16_l& acc, long tile, long npix, int c, int ch, int lh,
                                              float mu, float rs, float sc, float sh, const unsigned short (&zr)[16]) {
    if constexpr (BNF == 1) {
        if (tile * 32 + 32 <= npix) {                   // (wave-uniform) a whole tile: no row checks, the shift is set once
            if (st.a == 0.f) st.shift = __builtin_bit_cast(float, (unsigned)__builtin_bit_cast(unsigned short, (__bf16)acc[0]) << 16);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float v = __builtin_bit_cast(float, (unsigned)__builtin_bit_cast(unsigned short, (__bf16)acc[i]) << 16);
                const float d = v - st.shift;
                st.b += d; st.c += d * d;
            }
            st.a += 16.f;
            return;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const long pr = tile * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
            if (pr < npix) {
                const float v = __builtin_bit_cast(float, (unsigned)__builtin_bit_cast(unsigned short, (__bf16)acc[i]) << 16);
                if (st.a == 0.f) st.shift = v;
                const float d = v - st.shift;
                st.a += 1.f; st.b += d; st.c += d * d;
            }
        }
    } else if constexpr (BNF == 2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const long pr = tile * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
            const float z = __builtin_bit_cast(float, (unsigned)zr[i] << 16);
            float g = __builtin_bit_cast(float, (unsigned)__builtin_bit_cast(unsigned short, (__bf16)acc[i]) << 16);
            const bool on = pr < npix && (bn.sc == nullptr || __builtin_fmaf(z, sc, sh) > 0.f);
            g = on ? g : 0.f;
            acc[i] = g;
            st.a += g;
            st.b += g * ((z - mu) * rs);
        }
    }
}
template <int BNF>
__device__ __forceinline__ void gconv_bn_flush(const GconvBn& bn, const GconvAcc& st, int row, int c, int ch, int lh) {
    if constexpr (BNF == 1) {
        float n = st.a, mean = 0.f, m2 = 0.f;
        if (n > 0.f) { mean = st.shift + st.b / n; m2 = fmaxf(st.c - st.b * st.b / n, 0.f); }
        const float nb = __shfl_xor(n, 32), mb = __shfl_xor(mean, 32), qb = __shfl_xor(m2, 32);
        if (lh == 0) {
            chan_merge(n, mean, m2, nb, mb, qb);
            float* pp = bn.part + (long)row * 3 * c + ch;
            pp[0] = n; pp[c] = mean; pp[2 * (long)c] = m2;
        }
    } else if constexpr (BNF == 2) {
        const float a = st.a + __shfl_xor(st.a, 32), b = st.b + __shfl_xor(st.b, 32);
        if (lh == 0) { bn.part[(long)row * c + ch] = a; bn.pgx[(long)row * c + ch] = b; }
    }
}

// MODE 0: stride 1 (forward, or data gradient with the mirrored filter: the same index map); 1: forward at stride 2 (x is
// [n, hs, ws], y [n, h, wd] = the walked grid); 2: data gradient at stride 2 (x is dz [n, hs, ws], y is dx [n, h, wd]; tap (r, q) of
// the mirrored filter reads dz[(iy + pt - (2 - r)) / 2] when that is whole).  pt, pl = the TF-SAME pads before.
template <int MODE, bool H = false, int BNF = 0>      // H: x and y are bf16 in HBM (bf16 storage)
__global__ __launch_bounds__(256) void gconv3x3_mfma16_kernel(const float* __restrict__ x, const unsigned short* __restrict__ wpk,
                                                              float* __restrict__ y, int n, int h, int wd, int c,
                                                              int hs, int ws, int pt, int pl, GconvBn bn) {
    // the slice's 9 x [32][32] bf16 filter (18 KB) sits in LDS; a fragment is one ds_read_b128 right before its MFMA
    __shared__ __attribute__((aligned(16))) unsigned short wsh[9 * 32 * 32];
    // per wave: the 32 pixels x 128 B of one tap, twice (the next tap lands while this one multiplies).  A lane FETCHES 16-byte
    // piece (lane & 7) of pixels (lane >> 3) + 8 j -- one instruction = 8 whole 128-byte rows -- and READS the 32 bytes of ITS
    // pixel and k-half back as an MFMA operand.  (Fetching the operand layout directly -- 16 bytes of 64 different rows per
    // instruction -- made every row go through the L1 tags four times: 67 us for the 28x28x128 layer, address-bound.)
    // piece p of pixel r is stored at slot p ^ (r & 7): fetch-side stores and operand-side loads are both conflict-free.
    __shared__ __attribute__((aligned(16))) float stage[4][2][32 * 32];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int slice = blockIdx.y;
    {
        const uint4* src = reinterpret_cast<const uint4*>(wpk + (long)slice * 9 * 1024);
        uint4* dst = reinterpret_cast<uint4*>(wsh);
        for (int i = threadIdx.x; i < 9 * 1024 / 8; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const long npix = (long)n * h * wd;
    const long ntiles = (npix + 31) / 32;
    const long xs = slice * 32 + ((lane & 7) << 2);
    const int fr = lane >> 3;                         // fetch row inside a group of 8 pixels
    float* st = &stage[wv][0][0];
    GconvAcc bst = {0.f, 0.f, 0.f, 0.f};
    float bmu = 0.f, brs = 0.f, bsc = 0.f, bsh = 0.f;
    if constexpr (BNF == 2) {
        bmu = bn.mu[slice * 32 + li]; brs = bn.rs[slice * 32 + li];
        if (bn.sc) { bsc = bn.sc[slice * 32 + li]; bsh = bn.sh[slice * 32 + li]; }
    }
    for (long tile = (long)blockIdx.x * 4 + wv; tile < ntiles; tile += (long)gridDim.x * 4) {
        unsigned short zr[16];
        gconv_bn_prefetch<BNF>(bn, zr, tile, npix, c, slice * 32 + li, lh);
        // the four pixels this lane fetches for: coordinates once per tile
        int foy[4], fox[4];
        long fbase[4];
        bool fok[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long p = tile * 32 + fr + 8 * j;
            fok[j] = p < npix;
            const unsigned pu = (unsigned)(fok[j] ? p : 0);
            fox[j] = (int)(pu % (unsigned)wd);
            const unsigned t2 = pu / (unsigned)wd;
            foy[j] = (int)(t2 % (unsigned)h);
            fbase[j] = (long)(t2 / (unsigned)h) * hs;                     // image base in rows of the SOURCE
        }
        auto fetch = [&](int t, f32x4 (&v)[4]) {
            const int r = t / 3, q = t - 3 * r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int sy, sx;
                bool ok = fok[j];
                if (MODE == 0) { sy = foy[j] + r - 1; sx = fox[j] + q - 1; }
                else if (MODE == 1) { sy = 2 * foy[j] + r - pt; sx = 2 * fox[j] + q - pl; }
                else {
                    const int ny = foy[j] + pt - (2 - r), nx = fox[j] + pl - (2 - q);
                    ok = ok && !((ny | nx) & 1);
                    sy = ny >> 1; sx = nx >> 1;
                }
                ok = ok && sy >= 0 && sy < hs && sx >= 0 && sx < ws;
                const f32x4 val = ldq<H>(x, xs + ((fbase[j] + (ok ? sy : 0)) * ws + (ok ? sx : 0)) * c);
                v[j] = ok ? val : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        auto stash = [&](int buf, const f32x4 (&v)[4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = fr + 8 * j;
                *reinterpret_cast<f32x4*>(st + buf * 1024 + row * 32 + (((lane & 7) ^ (row & 7)) << 2)) = v[j];
            }
        };
        f32x16_l acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        f32x4 va[4], vb[4];
        fetch(0, va);
        fetch(1, vb);
        stash(0, va);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // registers: tap t + 1 is in (t odd ? va : vb) ... kept simple: two register sets alternate, LDS buffers alternate
            if (t + 2 < 9) { if (t & 1) fetch(t + 2, vb); else fetch(t + 2, va); }
            const float* sb = st + (t & 1) * 1024 + li * 32;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int p0 = 4 * ks + 2 * lh;                                  // 16-byte pieces p0, p0 + 1 of this lane's pixel
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(sb + ((p0 ^ (li & 7)) << 2));
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(sb + (((p0 + 1) ^ (li & 7)) << 2));
                bf16x8_l fa;
                const bf16x2_l c0 = __builtin_convertvector(f32x2_l{v0[0], v0[1]}, bf16x2_l), c1 = __builtin_convertvector(f32x2_l{v0[2], v0[3]}, bf16x2_l);
                const bf16x2_l c2 = __builtin_convertvector(f32x2_l{v1[0], v1[1]}, bf16x2_l), c3 = __builtin_convertvector(f32x2_l{v1[2], v1[3]}, bf16x2_l);
                fa[0] = c0[0]; fa[1] = c0[1]; fa[2] = c1[0]; fa[3] = c1[1]; fa[4] = c2[0]; fa[5] = c2[1]; fa[6] = c3[0]; fa[7] = c3[1];
                const bf16x8_l fb = *reinterpret_cast<const bf16x8_l*>(wsh + ((t * 32 + li) * 32 + 16 * ks + 8 * lh));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
            }
            if (t + 1 < 9) { if (t & 1) stash(0, va); else stash(1, vb); }      // tap t + 1 -> the other LDS buffer
        }
        // C layout: column (channel) = lane & 31, row (pixel) = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5): 32 lanes write one pixel's 128 B
        const long yo = slice * 32 + li;
        if constexpr (BNF != 0) gconv_bn_tile<BNF>(bn, bst, acc, tile, npix, c, (int)yo, lh, bmu, brs, bsc, bsh, zr);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const long pr = tile * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
            if (pr < npix) st1<H>(y, yo + pr * c, acc[i]);
        }
    }
    if constexpr (BNF != 0) gconv_bn_flush<BNF>(bn, bst, blockIdx.x * 4 + wv, c, slice * 32 + li, lh);
}

// Stride 1 (forward and data gradient), second formulation: the taps of one kernel ROW are shifted views of the same 34
// consecutive pixels.  In linear pixel space the source of (pixel p, tap (r, q)) is p + (r - 1) * W + (q - 1) whenever it lies
// inside the image, so per tile of 32 pixels and kernel row r a wave fetches the 34 rows [p0 - 1 + (r - 1) W, p0 + 32 + (r - 1) W]
// ONCE (coalesced 128-byte rows, rounded to bf16, 80-byte LDS rows), and the three taps q read window rows li + q; the image /
// row edges are a 9-bit mask per lane that zeroes the fragment.  gconv3x3_mfma16_kernel<0> fetched every tap on its own: 9 x the
// tensor through L2 -> L1 (461 MB for the 28x28x128 layer at 128 images, 52 us = the L2 rate); this one moves 3.2 x.
// a / d for 0 <= a < 2^24 with rd = 1 / d (one multiply and a correction instead of the integer division sequence)
__device__ __forceinline__ int gdiv(int a, int d, float rd) {
    int q = (int)((float)a * rd);
    const int r = a - q * d;
    q += (r >= d) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}
template <bool H = false, int BNF = 0, bool PRO = false>
__global__ __launch_bounds__(256) void gconv3x3_mfma16_win_kernel(const float* __restrict__ x, const unsigned short* __restrict__ wpk,
                                                                  float* __restrict__ y, int n, int h, int wd, int c, GconvBn bn) {
    __shared__ __attribute__((aligned(16))) unsigned short wsh[9 * 32 * 32];
    constexpr int WROW = 40, WBUF = 34 * WROW;                    // bf16 per window row (32 + 8 pad: conflict-free b128 reads), per buffer
    __shared__ __attribute__((aligned(16))) unsigned short win[4][3][WBUF];    // one buffer per kernel row: a tile's three rows are fetched together
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int slice = blockIdx.y;
    {
        const uint4* src = reinterpret_cast<const uint4*>(wpk + (long)slice * 9 * 1024);
        uint4* dst = reinterpret_cast<uint4*>(wsh);
        for (int i = threadIdx.x; i < 9 * 1024 / 8; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    // 32-bit pixel / byte offsets and buffer addressing (the launcher checks npix * c * element size < 2^31): an out-of-range window
    // entry is an out-of-range OFFSET, which the buffer load answers with zeros -- the kernel was bound by its own address arithmetic
    // (~1000 VALU instructions per 18 MFMAs with 64-bit offsets, range checks and the / and % of the edge masks)
    constexpr unsigned ES = H ? 2u : 4u, OOB = 0x80000000u;
    const int npix = n * h * wd;
    const int ntiles = (npix + 31) / 32;
    const int fr = lane >> 3, fp = lane & 7;
    const int xs = slice * 32 + (fp << 2);
    const unsigned cb = (unsigned)c * ES, xsb = (unsigned)xs * ES;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (unsigned)npix * cb, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y, 0, (unsigned)npix * cb, 0x00020000);
    unsigned short* wb = &win[wv][0][0];
    f32x4 isc4 = {0.f, 0.f, 0.f, 0.f}, ish4 = isc4;
    if constexpr (PRO) { isc4 = *reinterpret_cast<const f32x4*>(bn.isc + xs); ish4 = *reinterpret_cast<const f32x4*>(bn.ish + xs); }
    // a fetched window row stays in its memory format until it is stashed (bf16 storage: 2 registers per piece instead of 4)
    using raw_t = typename std::conditional<H, u32x2_l, f32x4>::type;
    auto fetch = [&](int tile, int r, raw_t (&v)[5]) {
        const int s0 = tile * 32 - 1 + (r - 1) * wd + fr;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int sp = s0 + 8 * i;
            const bool ok = fr + 8 * i < 34 && (unsigned)sp < (unsigned)npix;
            const unsigned voff = ok ? (unsigned)sp * cb + xsb : OOB;
            if constexpr (H) v[i] = __builtin_bit_cast(u32x2_l, __builtin_amdgcn_raw_buffer_load_b64(rx, voff, 0, 0));
            else v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff, 0, 0));
        }
    };
    auto stash = [&](int buf, int tile, int r, const raw_t (&v)[5]) {
        const int s0 = tile * 32 - 1 + (r - 1) * wd + fr;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int j = fr + 8 * i;
            if (j >= 34) continue;
            unsigned short* dst = wb + buf * WBUF + j * WROW + (fp << 2);
            if constexpr (H && !PRO) {
                *reinterpret_cast<u32x2_l*>(dst) = v[i];                  // (zeros where the entry lies outside the tensor)
            } else {
                f32x4 val;
                if constexpr (H) val = f32x4{__builtin_bit_cast(float, v[i][0] << 16), __builtin_bit_cast(float, v[i][0] & 0xffff0000u),
                                             __builtin_bit_cast(float, v[i][1] << 16), __builtin_bit_cast(float, v[i][1] & 0xffff0000u)};
                else val = v[i];
                if constexpr (PRO) {
                    const int sp = s0 + 8 * i;
                    const bool ok = (unsigned)sp < (unsigned)npix;
                    val = bn_affine(val, isc4, ish4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) val[e] = ok ? fmaxf(val[e], 0.f) : 0.f;
                    // the centre row's window entries 1 .. 32 are the tile's own pixels: each pixel of the tensor exactly once
                    if (r == 1 && ok && j >= 1 && j <= 32) stq<true>(reinterpret_cast<float*>(bn.yside), xs + (long)sp * c, val);
                }
                *reinterpret_cast<s16x4_l*>(dst) = __builtin_bit_cast(s16x4_l, __builtin_convertvector(val, bf16x4_l));
            }
        }
    };
    const bool fast_div = npix < (1 << 24);
    const float r_wd = 1.f / (float)wd, r_h = 1.f / (float)h;
    const int stride_t = (int)gridDim.x * 4;
    int tile = (int)blockIdx.x * 4 + wv;
    // all three window rows of a tile are in flight at once, and the NEXT tile's go out before this tile's MFMAs (one row at a time
    // left 2.5 KB in flight per wave: the launches ran at 0.6-1.2 TB/s on a chain of dependent round trips)
    raw_t va[5], vb[5], vc[5];
    if (tile < ntiles) { fetch(tile, 0, va); fetch(tile, 1, vb); fetch(tile, 2, vc); }
    GconvAcc bst = {0.f, 0.f, 0.f, 0.f};
    float bmu = 0.f, brs = 0.f, bsc = 0.f, bsh = 0.f;
    if constexpr (BNF == 2) {
        bmu = bn.mu[slice * 32 + li]; brs = bn.rs[slice * 32 + li];
        if (bn.sc) { bsc = bn.sc[slice * 32 + li]; bsh = bn.sh[slice * 32 + li]; }
    }
    for (; tile < ntiles; tile += stride_t) {
        unsigned short zr[16];
        gconv_bn_prefetch<BNF>(bn, zr, tile, npix, c, slice * 32 + li, lh);
        const int p = tile * 32 + li;
        const bool pok = p < npix;
        const int pu = pok ? p : 0;
        int ox, oy;
        if (fast_div) {                                            // (wave-uniform)
            const int q1 = gdiv(pu, wd, r_wd), q2 = gdiv(q1, h, r_h);
            ox = pu - q1 * wd; oy = q1 - q2 * h;
        } else {
            ox = (int)((unsigned)pu % (unsigned)wd);
            oy = (int)(((unsigned)pu / (unsigned)wd) % (unsigned)h);
        }
        unsigned vm = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int sy = oy + r - 1, sx = ox + q - 1;
                if (pok && sy >= 0 && sy < h && sx >= 0 && sx < wd) vm |= 1u << (3 * r + q);
            }
        f32x16_l acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        auto row = [&](int r, int buf) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int t = 3 * r + q;
                const bool ok = (vm >> t) & 1;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    s16x8_l raw = *reinterpret_cast<const s16x8_l*>(wb + buf * WBUF + (li + q) * WROW + 16 * ks + 8 * lh);
                    if (!ok) raw = s16x8_l{0, 0, 0, 0, 0, 0, 0, 0};
                    const bf16x8_l fa = __builtin_bit_cast(bf16x8_l, raw);
                    const bf16x8_l fb = *reinterpret_cast<const bf16x8_l*>(wsh + ((t * 32 + li) * 32 + 16 * ks + 8 * lh));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
                }
            }
        };
        stash(0, tile, 0, va);
        stash(1, tile, 1, vb);
        stash(2, tile, 2, vc);
        if (tile + stride_t < ntiles) { fetch(tile + stride_t, 0, va); fetch(tile + stride_t, 1, vb); fetch(tile + stride_t, 2, vc); }
        row(0, 0);
        row(1, 1);
        row(2, 2);
        const int yo = slice * 32 + li;
        if constexpr (BNF != 0) gconv_bn_tile<BNF>(bn, bst, acc, tile, npix, c, yo, lh, bmu, brs, bsc, bsh, zr);
        const unsigned ybase = (unsigned)(tile * 32 + 4 * lh) * cb + (unsigned)yo * ES;
        const int prow = tile * 32 + 4 * lh;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int d = (i & 3) + 8 * (i >> 2);
            const unsigned voff = prow + d < npix ? ybase + (unsigned)d * cb : OOB;      // (a store to an out-of-range offset is dropped)
            if constexpr (H) __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (__bf16)acc[i]), ry, voff, 0, 0);
            else { const float av = acc[i]; __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(av), ry, voff, 0, 0); }
        }
    }
    if constexpr (BNF != 0) gconv_bn_flush<BNF>(bn, bst, blockIdx.x * 4 + wv, c, slice * 32 + li, lh);
}

// Filter gradient of the same layers on the bf16 MFMA: per 32-channel slice and tap a dense [32 ic] x [32 oc] product over the
// pixels, D_tap[ic][oc] = sum_p x[p + tap][ic] * dz[p][oc] (the diagonal gw x gw blocks are the groups' gradients; the rest is
// discarded by the reduction).  Both operands need 8 consecutive PIXELS of one channel per lane while memory is channel-
// contiguous: rows are fetched whole (coalesced), rounded to bf16 and stored [pixel][channel] in LDS, and the fragments come
// back through ds_read_b64_tr_b16 -- gfx950's transposing LDS read (scripts/probes/ds_read_tr16.hip pins its lane map).
// Block = 3 waves, wave r owns kernel row r (3 taps = 3 accumulator blocks); 16 pixels per step; the next step's rows are in
// flight while this one multiplies.  Partials [chunk][slice][tap][32 oc][gw ic], summed in order by gconv_wgrad16_reduce_kernel.
template <int S, bool H = false>
__global__ __launch_bounds__(192) void gconv3x3_wgrad_mfma16_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                                    float* __restrict__ part, int n, int h, int wd, int c,
                                                                    int gw, long steps_per_chunk, int ho, int wo, int pt, int pl) {
    // S = stride: the walked pixels are dz's [n, ho, wo]; x is [n, h, wd]
    __shared__ __attribute__((aligned(16))) unsigned short img[3][4][16 * 32];      // per wave: 3 taps of x + dz, [16 pixels][32 channels]
    const int lane = threadIdx.x & 63, r = threadIdx.x >> 6;                        // r = kernel row of this wave
    const int li = lane & 31, lh = lane >> 5;
    const int slice = blockIdx.y;
    const long npix = (long)n * ho * wo;
    const long nsteps = (npix + 15) / 16;
    const long s0 = (long)blockIdx.x * steps_per_chunk, s1 = min(nsteps, s0 + steps_per_chunk);
    const int frow = lane >> 3, fpiece = lane & 7;                                  // fetch: rows frow, frow + 8; 16-byte piece
    const long xs = slice * 32 + (fpiece << 2);
    unsigned short* mine = &img[r][0][0];
    f32x16_l acc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
    // transposed-read addresses: group g = lane >> 4 reads the 4 x 16 block at (pixel 8 (g >> 1) [+4], channel 16 (g & 1));
    // lane 4 qq + pp of the group supplies row qq, columns 4 pp .. 4 pp + 3
    const int g4 = lane >> 4, idx = lane & 15;
    const int troff = (8 * (g4 >> 1) + (idx >> 2)) * 32 + 16 * (g4 & 1) + 4 * (idx & 3);
    f32x4 cur[4][2], nxt[4][2];                                                      // [x tap 0..2, dz][row half]
    auto fetch = [&](long step, f32x4 (&v)[4][2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long p = step * 16 + frow + 8 * j;
            const bool pok = p < npix;
            const unsigned pu = (unsigned)(pok ? p : 0);
            const int ox = (int)(pu % (unsigned)wo);
            const unsigned t2 = pu / (unsigned)wo;
            const int oy = (int)(t2 % (unsigned)ho);
            const long ib = (long)(t2 / (unsigned)ho) * h;
            const f32x4 dv = ldq<H>(dz, xs + (long)pu * c);
            v[3][j] = pok ? dv : f32x4{0.f, 0.f, 0.f, 0.f};
            const int sy = oy * S + r - pt;
            const bool rok = pok && sy >= 0 && sy < h;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int sx = ox * S + q - pl;
                const bool ok = rok && sx >= 0 && sx < wd;
                const f32x4 xv = ldq<H>(x, xs + ((ib + (ok ? sy : 0)) * wd + (ok ? sx : 0)) * c);
                v[q][j] = ok ? xv : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    if (s0 < s1) fetch(s0, cur);
    for (long step = s0; step < s1; ++step) {
        if (step + 1 < s1) fetch(step + 1, nxt);
        // round to bf16 and lay the four [16][32] images down ([pixel][channel], 64-byte rows)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 v = cur[t][j];
                const s16x4_l w = __builtin_bit_cast(s16x4_l, __builtin_convertvector(v, bf16x4_l));      // whole-vector casts: element-wise bit_casts of bf16 lanes were miscompiled
                *reinterpret_cast<s16x4_l*>(mine + t * 512 + (frow + 8 * j) * 32 + (fpiece << 2)) = w;
            }
        // one wave: its LDS accesses execute in order, no barrier -- but the compiler must not move the transposing reads (an
        // intrinsic on an LDS-address-space pointer) above the plain stores that fill the images
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        auto frag = [&](int t) -> bf16x8_l {
            const s16x4_l a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_l*)(mine + t * 512 + troff));
            const s16x4_l b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_l*)(mine + t * 512 + troff + 4 * 32));
            return __builtin_bit_cast(bf16x8_l, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
        };
        const bf16x8_l fz = frag(3);
#pragma unroll
        for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(q), fz, acc[q], 0, 0, 0);
        if (step + 1 < s1) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 2; ++j) cur[t][j] = nxt[t][j];
        }
    }
    // C layout: row (ic) = (i & 3) + 8 (i >> 2) + 4 lh, column (oc) = li.  Only the diagonal gw x gw blocks are groups: a lane keeps
    // the row quads of its column's group and writes them [oc][ic % gw] (32 * gw floats per tap instead of 1024)
    float* out = part + (((long)blockIdx.x * gridDim.y + slice) * 9 + r * 3) * 32 * gw + li * gw;
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row0 = 8 * j + 4 * lh;
            if (row0 / gw == li / gw)
                *reinterpret_cast<f32x4*>(out + q * 32 * gw + (row0 % gw)) = f32x4{acc[q][4 * j], acc[q][4 * j + 1], acc[q][4 * j + 2], acc[q][4 * j + 3]};
        }
}

// dW[g][tap][ic][oc] = sum over chunks of the slice partials [chunk][slice][tap][oc 32][ic gw]: 64 outputs x 4 chunk lanes per
// block, every lane sums its chunks (k = lane, lane + 4, ...) in order and the four are added in a fixed order
__global__ __launch_bounds__(256) void gconv_wgrad16_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                   int chunks, int slices, int gw, int groups) {
    const int total = groups * 9 * gw * gw;                      // = slices * 9 * 32 * gw
    const int kl = threadIdx.x & 3;
    const int i = blockIdx.x * 64 + (threadIdx.x >> 2);          // compact order: (slice, tap, oc 32, ic gw)
    const bool ok = i < total;
    const long stride = (long)slices * 9 * 32 * gw;
    float s = 0.f;
    if (ok)
        for (int k = kl; k < chunks; k += 4) s += part[(long)k * stride + i];
    const float s1 = __shfl_xor(s, 1);
    const float a = (kl & 1) ? s1 + s : s + s1;                  // (lane 0 + lane 1), (lane 2 + lane 3): the even lane's value first
    const float a2 = __shfl_xor(a, 2);
    const float t = (kl & 2) ? a2 + a : a + a2;
    if (ok && kl == 0) {
        const int ic = i % gw, col = (i / gw) & 31, tap = (i / (32 * gw)) % 9, slice = i / (9 * 32 * gw);
        const int ch = slice * 32 + col, g = ch / gw, oc = ch % gw;
        dw[(((long)g * 9 + tap) * gw + ic) * gw + oc] = t;
    }
}

// filter gradient: block = (pixel chunk, group set); thread = (group, kernel row r, ic quad, oc quad); partial [chunk][G*9*gw*gw].
// Per pixel a thread loads ONE float4 of dz (its oc quad) and THREE float4 of x (its ic quad at the three taps of row r) for
// 3 x 4 x 4 multiply-adds into 12 float4 accumulators: 12 FMAs per load instruction.  (A thread per (ic, oc quad) with nine scalar
// x loads per pixel -- 3.6 FMAs per load -- kept every layer at ~117 us whatever its size: bound by the count of 4-byte loads.)
template <int GW>
__global__ __launch_bounds__(256) void gconv3x3_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                             float* __restrict__ part, int n, int h, int wd, int c,
                                                             int ho, int wo, int stride, int pt, int pl, long pix_per_block, int groups) {
    constexpr int Q = GW / 4, TPG = 3 * Q * Q;               // threads per group
    constexpr int GPB = 256 / TPG > 0 ? 256 / TPG : 1;       // groups per block
    const int gl = threadIdx.x / TPG, rem = threadIdx.x % TPG;
    const int g = blockIdx.y * GPB + gl;
    const int r = rem / (Q * Q), iq = (rem / Q) % Q, oq = rem % Q;
    const long npix = (long)n * ho * wo;
    const long p0 = (long)blockIdx.x * pix_per_block, p1 = min(npix, p0 + pix_per_block);
    f32x4 acc[3][4];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[q][e] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (g < groups && gl < GPB) {
        // 4 pixels per trip: their 16 loads are independent and issued together (branch-free: clamped address, zero factor);
        // (img, oh, ow) of the running pixel is kept incrementally (a 64-bit index decomposition per pixel cost more than the FMAs)
        constexpr int UP = 4;
        int ow = (int)(p0 % wo);
        int oh = (int)((p0 / wo) % ho), img = (int)(p0 / ((long)wo * ho));
        const float* xg = x + g * GW + iq * 4;
        const float* dg = dz + g * GW + oq * 4;
        for (long pb = p0; pb < p1; pb += UP) {
            f32x4 d[UP], xs[UP][3];
#pragma unroll
            for (int u = 0; u < UP; ++u) {
                const long p = pb + u;
                const bool pok = p < p1;
                d[u] = *reinterpret_cast<const f32x4*>(dg + (pok ? p : p0) * c);
                if (!pok) d[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int im = pok ? img : 0;                      // pixels past the chunk read image 0 (in bounds) times zero
                const int ih = oh * stride + r - pt;
                const bool rok = pok && ih >= 0 && ih < h;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int iw = ow * stride + q - pl;
                    const bool ok = rok && iw >= 0 && iw < wd;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(xg + ((long)(im * h + (ok ? ih : 0)) * wd + (ok ? iw : 0)) * c);
                    xs[u][q] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
                }
                if (++ow == wo) { ow = 0; if (++oh == ho) { oh = 0; ++img; } }
            }
#pragma unroll
            for (int u = 0; u < UP; ++u)
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[q][e] += xs[u][q][e] * d[u];
        }
        float* out = part + ((long)blockIdx.x * groups + g) * 9 * GW * GW;
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                *reinterpret_cast<f32x4*>(out + ((r * 3 + q) * GW + iq * 4 + e) * GW + oq * 4) = acc[q][e];
    }
}

// ---- activations (SE gate: ReLU on the squeeze FC, sigmoid on the excitation FC) ------------------
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n, int kind) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = x[i];
        y[i] = kind == 0 ? fmaxf(v, 0.f) : 1.f / (1.f + expf(-v));
    }
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                                                      long n, int kind) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float o = y[i];
        dx[i] = kind == 0 ? (o > 0.f ? dy[i] : 0.f) : dy[i] * o * (1.f - o);
    }
}
// y[n,hw,c] = x * gate[n,c]
template <bool H = false>      // H: x / y (and dy / dx below) are bf16 in HBM; gate, dgate, dsq are [n, c] fp32
__global__ __launch_bounds__(256) void chscale_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gate,
                                                          float* __restrict__ y, int hwq, int cq) {
    // grid = (blocks over one image's hw * c / 4 quads, image); the stride is a multiple of c / 4 (chscale_grid), so a thread keeps its
    // gate quad and has four pieces in flight (a flat walk with a 64-bit / and % per piece ran the SE blocks' passes at 2.5-3 TB/s)
    const int step = (int)gridDim.x * 256;
    const long base = (long)blockIdx.y * hwq;
    int i = (int)blockIdx.x * 256 + threadIdx.x;
    if (i >= hwq) return;
    const f32x4 g = *reinterpret_cast<const f32x4*>(gate + ((long)blockIdx.y * cq + i % cq) * 4);
    for (; i + 3 * step < hwq; i += 4 * step) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ldq<H>(x, (base + i + u * step) * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) stq<H>(y, (base + i + u * step) * 4, v[u] * g);
    }
    for (; i < hwq; i += step) stq<H>(y, (base + i) * 4, ldq<H>(x, (base + i) * 4) * g);
}
// dx = dy * gate + dsq * scale: the whole gradient of an SE block's input in one pass (the gate path's dy * gate and the squeeze
// path's broadcast), after chscale_bwd_kernel's reduction has gone through the two dense layers.  With bf16 storage dx is written
// once, rounded once (the fp32 flow below writes dy * gate first and adds the broadcast in place).
template <bool H = false>
__global__ __launch_bounds__(256) void chscale_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ gate,
                                                                const float* __restrict__ dsq, float* __restrict__ dx,
                                                                int hwq, int cq, float scale) {
    const int step = (int)gridDim.x * 256;                   // (grid and walk as chscale_fwd_kernel)
    const long base = (long)blockIdx.y * hwq;
    int i = (int)blockIdx.x * 256 + threadIdx.x;
    if (i >= hwq) return;
    const long go = ((long)blockIdx.y * cq + i % cq) * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(gate + go), q = *reinterpret_cast<const f32x4*>(dsq + go) * scale;
    for (; i + 3 * step < hwq; i += 4 * step) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ldq<H>(dy, (base + i + u * step) * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) stq<H>(dx, (base + i + u * step) * 4, v[u] * g + q);
    }
    for (; i < hwq; i += step) stq<H>(dx, (base + i) * 4, ldq<H>(dy, (base + i) * 4) * g + q);
}
// dx = dy * gate ; dgate[n,c] = sum_hw dy * x   (block = one image x 16 channel quads x 16 row lanes, float4, fixed-order LDS sum)
template <bool H = false>      // dx may be NULL: reduction only
__global__ __launch_bounds__(256) void chscale_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ gate, float* __restrict__ dx,
                                                          float* __restrict__ dgate, int hw, int c, int pre_sigmoid) {
    __shared__ f32x4 sh[16][16];
    const int q = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int ch = (blockIdx.x * 16 + q) * 4, img = blockIdx.y;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    f32x4 gt = {0.f, 0.f, 0.f, 0.f};
    if (ch < c) {
        gt = *reinterpret_cast<const f32x4*>(gate + (long)img * c + ch);
        int r = rl;
        for (; r + 48 < hw; r += 64) {                          // four rows of both inputs in flight per lane
            f32x4 d[4], xv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long o = ((long)img * hw + r + 16 * u) * c + ch;
                d[u] = ldq<H>(dy, o); xv[u] = ldq<H>(x, o);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s += d[u] * xv[u];
                if (dx) stq<H>(dx, ((long)img * hw + r + 16 * u) * c + ch, d[u] * gt);
            }
        }
        for (; r < hw; r += 16) {
            const long o = ((long)img * hw + r) * c + ch;
            const f32x4 d = ldq<H>(dy, o);
            s += d * ldq<H>(x, o);
            if (dx) stq<H>(dx, o, d * gt);
        }
    }
    sh[rl][q] = s;
    __syncthreads();
    if (rl == 0 && ch < c) {
        f32x4 t = sh[0][q];
#pragma unroll
        for (int l = 1; l < 16; ++l) t += sh[l][q];
        if (pre_sigmoid) t = t * gt * (1.f - gt);                // gate = sigmoid(pre): the gradient w.r.t. pre
        *reinterpret_cast<f32x4*>(dgate + (long)img * c + ch) = t;
    }
}

template <bool DGRAD>
hipError_t gconv_launch(const float* x, const float* w, float* y, int n, int h, int wd, int c, int groups, int ho, int wo,
                        int stride, int pt, int pl, hipStream_t st) {
    const int gw = c / groups;
    if (stride != 1 && stride != 2) return hipErrorInvalidValue;
    const int oh_ = DGRAD ? h : ho, ow_ = DGRAD ? wd : wo;
    const long units = (long)n * oh_ * ((ow_ + 3) / 4);
    auto blocks = [&](int upb) { long b = (units + upb - 1) / upb; return (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); };
#define FTE_GCONV(GW_, GPBK_) do { \
        if (stride == 1) hipLaunchKernelGGL((gconv3x3_kernel<GW_, DGRAD, GPBK_, 1>), dim3(blocks(256 / ((GW_ / 4) * GPBK_)), groups / GPBK_), \
                                            dim3(256), 0, st, x, w, y, n, h, wd, c, ho, wo, pt, pl); \
        else hipLaunchKernelGGL((gconv3x3_kernel<GW_, DGRAD, GPBK_, 2>), dim3(blocks(256 / ((GW_ / 4) * GPBK_)), groups / GPBK_), \
                                dim3(256), 0, st, x, w, y, n, h, wd, c, ho, wo, pt, pl); } while (0)
    switch (gw) {
        case 4:  if (groups % 32 == 0) FTE_GCONV(4, 32); else FTE_GCONV(4, 1); break;
        case 8:  if (groups % 16 == 0) FTE_GCONV(8, 16); else FTE_GCONV(8, 1); break;
        case 16: if (groups % 4 == 0) FTE_GCONV(16, 4); else FTE_GCONV(16, 1); break;
        case 32: FTE_GCONV(32, 1); break;
        default: return hipErrorInvalidValue;
    }
#undef FTE_GCONV
    return hipGetLastError();
}

}  // namespace

hipError_t l_gconv_fwd(const float* x, const float* w, float* y, int n, int h, int wd, int c, int groups, int ho, int wo,
                       int stride, int pt, int pl, hipStream_t st) {
    return gconv_launch<false>(x, w, y, n, h, wd, c, groups, ho, wo, stride, pt, pl, st);
}
hipError_t l_gconv_dgrad(const float* dz, const float* w, float* dx, int n, int h, int wd, int c, int groups, int ho, int wo,
                         int stride, int pt, int pl, hipStream_t st) {
    return gconv_launch<true>(dz, w, dx, n, h, wd, c, groups, ho, wo, stride, pt, pl, st);
}
hipError_t l_gconv_pack16(const float* w, unsigned short* wf, unsigned short* wd, int c, int groups, hipStream_t st) {
    const int total = (c / 32) * 9 * 1024;
    hipLaunchKernelGGL(gconv_pack16_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, wf, wd, c, c / groups);
    return hipGetLastError();
}
// y [n, h, wd] is the walked grid, x [n, hs, ws] the source; mode as gconv3x3_mfma16_kernel's MODE
// blocks along x of a launch over [n, h, wd] output pixels (4 waves of 32-pixel tiles each); the "BN fusion" forms leave one partial
// row per wave, so their grid is capped at 512 blocks (2048 partial rows, still >= 8192 waves)
static long gconv16_blocks(long npix, int c, bool bn) {
    const long ntiles = (npix + 31) / 32;
    long bx = (ntiles + 3) / 4;
    long cap = 8192 / (c / 32) > 1 ? 8192 / (c / 32) : 1;
    if (bn && cap > 512) cap = 512;
    static const long total = getenv("FTE_GCONV_BLOCKS") ? atol(getenv("FTE_GCONV_BLOCKS")) : 512;    // blocks of the whole launch: two per CU, several tiles per wave (measured 2048 / 1024 / 768 / 512 / 384 / 256: DESIGN.md)
    if (total > 0 && cap > total / (c / 32)) cap = total / (c / 32) > 1 ? total / (c / 32) : 1;
    return bx > cap ? cap : bx;
}
int l_gconv_bn_rows(int n, int h, int wd, int c) { return (int)gconv16_blocks((long)n * h * wd, c, true) * 4; }
hipError_t l_gconv_mfma16(const float* x, const unsigned short* wpk, float* y, int n, int h, int wd, int c, int hs, int ws,
                          int mode, int pt, int pl, hipStream_t st, int h16) {
    const dim3 grid((unsigned)gconv16_blocks((long)n * h * wd, c, false), c / 32);
    static const bool win = !(getenv("FTE_GCONV_WIN") && atoi(getenv("FTE_GCONV_WIN")) == 0);     // A/B hook: 0 = a fetch per tap
    const GconvBn nobn = {};
    if (h16) {
        if (mode == 0 && win) hipLaunchKernelGGL(gconv3x3_mfma16_win_kernel<true>, grid, dim3(256), 0, st, x, wpk, y, n, h, wd, c, nobn);
        else if (mode == 0) hipLaunchKernelGGL((gconv3x3_mfma16_kernel<0, true>), grid, dim3(256), 0, st, x, wpk, y, n, h, wd, c, hs, ws, pt, pl, nobn);
        else if (mode == 1) hipLaunchKernelGGL((gconv3x3_mfma16_kernel<1, true>), grid, dim3(256), 0, st, x, wpk, y, n, h, wd, c, hs, ws, pt, pl, nobn);
        else hipLaunchKernelGGL((gconv3x3_mfma16_kernel<2, true>), grid, dim3(256), 0, st, x, wpk, y, n, h, wd, c, hs, ws, pt, pl, nobn);
        return hipGetLastError();
    }
    if (mode == 0 && win) hipLaunchKernelGGL(gconv3x3_mfma16_win_kernel<false>, grid, dim3(256), 0, st, x, wpk, y, n, h, wd, c, nobn);
    else if (mode == 0) hipLaunchKernelGGL((gconv3x3_mfma16_kernel<0, false>), grid, dim3(256), 0, st, x, wpk, y, n, h, wd, c, hs, ws, pt, pl, nobn);
    else if (mode == 1) hipLaunchKernelGGL((gconv3x3_mfma16_kernel<1, false>), grid, dim3(256), 0, st, x, wpk, y, n, h, wd, c, hs, ws, pt, pl, nobn);
    else hipLaunchKernelGGL((gconv3x3_mfma16_kernel<2, false>), grid, dim3(256), 0, st, x, wpk, y, n, h, wd, c, hs, ws, pt, pl, nobn);
    return hipGetLastError();
}
// the same launches (bf16 storage) with the BN work in the epilogue: bnf = 1 forward statistics (mode 0 / 1), bnf = 2 data gradient
// landing on a BN (+ ReLU) output (mode 0 / 2).  part (and pgx): l_gconv_bn_rows(n, h, wd, c) partial rows.
hipError_t l_gconv_mfma16_bn(const float* x, const unsigned short* wpk, float* y, int n, int h, int wd, int c, int hs, int ws,
                             int mode, int pt, int pl, int bnf, float* part, float* pgx, const unsigned short* zbn, const float* mu,
                             const float* rs, const float* sc, const float* sh, hipStream_t st, const float* isc, const float* ish,
                             unsigned short* yside) {
    const dim3 grid((unsigned)gconv16_blocks((long)n * h * wd, c, true), c / 32);
    const GconvBn bn = {part, pgx, zbn, mu, rs, sc, sh, isc, ish, yside};
    if (isc && !(bnf == 1 && mode == 0)) return hipErrorInvalidValue;
    if (bnf == 1 && mode == 0 && isc) hipLaunchKernelGGL((gconv3x3_mfma16_win_kernel<true, 1, true>), grid, dim3(256), 0, st, x, wpk, y, n, h, wd, c, bn);
    else if (bnf == 1 && mode == 0) hipLaunchKernelGGL((gconv3x3_mfma16_win_kernel<true, 1>), grid, dim3(256), 0, st, x, wpk, y, n, h, wd, c, bn);
    else if (bnf == 1 && mode == 1) hipLaunchKernelGGL((gconv3x3_mfma16_kernel<1, true, 1>), grid, dim3(256), 0, st, x, wpk, y, n, h, wd, c, hs, ws, pt, pl, bn);
    else if (bnf == 2 && mode == 0) hipLaunchKernelGGL((gconv3x3_mfma16_win_kernel<true, 2>), grid, dim3(256), 0, st, x, wpk, y, n, h, wd, c, bn);
    else if (bnf == 2 && mode == 2) hipLaunchKernelGGL((gconv3x3_mfma16_kernel<2, true, 2>), grid, dim3(256), 0, st, x, wpk, y, n, h, wd, c, hs, ws, pt, pl, bn);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
int l_gconv_wgrad16_chunks(long npix, int c) {
    const long steps = (npix + 15) / 16;
    long ch = 1024 / (c / 32);
    if (ch > steps / 8) ch = steps / 8;
    return (int)(ch < 1 ? 1 : ch);
}
hipError_t l_gconv_wgrad16(const float* x, const float* dz, float* part, float* dw, int n, int h, int wd, int c, int groups,
                           int ho, int wo, int stride, int pt, int pl, int chunks, hipStream_t st, int h16) {
    const long steps = ((long)n * ho * wo + 15) / 16, spc = (steps + chunks - 1) / chunks;
    const int gw = c / groups, total = groups * 9 * gw * gw;
    const dim3 grid(chunks, c / 32);
    if (h16) {
        if (stride == 1) hipLaunchKernelGGL((gconv3x3_wgrad_mfma16_kernel<1, true>), grid, dim3(192), 0, st, x, dz, part, n, h, wd, c, gw, spc, ho, wo, pt, pl);
        else hipLaunchKernelGGL((gconv3x3_wgrad_mfma16_kernel<2, true>), grid, dim3(192), 0, st, x, dz, part, n, h, wd, c, gw, spc, ho, wo, pt, pl);
    } else if (stride == 1) hipLaunchKernelGGL((gconv3x3_wgrad_mfma16_kernel<1, false>), grid, dim3(192), 0, st, x, dz, part, n, h, wd, c, gw, spc, ho, wo, pt, pl);
    else hipLaunchKernelGGL((gconv3x3_wgrad_mfma16_kernel<2, false>), grid, dim3(192), 0, st, x, dz, part, n, h, wd, c, gw, spc, ho, wo, pt, pl);
    hipLaunchKernelGGL(gconv_wgrad16_reduce_kernel, dim3((total + 63) / 64), dim3(256), 0, st, part, dw, chunks, c / 32, gw, groups);
    return hipGetLastError();
}
// pixel chunks of the grouped filter gradient: every thread walks its chunk serially (10 dependent-latency loads per
// pixel), so the chunk count IS the memory-level parallelism -- 64 pixels per chunk, bounded by a 96 MiB partial buffer
// (256 chunks of ~400 pixels ran the 28x28x128 layer of ResNeXt-50 at 800 us against an HBM time of 15 us)
int l_gconv_wgrad_chunks(long npix, int c, int gw) {
    long b = (npix + 63) / 64;
    const long cap = (96L << 20) / (9L * c * gw * (long)sizeof(float));
    if (b > cap) b = cap;
    if (b > 4096) b = 4096;
    return (int)(b < 1 ? 1 : b);
}
hipError_t l_gconv_wgrad(const float* x, const float* dz, float* part, int n, int h, int wd, int c, int groups, int ho, int wo,
                         int stride, int pt, int pl, int chunks, hipStream_t st) {
    const int gw = c / groups;
    const long npix = (long)n * ho * wo, ppb = (npix + chunks - 1) / chunks;
    const int tpg = 3 * (gw / 4) * (gw / 4), gpb = 256 / tpg > 0 ? 256 / tpg : 1;
    const int gy = (groups + gpb - 1) / gpb;
    const int used = (groups < gpb ? groups : gpb) * tpg;          // active threads: the block is cut to whole waves of them
    const dim3 grid(chunks, gy), blk((used + 63) / 64 * 64);
    switch (gw) {
        case 4:  hipLaunchKernelGGL(gconv3x3_wgrad_kernel<4>, grid, blk, 0, st, x, dz, part, n, h, wd, c, ho, wo, stride, pt, pl, ppb, groups); break;
        case 8:  hipLaunchKernelGGL(gconv3x3_wgrad_kernel<8>, grid, blk, 0, st, x, dz, part, n, h, wd, c, ho, wo, stride, pt, pl, ppb, groups); break;
        case 16: hipLaunchKernelGGL(gconv3x3_wgrad_kernel<16>, grid, blk, 0, st, x, dz, part, n, h, wd, c, ho, wo, stride, pt, pl, ppb, groups); break;
        case 32: hipLaunchKernelGGL(gconv3x3_wgrad_kernel<32>, grid, blk, 0, st, x, dz, part, n, h, wd, c, ho, wo, stride, pt, pl, ppb, groups); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
hipError_t l_act_fwd(const float* x, float* y, long n, int kind, hipStream_t st) {
    hipLaunchKernelGGL(act_fwd_kernel, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, st, x, y, n, kind);
    return hipGetLastError();
}
hipError_t l_act_bwd(const float* dy, const float* y, float* dx, long n, int kind, hipStream_t st) {
    hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, st, dy, y, dx, n, kind);
    return hipGetLastError();
}
// grid of the channel-scale passes: (blocks over one image's quads, images), ~1024 blocks in all, the stride (blocks.x * 256) a
// multiple of c / 4 so that a thread keeps its gate quad
static dim3 chscale_grid(int n, int hw, int c) {
    const long hwq = (long)hw * (c / 4), cq = c / 4;
    long a = cq, b = 256;
    while (b) { const long t = a % b; a = b; b = t; }
    const long m = cq / a;                                       // blocks.x must be a multiple of m
    long gx = (hwq + 1023) / 1024;                               // four pieces per thread
    const long want = (1024 + n - 1) / n;
    if (gx > want) gx = want;
    if (gx < 1) gx = 1;
    gx = (gx + m - 1) / m * m;
    return dim3((unsigned)gx, (unsigned)n);
}
hipError_t l_chscale_fwd(const float* x, const float* gate, float* y, int n, int hw, int c, hipStream_t st, int h16) {
    if (c % 4 || (long)hw * c / 4 >= (1L << 30)) return hipErrorInvalidValue;
    const dim3 grid = chscale_grid(n, hw, c);
    if (h16) hipLaunchKernelGGL(chscale_fwd_kernel<true>, grid, dim3(256), 0, st, x, gate, y, hw * (c / 4), c / 4);
    else hipLaunchKernelGGL(chscale_fwd_kernel<false>, grid, dim3(256), 0, st, x, gate, y, hw * (c / 4), c / 4);
    return hipGetLastError();
}
hipError_t l_chscale_bwd(const float* dy, const float* x, const float* gate, float* dx, float* dgate, int n, int hw, int c,
                         int pre_sigmoid, hipStream_t st, int h16) {
    if (h16) hipLaunchKernelGGL(chscale_bwd_kernel<true>, dim3((c / 4 + 15) / 16, n), dim3(256), 0, st, dy, x, gate, dx, dgate, hw, c, pre_sigmoid);
    else hipLaunchKernelGGL(chscale_bwd_kernel<false>, dim3((c / 4 + 15) / 16, n), dim3(256), 0, st, dy, x, gate, dx, dgate, hw, c, pre_sigmoid);
    return hipGetLastError();
}
hipError_t l_chscale_bwd_apply(const float* dy, const float* gate, const float* dsq, float* dx, int n, int hw, int c, float scale, hipStream_t st, int h16) {
    if (c % 4 || (long)hw * c / 4 >= (1L << 30)) return hipErrorInvalidValue;
    const dim3 grid = chscale_grid(n, hw, c);
    if (h16) hipLaunchKernelGGL(chscale_bwd_apply_kernel<true>, grid, dim3(256), 0, st, dy, gate, dsq, dx, hw * (c / 4), c / 4, scale);
    else hipLaunchKernelGGL(chscale_bwd_apply_kernel<false>, grid, dim3(256), 0, st, dy, gate, dsq, dx, hw * (c / 4), c / 4, scale);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// SE residual block in one forward pass over the tensor and two backward passes (round 5).  The block of nets/resnet.py:63-92 with the
// gate of nets/shufflenet_v2.py:79-85 is  z -> y = BN(z) (no activation) -> s = y * gate(mean_hw y) -> out = relu(s + shortcut).  As
// separate ops that was bn_apply (R z, W y), gap (R y), channel_scale (R y, W s), add + ReLU (R s, R shortcut, W out) forward and
// relu_bwd, channel_scale_bwd, channel_scale_bwd_apply, BN reduce, BN apply backward: 7 + 12 passes over the block's tensor.  Here y and
// s never exist in HBM:
//   forward   se_squeeze   R z            sq[n,c] = scale * mean_hw(z) + shift  (= mean_hw(y): BN without activation is affine per
//                                         channel), xm[n,c] = mean_hw(xhat) for the backward sums
//             (the gate's two dense layers on sq, as before)
//             se_apply     R z, R sc, W out     out = relu(fma(z, scale, shift) * gate + shortcut)
//   backward  se_bwd_gate  R dy, R out, R z, W g   g = dy * (out > 0) (the shortcut's gradient too); per image and channel
//                                         S1 = sum_hw g, S2 = sum_hw g * xhat; dgate_pre = (gamma * S2 + beta * S1) * gate * (1 - gate)
//                                         (sum_hw g * y with y = gamma * xhat + beta)
//             (the gate's dense layers backward -> dsq[n,c], the gradient of the squeeze)
//             se_bn_coef   [n,c] only     the BN backward sums of dy_bn = g * gate + dsq / hw from the per-image sums:
//                                         sum dy_bn = sum_n (gate * S1 + dsq), sum dy_bn * xhat = sum_n (gate * S2 + dsq * xm)
//                                         -> dgamma, dbeta, the coefficients of dz = A dy_bn + B z + C0 -- the BN reduce pass over the
//                                         tensor is gone
//             se_bn_apply  R g, R z, W dz   dz = (A * gate) g + B z + (A * dsq / hw + C0)
// 4 + 7 passes.  Sums in a fixed order (block-local, then over the images in order): bit-identical run to run.
// ---------------------------------------------------------------------------------------------------
template <bool ZH>
__global__ __launch_bounds__(256) void se_squeeze_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, float* __restrict__ sq, float* __restrict__ xm,
                                                         int hw, int c) {
    __shared__ f32x4 sh[16][16];
    const int q = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int ch = (blockIdx.x * 16 + q) * 4, img = blockIdx.y;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (ch < c) {
        const long px = (long)img * hw * c + ch;
        for (int r = rl; r < hw; r += 64) {                     // four rows in flight per lane (rows past the image: re-read row r, not added)
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = ldq<ZH>(z, px + (long)(r + 16 * u < hw ? r + 16 * u : r) * c);
#pragma unroll
            for (int u = 1; u < 4; ++u) if (r + 16 * u >= hw) v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            s0 += v[0] + v[2]; s1 += v[1] + v[3];
        }
    }
    sh[rl][q] = s0 + s1;
    __syncthreads();
    if (rl == 0 && ch < c) {
        f32x4 s = sh[0][q];
#pragma unroll
        for (int l = 1; l < 16; ++l) s += sh[l][q];
        const f32x4 zm = s / (float)hw;
        const long o = (long)img * c + ch;
        *reinterpret_cast<f32x4*>(sq + o) = bn_affine(zm, *reinterpret_cast<const f32x4*>(scale + ch), *reinterpret_cast<const f32x4*>(shift + ch));
        if (xm) *reinterpret_cast<f32x4*>(xm + o) = (zm - *reinterpret_cast<const f32x4*>(mean + ch)) * *reinterpret_cast<const f32x4*>(rstd + ch);
    }
}
// out = relu(fma(z, scale, shift) * gate[n,c] + shortcut)      (grid and walk as chscale_fwd_kernel: a thread keeps its channel quad)
template <bool ZH, bool AH>
__global__ __launch_bounds__(256) void se_apply_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ gate,
                                                       const float* __restrict__ res, float* __restrict__ out, int hwq, int cq) {
    const int step = (int)gridDim.x * 256;
    const long base = (long)blockIdx.y * hwq;
    int i = (int)blockIdx.x * 256 + threadIdx.x;
    if (i >= hwq) return;
    const int cqi = i % cq;
    const f32x4 g = *reinterpret_cast<const f32x4*>(gate + ((long)blockIdx.y * cq + cqi) * 4);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + cqi * 4), sf = *reinterpret_cast<const f32x4*>(shift + cqi * 4);
    auto f = [&](const f32x4 zz, const f32x4 rr) {
        f32x4 v = bn_affine(zz, sc, sf) * g + rr;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        return v;
    };
    for (; i + 3 * step < hwq; i += 4 * step) {
        f32x4 zv[4], rv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { zv[u] = ldq<ZH>(z, (base + i + u * step) * 4); rv[u] = ldq<AH>(res, (base + i + u * step) * 4); }
#pragma unroll
        for (int u = 0; u < 4; ++u) stq<AH>(out, (base + i + u * step) * 4, f(zv[u], rv[u]));
    }
    for (; i < hwq; i += step) stq<AH>(out, (base + i) * 4, f(ldq<ZH>(z, (base + i) * 4), ldq<AH>(res, (base + i) * 4)));
}
// g = dy * (out > 0) (stored: with bf16 storage the STORED value is what every sum below and the apply pass see);
// S1[n,c] = sum_hw g, S2[n,c] = sum_hw g * xhat, dgate[n,c] = (gamma * S2 + beta * S1) * gate * (1 - gate)
template <bool ZH, bool AH>
__global__ __launch_bounds__(256) void se_bwd_gate_kernel(const float* __restrict__ dy, const float* __restrict__ out,
                                                          const float* __restrict__ z, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ gate,
                                                          float* __restrict__ g, float* __restrict__ s1, float* __restrict__ s2,
                                                          float* __restrict__ dgate, int hw, int c) {
    __shared__ f32x4 sh[2][16][16];
    const int q = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int ch = (blockIdx.x * 16 + q) * 4, img = blockIdx.y;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
    if (ch < c) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + ch), rs = *reinterpret_cast<const f32x4*>(rstd + ch);
        auto one = [&](const long o, f32x4 d, const f32x4 ov, const f32x4 zz) {
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = ov[e] > 0.f ? d[e] : 0.f;
            if constexpr (AH) {
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = __builtin_bit_cast(float, (unsigned)__builtin_bit_cast(unsigned short, (__bf16)d[e]) << 16);
            }
            stq<AH>(g, o, d);
            a += d;
            b += d * ((zz - mu) * rs);
        };
        for (int r = rl; r < hw; r += 64) {                     // four rows of the three inputs in flight per lane, also in the last trip
            f32x4 d[4], ov[4], zv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long o = ((long)img * hw + (r + 16 * u < hw ? r + 16 * u : r)) * c + ch;
                d[u] = ldq<AH>(dy, o); ov[u] = ldq<AH>(out, o); zv[u] = ldq<ZH>(z, o);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (r + 16 * u < hw) one(((long)img * hw + r + 16 * u) * c + ch, d[u], ov[u], zv[u]);
        }
    }
    sh[0][rl][q] = a; sh[1][rl][q] = b;
    __syncthreads();
    if (rl == 0 && ch < c) {
        f32x4 t1 = sh[0][0][q], t2 = sh[1][0][q];
#pragma unroll
        for (int l = 1; l < 16; ++l) { t1 += sh[0][l][q]; t2 += sh[1][l][q]; }
        const long o = (long)img * c + ch;
        *reinterpret_cast<f32x4*>(s1 + o) = t1;
        *reinterpret_cast<f32x4*>(s2 + o) = t2;
        const f32x4 gt = *reinterpret_cast<const f32x4*>(gate + o);
        const f32x4 gy = *reinterpret_cast<const f32x4*>(gamma + ch) * t2 + *reinterpret_cast<const f32x4*>(beta + ch) * t1;      // sum_hw g * y
        *reinterpret_cast<f32x4*>(dgate + o) = gy * gt * (1.f - gt);
    }
}
// block = 16 channel quads x 16 image lanes (image i on lane i % 16, every lane's loads independent), the lanes' sums added in lane
// order: the two sums of the BN backward, then bn_bwd_finalize_channel's arithmetic
__global__ __launch_bounds__(256) void se_bn_coef_kernel(const float* __restrict__ s1, const float* __restrict__ s2,
                                                         const float* __restrict__ gate, const float* __restrict__ dsq,
                                                         const float* __restrict__ xm, const float* __restrict__ gamma,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef,
                                                         int n, int C, float count) {
    __shared__ f32x4 sh[2][16][16];
    const int q = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int ch = (blockIdx.x * 16 + q) * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
    if (ch < C) {
        for (int i0 = rl; i0 < n; i0 += 64) {                   // four images' five quads in flight per lane
            f32x4 v1[4], v2[4], vg[4], vd[4], vx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long o = (long)(i0 + 16 * u < n ? i0 + 16 * u : i0) * C + ch;
                v1[u] = *reinterpret_cast<const f32x4*>(s1 + o); v2[u] = *reinterpret_cast<const f32x4*>(s2 + o);
                vg[u] = *reinterpret_cast<const f32x4*>(gate + o); vd[u] = *reinterpret_cast<const f32x4*>(dsq + o);
                vx[u] = *reinterpret_cast<const f32x4*>(xm + o);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + 16 * u < n) { a += vg[u] * v1[u] + vd[u]; b += vg[u] * v2[u] + vd[u] * vx[u]; }
        }
    }
    sh[0][rl][q] = a; sh[1][rl][q] = b;
    __syncthreads();
    if (rl != 0 || ch >= C) return;
#pragma unroll
    for (int l = 1; l < 16; ++l) { a += sh[0][l][q]; b += sh[1][l][q]; }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c1 = ch + e;
        dbeta[c1] = a[e];
        dgamma[c1] = b[e];
        const float gr = gamma[c1] * rstd[c1];
        const float bb = -gr * rstd[c1] * b[e] / count;
        coef[c1] = gr;
        coef[C + c1] = bb;
        coef[2 * C + c1] = -gr * a[e] / count - bb * mean[c1];
    }
}
// dz = A * (g * gate + dsq / hw) + B * z + C0      (grid and walk as chscale_fwd_kernel)
template <bool ZH, bool AH>
__global__ __launch_bounds__(256) void se_bn_apply_kernel(const float* __restrict__ g, const float* __restrict__ z,
                                                          const float* __restrict__ coef, const float* __restrict__ gate,
                                                          const float* __restrict__ dsq, float* __restrict__ dz, int hwq, int cq, float inv_hw) {
    const int step = (int)gridDim.x * 256;
    const long base = (long)blockIdx.y * hwq;
    int i = (int)blockIdx.x * 256 + threadIdx.x;
    if (i >= hwq) return;
    const int cqi = i % cq, C = cq * 4;
    const long go = ((long)blockIdx.y * cq + cqi) * 4;
    const f32x4 A = *reinterpret_cast<const f32x4*>(coef + cqi * 4), B = *reinterpret_cast<const f32x4*>(coef + C + cqi * 4);
    const f32x4 ag = A * *reinterpret_cast<const f32x4*>(gate + go);
    const f32x4 k0 = A * (*reinterpret_cast<const f32x4*>(dsq + go) * inv_hw) + *reinterpret_cast<const f32x4*>(coef + 2 * C + cqi * 4);
    for (; i + 3 * step < hwq; i += 4 * step) {
        f32x4 gv[4], zv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { gv[u] = ldq<AH>(g, (base + i + u * step) * 4); zv[u] = ldq<ZH>(z, (base + i + u * step) * 4); }
#pragma unroll
        for (int u = 0; u < 4; ++u) stq<ZH>(dz, (base + i + u * step) * 4, ag * gv[u] + B * zv[u] + k0);
    }
    for (; i < hwq; i += step) stq<ZH>(dz, (base + i) * 4, ag * ldq<AH>(g, (base + i) * 4) + B * ldq<ZH>(z, (base + i) * 4) + k0);
}
hipError_t l_se_squeeze(const float* z, const float* scale, const float* shift, const float* mean, const float* rstd, float* sq, float* xm,
                        int n, int hw, int c, hipStream_t st, int flags) {
    if (c % 4) return hipErrorInvalidValue;
    const dim3 grid((c / 4 + 15) / 16, n);
    if (flags & 1) hipLaunchKernelGGL(se_squeeze_kernel<true>, grid, dim3(256), 0, st, z, scale, shift, mean, rstd, sq, xm, hw, c);
    else hipLaunchKernelGGL(se_squeeze_kernel<false>, grid, dim3(256), 0, st, z, scale, shift, mean, rstd, sq, xm, hw, c);
    return hipGetLastError();
}
hipError_t l_se_apply(const float* z, const float* scale, const float* shift, const float* gate, const float* res, float* out,
                      int n, int hw, int c, hipStream_t st, int flags) {
    if (c % 4 || (long)hw * c / 4 >= (1L << 30)) return hipErrorInvalidValue;
    const dim3 grid = chscale_grid(n, hw, c);
    FTE_ZA(flags & 1, flags & 2, hipLaunchKernelGGL((se_apply_kernel<ZH, AH>), grid, dim3(256), 0, st, z, scale, shift, gate, res, out, hw * (c / 4), c / 4));
    return hipGetLastError();
}
hipError_t l_se_bwd_gate(const float* dy, const float* out, const float* z, const float* gamma, const float* beta, const float* mean,
                         const float* rstd, const float* gate, float* g, float* s1, float* s2, float* dgate, int n, int hw, int c,
                         hipStream_t st, int flags) {
    if (c % 4) return hipErrorInvalidValue;
    const dim3 grid((c / 4 + 15) / 16, n);
    FTE_ZA(flags & 1, flags & 2, hipLaunchKernelGGL((se_bwd_gate_kernel<ZH, AH>), grid, dim3(256), 0, st, dy, out, z, gamma, beta, mean, rstd, gate, g, s1, s2, dgate, hw, c));
    return hipGetLastError();
}
hipError_t l_se_bn_coef(const float* s1, const float* s2, const float* gate, const float* dsq, const float* xm, const float* gamma,
                        const float* mean, const float* rstd, float* dgamma, float* dbeta, float* coef, int n, int hw, int c, hipStream_t st) {
    hipLaunchKernelGGL(se_bn_coef_kernel, dim3((c / 4 + 15) / 16), dim3(256), 0, st, s1, s2, gate, dsq, xm, gamma, mean, rstd, dgamma, dbeta, coef,
                       n, c, (float)((long)n * hw));
    return hipGetLastError();
}
hipError_t l_se_bn_apply(const float* g, const float* z, const float* coef, const float* gate, const float* dsq, float* dz,
                         int n, int hw, int c, hipStream_t st, int flags) {
    if (c % 4 || (long)hw * c / 4 >= (1L << 30)) return hipErrorInvalidValue;
    const dim3 grid = chscale_grid(n, hw, c);
    FTE_ZA(flags & 1, flags & 2, hipLaunchKernelGGL((se_bn_apply_kernel<ZH, AH>), grid, dim3(256), 0, st, g, z, coef, gate, dsq, dz, hw * (c / 4), c / 4, 1.f / (float)hw));
    return hipGetLastError();
}

// dx[n,hw,c] += v[n,c] * scale   (gradient of the SE squeeze: d(mean over hw) broadcast back)
namespace {
__global__ __launch_bounds__(256) void bcast_add_kernel(float* __restrict__ dx, const float* __restrict__ v, long n4, int hw, int c, float scale) {
    f32x4* d4 = reinterpret_cast<f32x4*>(dx);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long e = i * 4;
        const int ch = (int)(e % c);
        const long img = e / ((long)hw * c);
        d4[i] += *reinterpret_cast<const f32x4*>(v + img * c + ch) * scale;
    }
}
}  // namespace
hipError_t l_bcast_add(float* dx, const float* v, int n, int hw, int c, float scale, hipStream_t st) {
    const long n4 = (long)n * hw * c / 4;
    hipLaunchKernelGGL(bcast_add_kernel, dim3((unsigned)((n4 + 255) / 256 > 8192 ? 8192 : (n4 + 255) / 256)), dim3(256), 0, st, dx, v, n4, hw, c, scale);
    return hipGetLastError();
}

// ===================================================================================================
// ShuffleNet-v2 pieces (nets/shufflenet_v2.py): depthwise 3x3 (the DepthwiseConv2dNative half of
// layers.separable_conv2d, :98,104) and a table-driven channel gather that implements _channel_split
// (:60-64), tf.concat + _channel_shuffle (:66-77, :112-113) and their gradients without ever
// materialising the concatenated tensor.  All HBM-bound: ~9 MAC per element.
// ===================================================================================================
namespace {

// y[n,oh,ow,c] = sum_{r,q} x[n, oh*s + r - pt, ow*s + q - pl, c] * w[r,q,c]        (DGRAD: the transposed gather)
template <bool DGRAD, int S>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        float* __restrict__ y, int n, int h, int wd, int c,
                                                        int ho, int wo, int pt, int pl) {
    constexpr int stride = S;                 // compile-time: the dgrad gather divides by it
    const int c4n = c >> 2;
    const int oh_ = DGRAD ? h : ho, ow_ = DGRAD ? wd : wo;
    const long total = (long)n * oh_ * ow_ * c4n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int c4, ox, oy, img;
        unflat4(i, total <= 0xffffffffL, c4n, ow_, oh_, c4, ox, oy, img);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            int sy; bool rok;
            if (!DGRAD) { sy = oy * stride + r - pt; rok = sy >= 0 && sy < h; }
            else { const int num = oy + pt - r; sy = num / stride; rok = num >= 0 && num % stride == 0 && sy < ho; }
            if (!rok) continue;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                int sx; bool ok;
                if (!DGRAD) { sx = ox * stride + q - pl; ok = sx >= 0 && sx < wd; }
                else { const int num = ox + pl - q; sx = num / stride; ok = num >= 0 && num % stride == 0 && sx < wo; }
                if (!ok) continue;
                const long src = DGRAD ? ((long)(img * ho + sy) * wo + sx) : ((long)(img * h + sy) * wd + sx);
                acc += *reinterpret_cast<const f32x4*>(x + src * c + c4 * 4) * *reinterpret_cast<const f32x4*>(w + (r * 3 + q) * c + c4 * 4);
            }
        }
        *reinterpret_cast<f32x4*>(y + i * 4) = acc;
    }
}

// Sliding-window variant for the unit-gather cases (forward at stride 1 / 2, dgrad at stride 1): a thread owns PX
// consecutive output pixels of one row and one channel quad, keeps the nine weight vectors in registers and loads every
// source column of the window once -- (PX-1)*S + 3 loads per row instead of 3*PX, and no weight re-loads (the
// one-pixel-per-thread loop issued 18 loads per output vector and ran at 25-35 % of HBM speed).
// DGRAD at stride 1 is the same gather with the taps mirrored: dx[iy,ix] = sum dy[iy+pt-r, ix+pl-q] * w[r,q].
template <bool DGRAD, int S, bool H = false>      // H: x and y are bf16 in HBM (bf16 storage); the filter stays fp32
__global__ __launch_bounds__(256) void dwconv3x3_win_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            float* __restrict__ y, int n, int hs, int ws_, int c,
                                                            int ho, int wo, int pt, int pl) {
    // hs x ws_: source image; ho x wo: destination image (for DGRAD the caller passes dy's size as the source)
    constexpr int PX = 4, NC = (PX - 1) * S + 3;
    const int c4n = c >> 2;
    const int wq = (wo + PX - 1) / PX;
    const long total = (long)n * ho * wq * c4n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int c4, xq, oy, img;
        unflat4(i, total <= 0xffffffffL, c4n, wq, ho, c4, xq, oy, img);
        const int ox0 = xq * PX;
        f32x4 wv[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wv[k] = *reinterpret_cast<const f32x4*>(w + k * c + c4 * 4);
        f32x4 acc[PX];
#pragma unroll
        for (int p = 0; p < PX; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        // source column of window slot j: forward sx = ox0*S - pl + j; dgrad (S = 1) sx = ox0 + pl - 2 + j
        const int sx0 = DGRAD ? ox0 + pl - 2 : ox0 * S - pl;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int sy = DGRAD ? oy + pt - r : oy * S + r - pt;
            if (sy < 0 || sy >= hs) continue;
            const long rowp = ((long)(img * hs + sy) * ws_) * c + c4 * 4;
            f32x4 v[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const int sx = sx0 + j;
                v[j] = (sx >= 0 && sx < ws_) ? ldq<H>(x, rowp + (long)sx * c) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int p = 0; p < PX; ++p)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    // forward: slot p*S + q, tap (r, q); dgrad: sx = ox0 + p + pl - q -> slot p + 2 - q, tap (r, q)
                    const int j = DGRAD ? p + 2 - q : p * S + q;
                    acc[p] += v[j] * wv[r * 3 + q];
                }
        }
#pragma unroll
        for (int p = 0; p < PX; ++p)
            if (ox0 + p < wo) stq<H>(y, (((long)(img * ho + oy) * wo + ox0 + p) * c) + c4 * 4, acc[p]);
    }
}

// Data gradient at stride 2 by PARITY PATCHES.  dx[iy,ix] = sum_{r,q} dy[(iy+pt-r)/2, (ix+pl-q)/2] * w[r,q] over the taps for which
// both quotients are whole: the four pixels of a 2x2 patch of dx (rows 2a, 2a+1; columns 2b, 2b+1) draw on the SAME 2x2 window of dy
// -- rows a-1+pt, a+pt, columns b-1+pl, b+pl -- with 4 / 2 / 2 / 1 of the nine taps.  A thread owns two horizontally adjacent
// patches of one channel quad: 6 dy vectors in, 8 dx vectors out, every tap's weight held in registers as a [patch row][dy row] x
// [patch column][dy column] table with zeros where the tap does not exist (16 multiply-adds per patch, 9 of them real): no
// division, no branch on the parity, and 0.75 loads per output vector instead of the 2.25 + 2.25 (weights) of the
// pixel-per-thread gather above (44-52 % of the HBM roofline; the grid stride is a multiple of the channel-quad count, so the
// weights are loaded once per thread).
template <bool H = false>
__global__ __launch_bounds__(256) void dwconv3x3_dgrad_s2_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                 float* __restrict__ dx, int n, int h, int wd, int c,
                                                                 int ho, int wo, int pt, int pl) {
    const int c4n = c >> 2;
    const int ph = (h + 1) >> 1, pw2 = (((wd + 1) >> 1) + 1) >> 1;        // patch rows, PAIRS of patch columns
    const long total = (long)n * ph * pw2 * c4n;
    const long step = (long)gridDim.x * 256;
    const bool inv = step % c4n == 0;                                    // this thread keeps its channel quad
    f32x4 wt[2][2][2][2];                                                // [dx row parity][dy row k][dx column parity][dy column j]
    auto load_w = [&](int c4) {
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int r = py + pt - 2 * (k - 1 + pt);               // iy = 2a + py, sy = a - 1 + pt + k  ->  r = iy + pt - 2 sy
#pragma unroll
                for (int px = 0; px < 2; ++px)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int q = px + pl - 2 * (j - 1 + pl);
                        const bool ok = r >= 0 && r < 3 && q >= 0 && q < 3;
                        wt[py][k][px][j] = ok ? *reinterpret_cast<const f32x4*>(w + (r * 3 + q) * c + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
            }
    };
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (inv && i < total) load_w((int)(i % c4n));
    for (; i < total; i += step) {
        int c4, bq, a, img;
        unflat4(i, total <= 0xffffffffL, c4n, pw2, ph, c4, bq, a, img);
        if (!inv) load_w(c4);
        const int b0 = 2 * bq;                                           // first of the two patch columns
        f32x4 v[2][3];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int sy = a - 1 + pt + k;
            const bool rok = sy >= 0 && sy < ho;
            const long rowp = ((long)(img * ho + (rok ? sy : 0)) * wo) * c + c4 * 4;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int sx = b0 - 1 + pl + j;
                v[k][j] = (rok && sx >= 0 && sx < wo) ? ldq<H>(dy, rowp + (long)sx * c) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int py = 0; py < 2; ++py) {
            const int iy = 2 * a + py;
            if (iy >= h) continue;
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    const int ix = 2 * (b0 + pb) + px;
                    if (ix >= wd) continue;
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc += v[k][pb + j] * wt[py][k][px][j];
                    stq<H>(dx, (((long)(img * h + iy) * wd + ix) * c) + c4 * 4, acc);
                }
        }
    }
}

// dw[r,q,c] partials: block = Q channel quads x (256/Q) lanes over one chunk of 4-pixel groups (4 consecutive outputs of
// one row): a lane loads the group's 4 dy vectors and each source column of the 3-row window once (3*(3*S+3) + 4 loads
// per 4 pixels instead of 40)
template <int Q, int S, bool H = false>
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ part, int n, int h, int wd, int c,
                                                              int ho, int wo, int pt, int pl, long grp_per_split) {
    constexpr int RL = 256 / Q, PX = 4, NC = (PX - 1) * S + 3;
    __shared__ f32x4 sh[RL][Q];
    const int q4 = threadIdx.x % Q, rl = threadIdx.x / Q;
    const int ch = (blockIdx.x * Q + q4) * 4;
    const int wq = (wo + PX - 1) / PX;
    const long ngrp = (long)n * ho * wq;
    const long g0 = (long)blockIdx.y * grp_per_split, g1 = min(ngrp, g0 + grp_per_split);
    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (ch < c) {
        for (long gi = g0 + rl; gi < g1; gi += RL) {
            const int xq = (int)(gi % wq);
            const long t2 = gi / wq;
            const int oh = (int)(t2 % ho), img = (int)(t2 / ho);
            const int ow0 = xq * PX;
            f32x4 d[PX];
#pragma unroll
            for (int p = 0; p < PX; ++p)
                d[p] = (ow0 + p < wo) ? ldq<H>(dy, (((long)(img * ho + oh) * wo + ow0 + p) * c) + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
            const int sx0 = ow0 * S - pl;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int ih = oh * S + r - pt;
                if (ih < 0 || ih >= h) continue;
                const long rowp = ((long)(img * h + ih) * wd) * c + ch;
                f32x4 v[NC];
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    const int sx = sx0 + j;
                    v[j] = (sx >= 0 && sx < wd) ? ldq<H>(x, rowp + (long)sx * c) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int p = 0; p < PX; ++p) acc[r * 3 + q] += v[p * S + q] * d[p];
            }
        }
    }
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
        sh[rl][q4] = acc[t];
        __syncthreads();
        if (rl == 0 && ch < c) {
            f32x4 s = sh[0][q4];
            for (int l = 1; l < RL; ++l) s += sh[l][q4];
            *reinterpret_cast<f32x4*>(part + ((long)blockIdx.y * 9 + t) * c + ch) = s;
        }
    }
}

// out[row, k] = (table[k] < 0) ? 0 : src[table[k] >> 16][row, table[k] & 0xffff]      (src 0 = a, 1 = b)
// thread = 4 consecutive output channels: four 4-byte gathers (neighbouring lanes read neighbouring source channels), one
// 16-byte store
template <bool H>
__device__ __forceinline__ float ld1(const float* p, long off) {
    if constexpr (H) return __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short*>(p)[off] << 16);
    else return p[off];
}
template <bool H = false>
__global__ __launch_bounds__(256) void channel_gather_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             float* __restrict__ out, const int* __restrict__ table,
                                                             long rows, int ca, int cb, int co) {
    const int q = co >> 2;
    const long total = rows * q;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int k4; long row;
        unflat2(i, total <= 0xffffffffL, q, k4, row);
        const int4 t = *reinterpret_cast<const int4*>(table + 4 * k4);
        const int tt[4] = {t.x, t.y, t.z, t.w};
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = 0.f;
            if (tt[e] >= 0) {
                const int ch = tt[e] & 0xffff;
                x = (tt[e] >> 16) ? ld1<H>(b, row * cb + ch) : ld1<H>(a, row * ca + ch);
            }
            v[e] = x;
        }
        stq<H>(out, row * co + 4 * k4, v);
    }
}

// the same gather with batch norm (+ ReLU) applied to a source on the way: v = [relu](fma(src, scale[ch], shift[ch])) for a
// source whose scale is given.  conv3_1x1's BN+ReLU output (and the stride-2 shortcut's) is consumed only by the concat /
// shuffle / split that follows (nets/shufflenet_v2.py:110-113): it is never written to HBM.  Two outputs in one launch
// (out1 / table1 / co1, optional): the two halves a block hands to the next one, or the gradients of both sources.
template <bool H = false>
__global__ __launch_bounds__(256) void channel_gather_affine_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                    float* __restrict__ out0, const int* __restrict__ table0, int co0,
                                                                    float* __restrict__ out1, const int* __restrict__ table1, int co1,
                                                                    long rows, int ca, int cb,
                                                                    const float* __restrict__ sca, const float* __restrict__ sfa, int relu_a,
                                                                    const float* __restrict__ scb, const float* __restrict__ sfb, int relu_b) {
    const int q0 = co0 >> 2, q = q0 + (co1 >> 2);
    const long total = rows * q;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int k4; long row;
        unflat2(i, total <= 0xffffffffL, q, k4, row);
        const bool second = k4 >= q0;
        if (second) k4 -= q0;
        const int4 t = *reinterpret_cast<const int4*>((second ? table1 : table0) + 4 * k4);
        const int tt[4] = {t.x, t.y, t.z, t.w};
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = 0.f;
            if (tt[e] >= 0) {
                const int ch = tt[e] & 0xffff;
                if (tt[e] >> 16) {
                    x = ld1<H>(b, row * cb + ch);
                    if (scb) { x = __builtin_fmaf(x, scb[ch], sfb[ch]); if (relu_b) x = fmaxf(x, 0.f); }
                } else {
                    x = ld1<H>(a, row * ca + ch);
                    if (sca) { x = __builtin_fmaf(x, sca[ch], sfa[ch]); if (relu_a) x = fmaxf(x, 0.f); }
                }
            }
            v[e] = x;
        }
        if (second) stq<H>(out1, row * co1 + 4 * k4, v);
        else stq<H>(out0, row * co0 + 4 * k4, v);
    }
}


// The same gather through LDS, for rows of a power-of-two number of 16-byte quads (ShuffleNet-v2's padded halves: 128 / 256 / 512
// channels per source).  The kernels above fetch every element with its own 4-byte (2-byte) load -- 64 addresses per instruction,
// 3.0 TB/s however the loop around them is arranged (fixed quad per thread, four rows in flight: 20.3 vs 20.9 us).  Here a block
// moves GROUPS of rows: whole source rows come in as 16-byte (8-byte) pieces, four per thread in flight, get the batch-norm
// affine of their source on the way (the same fma / max as above: bit-identical results) and land in LDS as [row][a | b]; the
// permutation is applied by the LDS reads (4-byte reads at the table's positions, 2-way conflicts for the shuffle) and the
// outputs leave as 16-byte pieces.  Every per-thread quantity (source, channel quad, coefficients, table entries, LDS
// positions) is fixed for the whole walk.
template <bool H = false>
__global__ __launch_bounds__(256) void channel_gather_lds_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                 float* __restrict__ out0, const int* __restrict__ table0, int co0,
                                                                 float* __restrict__ out1, const int* __restrict__ table1, int co1,
                                                                 long rows, int ca, int cb,
                                                                 const float* __restrict__ sca, const float* __restrict__ sfa, int relu_a,
                                                                 const float* __restrict__ scb, const float* __restrict__ sfb, int relu_b,
                                                                 int qin, int qout, int rg) {
    extern __shared__ __attribute__((aligned(16))) float gsh[];          // [rg][4 * qin]
    const int tid = threadIdx.x, W = 4 * qin;
    // load side: this thread's source quad
    const int iq = tid & (qin - 1), ri = tid / qin, rin = 256 / qin, qa = ca >> 2;
    const bool fb = iq >= qa;
    const int chi = 4 * (fb ? iq - qa : iq);
    const float* const srcp = fb ? b : a;
    const long ldi = fb ? cb : ca;
    const float* const scp = fb ? scb : sca;
    const float* const sfp = fb ? sfb : sfa;
    const bool aff = scp != nullptr, relu = fb ? relu_b != 0 : relu_a != 0;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sf = {0.f, 0.f, 0.f, 0.f};
    if (aff) { sc = *reinterpret_cast<const f32x4*>(scp + chi); sf = *reinterpret_cast<const f32x4*>(sfp + chi); }
    // store side: this thread's output quad and where its four elements sit in a staged row
    const int oq = tid & (qout - 1), ro = tid / qout, rout = 256 / qout, q0 = co0 >> 2;
    const bool second = oq >= q0;
    const int k4 = second ? oq - q0 : oq;
    const int4 t = *reinterpret_cast<const int4*>((second ? table1 : table0) + 4 * k4);
    const int tt[4] = {t.x, t.y, t.z, t.w};
    int pos[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) pos[e] = tt[e] < 0 ? -1 : ((tt[e] >> 16) ? ca : 0) + (tt[e] & 0xffff);
    float* const outp = second ? out1 : out0;
    const long ldo = second ? co1 : co0;
    const int nin = rg / rin, nout = rg / rout;
    for (long g0 = (long)blockIdx.x * rg; g0 < rows; g0 += (long)gridDim.x * rg) {
        for (int s0 = 0; s0 < nin; s0 += 4) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long r = g0 + (long)(s0 + u) * rin + ri;
                v[u] = (s0 + u < nin && r < rows) ? ldq<H>(srcp, r * ldi + chi) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (s0 + u >= nin) continue;
                f32x4 x = v[u];
                if (aff) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        x[e] = __builtin_fmaf(x[e], sc[e], sf[e]);
                        if (relu) x[e] = fmaxf(x[e], 0.f);
                    }
                }
                *reinterpret_cast<f32x4*>(gsh + ((s0 + u) * rin + ri) * W + 4 * iq) = x;
            }
        }
        __syncthreads();
        for (int s = 0; s < nout; ++s) {
            const int rl = s * rout + ro;
            const long r = g0 + rl;
            if (r >= rows) break;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = pos[e] < 0 ? 0.f : gsh[rl * W + pos[e]];
            stq<H>(outp, r * ldo + 4 * k4, o);
        }
        __syncthreads();
    }
}

}  // namespace

hipError_t l_dwconv_fwd(const float* x, const float* w, float* y, int n, int h, int wd, int c, int ho, int wo, int stride, int pt, int pl, hipStream_t st, int h16) {
    const long total = (long)n * ho * ((wo + 3) / 4) * (c / 4);
    const dim3 grid((unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256));
    if (h16) {
        if (stride == 1) hipLaunchKernelGGL((dwconv3x3_win_kernel<false, 1, true>), grid, dim3(256), 0, st, x, w, y, n, h, wd, c, ho, wo, pt, pl);
        else hipLaunchKernelGGL((dwconv3x3_win_kernel<false, 2, true>), grid, dim3(256), 0, st, x, w, y, n, h, wd, c, ho, wo, pt, pl);
    } else if (stride == 1) hipLaunchKernelGGL((dwconv3x3_win_kernel<false, 1, false>), grid, dim3(256), 0, st, x, w, y, n, h, wd, c, ho, wo, pt, pl);
    else hipLaunchKernelGGL((dwconv3x3_win_kernel<false, 2, false>), grid, dim3(256), 0, st, x, w, y, n, h, wd, c, ho, wo, pt, pl);
    return hipGetLastError();
}
hipError_t l_dwconv_dgrad(const float* dy, const float* w, float* dx, int n, int h, int wd, int c, int ho, int wo, int stride, int pt, int pl, hipStream_t st, int h16) {
    if (stride == 1) {                       // source = dy (ho x wo = h x wd at stride 1), destination = dx
        const long total = (long)n * h * ((wd + 3) / 4) * (c / 4);
        const dim3 grid((unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256));
        if (h16) hipLaunchKernelGGL((dwconv3x3_win_kernel<true, 1, true>), grid, dim3(256), 0, st, dy, w, dx, n, ho, wo, c, h, wd, pt, pl);
        else hipLaunchKernelGGL((dwconv3x3_win_kernel<true, 1, false>), grid, dim3(256), 0, st, dy, w, dx, n, ho, wo, c, h, wd, pt, pl);
        return hipGetLastError();
    }
    static const bool old_gather = getenv("FTE_DW_DGRAD_S2") && atoi(getenv("FTE_DW_DGRAD_S2")) == 0;      // A/B hook: the pixel-per-thread gather (fp32 tensors only)
    if (old_gather && !h16) {
        const long total = (long)n * h * wd * (c / 4);
        const dim3 grid((unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256));
        hipLaunchKernelGGL((dwconv3x3_kernel<true, 2>), grid, dim3(256), 0, st, dy, w, dx, n, h, wd, c, ho, wo, pt, pl);
        return hipGetLastError();
    }
    const long total = (long)n * ((h + 1) / 2) * ((((wd + 1) / 2) + 1) / 2) * (c / 4);
    if (h16) hipLaunchKernelGGL(dwconv3x3_dgrad_s2_kernel<true>, dim3(grid_for_c(total, c)), dim3(256), 0, st, dy, w, dx, n, h, wd, c, ho, wo, pt, pl);
    else hipLaunchKernelGGL(dwconv3x3_dgrad_s2_kernel<false>, dim3(grid_for_c(total, c)), dim3(256), 0, st, dy, w, dx, n, h, wd, c, ho, wo, pt, pl);
    return hipGetLastError();
}
static int dw_quads(int c) { return c >= 256 ? 64 : (c >= 128 ? 32 : 16); }
int l_dwconv_wgrad_splits(long npix, int c) {
    const int Q = dw_quads(c);
    const long cb = (c / 4 + Q - 1) / Q;
    long s = 2048 / cb;
    if (s > npix / (256 / Q * 4)) s = npix / (256 / Q * 4);
    if (s > 1024) s = 1024;
    return (int)(s < 1 ? 1 : s);
}
hipError_t l_dwconv_wgrad(const float* x, const float* dy, float* part, int n, int h, int wd, int c, int ho, int wo, int stride,
                          int pt, int pl, int splits, hipStream_t st, int h16) {
    const long ngrp = (long)n * ho * ((wo + 3) / 4), gps = (ngrp + splits - 1) / splits;
    const int Q = dw_quads(c);
    const dim3 grid((c / 4 + Q - 1) / Q, splits);
#define FTE_DWW(Q_, S_) do { if (h16) hipLaunchKernelGGL((dwconv3x3_wgrad_kernel<Q_, S_, true>), grid, dim3(256), 0, st, x, dy, part, n, h, wd, c, ho, wo, pt, pl, gps); \
                             else hipLaunchKernelGGL((dwconv3x3_wgrad_kernel<Q_, S_, false>), grid, dim3(256), 0, st, x, dy, part, n, h, wd, c, ho, wo, pt, pl, gps); } while (0)
    if (stride == 1) { switch (Q) { case 64: FTE_DWW(64, 1); break; case 32: FTE_DWW(32, 1); break; default: FTE_DWW(16, 1); } }
    else { switch (Q) { case 64: FTE_DWW(64, 2); break; case 32: FTE_DWW(32, 2); break; default: FTE_DWW(16, 2); } }
#undef FTE_DWW
    return hipGetLastError();
}
// the LDS form when a row is a power-of-two number of quads on both sides (<= 256) and holds at most 16 KB of staging per 4 load steps
static bool gather_lds(const float* a, const float* b, float* out0, const int* table0, int co0, float* out1, const int* table1, int co1,
                       long rows, int ca, int cb, const float* sca, const float* sfa, int relu_a, const float* scb, const float* sfb, int relu_b,
                       hipStream_t st, int h16) {
    static const bool off = getenv("FTE_GATHER_LDS") && atoi(getenv("FTE_GATHER_LDS")) == 0;      // A/B hook: the element-gather kernels
    if (off || (ca & 3) || (cb & 3) || (co0 & 3) || (co1 & 3) || (cb && !b) || (co1 && !out1)) return false;
    const uintptr_t ptrs = (uintptr_t)a | (uintptr_t)b | (uintptr_t)out0 | (uintptr_t)out1 | (uintptr_t)sca | (uintptr_t)sfa | (uintptr_t)scb | (uintptr_t)sfb;
    if (ptrs & 15) return false;                          // whole 16-byte pieces on both sides (the element kernels take any 4-byte alignment of the sources)
    const int qin = (ca + cb) / 4, qout = (co0 + co1) / 4;
    if (qin < 1 || qin > 256 || (qin & (qin - 1)) || qout < 1 || qout > 256 || (qout & (qout - 1))) return false;
    static const int U = getenv("FTE_GATHER_U") ? atoi(getenv("FTE_GATHER_U")) : 4;
    static const long cap = getenv("FTE_GATHER_BLOCKS") ? atol(getenv("FTE_GATHER_BLOCKS")) : 2048;
    const int rin = 256 / qin, rout = 256 / qout, rg = U * (rin > rout ? rin : rout);
    const size_t lds = (size_t)rg * qin * 16;
    if (lds > 65536) return false;
    const long groups = (rows + rg - 1) / rg;
    const dim3 grid((unsigned)(groups > cap ? cap : (groups < 1 ? 1 : groups)));
    if (h16) hipLaunchKernelGGL(channel_gather_lds_kernel<true>, grid, dim3(256), lds, st, a, b, out0, table0, co0, out1, table1, co1, rows, ca, cb, sca, sfa, relu_a, scb, sfb, relu_b, qin, qout, rg);
    else hipLaunchKernelGGL(channel_gather_lds_kernel<false>, grid, dim3(256), lds, st, a, b, out0, table0, co0, out1, table1, co1, rows, ca, cb, sca, sfa, relu_a, scb, sfb, relu_b, qin, qout, rg);
    return true;
}
hipError_t l_channel_gather_affine(const float* a, const float* b, float* out0, const int* table0, int co0,
                                   float* out1, const int* table1, int co1, long rows, int ca, int cb,
                                   const float* sca, const float* sfa, int relu_a, const float* scb, const float* sfb, int relu_b, hipStream_t st, int h16) {
    if (gather_lds(a, b, out0, table0, co0, out1, table1, co1, rows, ca, cb, sca, sfa, relu_a, scb, sfb, relu_b, st, h16)) return hipGetLastError();
    const long total = rows * ((co0 + co1) / 4);
    const dim3 grid((unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256));
    if (h16) hipLaunchKernelGGL(channel_gather_affine_kernel<true>, grid, dim3(256), 0, st, a, b, out0, table0, co0, out1, table1, co1, rows, ca, cb, sca, sfa, relu_a, scb, sfb, relu_b);
    else hipLaunchKernelGGL(channel_gather_affine_kernel<false>, grid, dim3(256), 0, st, a, b, out0, table0, co0, out1, table1, co1, rows, ca, cb, sca, sfa, relu_a, scb, sfb, relu_b);
    return hipGetLastError();
}
hipError_t l_channel_gather(const float* a, const float* b, float* out, const int* table, long rows, int ca, int cb, int co, hipStream_t st, int h16) {
    if (gather_lds(a, b, out, table, co, nullptr, nullptr, 0, rows, ca, cb, nullptr, nullptr, 0, nullptr, nullptr, 0, st, h16)) return hipGetLastError();
    const long total = rows * (co / 4);
    const dim3 grid((unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256));
    if (h16) hipLaunchKernelGGL(channel_gather_kernel<true>, grid, dim3(256), 0, st, a, b, out, table, rows, ca, cb, co);
    else hipLaunchKernelGGL(channel_gather_kernel<false>, grid, dim3(256), 0, st, a, b, out, table, rows, ca, cb, co);
    return hipGetLastError();
}
