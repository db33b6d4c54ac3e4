// igemm_dev.h -- device-side pieces shared by the gathered-GEMM kernels (igemm.hip: fp32 / bf16-operand family;
// igemm16.hip: the bf16 LDS-DMA kernel): vector types, index helpers and the fused, LDS-transposed epilogue.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "igemm.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace igemm_dev {

constexpr int BK = 32;

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ int swz(int row, int chunk) { return (chunk ^ ((row >> 1) & 7)) << 2; }

__device__ __forceinline__ unsigned pkbf(float a, float b) {      // two floats -> packed bf16 pair, round to nearest even
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ unsigned short tobf(float a) {
    return __builtin_bit_cast(unsigned short, (__bf16)a);
}

// a / d for 0 <= a < 2^24 and d > 0 with rd = 1.f / d: float multiply, truncate, one-step fix-up (the compiler's exact
// 32-bit division is ~35 VALU instructions; the wgrad loaders decompose a pixel index on EVERY K-step)
__device__ __forceinline__ int fdiv(int a, int d, float rd) {
    int q = (int)((float)a * rd);
    const int r = a - q * d;
    q += (r >= d) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}

// four consecutive bf16 values (8 bytes) <-> f32x4
__device__ __forceinline__ f32x4 ld_bf4(const unsigned short* q) {
    const u32x2 u = *reinterpret_cast<const u32x2*>(q);
    return f32x4{__builtin_bit_cast(float, u[0] << 16), __builtin_bit_cast(float, u[0] & 0xffff0000u),
                 __builtin_bit_cast(float, u[1] << 16), __builtin_bit_cast(float, u[1] & 0xffff0000u)};
}
__device__ __forceinline__ void st_bf4(unsigned short* q, const f32x4 v) {
    *reinterpret_cast<u32x2*>(q) = u32x2{pkbf(v[0], v[1]), pkbf(v[2], v[3])};
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
__device__ __forceinline__ float rbf(float a) { return bf2f(tobf(a)); }      // the value a bf16 store of `a` leaves in memory

__device__ __forceinline__ float prelu_slope(float z, float a) {
    // d/dz [relu(z) + a*(z-|z|)/2]; TF's grad of relu(0) and sign(0) are 0 -> a/2 at exactly 0.
    return z > 0.f ? 1.f : (z == 0.f ? 0.5f * a : a);
}

// ---- epilogue of one BM x BN output tile held as (TM x TN) 32x32 MFMA accumulator blocks per wave (WM x WN waves) ----------
// acc[i][j][r]: row wm*TM*32 + i*32 + (r&3) + 8*(r>>2) + 4*(lane>>5), column wn*TN*32 + j*32 + (lane&31) (the C/D map of
// every 32x32 MFMA).  `smem` (dynamic LDS, free for reuse -- the caller has passed a barrier after its last operand read)
// must hold BM + WM*WN*32*36 + 2*WM*BN floats.  bid = tile index of the launch (after the XCD remap), mt = its row-tile index.
// CAP16 (the bf16-source kernels): the tensors of the epilogue may live in HBM as bf16 ("bf16 storage", fte_conv2d_*_s16): R16 /
// ADD16 / Zin16 replace the fp32 inputs when set, Z16 / RAW16 (and Y16 / DZ16) are written, and the fp32 outputs Y / DZ are
// optional.  The arithmetic stays fp32; every stored value is rounded once, to nearest even, where it is written.
// BNM (the graph nets' conv -> BN pairs, igemm.h "BN fusion"): the forward epilogue also leaves per-tile column statistics of the
// stored output (SP), the data-gradient epilogue applies the ReLU mask of the BN layer below and leaves the two sums of its
// backward pass as column partials (bn_mu).  Separate instantiations: the kernels of the BN-free nets compile to the code they had.
template <int BM, int BN, int WM, int WN, int EPI, bool CAP16 = false, bool BNM = false>
__device__ __forceinline__ void igemm_epilogue(const IgemmParams& p, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], float* smem,
                                               int bid, int split, int m0, int n0, int mt, int c_ph, int c_pw, int prow, int tid_in = -1) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32, NTH = 64 * WM * WN;
    const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x;      // (the stream-K loop passes its opaque copy of the id: igemm.hip)
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int li = lane & 31, lh = lane >> 5;
    // ---- split-K partial tile: raw accumulators to the workspace, epilogue happens in igemm_fixup ----
    if (p.PW) {
        float* W = p.PW + ((long)split * gridDim.x + bid) * (BM * BN);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int cl = wn * (TN * 32) + j * 32 + li;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    W[rl * BN + cl] = acc[i][j][r];
                }
        }
        return;
    }

    // ---- epilogue -----------------------------------------------------------------
    int* rowoff = reinterpret_cast<int*>(smem);
    for (int r = tid; r < BM; r += NTH) {
        const int m = m0 + r;
        int off = -1;
        if (m < p.M) {
            if (p.c_OH == 0) {
                off = m * p.c_ld;
            } else {
                const int hw = p.c_OH * p.c_OW;
                const int n = fdiv(m, hw, 1.f / (float)hw), rem = m - n * hw;
                const int oh = fdiv(rem, p.c_OW, 1.f / (float)p.c_OW), ow = rem - oh * p.c_OW;
                off = ((n * p.c_FH + oh * p.c_step + c_ph) * p.c_FW + ow * p.c_step + c_pw) * p.c_ld;
            }
        }
        rowoff[r] = off;
    }
    __syncthreads();

    if constexpr (BNM && EPI == EPI_FWD) {
        if (p.SP) {
            // ---- batch-norm statistics of this tile's stored values, straight from the accumulators (lane = column, registers =
            // rows): two passes over the registers per column -- mean of the wave's valid rows, then the squared deviations from
            // it -- the two half-waves meet through one shuffle, the WM row-waves through LDS with Chan's merge in wave order.
            // No E[x^2] - E[x]^2 anywhere; the value counted is the one the store below leaves in memory (bf16-rounded when the
            // output tensor is bf16).
            float* red = smem + BM + (WM * WN) * (32 * 36);      // [2][WM][BN], past the patches
            bool rnd = false;
            if constexpr (CAP16) rnd = p.Y16 != nullptr && p.Y == nullptr;
            const int lim = p.M - m0 - wm * (TM * 32) - 4 * lh;            // valid: i * 32 + (r & 3) + 8 * (r >> 2) < lim
            const int cntw = min(max(p.M - m0 - wm * (TM * 32), 0), TM * 32);
            const float rcnt = cntw > 0 ? 1.f / (float)cntw : 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int cl = wn * (TN * 32) + j * 32 + li;
                const float b = p.bias ? p.bias[n0 + cl] : 0.f;
                float s1 = 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[i][j][r] + b;
                        if (rnd) v = rbf(v);
                        s1 += (i * 32 + (r & 3) + 8 * (r >> 2) < lim) ? v : 0.f;
                    }
                s1 += __shfl_xor(s1, 32);
                const float mw = s1 * rcnt;
                float s2 = 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[i][j][r] + b;
                        if (rnd) v = rbf(v);
                        const float d = v - mw;
                        s2 += (i * 32 + (r & 3) + 8 * (r >> 2) < lim) ? d * d : 0.f;
                    }
                s2 += __shfl_xor(s2, 32);
                if (lh == 0) { red[wm * BN + cl] = mw; red[(WM + wm) * BN + cl] = s2; }
            }
            __syncthreads();
            for (int c = tid; c < BN; c += NTH) {
                float n = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) {
                    const float nb = (float)min(max(p.M - m0 - w * (TM * 32), 0), TM * 32);
                    if (nb > 0.f) {
                        const float mb = red[w * BN + c], qb = red[(WM + w) * BN + c];
                        const float tot = n + nb, d = mb - mean;
                        mean += d * (nb / tot);
                        m2 += qb + d * d * (n * nb / tot);
                        n = tot;
                    }
                }
                float* sp = p.SP + (long)(prow + mt) * 3 * p.N + n0 + c;
                sp[0] = n; sp[p.N] = mean; sp[2 * (long)p.N] = m2;
            }
        }
    }

    {
        // ---- LDS-staged epilogue, 16 bytes per lane --------------------------------------------------------------
        // The accumulator layout (lane = column) gives 4-byte global accesses, 256 B per wave instruction -- a quarter
        // of what the texture-address path moves per clock.  Each wave transposes its 32x32 blocks through a private
        // 32 x 36 float LDS patch and then touches global memory as 8 rows x 128 B per instruction.  With bf16 MFMAs the
        // epilogue is no longer hidden under other blocks' matrix work (56x56x64 layer 0.87 -> 0.65 ms); the fp32
        // kernels gain 1-3 %.
        float* patch = smem + BM + wid * (32 * 36);              // after the BM row offsets
        const int prw = lane >> 3, pc4 = lane & 7;
        f32x4 sa4[TN], sb4[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) { sa4[j] = f32x4{0.f, 0.f, 0.f, 0.f}; sb4[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        float* Y = nullptr;
        if constexpr (EPI == EPI_FWD) Y = p.Y + (long)split * p.slab;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (TN * 32) + j * 32 + 4 * pc4;
            f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, al4 = {1.f, 1.f, 1.f, 1.f};
            f32x4 mu4 = bias4, rs4 = bias4, sh4 = bias4;      // BNM: mean, rstd, shift of the BN layer below (its scale rides in al4)
            bool act = false;
            if constexpr (EPI == EPI_FWD) {
                if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + col);
                act = p.alpha != nullptr;
                if (act) al4 = *reinterpret_cast<const f32x4*>(p.alpha + col);
            } else {
                act = p.Zin != nullptr;
                if constexpr (CAP16) act = act || p.Zin16 != nullptr;
                if constexpr (BNM) {
                    if (p.bn_mu) {           // BN mode: the mask comes from z (or the stored output), never from PReLU
                        act = false;
                        mu4 = *reinterpret_cast<const f32x4*>(p.bn_mu + col);
                        rs4 = *reinterpret_cast<const f32x4*>(p.bn_rs + col);
                        if (p.bn_sc) { al4 = *reinterpret_cast<const f32x4*>(p.bn_sc + col); sh4 = *reinterpret_cast<const f32x4*>(p.bn_sh + col); }
                    }
                }
                if (act) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) al4[e] = p.alpha[(col + e) % p.amod];
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                // the global inputs of the four passes (shortcut / skip gradient / z) are fetched BEFORE the transpose: one
                // memory round trip per 32x32 block instead of one per pass
                int offs[4];
                f32x4 in0[4], in1[4];
                f32x4 in2[BNM ? 4 : 1];
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    offs[ps] = rowoff[wm * (TM * 32) + i * 32 + prw + 8 * ps];
                    const long o = (long)(offs[ps] < 0 ? 0 : offs[ps]) + col;
                    in0[ps] = f32x4{0.f, 0.f, 0.f, 0.f};
                    in1[ps] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if constexpr (BNM && EPI == EPI_DGRAD) {
                        in2[ps] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if constexpr (CAP16) { if (p.Zx16 && offs[ps] >= 0) in2[ps] = ld_bf4(p.Zx16 + o); }
                        if (p.Zx && offs[ps] >= 0) in2[ps] = *reinterpret_cast<const f32x4*>(p.Zx + o);
                    }
                    if constexpr (EPI == EPI_FWD) {
                        if constexpr (CAP16) { if (p.R16 && offs[ps] >= 0) in0[ps] = ld_bf4(p.R16 + o); }
                        if (p.R && offs[ps] >= 0) in0[ps] = *reinterpret_cast<const f32x4*>(p.R + o);
                    } else {
                        if constexpr (CAP16) {
                            if (p.ADD16 && offs[ps] >= 0) in0[ps] = ld_bf4(p.ADD16 + o);
                            if (p.Zin16 && offs[ps] >= 0) in1[ps] = ld_bf4(p.Zin16 + o);
                        }
                        if (p.ADD && offs[ps] >= 0) in0[ps] = *reinterpret_cast<const f32x4*>(p.ADD + o);
                        if (p.Zin && offs[ps] >= 0) in1[ps] = *reinterpret_cast<const f32x4*>(p.Zin + o);
                    }
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + li] = acc[i][j][r];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int rr = prw + 8 * ps;
                    const int off = offs[ps];
                    f32x4 v = *reinterpret_cast<const f32x4*>(patch + rr * 36 + 4 * pc4);
                    if (off < 0) continue;
                    const long o = (long)off + col;
                    if constexpr (EPI == EPI_FWD) {
                        v += bias4;
                        if (p.Z) *reinterpret_cast<f32x4*>(p.Z + o) = v;
                        if constexpr (CAP16) { if (p.Z16) st_bf4(p.Z16 + o, v); }
                        if (act) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : al4[e] * v[e];
                        }
                        v += in0[ps];
                        if constexpr (CAP16) { if (p.Y) *reinterpret_cast<f32x4*>(Y + o) = v; }
                        else *reinterpret_cast<f32x4*>(Y + o) = v;
                        if (p.Y16) st_bf4(p.Y16 + o, v);
                    } else {
                        v += in0[ps];
                        if (p.RAW) *reinterpret_cast<f32x4*>(p.RAW + o) = v;
                        if constexpr (CAP16) { if (p.RAW16) st_bf4(p.RAW16 + o, v); }
                        if (act) {
                            const f32x4 z = in1[ps];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                sa4[j][e] += v[e] * fminf(z[e], 0.f);
                                v[e] *= prelu_slope(z[e], al4[e]);
                                sb4[j][e] += v[e];
                            }
                        }
                        if constexpr (BNM) {
                            if (p.bn_mu) {
                                bool rnd = false, two = p.Zx != nullptr;
                                if constexpr (CAP16) { rnd = p.DZ16 != nullptr && p.DZ == nullptr; two = two || p.Zx16 != nullptr; }
                                const f32x4 z = in1[ps], zb = two ? in2[ps] : in1[ps];
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    float g = rnd ? rbf(v[e]) : v[e];
                                    const bool on = p.bn_sc ? __builtin_fmaf(z[e], al4[e], sh4[e]) > 0.f : (two ? z[e] > 0.f : true);
                                    g = on ? g : 0.f;
                                    v[e] = g;
                                    sb4[j][e] += g;
                                    sa4[j][e] += g * ((zb[e] - mu4[e]) * rs4[e]);
                                }
                            }
                        }
                        if constexpr (CAP16) { if (p.DZ) *reinterpret_cast<f32x4*>(p.DZ + o) = v; }
                        else *reinterpret_cast<f32x4*>(p.DZ + o) = v;
                        if (p.DZ16) st_bf4(p.DZ16 + o, v);
                    }
                }
            }
        }
        if constexpr (EPI == EPI_DGRAD) {
            if (p.PA) {    // per-block column partials (dalpha, dbias), reduced later in a fixed order
                __syncthreads();
                float* red = smem + BM + (WM * WN) * (32 * 36);      // [2][WM][BN], past the patches
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float a = sa4[j][e], b = sb4[j][e];
                        a += __shfl_xor(a, 8); b += __shfl_xor(b, 8);
                        a += __shfl_xor(a, 16); b += __shfl_xor(b, 16);
                        a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
                        if (prw == 0) {
                            const int c = wn * (TN * 32) + j * 32 + 4 * pc4 + e;
                            red[wm * BN + c] = a;
                            red[(WM + wm) * BN + c] = b;
                        }
                    }
                }
                __syncthreads();
                for (int c = tid; c < BN; c += NTH) {
                    float a = 0.f, b = 0.f;
#pragma unroll
                    for (int w = 0; w < WM; ++w) {
                        a += red[w * BN + c];
                        b += red[(WM + w) * BN + c];
                    }
                    const long o = (long)(prow + mt) * p.N + n0 + c;
                    p.PA[o] = a;
                    if (p.PB) p.PB[o] = b;
                }
            }
        }
    }
}

}  // namespace igemm_dev
