// igemm.hip -- fp32-MFMA gathered-GEMM kernel family for gfx950 (MI355X).
//
// One templated kernel computes  C[M,N] = sum_k A(m,k) * B(k,n)  where A and B
// are *gathered* operands, so that the same main loop serves
//   conv3x3 forward   : A = im2col rows of X   (k-contiguous), B = W [K,N]
//   conv3x3 dgrad     : A = im2col rows of dZ  (k-contiguous), B = W^T per tap (k-contiguous)
//   conv3x3 wgrad     : A = X^T (m-contiguous, k = pixel)    , B = dZ [K,N]
//   dense nn / nt / tn: the degenerate 1-tap, 1x1-image cases of the three above.
// It replaces the cuDNN/cuBLAS calls behind layers.conv2d / fully_connected and
// tf.gradients in the reference (nets/sphere.py:41-42,57-74; data_parallel.py:33).
//
// Design (CDNA4): 256 threads = 4 waves; each wave owns a (TM*32)x(TN*32) block
// of v_mfma_f32_32x32x2_f32 accumulators; BK = 32 floats per K-step; operand
// tiles are staged global -> VGPR -> LDS (double buffered, one barrier per
// K-step) so that HBM/L2 latency hides under the 64-cycle fp32 MFMAs.
// LDS images:
//   "MK" (k-contiguous operands): [row][32 floats] with the 16-byte chunk index
//        XOR-swizzled by (row>>1)&7 -> ds_read_b128 fragment reads are
//        conflict-free (16 lanes of a b128 group hit 16 distinct 16-B slots);
//   "KM" (m/n-contiguous operands): [k][BM floats], fragments by ds_read_b32.
// The reduction index inside a K-step is permuted (k = 8u + 4*half + t) so that
// one b128 read feeds four consecutive MFMAs; A and B use the same permutation.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "igemm.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 32;

__device__ __forceinline__ int swz(int row, int chunk) { return (chunk ^ ((row >> 1) & 7)) << 2; }

__device__ __forceinline__ float prelu_slope(float z, float a) {
    // d/dz [relu(z) + a*(z-|z|)/2]; TF's grad of relu(0) and sign(0) are 0 -> a/2 at exactly 0.
    return z > 0.f ? 1.f : (z == 0.f ? 0.5f * a : a);
}

template <int BM, int BN, int WM, int WN, int AL, int BL, int EPI>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const IgemmParams p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "4 waves");
    constexpr int A_CH = BM / 32;                 // 16-byte chunks each thread stages for A per K-step
    constexpr int B_CH = BN / 32;
    constexpr int STAGE = (BM + BN) * BK;         // floats per LDS stage
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int li = lane & 31, lh = lane >> 5;

    // ---- tile coordinates: XCD-aware bijective remap, n-tiles fastest --------
    const int ntn = p.N / BN;
    const int ntiles = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int mt = bid / ntn, nt_ = bid - mt * ntn;
    const int m0 = mt * BM, n0 = nt_ * BN;
    const int kbeg = blockIdx.y * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int nsteps = (kend - kbeg + BK - 1) / BK;

    // ---- per-thread loader state ------------------------------------------------
    // A, MK layout (im2col rows): chunk column = tid&7, rows (tid>>3) + 32*i
    int a_base[A_CH];
    int a_mask[A_CH];
    // A, KM layout (k = pixel): k-row = tid>>3, m-chunks (tid&7) + 8*i
    int a_c[A_CH];        // channel offset inside the source pixel
    int a_dhw[A_CH];      // packed (dh+8) | (dw+8)<<8 | valid<<16
    if constexpr (AL == AL_MK) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int m = m0 + (tid >> 3) + 32 * i;
            int base = 0, mask = 0;
            if (m < p.M) {
                const int hw = p.a_OH * p.a_OW;
                const int n = m / hw, rem = m - n * hw;
                const int oh = rem / p.a_OW, ow = rem - oh * p.a_OW;
                const int ih0 = oh * p.a_stride, iw0 = ow * p.a_stride;
                base = ((n * p.a_IH + ih0) * p.a_IW + iw0) * p.a_ld;
                for (int t = 0; t < p.a_NT; ++t) {
                    const int ih = ih0 + p.a_dh[t], iw = iw0 + p.a_dw[t];
                    if (ih >= 0 && ih < p.a_IH && iw >= 0 && iw < p.a_IW) mask |= 1 << t;
                }
            }
            a_base[i] = base;
            a_mask[i] = mask;
        }
    } else {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int m = m0 + 4 * ((tid & 7) + 8 * i);
            const int t = m / p.a_KC;
            a_c[i] = m - t * p.a_KC;
            const int tt = t < p.a_NT ? t : 0;
            a_dhw[i] = (p.a_dh[tt] + 8) | ((p.a_dw[tt] + 8) << 8) | ((m < p.M ? 1 : 0) << 16);
        }
    }

    f32x4 ra[A_CH], rb[B_CH];

    auto load_tiles = [&](int k0) {
        // ---------------- A ----------------
        if constexpr (AL == AL_MK) {
            const int tap = k0 / p.a_KC;
            const int kc0 = k0 - tap * p.a_KC;
            const int toff = (p.a_dh[tap] * p.a_IW + p.a_dw[tap]) * p.a_ld + kc0 + ((tid & 7) << 2);
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if ((a_mask[i] >> tap) & 1) v = *reinterpret_cast<const f32x4*>(p.A + (long)(a_base[i] + toff));
                ra[i] = v;
            }
        } else {
            const int pix = k0 + (tid >> 3);
            int n = 0, ih0 = 0, iw0 = 0;
            const bool kin = pix < kend;
            if (kin) {
                const int hw = p.a_OH * p.a_OW;
                n = pix / hw;
                const int rem = pix - n * hw;
                const int oh = rem / p.a_OW;
                ih0 = oh * p.a_stride;
                iw0 = (rem - oh * p.a_OW) * p.a_stride;
            }
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                const int ih = ih0 + (a_dhw[i] & 0xff) - 8, iw = iw0 + ((a_dhw[i] >> 8) & 0xff) - 8;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (kin && (a_dhw[i] >> 16) && ih >= 0 && ih < p.a_IH && iw >= 0 && iw < p.a_IW)
                    v = *reinterpret_cast<const f32x4*>(p.A + ((long)((n * p.a_IH + ih) * p.a_IW + iw) * p.a_ld + a_c[i]));
                ra[i] = v;
            }
        }
        // ---------------- B ----------------
        if constexpr (BL == BL_KN) {
            constexpr int CPR = BN / 4;               // chunks per k-row
            constexpr int RPP = 256 / CPR;            // k-rows per pass
#pragma unroll
            for (int i = 0; i < B_CH; ++i) {
                const int k = k0 + tid / CPR + RPP * i;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (k < kend) v = *reinterpret_cast<const f32x4*>(p.B + ((long)k * p.b_ld + n0 + ((tid % CPR) << 2)));
                rb[i] = v;
            }
        } else {
            const int tap = k0 / p.a_KC;
            const int kc0 = k0 - tap * p.a_KC;
            const long toff = (long)p.b_tapoff[tap] + kc0 + ((tid & 7) << 2);
#pragma unroll
            for (int i = 0; i < B_CH; ++i) {
                const int nn = n0 + (tid >> 3) + 32 * i;
                rb[i] = *reinterpret_cast<const f32x4*>(p.B + ((long)nn * p.b_ld + toff));
            }
        }
    };

    auto store_tiles = [&](int stage) {
        float* As = smem + stage * STAGE;
        float* Bs = As + BM * BK;
        if constexpr (AL == AL_MK) {
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                const int r = (tid >> 3) + 32 * i;
                *reinterpret_cast<f32x4*>(As + r * BK + swz(r, tid & 7)) = ra[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_CH; ++i)
                *reinterpret_cast<f32x4*>(As + (tid >> 3) * BM + (((tid & 7) + 8 * i) << 2)) = ra[i];
        }
        if constexpr (BL == BL_KN) {
            constexpr int CPR = BN / 4, RPP = 256 / CPR;
#pragma unroll
            for (int i = 0; i < B_CH; ++i)
                *reinterpret_cast<f32x4*>(Bs + (tid / CPR + RPP * i) * BN + ((tid % CPR) << 2)) = rb[i];
        } else {
#pragma unroll
            for (int i = 0; i < B_CH; ++i) {
                const int r = (tid >> 3) + 32 * i;
                *reinterpret_cast<f32x4*>(Bs + r * BK + swz(r, tid & 7)) = rb[i];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int stage) {
        const float* As = smem + stage * STAGE;
        const float* Bs = As + BM * BK;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * (TM * 32) + i * 32 + li;
                if constexpr (AL == AL_MK) {
                    fa[i] = *reinterpret_cast<const f32x4*>(As + row * BK + swz(row, 2 * u + lh));
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) fa[i][t] = As[(8 * u + 4 * lh + t) * BM + row];
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = wn * (TN * 32) + j * 32 + li;
                if constexpr (BL == BL_KN) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) fb[j][t] = Bs[(8 * u + 4 * lh + t) * BN + col];
                } else {
                    fb[j] = *reinterpret_cast<const f32x4*>(Bs + col * BK + swz(col, 2 * u + lh));
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][t], fb[j][t], acc[i][j], 0, 0, 0);
        }
    };

    // ---- main loop: one barrier per K-step, loads for step s+1 in flight under step s ----
    if (nsteps > 0) {
        load_tiles(kbeg);
        store_tiles(0);
        __syncthreads();
        for (int s = 0; s < nsteps; ++s) {
            const int cur = s & 1;
            const bool more = s + 1 < nsteps;
            if (more) load_tiles(kbeg + (s + 1) * BK);
            compute(cur);
            if (more) store_tiles(cur ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue -----------------------------------------------------------------
    int* rowoff = reinterpret_cast<int*>(smem);
    for (int r = tid; r < BM; r += 256) {
        const int m = m0 + r;
        int off = -1;
        if (m < p.M) {
            if (p.c_OH == 0) {
                off = m * p.c_ld;
            } else {
                const int hw = p.c_OH * p.c_OW;
                const int n = m / hw, rem = m - n * hw;
                const int oh = rem / p.c_OW, ow = rem - oh * p.c_OW;
                off = ((n * p.c_FH + oh * p.c_step + p.c_ph) * p.c_FW + ow * p.c_step + p.c_pw) * p.c_ld;
            }
        }
        rowoff[r] = off;
    }
    __syncthreads();

    if constexpr (EPI == EPI_FWD) {
        float* Y = p.Y + (long)blockIdx.y * p.slab;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (TN * 32) + j * 32 + li;
            const float bias = p.bias ? p.bias[col] : 0.f;
            const bool act = p.alpha != nullptr;
            const float al = act ? p.alpha[col] : 1.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int off = rowoff[rl];
                    if (off < 0) continue;
                    const long o = (long)off + col;
                    float v = acc[i][j][r] + bias;
                    if (p.Z) p.Z[o] = v;
                    if (act) v = v > 0.f ? v : al * v;
                    if (p.R) v += p.R[o];
                    Y[o] = v;
                }
            }
        }
    } else {   // EPI_DGRAD
        float sa[TN], sb[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            sa[j] = 0.f;
            sb[j] = 0.f;
            const int col = n0 + wn * (TN * 32) + j * 32 + li;
            const bool msk = p.Zin != nullptr;
            const float al = msk ? p.alpha[col % p.amod] : 1.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int off = rowoff[rl];
                    if (off < 0) continue;
                    const long o = (long)off + col;
                    float v = acc[i][j][r];
                    if (p.ADD) v += p.ADD[o];
                    if (p.RAW) p.RAW[o] = v;
                    if (msk) {
                        const float z = p.Zin[o];
                        sa[j] += v * fminf(z, 0.f);
                        v *= prelu_slope(z, al);
                        sb[j] += v;
                    }
                    p.DZ[o] = v;
                }
            }
        }
        if (p.PA) {        // per-block column partials (dalpha, dbias), reduced later in a fixed order
            __syncthreads();
            float* red = smem + 256;                      // [2][WM][BN], past rowoff
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                sa[j] += __shfl_xor(sa[j], 32);
                sb[j] += __shfl_xor(sb[j], 32);
                if (lh == 0) {
                    const int c = wn * (TN * 32) + j * 32 + li;
                    red[wm * BN + c] = sa[j];
                    red[(WM + wm) * BN + c] = sb[j];
                }
            }
            __syncthreads();
            for (int c = tid; c < BN; c += 256) {
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) {
                    a += red[w * BN + c];
                    b += red[(WM + w) * BN + c];
                }
                const long o = (long)(p.prow0 + mt) * p.N + n0 + c;
                p.PA[o] = a;
                if (p.PB) p.PB[o] = b;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int AL, int BL, int EPI>
hipError_t launch_cfg(const IgemmParams& p, int splits, hipStream_t st) {
    const int mt = (p.M + BM - 1) / BM, nt = p.N / BN;
    const size_t lds = 2 * (size_t)(BM + BN) * BK * sizeof(float);
    auto kern = igemm_kernel<BM, BN, WM, WN, AL, BL, EPI>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(mt * nt, splits), dim3(256), lds, st, p);
    return hipGetLastError();
}

template <int AL, int BL, int EPI>
hipError_t launch_tile(const IgemmParams& p, int tile, int splits, hipStream_t st) {
    switch (tile) {
        case TILE_128x128: return launch_cfg<128, 128, 2, 2, AL, BL, EPI>(p, splits, st);
        case TILE_256x64:  return launch_cfg<256, 64, 4, 1, AL, BL, EPI>(p, splits, st);
        case TILE_128x64:  return launch_cfg<128, 64, 2, 2, AL, BL, EPI>(p, splits, st);
        case TILE_64x64:   return launch_cfg<64, 64, 2, 2, AL, BL, EPI>(p, splits, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace

hipError_t igemm_launch(const IgemmParams& p, int al, int bl, int epi, int tile, int splits, hipStream_t st) {
    if (al == AL_MK && bl == BL_KN && epi == EPI_FWD) return launch_tile<AL_MK, BL_KN, EPI_FWD>(p, tile, splits, st);
    if (al == AL_MK && bl == BL_NK && epi == EPI_DGRAD) return launch_tile<AL_MK, BL_NK, EPI_DGRAD>(p, tile, splits, st);
    if (al == AL_MK && bl == BL_NK && epi == EPI_FWD) return launch_tile<AL_MK, BL_NK, EPI_FWD>(p, tile, splits, st);
    if (al == AL_KM && bl == BL_KN && epi == EPI_FWD) return launch_tile<AL_KM, BL_KN, EPI_FWD>(p, tile, splits, st);
    return hipErrorInvalidValue;
}

void igemm_tile_dims(int tile, int* bm, int* bn) {
    switch (tile) {
        case TILE_128x128: *bm = 128; *bn = 128; break;
        case TILE_256x64:  *bm = 256; *bn = 64; break;
        case TILE_128x64:  *bm = 128; *bn = 64; break;
        default:           *bm = 64;  *bn = 64; break;
    }
}
