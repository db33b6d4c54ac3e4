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
// Design (CDNA4): 256 threads = 4 waves; each wave owns a (TM*32)x(TN*32) block of
// v_mfma_f32_32x32x2_f32 accumulators; BK = 32 floats per K-step; operand tiles are staged
// global -> VGPR -> LDS by raw buffer loads (out-of-range lanes read 0: no branches) into ONE LDS
// stage (16 KiB for the 64x64 tile) with two barriers per K-step, so that up to 6 blocks share a CU:
// fp32 MFMAs are 64 cycles each and what these kernels live on is co-resident blocks hiding each
// other's prologue / epilogue / barrier stalls, not operand reuse (measurements: DESIGN.md 4.1).
// LDS images:
//   "MK" (k-contiguous operands): [row][32 floats] with the 16-byte chunk index
//        XOR-swizzled by (row>>1)&7 -> ds_read_b128 fragment reads are
//        conflict-free (16 lanes of a b128 group hit 16 distinct 16-B slots);
//   "KM" (m/n-contiguous operands): [k][BM floats], fragments by ds_read_b32.
// The reduction index inside a K-step is permuted (k = 8u + 4*half + t) so that
// one b128 read feeds four consecutive MFMAs; A and B use the same permutation.
// The K-step is hand-slotted: every MFMA is followed by at most one payload operation (a buffer
// load of the next tile, a fragment-read group, an LDS write) and a scheduling fence.
// Epilogue: every wave transposes its 32x32 accumulator blocks through a private LDS patch so that all global
// accesses of the epilogue are 16 bytes per lane (8 rows x 128 B per wave instruction).
// BF = 1 (fte_set_mfma_dtype(FTE_MFMA_BF16)): same gathers / epilogues / fp32 accumulators, operand tiles rounded to
// bf16 on the way into LDS ([row][32 bf16] images for BOTH operands, row-contiguous sources transposed in the write
// pass) and multiplied by v_mfma_f32_32x32x16_bf16.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <type_traits>
#include <vector>

#include "igemm.h"

#include "igemm_dev.h"

// LDS staging: 1 = ONE stage (32 KiB for the 128x128 tile) with a second barrier in the middle of the K-step,
// 3 blocks per CU; 0 = two stages (64 KiB), one barrier per K-step, 2 blocks per CU.  Measured on MI355X at batch
// 512: the extra co-resident block is worth more than the saved barrier (conv time per step 60.7 -> 56.3 ms).
#ifndef FTE_SINGLE
#define FTE_SINGLE 1
#endif
// wave priority outside the K loop (s_setprio; 0 = leave it alone): see the kernel's prologue
#ifndef FTE_PRIO_PROLOGUE
#define FTE_PRIO_PROLOGUE 3
#endif
#ifndef FTE_PRIO_EPILOGUE
#define FTE_PRIO_EPILOGUE 2
#endif
// K order of the fp32 forward / dgrad kernels: 1 = tap outer, channel chunk inner (no VALU addressing in the K loop)
#ifndef FTE_TAP_OUTER
#define FTE_TAP_OUTER 1
#endif
#ifndef FTE_SUPER_CHUNK
#define FTE_SUPER_CHUNK 128
#endif
// filter-gradient addressing: 1 = incremental pixel tracking with wave-uniform taps where the shape allows (see the K loop)
// 1: row-contiguous bf16 sources (the filter gradient's operands) are stored to LDS as loaded and read back with ds_read_b64_tr_b16;
// 0: packed into k-pairs on the way in (round 2's path; A/B variant through scripts/build_variant.sh)
#ifndef FTE_TR16
#define FTE_TR16 1
#endif
#ifndef FTE_KM_FAST
#define FTE_KM_FAST 1
#endif


#ifdef FTE_STAMP
// DIAGNOSTIC BUILD ONLY (scripts/build_stamp_variant.sh -> variants/libfte_stamp.so; never the product library): every block of
// the fp32 main loop stamps s_memtime (shader clock) and s_memrealtime (100 MHz) around its K loop into a buffer of its own,
// from which scripts/clock_probe.py derives the clock the chip holds under this kernel and the cycles per K-step
// (MI355X_MICROARCH.md, "DVFS give-back" item 6).  The stamps go to memory nothing else reads; no output depends on them.
__device__ unsigned long long* g_stamp_buf = nullptr;
extern "C" int fte_debug_set_stamp(void* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &buf, sizeof(buf));
}
#endif

namespace {

using namespace igemm_dev;

// BF = 1: the same gathers, epilogues and fp32 accumulators, but the operand tiles are rounded to bf16 (RNE) on their way
// into LDS and multiplied by v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate): storage stays fp32 in HBM, the kernel
// turns from MFMA-bound into staging / HBM-bound.  See the BF branch of the main loop.
// (the body is shared by two kernel symbols: igemm_kernel, and igemm_bn_kernel = the same with the BN-fusion epilogue compiled in)
// SK (stream-K, see below the kernels): the body computes K-steps [sk_k0, sk_k1) of tile sk_tile for worker sk_worker
template <int BM, int BN, int WM, int WN, int AL, int BL, int EPI, int BF, bool BNM, bool SK = false>
__device__ __forceinline__ void igemm_body(const IgemmParams& p, int sk_tile = 0, int sk_k0 = 0, int sk_k1 = 0, int sk_worker = 0) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    // NTH threads = WM x WN waves.  4 waves is the general case; 1 or 2 waves per block (fp32, k-contiguous A only) trade
    // operand reuse across waves for fewer waves coupled by each barrier: with ONE wave per block there is no barrier at all
    // (hipcc drops s_barrier for a workgroup of one wavefront) and the wave's LDS image is private.
    constexpr int NTH = 64 * WM * WN, RP = NTH / 8;   // RP: rows of a k-contiguous operand staged per pass
    static_assert((NTH == 256 || (BF == 0 && AL == AL_MK)) && TM >= 1 && TN >= 1, "1 / 2-wave blocks: fp32 forward / dgrad only");
    constexpr int A_CH = BM * 8 / NTH;            // 16-byte chunks each thread stages for A per K-step
    constexpr int B_CH = BN * 8 / NTH;
    constexpr int STAGE = (BM + BN) * BK;         // floats per LDS stage
    extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef FTE_STAMP
    const unsigned long long st_entry = g_stamp_buf ? __builtin_amdgcn_s_memtime() : 0ull;
#endif

    int tid_ = threadIdx.x;
    // stream-K: the body sits in a loop over the worker's parts.  Everything derived from the thread id is loop-invariant there and
    // the compiler would keep it all in registers across the parts (+30 VGPRs: 6 -> 4 blocks per CU on the 64x64 tile); an opaque
    // copy of the id per part makes it recompute them (a few dozen instructions per part).
    if constexpr (SK) asm volatile("" : "+v"(tid_));
    const int tid = tid_;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int li = lane & 31, lh = lane >> 5;
    // The prologue (index decomposition, tap masks, first tile) is a few hundred VALU / SALU instructions.  A new block's waves
    // are the YOUNGEST on their SIMDs and lose every issue arbitration to the older blocks' MFMA streams: measured with in-kernel
    // stamps at batch 512, a block spent 23 us (56x56x64) to 69 us (7x7x512) between entry and its first K-step -- as long as a
    // quarter of its K loop -- with only 2.5 - 4 of the 6 resident blocks inside their loops.  Raised priority gets the block to
    // its MFMAs at once; it drops back to 0 for the loop (FTE_PRIO tuning hook: 0 = off).
    if (FTE_PRIO_PROLOGUE) __builtin_amdgcn_s_setprio(FTE_PRIO_PROLOGUE);

    // ---- tile coordinates: XCD-aware bijective remap, n-tiles fastest --------
    const int ntn = p.N / BN;
    int bid = blockIdx.x, split = blockIdx.y;
    if constexpr (SK) {
        bid = sk_tile; split = 0;             // tiles in natural order (n-tiles fastest); the WORKER ids carry the XCD remap
    } else if (p.split_major > 0) {
        // Split-K filter gradients: every tile of one K range (pixel range) reads the SAME x and dz rows.  Blocks are dealt
        // round-robin over the 8 XCDs, each with its own L2: ids are laid out in groups of 8 splits x all tiles, the split
        // fastest, so that the tiles of one split are 8 ids apart -- same XCD, dispatched together -- and x / dz leave HBM
        // once per split instead of once per tile (placement is a speed / traffic matter only, never correctness).
        const int S = p.split_major, T = gridDim.x / S;
        const int g = bid / (8 * T), rem = bid - g * (8 * T);
        const int w = min(8, S - 8 * g);                      // splits in this group (the last group may be narrower)
        const int tile = rem / w;
        split = 8 * g + (rem - tile * w);
        bid = tile;
    } else {
        const int ntiles = gridDim.x;
        const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    // stride-2 dgrad: the (up to) four output-parity classes share one launch.  Class c owns taps
    // [cls_tap0[c], cls_tap0[c+1]) of the concatenated tap tables and its own K = taps * KC.
    int tap0 = 0, NT = p.a_NT, Kc = p.K, c_ph = p.c_ph, c_pw = p.c_pw, prow = p.prow0;
    if constexpr (EPI == EPI_DGRAD) {
        if (p.ncls > 1) {
            // class = fastest index of the remapped id: the XCD remap hands each XCD a CONTIGUOUS range of ids, so
            // "class c owns ids [c*T, (c+1)*T)" put the whole 4-tap class on two XCDs and the 1-tap class on two others
            // (merged launch 25 % slower than four separate ones); interleaved, every XCD gets 1/8 of every class
            // ... and ROTATED by the group index: blocks go to an XCD's 32 CUs round-robin, so with class = id % 4 alone a CU received
            // tiles of ONE class only (the CUs of the 4-tap class 4x the K-steps of the 1-tap class's: 28x28x128 <- 256 at 64 images ran
            // 127 us for 46 us of MFMA work); rotating the class by (group / 8) deals every CU the classes in turn
            const int grp = bid / p.ncls;
            const int cls = (bid - grp * p.ncls + ((grp >> p.cls_rot) & 3)) % p.ncls;
            bid = grp;
            tap0 = p.cls_tap0[cls];
            NT = p.cls_tap0[cls + 1] - tap0;
            Kc = NT * p.a_KC;
            c_ph = p.cls_ph[cls]; c_pw = p.cls_pw[cls];
            prow += cls * p.cls_mtiles;
        }
    }
    const int mt = bid / ntn, nt_ = bid - mt * ntn;
    const int m0 = p.m_base + mt * BM, n0 = nt_ * BN;
    const int kbeg = SK ? sk_k0 * BK : split * p.kchunk;
    const int kend = SK ? sk_k1 * BK : min(Kc, kbeg + p.kchunk);
    const int nsteps = (kend - kbeg + BK - 1) / BK;
    // index decompositions use fdiv(): every decomposed index (GEMM row m < M, pixel < K) is below 2^24 for the tensors
    // the 2-GiB buffer range admits with >= 32 channels; the launcher checks it (igemm_launch)
    const int a_hw = p.a_OH * p.a_OW;
    const float r_ahw = 1.f / (float)a_hw, r_aow = 1.f / (float)p.a_OW;

    // ---- per-thread loader state ------------------------------------------------
    // Operand tiles are fetched with raw buffer loads: a lane whose element is padding / out of
    // range gets the byte offset OOB (>= num_records), for which the hardware returns 0.  No
    // branches, no selects -- the whole K-step is one basic block and hipcc is free to spread the
    // address arithmetic, the loads and the LDS writes between the 64-cycle MFMAs.
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, p.b_bytes, 0x00020000);
    // A, MK layout (im2col rows): chunk column = tid&7, rows (tid>>3) + 32*i
    unsigned a_base[A_CH];   // byte offset of the row's (tap 0,0) source pixel + this thread's chunk
    int a_mask[A_CH];        // bit t: tap t is inside the image for this row
    // A, KM layout (k = pixel): k-row = tid>>3, m-chunks (tid&7) + 8*i
    int a_c[A_CH];           // byte offset inside the source pixel
    int a_dhw[A_CH];         // packed (dh+8) | (dw+8)<<8 | valid<<16
    if constexpr (AL == AL_MK) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int m = m0 + (tid >> 3) + RP * i;
            int base = 0, mask = 0;
            if (m < p.M) {
                const int n = fdiv(m, a_hw, r_ahw), rem = m - n * a_hw;
                const int oh = fdiv(rem, p.a_OW, r_aow), ow = rem - oh * p.a_OW;
                const int ih0 = oh * p.a_stride, iw0 = ow * p.a_stride;
                base = ((n * p.a_IH + ih0) * p.a_IW + iw0) * p.a_ld;
                for (int t = 0; t < NT; ++t) {
                    const int ih = ih0 + p.a_dh[tap0 + t], iw = iw0 + p.a_dw[tap0 + t];
                    if (ih >= 0 && ih < p.a_IH && iw >= 0 && iw < p.a_IW) mask |= 1 << t;
                }
            }
            a_base[i] = (unsigned)(base + ((tid & 7) << 2)) * 4u;
            a_mask[i] = mask;
        }
    } else {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            // fp32: k-row = tid>>3, m-chunks (tid&7) + 8i.  bf16: k-PAIR = tid>>4, m-chunks (tid&15) + 16i (i < A_CH/2),
            // so that a thread holds rows k and k+1 of the same 4 m and packs them into 32-bit LDS words.
            const int m = BF ? m0 + 4 * ((tid & 15) + 16 * i) : m0 + 4 * ((tid & 7) + 8 * i);
            const int t = m / p.a_KC;
            a_c[i] = (m - t * p.a_KC) * 4;
            const int tt = t < p.a_NT ? t : 0;
            a_dhw[i] = (p.a_dh[tt] + 8) | ((p.a_dw[tt] + 8) << 8) | ((m < p.M ? 1 : 0) << 16);
        }
    }
    // B: per-thread constant part of the byte offset
    unsigned b_base[B_CH];
    if constexpr (BL == BL_KN) {
        constexpr int CPR = BN / 4, RPP = NTH / CPR;
#pragma unroll
        for (int i = 0; i < B_CH; ++i) {
            if constexpr (BF)      // entry 2j + e: row 2*(tid>>4) + e, n-chunk (tid&15) + 16j
                b_base[i] = (unsigned)((2 * (tid >> 4) + (i & 1)) * p.b_ld + n0 + (((tid & 15) + 16 * (i >> 1)) << 2)) * 4u;
            else
                b_base[i] = (unsigned)((tid / CPR + RPP * i) * p.b_ld + n0 + ((tid % CPR) << 2)) * 4u;
        }
    } else {
#pragma unroll
        for (int i = 0; i < B_CH; ++i)
            b_base[i] = (unsigned)((n0 + (tid >> 3) + RP * i) * p.b_ld + ((tid & 7) << 2)) * 4u;
    }

    // Reduction order for tap-structured operands: K-step sigma = k0/32 covers channel chunk
    // sigma / NT of tap sigma % NT (chunk outer, tap inner).  The NT taps of one 32-channel chunk
    // touch ~1.3x one window of 128-B row pieces (~21 KB for a 128-row tile): they hit in the CU's L1
    // instead of streaming the whole Cin-wide window once per tap from L2 / Infinity Cache.
    // Weight / B rows are addressed accordingly (row tap*KC + chunk*32); sums are order-independent.
    // fp32, k-contiguous A (TAPO): K-step sigma covers tap sigma / (KC/32), channel chunk sigma % (KC/32) -- TAP OUTER.  Inside a
    // tap the per-thread source offsets do not change (the wave-uniform channel offset rides in the buffer load's scalar offset),
    // so the K loop issues NO vector ALU instruction for addressing: every VALU instruction between two dependent MFMAs holds
    // the SIMD's issue port and idles the matrix pipe ~3 cycles (scripts/probes/mfma_f32_waves.hip: 2 VALU per MFMA -> 90 % of the
    // bare-loop rate, 6 -> 79 %, at every occupancy); the chunk-outer order needed ~1.5 per MFMA for the tap mask / offset selects.
    constexpr bool TAPO = FTE_TAP_OUTER && BF == 0 && AL == AL_MK;
    // ... in SUPER-CHUNKS of SC channels: (super-chunk outer, tap, 32-channel step inner).  With the taps of ALL channels outermost a
    // block re-reads its rows at a distance of rows x KC x 4 bytes (the co-resident blocks of an XCD then cycle 12 MB through a
    // 4 MB L2: 897 MB per launch left L2 on the 64x64 forward symbol against 530 MB algorithmic); SC = 128 channels keeps the nine
    // taps of a super-chunk within L2 and costs one address recomputation (~10 VALU) per 4 K-steps.
    const int SC = TAPO ? ((p.a_KC % FTE_SUPER_CHUNK == 0) ? FTE_SUPER_CHUNK : p.a_KC) : BK;
    const int SCS = SC / BK;                                    // K-steps per (super-chunk, tap) visit
    auto ktap = [&](int k0) -> int {
        if constexpr (TAPO) return ((k0 / BK) % (NT * SCS)) / SCS;
        const int sg = k0 >> 5;
        int t = sg % NT;
        return t;
    };
    auto kchan = [&](int k0) -> int {
        if constexpr (TAPO) return ((k0 / BK) / (NT * SCS)) * SC + ((k0 / BK) % SCS) * BK;
        return ((k0 >> 5) / NT) << 5;
    };

    f32x4 ra0[A_CH], rb0[B_CH];

    auto ldg = [&](const __amdgpu_buffer_rsrc_t& r, unsigned voff, unsigned soff) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    };

    // (tap, kc0) = (ktap(k0), kchan(k0)): the bf16 loop keeps them incrementally -- the two integer divisions cost ~40
    // scalar instructions per K-step, invisible next to sixteen 64-cycle fp32 MFMAs but not next to two 32-cycle bf16 ones
    auto load_tiles = [&](int k0, int tap, int kc0, f32x4 (&ra)[A_CH], f32x4 (&rb)[B_CH]) {
        // ---------------- A ----------------
        if constexpr (AL == AL_MK) {
            const unsigned toff = (unsigned)((p.a_dh[tap0 + tap] * p.a_IW + p.a_dw[tap0 + tap]) * p.a_ld + kc0) * 4u;   // wave-uniform
#pragma unroll
            for (int i = 0; i < A_CH; ++i)
                ra[i] = ldg(rsrcA, ((a_mask[i] >> tap) & 1) ? a_base[i] + toff : OOB, 0);
        } else if constexpr (BF) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {                  // ra[2j + e]: pixel k0 + 2*(tid>>4) + e, m-chunk j
                const int pix = k0 + 2 * (tid >> 4) + e;
                const bool kin = pix < kend;
                const int n = fdiv(pix, a_hw, r_ahw);
                const int rem = pix - n * a_hw;
                const int oh = fdiv(rem, p.a_OW, r_aow);
                const int ih0 = oh * p.a_stride;
                const int iw0 = (rem - oh * p.a_OW) * p.a_stride;
#pragma unroll
                for (int j = 0; j < A_CH / 2; ++j) {
                    const int ih = ih0 + (a_dhw[j] & 0xff) - 8, iw = iw0 + ((a_dhw[j] >> 8) & 0xff) - 8;
                    const bool ok = kin && (a_dhw[j] >> 16) && ih >= 0 && ih < p.a_IH && iw >= 0 && iw < p.a_IW;
                    ra[2 * j + e] = ldg(rsrcA, ok ? (unsigned)(((n * p.a_IH + ih) * p.a_IW + iw) * p.a_ld) * 4u + a_c[j] : OOB, 0);
                }
            }
        } else {
            const int pix = k0 + (tid >> 3);
            const bool kin = pix < kend;
            const int n = fdiv(pix, a_hw, r_ahw);
            const int rem = pix - n * a_hw;
            const int oh = fdiv(rem, p.a_OW, r_aow);
            const int ih0 = oh * p.a_stride;
            const int iw0 = (rem - oh * p.a_OW) * p.a_stride;
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                const int ih = ih0 + (a_dhw[i] & 0xff) - 8, iw = iw0 + ((a_dhw[i] >> 8) & 0xff) - 8;
                const bool ok = kin && (a_dhw[i] >> 16) && ih >= 0 && ih < p.a_IH && iw >= 0 && iw < p.a_IW;
                ra[i] = ldg(rsrcA, ok ? (unsigned)(((n * p.a_IH + ih) * p.a_IW + iw) * p.a_ld) * 4u + a_c[i] : OOB, 0);
            }
        }
        // ---------------- B ----------------
        if constexpr (BL == BL_KN) {
            constexpr int CPR = BN / 4, RPP = NTH / CPR;
            const unsigned koff = (unsigned)((AL == AL_MK ? tap * p.a_KC + kc0 : k0) * p.b_ld) * 4u;   // wave-uniform
#pragma unroll
            for (int i = 0; i < B_CH; ++i) {
                if constexpr (BF) rb[i] = ldg(rsrcB, (k0 + 2 * (tid >> 4) + (i & 1) < kend) ? b_base[i] : OOB, koff);
                else rb[i] = ldg(rsrcB, (k0 + tid / CPR + RPP * i < kend) ? b_base[i] : OOB, koff);
            }
        } else {
            const unsigned toff = (unsigned)(p.b_tapoff[tap0 + tap] + kc0) * 4u;                        // wave-uniform
#pragma unroll
            for (int i = 0; i < B_CH; ++i) rb[i] = ldg(rsrcB, b_base[i], toff);
        }
    };

    f32x4 (&ra)[A_CH] = ra0;
    f32x4 (&rb)[B_CH] = rb0;
    auto store_tiles = [&](int stage) {
        float* As = smem + stage * STAGE;
        float* Bs = As + BM * BK;
        if constexpr (AL == AL_MK) {
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                const int r = (tid >> 3) + RP * i;
                *reinterpret_cast<f32x4*>(As + r * BK + swz(r, tid & 7)) = ra[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_CH; ++i)
                *reinterpret_cast<f32x4*>(As + (tid >> 3) * BM + (((tid & 7) + 8 * i) << 2)) = ra[i];
        }
        if constexpr (BL == BL_KN) {
            constexpr int CPR = BN / 4, RPP = NTH / CPR;
#pragma unroll
            for (int i = 0; i < B_CH; ++i)
                *reinterpret_cast<f32x4*>(Bs + (tid / CPR + RPP * i) * BN + ((tid % CPR) << 2)) = rb[i];
        } else {
#pragma unroll
            for (int i = 0; i < B_CH; ++i) {
                const int r = (tid >> 3) + RP * i;
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

    // Fragment reads are software-pipelined one u-step (8 k) ahead of the MFMAs that consume
    // them: left to itself hipcc sinks each ds_read next to its consumer and every group of four
    // 64-cycle MFMAs then starts with an exposed LDS round trip.
    auto read_frags = [&](const float* As, const float* Bs, int u, f32x4 (&fa)[TM], f32x4 (&fb)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = wm * (TM * 32) + i * 32 + li;
            if constexpr (AL == AL_MK) {
                fa[i] = *reinterpret_cast<const f32x4*>(As + row * BK + swz(row, 2 * u + lh));
            } else {
                // (integer addresses of the LDS address space: lane part + stage once, the k index in the read's immediate offset --
                // through `As[...]` on the generic pointer every read pair paid a vector add: 29 of the 73 vector instructions beside
                // the 64 MFMAs of a filter-gradient K-step, and each costs the matrix pipe issue cycles)
                const unsigned ab = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) float*)As + (unsigned)((4 * lh * BM + row) * 4);
#pragma unroll
                for (int t = 0; t < 4; ++t) fa[i][t] = *(const __attribute__((address_space(3))) float*)(ab + (unsigned)((8 * u + t) * BM * 4));
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = wn * (TN * 32) + j * 32 + li;
            if constexpr (BL == BL_KN) {
                const unsigned bb = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) float*)Bs + (unsigned)((4 * lh * BN + col) * 4);
#pragma unroll
                for (int t = 0; t < 4; ++t) fb[j][t] = *(const __attribute__((address_space(3))) float*)(bb + (unsigned)((8 * u + t) * BN * 4));
            } else {
                fb[j] = *reinterpret_cast<const f32x4*>(Bs + col * BK + swz(col, 2 * u + lh));
            }
        }
    };
    // ---- bf16-operand main loop ------------------------------------------------------------------
    // LDS image of BOTH operands: [row][32 bf16] (64-byte rows), 16-byte chunk c stored at c ^ ((row>>2)&3): the
    // ds_read_b128 fragment reads (lane -> row li, chunk 2s + lh = k 16s + 8lh .. +7, exactly the 32x32x16 operand
    // map) are conflict-free.  k-contiguous sources (MK / NK) convert 4 floats and write 8 bytes; row-contiguous
    // sources (KM / KN) hold rows k, k+1 of 4 consecutive m (n) and write four 32-bit words -- the transpose
    // happens in the write pass.  One LDS image per macro step (two barriers); the next macro step's buffer loads are in
    // flight under the MFMAs.  No k permutation.
    if constexpr (BF != 0) {
        if (FTE_PRIO_PROLOGUE) __builtin_amdgcn_s_setprio(0);      // the bf16 loops keep the default priority throughout
    }
    if constexpr (BF == 2) {
        // ---- bf16 SOURCES (fte_*16 entry points): the operands already live in HBM as bf16 copies -- activations written
        // by the producing epilogue (Y16 / DZ16), weights packed once per step -- so a K-step moves half the bytes through
        // the texture-address path, needs no conversion and writes 16 bytes per lane into the same LDS image as BF = 1:
        //   k-contiguous sources (MK / NK): lane = (row = tid>>2 + 64 i, 16-byte chunk = tid&3 = 8 k values) -> ds_write_b128
        //   row-contiguous sources (KM / KN): lane = (k pair = tid>>4, 8 consecutive m / n) loads rows k and k+1 and writes
        //   eight packed 32-bit words (v_perm_b32) -- the transpose happens in the write pass, as in BF = 1.
        // Two LDS images, one barrier per K-step.
        static_assert(!(AL == AL_MK && BL == BL_KN), "forward with bf16 sources takes the transposed weight pack (NK)");
        char* lds = reinterpret_cast<char*>(smem);
        auto off16 = [](int row, int chunk) -> int { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); };
        constexpr int SUB_B = (BM + BN) * 64;
        constexpr int AR = (BM + 63) / 64, BR = (BN + 63) / 64;          // 16-byte loads per thread, k-contiguous sources
        constexpr int AP = (BM + 127) / 128, BP = (BN + 127) / 128;      // k-pair loads per thread, row-contiguous sources
        const int r4 = tid >> 2, c4 = tid & 3, kp = tid >> 4, c16 = tid & 15;
        auto ldg16 = [&](const __amdgpu_buffer_rsrc_t& r, unsigned voff, unsigned soff) -> u32x4 {
            return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
        };
        unsigned a16_base[AR]; int a16_mask[AR];        // MK
        int a16_c[AP], a16_dhw[AP];                     // KM
        if constexpr (AL == AL_MK) {
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const int m = m0 + r4 + 64 * i;
                int base = 0, mask = 0;
                if (m < p.M && r4 + 64 * i < BM) {
                    const int n = fdiv(m, a_hw, r_ahw), rem = m - n * a_hw;
                    const int oh = fdiv(rem, p.a_OW, r_aow), ow = rem - oh * p.a_OW;
                    const int ih0 = oh * p.a_stride, iw0 = ow * p.a_stride;
                    base = ((n * p.a_IH + ih0) * p.a_IW + iw0) * p.a_ld;
                    for (int t = 0; t < NT; ++t) {
                        const int ih = ih0 + p.a_dh[tap0 + t], iw = iw0 + p.a_dw[tap0 + t];
                        if (ih >= 0 && ih < p.a_IH && iw >= 0 && iw < p.a_IW) mask |= 1 << t;
                    }
                }
                a16_base[i] = (unsigned)(base + (c4 << 3)) * 2u;
                a16_mask[i] = mask;
            }
        } else {
#pragma unroll
            for (int j = 0; j < AP; ++j) {
                const int ml = 8 * (c16 + 16 * j);
                const int m = m0 + ml;
                const int t = m / p.a_KC;
                a16_c[j] = (m - t * p.a_KC) * 2;
                const int tt = t < p.a_NT ? t : 0;
                a16_dhw[j] = (p.a_dh[tt] + 8) | ((p.a_dw[tt] + 8) << 8) | ((m < p.M && ml < BM ? 1 : 0) << 16);
            }
        }
        unsigned b16_base[BL == BL_NK ? BR : 2 * BP];
        if constexpr (BL == BL_NK) {
#pragma unroll
            for (int i = 0; i < BR; ++i)
                b16_base[i] = (r4 + 64 * i < BN) ? (unsigned)((n0 + r4 + 64 * i) * p.b_ld + (c4 << 3)) * 2u : OOB;
        } else {
#pragma unroll
            for (int j = 0; j < BP; ++j)
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    b16_base[2 * j + e] = (8 * (c16 + 16 * j) < BN) ? (unsigned)((2 * kp + e) * p.b_ld + n0 + 8 * (c16 + 16 * j)) * 2u : OOB;
        }
        u32x4 ga[AL == AL_MK ? AR : 2 * AP], gb[BL == BL_NK ? BR : 2 * BP];
        int ptap = 0, pkc = 0;
        if constexpr (AL == AL_MK || BL == BL_NK) { ptap = ktap(kbeg); pkc = kchan(kbeg); }
        auto load16 = [&](int k0) {
            if constexpr (AL == AL_MK) {
                const unsigned toff = (unsigned)((p.a_dh[tap0 + ptap] * p.a_IW + p.a_dw[tap0 + ptap]) * p.a_ld + pkc) * 2u;
#pragma unroll
                for (int i = 0; i < AR; ++i) ga[i] = ldg16(rsrcA, ((a16_mask[i] >> ptap) & 1) ? a16_base[i] + toff : OOB, 0);
            } else {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int pix = k0 + 2 * kp + e;
                    const bool kin = pix < kend;
                    const int n = fdiv(pix, a_hw, r_ahw);
                    const int rem = pix - n * a_hw;
                    const int oh = fdiv(rem, p.a_OW, r_aow);
                    const int ih0 = oh * p.a_stride;
                    const int iw0 = (rem - oh * p.a_OW) * p.a_stride;
#pragma unroll
                    for (int j = 0; j < AP; ++j) {
                        const int ih = ih0 + (a16_dhw[j] & 0xff) - 8, iw = iw0 + ((a16_dhw[j] >> 8) & 0xff) - 8;
                        const bool ok = kin && (a16_dhw[j] >> 16) && ih >= 0 && ih < p.a_IH && iw >= 0 && iw < p.a_IW;
                        ga[2 * j + e] = ldg16(rsrcA, ok ? (unsigned)(((n * p.a_IH + ih) * p.a_IW + iw) * p.a_ld) * 2u + a16_c[j] : OOB, 0);
                    }
                }
            }
            if constexpr (BL == BL_NK) {
                const unsigned toff = (unsigned)(p.b_tapoff[tap0 + ptap] + pkc) * 2u;
#pragma unroll
                for (int i = 0; i < BR; ++i) gb[i] = ldg16(rsrcB, b16_base[i], toff);
            } else {
                const unsigned koff = (unsigned)(k0 * p.b_ld) * 2u;
#pragma unroll
                for (int i = 0; i < 2 * BP; ++i) gb[i] = ldg16(rsrcB, (k0 + 2 * kp + (i & 1) < kend) ? b16_base[i] : OOB, koff);
            }
            if constexpr (AL == AL_MK || BL == BL_NK) {
                if (++ptap == NT) { ptap = 0; pkc += BK; }
            }
        };
        // Row-contiguous sources (KM / KN: the filter gradient's x and dz, k = pixel), FTE_TR16: the rows go to LDS AS THEY ARE --
        // image [32 k][rows] bf16, one ds_write_b128 per loaded row piece (16-byte chunk c of k-row k at slot c ^ 2 (k & 3)) -- and the
        // 32x32x16 operand (8 consecutive k of one m per lane) comes back through ds_read_b64_tr_b16, gfx950's transposing LDS read:
        // of each 16 lanes, lane 4 q + p addresses k-row q, columns 4 p .. 4 p + 3, and lane i receives column i of the four rows
        // (scripts/probes/ds_read_tr16.hip; the grouped filter gradient of layers.hip reads its fragments the same way).  Two reads
        // (k-rows 0-3, 4-7 of the lane's k-half) make one operand.  Replaces eight v_perm_b32 + eight ds_write_b32 per loaded pair of
        // rows; the values every lane feeds its MFMAs are the same, so results are bit-identical to the packed-pair image.
        typedef short s16x4t __attribute__((ext_vector_type(4)));
        auto offk = [](int k, int chunk, int rowbytes) -> int { return k * rowbytes + ((chunk ^ ((k & 3) << 1)) << 4); };
        auto store_rows = [&](char* T, int rows, const u32x4& v0, const u32x4& v1, int j) {
            const int chunk = c16 + 16 * j;
            if (8 * chunk < rows) {
                *reinterpret_cast<u32x4*>(T + offk(2 * kp, chunk, rows * 2)) = v0;
                *reinterpret_cast<u32x4*>(T + offk(2 * kp + 1, chunk, rows * 2)) = v1;
            }
        };
        auto frag_tr = [&](const char* T, int rows, int row0, int h) -> bf16x8 {
            const int g4 = lane >> 4, idx = lane & 15;
            const int k = 16 * h + 8 * (g4 >> 1) + (idx >> 2);
            const int m = row0 + 16 * (g4 & 1) + 4 * (idx & 3);
            const char* q = T + offk(k, m >> 3, rows * 2) + (m & 7) * 2;
            const s16x4t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4t*)(q));
            const s16x4t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4t*)(q + 4 * rows * 2));
            return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
        };
        auto store_pairs = [&](char* T, int rows, const u32x4& v0, const u32x4& v1, int j) {
            if constexpr (FTE_TR16) { store_rows(T, rows, v0, v1, j); return; }
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int r = 8 * (c16 + 16 * j) + 2 * w;
                if (r < rows) {
                    *reinterpret_cast<unsigned*>(T + off16(r, kp >> 2) + (kp & 3) * 4) = __builtin_amdgcn_perm(v1[w], v0[w], 0x05040100u);
                    *reinterpret_cast<unsigned*>(T + off16(r + 1, kp >> 2) + (kp & 3) * 4) = __builtin_amdgcn_perm(v1[w], v0[w], 0x07060302u);
                }
            }
        };
        auto store16 = [&](int stage) {
            char* As = lds + stage * SUB_B;
            char* Bs = As + BM * 64;
            if constexpr (AL == AL_MK) {
#pragma unroll
                for (int i = 0; i < AR; ++i)
                    if (r4 + 64 * i < BM) *reinterpret_cast<u32x4*>(As + off16(r4 + 64 * i, c4)) = ga[i];
            } else {
#pragma unroll
                for (int j = 0; j < AP; ++j) store_pairs(As, BM, ga[2 * j], ga[2 * j + 1], j);
            }
            if constexpr (BL == BL_NK) {
#pragma unroll
                for (int i = 0; i < BR; ++i)
                    if (r4 + 64 * i < BN) *reinterpret_cast<u32x4*>(Bs + off16(r4 + 64 * i, c4)) = gb[i];
            } else {
#pragma unroll
                for (int j = 0; j < BP; ++j) store_pairs(Bs, BN, gb[2 * j], gb[2 * j + 1], j);
            }
        };
        if (nsteps > 0) load16(kbeg);
        for (int st = 0; st < nsteps; ++st) {
            store16(st & 1);
            __syncthreads();
            if (st + 1 < nsteps) load16(kbeg + (st + 1) * BK);
            const char* As = lds + (st & 1) * SUB_B;
            const char* Bs = As + BM * 64;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                bf16x8 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if constexpr (AL == AL_KM && FTE_TR16) fa[i] = frag_tr(As, BM, wm * (TM * 32) + i * 32, h);
                    else fa[i] = *reinterpret_cast<const bf16x8*>(As + off16(wm * (TM * 32) + i * 32 + li, 2 * h + lh));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (BL == BL_KN && FTE_TR16) fb[j] = frag_tr(Bs, BN, wn * (TN * 32) + j * 32, h);
                    else fb[j] = *reinterpret_cast<const bf16x8*>(Bs + off16(wn * (TN * 32) + j * 32 + li, 2 * h + lh));
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();           // the epilogue reuses the LDS
    } else if constexpr (BF) {
        char* lds = reinterpret_cast<char*>(smem);
        auto off16 = [](int row, int chunk) -> int { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); };
        auto pk = [](float a, float b) -> unsigned {
            return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
        };
        // KS sub-steps of 32 k form one macro step: all of its buffer loads are issued together and stay in flight
        // under the previous macro step's MFMAs.  PMC on the 14x14x256 forward layer: 17 VALU + 20 SALU + 3.6 LDS + 2
        // VMEM instructions per MFMA, MFMA pipe 14 % busy -- the loop is instruction-issue bound, not latency bound
        // (KS = 1 / 2 / 4 run within 1 % of each other on the 64x64 tile).
        // 64x64 tiles: KS = 4 sub-steps per macro step in ONE LDS image (two barriers per macro step).  Bigger tiles:
        // KS = 1 with TWO images and one barrier per step -- with KS = 2 the 128x128 wgrad kernel spills and ran the
        // 14x14x256 filter gradient at 0.53 ms instead of 0.31.
        constexpr int KS = (BM + BN) <= 128 ? 4 : 1;
        constexpr int NST = KS == 1 ? 2 : 1;
        constexpr int SUB_B = (BM + BN) * 64;                          // bytes per sub-step image
        f32x4 qa[KS][A_CH], qb[KS][B_CH];
        int ptap = 0, pkc = 0;                                         // (tap, channel chunk) of the next sub-step to load
        if constexpr (AL == AL_MK || BL == BL_NK) { ptap = ktap(kbeg); pkc = kchan(kbeg); }
        auto load_macro = [&](int k0) {
#pragma unroll
            for (int u = 0; u < KS; ++u) {
                if (k0 + u * BK < kend) {
                    load_tiles(k0 + u * BK, ptap, pkc, qa[u], qb[u]);
                    if constexpr (AL == AL_MK || BL == BL_NK) {
                        if (++ptap == NT) { ptap = 0; pkc += BK; }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < A_CH; ++i) qa[u][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < B_CH; ++i) qb[u][i] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        };
        auto store_bf = [&](int u, int stage) {
            char* As = lds + (stage * KS + u) * SUB_B;
            char* Bs = As + BM * 64;
            if constexpr (AL == AL_MK) {
#pragma unroll
                for (int i = 0; i < A_CH; ++i) {
                    const int r = (tid >> 3) + 32 * i, c8 = tid & 7;
                    *reinterpret_cast<u32x2*>(As + off16(r, c8 >> 1) + (c8 & 1) * 8) = u32x2{pk(qa[u][i][0], qa[u][i][1]), pk(qa[u][i][2], qa[u][i][3])};
                }
            } else {
                const int kp = tid >> 4;
#pragma unroll
                for (int j = 0; j < A_CH / 2; ++j)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int r = 4 * ((tid & 15) + 16 * j) + t;
                        *reinterpret_cast<unsigned*>(As + off16(r, kp >> 2) + (kp & 3) * 4) = pk(qa[u][2 * j][t], qa[u][2 * j + 1][t]);
                    }
            }
            if constexpr (BL == BL_NK) {
#pragma unroll
                for (int i = 0; i < B_CH; ++i) {
                    const int r = (tid >> 3) + 32 * i, c8 = tid & 7;
                    *reinterpret_cast<u32x2*>(Bs + off16(r, c8 >> 1) + (c8 & 1) * 8) = u32x2{pk(qb[u][i][0], qb[u][i][1]), pk(qb[u][i][2], qb[u][i][3])};
                }
            } else {
                const int kp = tid >> 4;
#pragma unroll
                for (int j = 0; j < B_CH / 2; ++j)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int r = 4 * ((tid & 15) + 16 * j) + t;
                        *reinterpret_cast<unsigned*>(Bs + off16(r, kp >> 2) + (kp & 3) * 4) = pk(qb[u][2 * j][t], qb[u][2 * j + 1][t]);
                    }
            }
        };
        const int nmacro = (nsteps + KS - 1) / KS;
        if (nmacro > 0) load_macro(kbeg);
        for (int ms = 0; ms < nmacro; ++ms) {
            const int stage = NST == 2 ? (ms & 1) : 0;
#pragma unroll
            for (int u = 0; u < KS; ++u) store_bf(u, stage);
            __syncthreads();
            if (ms + 1 < nmacro) load_macro(kbeg + (ms + 1) * KS * BK);
#pragma unroll
            for (int u = 0; u < KS; ++u) {
                const char* As = lds + (stage * KS + u) * SUB_B;
                const char* Bs = As + BM * 64;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    bf16x8 fa[TM], fb[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        fa[i] = *reinterpret_cast<const bf16x8*>(As + off16(wm * (TM * 32) + i * 32 + li, 2 * h + lh));
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        fb[j] = *reinterpret_cast<const bf16x8*>(Bs + off16(wn * (TN * 32) + j * 32 + li, 2 * h + lh));
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                }
            }
            if constexpr (NST == 1) __syncthreads();       // every wave has read this image before it is overwritten
        }
        __syncthreads();           // the epilogue reuses the LDS
    } else
    // ---- main loop, hand-slotted --------------------------------------------------------------
    // One K-step = NM MFMAs per wave.  Each MFMA is followed by at most one "payload" operation and a
    // scheduling fence, so the issue order below is the order in the binary:
    //   slots 0 .. NL-1        : the NL buffer loads of tile s+1 (address math included)
    //   slot  Q   /  slot 2Q   : fragment reads of u-step 2 / 3 (into the set the previous u-step freed)
    //   slots NM-NL .. NM-1    : LDS writes of tile s+1 into the other stage (loads had >= NM-2NL MFMAs to land)
    // Everything except the two leading fragment reads and the barrier issues in the shadow of a
    // 64-cycle MFMA.  The step after the last one loads a tile nobody uses (clamped / zero-filled).
    {
        constexpr int NM = 16 * TM * TN, Q = 4 * TM * TN, NL = A_CH + B_CH;
        static_assert(NM - NL >= 2 * Q + 1, "payload slots overlap");
        if (nsteps > 0) {
            load_tiles(kbeg, ktap(kbeg), kchan(kbeg), ra0, rb0);
            store_tiles(0);
            __syncthreads();
        }
        if (FTE_PRIO_PROLOGUE) __builtin_amdgcn_s_setprio(0);
        int stap = 0, skc = 0;                                      // position of the tile in LDS (step s): advanced to s + 1 below
        if constexpr (AL == AL_MK || BL == BL_NK) { stap = ktap(kbeg); skc = kchan(kbeg); }
        unsigned va[A_CH];                                          // TAPO: per-thread source offsets of the current tap
        int sstep = 0, scbase = 0;                                  // TAPO: step inside the (super-chunk, tap) visit; first channel of the super-chunk
        if constexpr (TAPO) {
            sstep = (kbeg / BK) % SCS;
            scbase = skc - sstep * BK;
            const unsigned toff = (unsigned)((p.a_dh[tap0 + stap] * p.a_IW + p.a_dw[tap0 + stap]) * p.a_ld) * 4u;
#pragma unroll
            for (int i = 0; i < A_CH; ++i) va[i] = ((a_mask[i] >> stap) & 1) ? a_base[i] + toff : OOB;
        }
#ifdef FTE_STAMP
        unsigned long long st_t0 = 0, st_r0 = 0;
        if (g_stamp_buf) { st_t0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_s_waitcnt(0xC07F); }
#endif
        // AL_KM (filter gradient / dense tn), KMF = true: the fast addressing path, taken when every 32-wide m group of the
        // tile lies inside ONE tap (KC % 32 == 0: the i-th load of a wave is then a single tap -- its (dh, dw) are scalars), the K
        // range is whole 32-pixel steps, and the image is at least 32 / OW + 1 rows high (or 1 x 1: dense).  The thread's pixel
        // (n, oh, ow) is advanced by 32 with two carries instead of being re-derived by two divisions per step, and its tap
        // shifts are scalar adds: ~40 instead of ~110 vector-ALU instructions per K-step (each costs the matrix pipe 3-4 cycles).
        int kn_ = 0, koh_ = 0, kow_ = 0;                            // KMF: (n, oh, ow) of pixel k0 + (tid >> 3)
        bool kmf = false;
        if constexpr (AL == AL_KM && BF == 0) {
            const bool dense = p.a_OH * p.a_OW == 1;
            kmf = FTE_KM_FAST && (p.a_KC % 32 == 0) && (p.M % 32 == 0) && ((kend - kbeg) % BK == 0) && nsteps > 0 &&
                  (dense || (BK / p.a_OW + 1 <= p.a_OH));
            if (kmf) {
                const int pix = kbeg + (tid >> 3);                  // pixel of the tile in LDS; the loop advances it to step s + 1
                kn_ = fdiv(pix, a_hw, r_ahw);
                const int rem = pix - kn_ * a_hw;
                koh_ = fdiv(rem, p.a_OW, r_aow);
                kow_ = rem - koh_ * p.a_OW;
            }
        }
        // KMF, hoisted out of the K loop (round 3: the loop had 73 vector instructions beside its 64 MFMAs, nine of them quarter-rate
        // multiplies -- the tap of a load's 32-wide m group was re-derived by a division EVERY step and the source offset rebuilt from
        // (n, oh, ow) with three multiplies; each vector instruction costs the matrix pipe issue cycles):
        //   * per load i, wave-uniform and constant over the loop: its tap's (dh, dw), source offset and validity;
        //   * the thread's source offset is ADVANCED: a step moves the pixel by 32 = q32 rows + r32 columns plus at most one row carry
        //     and one image carry, i.e. by kd0 + (carry ? kdA : 0) + (image carry ? kdB : 0) bytes.
        int kq_dh[A_CH], kq_dw[A_CH], kq_off[A_CH];
        bool kq_ok[A_CH];
        unsigned kbase = 0;
        int kd0 = 0, kdA = 0, kdB = 0;
        if constexpr (AL == AL_KM && BF == 0) {
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                const int mg = m0 + 32 * i;                         // wave-uniform: first m of this load's 32-wide group
                const int t = mg / p.a_KC, c0 = mg - t * p.a_KC;
                const int tt = t < p.a_NT ? t : 0;
                kq_dh[i] = p.a_dh[tt]; kq_dw[i] = p.a_dw[tt];
                kq_ok[i] = mg < p.M;
                kq_off[i] = ((kq_dh[i] * p.a_IW + kq_dw[i]) * p.a_ld + c0) * 4;
            }
            kbase = (unsigned)(((kn_ * p.a_IH + koh_ * p.a_stride) * p.a_IW + kow_ * p.a_stride) * p.a_ld + ((tid & 7) << 2)) * 4u;
            if (p.a_OH * p.a_OW == 1) {
                kd0 = BK * p.a_IH * p.a_IW * p.a_ld * 4;            // 1 x 1 image: the pixel index IS n
            } else {
                const int q32 = BK / p.a_OW, r32 = BK - q32 * p.a_OW;
                kd0 = (q32 * p.a_stride * p.a_IW + r32 * p.a_stride) * p.a_ld * 4;
                kdA = (p.a_stride * p.a_IW - p.a_OW * p.a_stride) * p.a_ld * 4;
                kdB = (p.a_IH - p.a_OH * p.a_stride) * p.a_IW * p.a_ld * 4;
            }
        }
        auto run_loop = [&](auto KMF_) {
        constexpr bool KMF = decltype(KMF_)::value;
        for (int s = 0; s < nsteps; ++s) {
            const int cur = FTE_SINGLE ? 0 : (s & 1);
            const int k0 = kbeg + (s + 1) * BK;
            const float* As = smem + cur * STAGE;
            const float* Bs = As + BM * BK;
            float* Asn = smem + (FTE_SINGLE ? 0 : (cur ^ 1)) * STAGE;
            float* Bsn = Asn + BM * BK;
            // ---- per-step operand addressing (scalar / per-thread, no memory access yet) ----
            int tap = 0, kc0 = 0;
            unsigned a_toff = 0, b_soff = 0, a_soff = 0;
            if constexpr (TAPO) {
                // tile of step s + 1; the step after the last one re-stages the last tile (nobody reads it): no selects needed
                if (k0 < kend) {
                    skc += BK;
                    if (++sstep == SCS) {                       // next tap (of this super-chunk, or the first of the next one):
                        sstep = 0;                              // the only vector instructions of the K loop's addressing
                        if (++stap == NT) { stap = 0; scbase += SC; }
                        skc = scbase;
                        const unsigned toff = (unsigned)((p.a_dh[tap0 + stap] * p.a_IW + p.a_dw[tap0 + stap]) * p.a_ld) * 4u;
#pragma unroll
                        for (int i = 0; i < A_CH; ++i) va[i] = ((a_mask[i] >> stap) & 1) ? a_base[i] + toff : OOB;
                    }
                }
                tap = stap; kc0 = skc;
                a_soff = (unsigned)skc * 4u;
            } else if constexpr (AL == AL_MK || BL == BL_NK) {
                if (++stap == NT) { stap = 0; skc += BK; }      // (ktap, kchan) of k0, kept incrementally: no division
                tap = stap;
                kc0 = skc < p.a_KC ? skc : 0;                   // the unused tile after the last step
            }
            bool kin = false;
            int kn = 0, kih0 = 0, kiw0 = 0;
            unsigned vk[A_CH];                                  // KMF: this step's source offsets
            if constexpr (TAPO) {
            } else if constexpr (AL == AL_MK) {
                a_toff = (unsigned)((p.a_dh[tap0 + tap] * p.a_IW + p.a_dw[tap0 + tap]) * p.a_ld + kc0) * 4u;
            } else if constexpr (KMF) {
                // advance the thread's pixel by one K-step (the step after the last one re-stages the last tile: no selects)
                const bool adv = k0 < kend;
                if (adv && !(p.a_OH * p.a_OW == 1)) {
                    const int q32 = BK / p.a_OW, r32 = BK - q32 * p.a_OW;              // scalars
                    kow_ += r32;
                    const bool c1 = kow_ >= p.a_OW;
                    kow_ -= c1 ? p.a_OW : 0;
                    koh_ += q32 + (c1 ? 1 : 0);
                    const bool c2 = koh_ >= p.a_OH;
                    koh_ -= c2 ? p.a_OH : 0;
                    kbase += (unsigned)(kd0 + (c1 ? kdA : 0) + (c2 ? kdB : 0));
                } else if (adv) {
                    kbase += (unsigned)kd0;
                }
                kih0 = __mul24(koh_, p.a_stride);
                kiw0 = __mul24(kow_, p.a_stride);
#pragma unroll
                for (int i = 0; i < A_CH; ++i) {
                    const bool ok = kq_ok[i] && (unsigned)(kih0 + kq_dh[i]) < (unsigned)p.a_IH && (unsigned)(kiw0 + kq_dw[i]) < (unsigned)p.a_IW;
                    vk[i] = ok ? kbase + (unsigned)kq_off[i] : OOB;
                }
            } else {
                const int pix = k0 + (tid >> 3);
                kin = pix < kend;
                kn = fdiv(pix, a_hw, r_ahw);
                const int rem = pix - kn * a_hw;
                const int oh = fdiv(rem, p.a_OW, r_aow);
                kih0 = oh * p.a_stride;
                kiw0 = (rem - oh * p.a_OW) * p.a_stride;
            }
            if constexpr (BL == BL_KN) b_soff = (unsigned)((AL == AL_MK ? tap * p.a_KC + kc0 : k0) * p.b_ld) * 4u;
            else b_soff = (unsigned)(p.b_tapoff[tap0 + tap] + kc0) * 4u;

            auto load_slot = [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i < A_CH) {
                    if constexpr (TAPO) {
                        ra[i] = ldg(rsrcA, va[i], a_soff);
                    } else if constexpr (AL == AL_MK) {
                        ra[i] = ldg(rsrcA, ((a_mask[i] >> tap) & 1) && k0 < kend ? a_base[i] + a_toff : OOB, 0);
                    } else if constexpr (KMF) {
                        ra[i] = ldg(rsrcA, vk[i], 0);
                    } else {
                        const int ih = kih0 + (a_dhw[i] & 0xff) - 8, iw = kiw0 + ((a_dhw[i] >> 8) & 0xff) - 8;
                        const bool ok = kin && (a_dhw[i] >> 16) && ih >= 0 && ih < p.a_IH && iw >= 0 && iw < p.a_IW;
                        ra[i] = ldg(rsrcA, ok ? (unsigned)(((kn * p.a_IH + ih) * p.a_IW + iw) * p.a_ld) * 4u + a_c[i] : OOB, 0);
                    }
                } else {
                    constexpr int j = i - A_CH;
                    if constexpr (TAPO || KMF) {
                        rb[j] = ldg(rsrcB, b_base[j], b_soff);      // K ranges are whole 32-deep steps (the launcher / kmf check): no row test
                    } else if constexpr (BL == BL_KN) {
                        constexpr int CPR = BN / 4, RPP = NTH / CPR;
                        rb[j] = ldg(rsrcB, (k0 + tid / CPR + RPP * j < kend) ? b_base[j] : OOB, b_soff);
                    } else {
                        rb[j] = ldg(rsrcB, k0 < kend ? b_base[j] : OOB, b_soff);
                    }
                }
            };
            auto store_slot = [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i < A_CH) {
                    if constexpr (AL == AL_MK) {
                        const int r = (tid >> 3) + RP * i;
                        *reinterpret_cast<f32x4*>(Asn + r * BK + swz(r, tid & 7)) = ra[i];
                    } else {
                        *reinterpret_cast<f32x4*>(Asn + (tid >> 3) * BM + (((tid & 7) + 8 * i) << 2)) = ra[i];
                    }
                } else {
                    constexpr int j = i - A_CH;
                    if constexpr (BL == BL_KN) {
                        constexpr int CPR = BN / 4, RPP = NTH / CPR;
                        *reinterpret_cast<f32x4*>(Bsn + (tid / CPR + RPP * j) * BN + ((tid % CPR) << 2)) = rb[j];
                    } else {
                        const int r = (tid >> 3) + RP * j;
                        *reinterpret_cast<f32x4*>(Bsn + r * BK + swz(r, tid & 7)) = rb[j];
                    }
                }
            };

            f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
            read_frags(As, Bs, 0, fa0, fb0);
            read_frags(As, Bs, 1, fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, NM>([&](auto ic) {
                constexpr int idx = decltype(ic)::value;
                constexpr int u = idx / Q, w = idx % Q, t = w / (TM * TN), i = (w % (TM * TN)) / TN, j = w % TN;
                if constexpr ((u & 1) == 0)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[i][t], fb0[j][t], acc[i][j], 0, 0, 0);
                else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[i][t], fb1[j][t], acc[i][j], 0, 0, 0);
                if constexpr (idx < NL) load_slot(std::integral_constant<int, idx>{});
                if constexpr (idx == Q) read_frags(As, Bs, 2, fa0, fb0);
                if constexpr (idx == 2 * Q) read_frags(As, Bs, 3, fa1, fb1);
#if FTE_SINGLE
                if constexpr (idx == 3 * Q) __syncthreads();       // every wave has fetched its last fragments of this tile
#endif
                if constexpr (idx >= NM - NL) store_slot(std::integral_constant<int, idx - (NM - NL)>{});
                __builtin_amdgcn_sched_barrier(0);
            });
            __syncthreads();
        }
        };
        if constexpr (AL == AL_KM && BF == 0) {
            if (kmf) run_loop(std::true_type{});
            else run_loop(std::false_type{});
        } else {
            run_loop(std::false_type{});
        }
#ifdef FTE_STAMP
        if (g_stamp_buf && tid == 0) {
            const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
            unsigned long long* o = g_stamp_buf + 8 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
            o[0] = t1 - st_t0; o[1] = r1 - st_r0; o[2] = (unsigned long long)nsteps; o[3] = st_r0;
            o[4] = st_t0 - st_entry;            // prologue: entry -> first K-step
            o[5] = t1;                          // loop end (the epilogue's duration is taken against the stamp below)
        }
#endif
    }

    if (FTE_PRIO_EPILOGUE && BF == 0) __builtin_amdgcn_s_setprio(FTE_PRIO_EPILOGUE);      // drain quickly, free the slot
    if constexpr (SK) {
        // ---- stream-K hand-over (MI355X guide, "inter-workgroup communication": write-through payload + drained flag) ----------
        // slab layout [TM*TN*4 quads][NTH lanes] of 16 bytes: every store / load instruction of a wave covers 1 KiB contiguous
        constexpr int NQ = TM * TN * 4;
        constexpr unsigned SLAB_BYTES = (unsigned)NQ * NTH * 16u;
        if (sk_k0 > 0) {
            // PRODUCER: this worker started inside the tile.  Raw accumulators -> its slab with sc1 (write-through) stores, every wave
            // drains its stores, the workgroup meets, ONE lane raises the flag with an agent-scope store.
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.SKW + (size_t)sk_worker * (BM * BN), 0, SLAB_BYTES, 0x00020000);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (unsigned)((((i * TN + j) * 4 + q) * NTH + tid) * 16), 0, 16);
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(p.SKF + sk_worker, (unsigned)p.sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (FTE_PRIO_EPILOGUE) __builtin_amdgcn_s_setprio(0);
            return;
        }
        const int ksteps = Kc / BK;
        if (sk_k1 < ksteps) {
            // FINISHER: holds K-steps [0, sk_k1) of the tile; the following workers hold the rest, in worker order (each published its
            // part as the FIRST thing it did).  One lane polls the flag (relaxed, agent scope), the workgroup meets, then every lane
            // reads its quads with sc1 loads (they bypass this CU's L1; the slab was stored write-through) and adds them IN WORKER
            // ORDER: the sum does not depend on timing or placement.
            const int tile_end = (bid + 1) * ksteps;
            for (int w2 = sk_worker + 1; w2 < p.sk_workers; ++w2) {
                const int it0 = w2 * p.sk_base + min(w2, p.sk_rem);
                if (it0 >= tile_end) break;
                if (tid == 0) {
                    while (__hip_atomic_load(p.SKF + w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)p.sk_epoch) __builtin_amdgcn_s_sleep(1);
                }
                if (p.sk_acq) {
                    if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.SKW + (size_t)w2 * (BM * BN), 0, SLAB_BYTES, 0x00020000);
                constexpr int QC = NQ < 8 ? NQ : 8;        // quads in flight per lane
#pragma unroll
                for (int q0 = 0; q0 < NQ; q0 += QC) {
                    f32x4 part[QC];
#pragma unroll
                    for (int q = 0; q < QC; ++q)
                        part[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(((q0 + q) * NTH + tid) * 16), 0, 16));
#pragma unroll
                    for (int q = 0; q < QC; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[(q0 + q) / 4 / TN][((q0 + q) / 4) % TN][4 * ((q0 + q) & 3) + e] += part[q][e];
                }
            }
        }
    }
    igemm_epilogue<BM, BN, WM, WN, EPI, BF == 2, BNM>(p, acc, smem, bid, split, m0, n0, mt, c_ph, c_pw, prow, SK ? tid : -1);
    if constexpr (SK) { if (FTE_PRIO_EPILOGUE) __builtin_amdgcn_s_setprio(0); }
#ifdef FTE_STAMP
    if constexpr (BF == 0) {
        __syncthreads();
        if (g_stamp_buf && tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned long long* o = g_stamp_buf + 8 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
            o[6] = __builtin_amdgcn_s_memtime() - o[5];      // epilogue incl. its stores' completion
            o[7] = __builtin_amdgcn_s_memrealtime();
        }
    }
#endif
}

template <int BM, int BN, int WM, int WN, int AL, int BL, int EPI, int BF = 0>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN < 4) ? 2 : ((FTE_SINGLE && BM <= 192) ? 3 : 2)) void igemm_kernel(const IgemmParams p) {
    igemm_body<BM, BN, WM, WN, AL, BL, EPI, BF, false>(p);
}
// conv -> BN pairs of the graph nets (igemm.h "BN fusion"): forward with tile statistics, data gradient with the BN mask / sums
template <int BM, int BN, int WM, int WN, int AL, int BL, int EPI, int BF = 0>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN < 4) ? 2 : ((FTE_SINGLE && BM <= 192) ? 3 : 2)) void igemm_bn_kernel(const IgemmParams p) {
    igemm_body<BM, BN, WM, WN, AL, BL, EPI, BF, true>(p);
}

// ---- stream-K -----------------------------------------------------------------------------------------------------------------
// A launch whose tiles make 1-2.5 rounds of the chip quantises: 392 tiles of 128x64 on 256 CUs run for two tiles' time and do 1.53
// tiles' work per CU (the 14x14 layers of SphereNet at a 64-image shard: 77 %).  Here the launch is ONE grid of resident workers
// (p.sk_workers = CUs x blocks per CU) and the linear space of (tile, K-step) iterations is dealt out in equal contiguous shares.  A
// share that starts inside a tile leaves a raw partial tile (producer), the share that holds the tile's first K-step collects the
// partials and runs the fused epilogue (finisher): a producer part is always the FIRST thing its worker does and a finisher part the
// LAST, so a finisher never waits for work that has not started, whatever the dispatch order.  The partials are added in worker order:
// results are bit-identical run to run (they differ from the one-block-per-tile schedule by the summation order of the split tiles only).
// Worker ids are the XCD-aware remap of the block id: consecutive workers -- consecutive tiles, the same A rows -- share an XCD.
// blocks per CU the stream-K symbols are compiled for (= igemm_sk_blocks_per_cu(): the planner's worker count per CU)
#define SK_MINB(BM, BN, EPI) ((BM) * (BN) <= 64 * 64 ? ((EPI) == EPI_FWD ? 6 : 5) : ((BM) * (BN) <= 128 * 64 ? 4 : 3))
template <int BM, int BN, int WM, int WN, int AL, int BL, int EPI, bool BNM>
__device__ __forceinline__ void igemm_sk_body(const IgemmParams& p) {
    const int W = gridDim.x;
    const int q = W >> 3, r = W & 7, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int ksteps = p.K / BK;
    const int it0 = w * p.sk_base + min(w, p.sk_rem);
    const int it1 = it0 + p.sk_base + (w < p.sk_rem ? 1 : 0);
    for (int it = it0; it < it1;) {
        const int tile = it / ksteps, k0 = it - tile * ksteps;
        const int k1 = min(ksteps, k0 + (it1 - it));
        if (it != it0) __syncthreads();            // the previous part's epilogue is done with the LDS
        igemm_body<BM, BN, WM, WN, AL, BL, EPI, 0, BNM, true>(p, tile, k0, k1, w);
        it += k1 - k0;
    }
}
template <int BM, int BN, int WM, int WN, int AL, int BL, int EPI>
__global__ __launch_bounds__(256, SK_MINB(BM, BN, EPI)) void igemm_sk_kernel(const IgemmParams p) {
    igemm_sk_body<BM, BN, WM, WN, AL, BL, EPI, false>(p);
}
template <int BM, int BN, int WM, int WN, int AL, int BL, int EPI>
__global__ __launch_bounds__(256, SK_MINB(BM, BN, EPI)) void igemm_bn_sk_kernel(const IgemmParams p) {
    igemm_sk_body<BM, BN, WM, WN, AL, BL, EPI, true>(p);
}

// ---- fix-up: sum the split-K partial tiles of one output tile and apply the fused epilogue -------------
// grid = tiles of the launch that wrote PW (same tile numbering, no XCD remap needed: pure streaming);
// thread = one column (tid % BN) and every (256/BN)-th row of the tile: 512-B coalesced rows.
template <int BM, int BN, int EPI>
__global__ __launch_bounds__(256) void igemm_fixup_kernel(const IgemmParams p, int splits) {
    __shared__ int rowoff[BM];
    __shared__ float red[2][256 / BN][BN];
    const int tid = threadIdx.x;
    const int ntn = p.N / BN;
    const int bid = blockIdx.x / FIXUP_CHUNKS, chunk = blockIdx.x % FIXUP_CHUNKS;   // tile, row chunk of the tile
    constexpr int CR = BM / FIXUP_CHUNKS;
    const int mt = bid / ntn, nt_ = bid - mt * ntn;
    const int m0 = p.m_base + mt * BM, n0 = nt_ * BN;
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
    constexpr int RG = 256 / BN;                       // row groups
    const int cl = tid % BN, rg = tid / BN;
    const int col = n0 + cl;
    const long tile_stride = (long)(gridDim.x / FIXUP_CHUNKS) * (BM * BN);
    const float* W = p.PW + (long)bid * (BM * BN) + cl;
    float sa = 0.f, sb = 0.f;
    // rows of this thread: chunk*CR + rg + RG*i, i < NR; all partial-tile loads of UNR rows are issued
    // before any of them is used (the loop is latency-bound otherwise)
    constexpr int NR = CR / RG;
    constexpr int UNR = NR < 4 ? NR : 4;
    static_assert(NR % UNR == 0, "row unroll");
    if constexpr (EPI == EPI_FWD) {
        const float bias = p.bias ? p.bias[col] : 0.f;
        const bool act = p.alpha != nullptr;
        const float al = act ? p.alpha[col] : 1.f;
        for (int i0 = 0; i0 < NR; i0 += UNR) {
            float v[UNR], rres[UNR];
            int off[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int rl = chunk * CR + rg + RG * (i0 + u);
                off[u] = rowoff[rl];
                v[u] = 0.f;
                rres[u] = 0.f;
                if (off[u] >= 0) {
                    for (int sidx = 0; sidx < splits; ++sidx) v[u] += W[sidx * tile_stride + rl * BN];
                    if (p.R) rres[u] = p.R[(long)off[u] + col];
                    else if (p.R16) rres[u] = bf2f(p.R16[(long)off[u] + col]);
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (off[u] < 0) continue;
                const long o = (long)off[u] + col;
                float x = v[u] + bias;
                if (p.Z) p.Z[o] = x;
                if (p.Z16) p.Z16[o] = tobf(x);
                if (act) x = x > 0.f ? x : al * x;
                if (p.Y) p.Y[o] = x + rres[u];
                if (p.Y16) p.Y16[o] = tobf(x + rres[u]);
            }
        }
    } else {
        const bool msk = p.Zin != nullptr || p.Zin16 != nullptr;
        const float al = msk ? p.alpha[col % p.amod] : 1.f;
        for (int i0 = 0; i0 < NR; i0 += UNR) {
            float v[UNR], zz[UNR];
            int off[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int rl = chunk * CR + rg + RG * (i0 + u);
                off[u] = rowoff[rl];
                v[u] = 0.f;
                zz[u] = 1.f;
                if (off[u] >= 0) {
                    for (int sidx = 0; sidx < splits; ++sidx) v[u] += W[sidx * tile_stride + rl * BN];
                    const long o = (long)off[u] + col;
                    if (p.ADD) v[u] += p.ADD[o];
                    else if (p.ADD16) v[u] += bf2f(p.ADD16[o]);
                    if (p.Zin) zz[u] = p.Zin[o];
                    else if (p.Zin16) zz[u] = bf2f(p.Zin16[o]);
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (off[u] < 0) continue;
                const long o = (long)off[u] + col;
                float x = v[u];
                if (p.RAW) p.RAW[o] = x;
                if (p.RAW16) p.RAW16[o] = tobf(x);
                if (msk) {
                    sa += x * fminf(zz[u], 0.f);
                    x *= prelu_slope(zz[u], al);
                    sb += x;
                }
                if (p.DZ) p.DZ[o] = x;
                if (p.DZ16) p.DZ16[o] = tobf(x);
            }
        }
        if (p.PA) {
            red[0][rg][cl] = sa;
            red[1][rg][cl] = sb;
            __syncthreads();
            if (rg == 0) {
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int g = 0; g < RG; ++g) { a += red[0][g][cl]; b += red[1][g][cl]; }
                const long o = (long)(p.prow0 + mt * FIXUP_CHUNKS + chunk) * p.N + col;      // one partial row per (tile row, chunk)
                p.PA[o] = a;
                if (p.PB) p.PB[o] = b;
            }
        }
    }
}

template <int EPI>
hipError_t fixup_tile(const IgemmParams& p, int tile, int splits, hipStream_t st) {
    int bm, bn;
    igemm_tile_dims(tile, &bm, &bn);
    const int tiles = ((p.M - p.m_base + bm - 1) / bm) * (p.N / bn) * FIXUP_CHUNKS;
    switch (tile) {
        case TILE_128x128: hipLaunchKernelGGL((igemm_fixup_kernel<128, 128, EPI>), dim3(tiles), dim3(256), 0, st, p, splits); break;
        case TILE_256x64:  hipLaunchKernelGGL((igemm_fixup_kernel<256, 64, EPI>), dim3(tiles), dim3(256), 0, st, p, splits); break;
        case TILE_128x64:  hipLaunchKernelGGL((igemm_fixup_kernel<128, 64, EPI>), dim3(tiles), dim3(256), 0, st, p, splits); break;
        case TILE_192x64:  hipLaunchKernelGGL((igemm_fixup_kernel<192, 64, EPI>), dim3(tiles), dim3(256), 0, st, p, splits); break;
        default:           hipLaunchKernelGGL((igemm_fixup_kernel<64, 64, EPI>), dim3(tiles), dim3(256), 0, st, p, splits); break;
    }
    return hipGetLastError();
}

// Flag words of the stream-K hand-over.  A launch needs its workers' words to differ from its epoch when it starts; zeroing words in the
// caller's workspace is a fill kernel of ~6 us in front of EVERY stream-K launch (measured: 31 per SphereNet step at 64 images).
// Instead the library owns one row of words per STREAM (launches of one stream never overlap, so a row has one user at a time) and
// tags them with a per-row launch counter: a word equals the current epoch only after this launch's producer has stored it.  Rows are
// handed out to stream handles first come, first served; a 17th concurrent stream falls back to words in the workspace + a memset
// (a row is never taken from its stream: a launch of that stream may still be polling its words -- the price of cycling through more
// than 16 stream handles is the ~6 us fill in front of their stream-K launches, never a wrong flag).
constexpr int SK_ROWS = 16, SK_ROW_WORDS = 2048;
__device__ unsigned g_sk_words[SK_ROWS][SK_ROW_WORDS];
struct SkRow { hipStream_t st; int dev; unsigned epoch; bool used; };
SkRow g_sk_rows[SK_ROWS];
std::mutex g_sk_mu;
constexpr int SK_DEVS = 16;
unsigned* g_sk_base[SK_DEVS];      // the words' address on each device of this process (one process per GPU is the design; this keeps two honest)
hipError_t sk_flags(IgemmParams* q, hipStream_t st) {
    static const bool force_ws = getenv("FTE_SK_WS_FLAGS") != nullptr;      // A/B hook: the memset form
    // A stream being CAPTURED into a graph replays this launch with its arguments frozen -- the epoch too, which would then equal what
    // the previous replay left in the words: captured launches take the workspace words and a memset node (replayed every time).
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    int dev = -1;
    if (!force_ws && !capturing && q->sk_workers <= SK_ROW_WORDS && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < SK_DEVS) {
        std::lock_guard<std::mutex> lock(g_sk_mu);
        if (!g_sk_base[dev]) {
            hipError_t e = hipGetSymbolAddress(reinterpret_cast<void**>(&g_sk_base[dev]), HIP_SYMBOL(g_sk_words));
            if (e != hipSuccess) return e;
        }
        int row = -1;
        for (int i = 0; i < SK_ROWS; ++i) if (g_sk_rows[i].used && g_sk_rows[i].st == st && g_sk_rows[i].dev == dev) { row = i; break; }
        if (row < 0) for (int i = 0; i < SK_ROWS; ++i) if (!g_sk_rows[i].used) { row = i; g_sk_rows[i].used = true; g_sk_rows[i].st = st; g_sk_rows[i].dev = dev; g_sk_rows[i].epoch = 0; break; }
        if (row >= 0) {
            if (++g_sk_rows[row].epoch == 0u) g_sk_rows[row].epoch = 1u;      // 0 is the value the words are loaded with
            q->SKF = g_sk_base[dev] + (size_t)row * SK_ROW_WORDS;
            q->sk_epoch = (int)g_sk_rows[row].epoch;
            return hipSuccess;
        }
    }
    q->sk_epoch = 1;
    return hipMemsetAsync(q->SKF, 0, ((size_t)q->sk_workers * 4 + 15) & ~(size_t)15, st);
}

template <int BM, int BN, int WM, int WN, int AL, int BL, int EPI, int BF>
hipError_t launch_cfg_p(const IgemmParams& p, int splits, hipStream_t st) {
    const int mt = (p.M - p.m_base + BM - 1) / BM, nt = p.N / BN;
    // fp32: one stage of (BM+BN) x 32 floats; bf16: KS sub-step images of (BM+BN) x 64 bytes (one macro step); both hold
    // at least the BM ints of the epilogue's row-offset table
    constexpr int KSL = (BM + BN) <= 128 ? 4 : 2;               // sub-step images: KS = 4 x 1 stage, or KS = 1 x 2 stages
    const size_t epi = (size_t)(BM + WM * WN * 32 * 36 + 2 * WM * BN) * sizeof(float);      // row offsets + one transpose patch per wave + column partials
    const size_t loop = (size_t)KSL * (BM + BN) * 64;
    const size_t loop32 = (FTE_SINGLE ? 1 : 2) * (size_t)(BM + BN) * BK * sizeof(float);
    const size_t loop16 = 2 * (size_t)(BM + BN) * 64;
    const size_t lds = BF == 2 ? (loop16 > epi ? loop16 : epi) : (BF ? (loop > epi ? loop : epi) : (epi > loop32 ? epi : loop32));
    static const size_t lds_pad = getenv("FTE_LDS_PAD") ? (size_t)atoi(getenv("FTE_LDS_PAD")) : 0;      // tuning hook: fewer blocks per CU
    const size_t lds_x = lds + lds_pad;
    // the BN-fusion symbol exists for the 4-wave forward / data-gradient kernels only (k-contiguous A)
    constexpr bool BN_OK = AL == AL_MK && WM * WN == 4;
    const bool bnm = p.SP != nullptr || p.bn_mu != nullptr;
    if (bnm && !BN_OK) return hipErrorInvalidValue;
    if (p.sk_workers > 0) {
        // stream-K: fp32, k-contiguous A, four waves, the three square-ish tiles
        if constexpr (AL == AL_MK && BF == 0 && WM * WN == 4 && BM <= 128) {
            if (p.ncls > 1 || p.split_major > 0 || splits != 1 || p.PW || !p.SKW || !p.SKF || p.m_base != 0) return hipErrorInvalidValue;
            auto sk = igemm_sk_kernel<BM, BN, WM, WN, AL, BL, EPI>;
            if (bnm) sk = igemm_bn_sk_kernel<BM, BN, WM, WN, AL, BL, EPI>;
            if (igemm_prof_on()) { const int ta[7] = {BM, BN, WM, WN, AL, BL, EPI}; igemm_note_symbol(bnm ? "igemm_bn_sk_kernel" : "igemm_sk_kernel", ta, 7); }
            // (per device and symbol, set-once with an atomic mask: two host threads driving two devices of one process both get their
            // attribute; setting it twice is harmless)
            static std::atomic<unsigned> sk_attr[2] = {{0u}, {0u}};
            int adev = 0;
            (void)hipGetDevice(&adev);
            const unsigned abit = 1u << (adev & 31);
            if (!(sk_attr[bnm].load(std::memory_order_acquire) & abit)) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + 65536));
                if (e != hipSuccess) return e;
                sk_attr[bnm].fetch_or(abit, std::memory_order_release);
            }
            static const bool sk_dbg = getenv("FTE_SK_DEBUG") != nullptr;
            if (sk_dbg) {
                int nb = 0;
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(sk), 256, lds_x);
                fprintf(stderr, "[sk] %dx%d epi %d bn %d: workers %d (base %d rem %d), M %d N %d K %d, occupancy %d blocks/CU, lds %zu\n", BM, BN, EPI, (int)bnm,
                        p.sk_workers, p.sk_base, p.sk_rem, p.M, p.N, p.K, nb, lds_x);
            }
            IgemmParams q = p;
            static const int sk_acq = getenv("FTE_SK_ACQ") ? atoi(getenv("FTE_SK_ACQ")) : 0;
            q.sk_acq = sk_acq;
            hipError_t e = sk_flags(&q, st);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(sk, dim3(q.sk_workers), dim3(256), lds_x, st, q);
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    }
    auto kern = igemm_kernel<BM, BN, WM, WN, AL, BL, EPI, BF>;
    if constexpr (BN_OK) { if (bnm) kern = igemm_bn_kernel<BM, BN, WM, WN, AL, BL, EPI, BF>; }
    if (igemm_prof_on()) { const int ta[8] = {BM, BN, WM, WN, AL, BL, EPI, BF}; igemm_note_symbol(bnm ? "igemm_bn_kernel" : "igemm_kernel", ta, 8); }
    static bool attr_done[2] = {false, false};
    if (!attr_done[bnm]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + 65536 > 160 * 1024 ? lds : lds + 65536));
        if (e != hipSuccess) return e;
        attr_done[bnm] = true;
    }
    if (p.ncls > 1) {                       // merged parity classes: every class has the same M x N tile grid
        IgemmParams q = p;
        q.cls_tiles = mt * nt; q.cls_mtiles = mt;
        static const int cls_rot = getenv("FTE_CLS_ROT") ? atoi(getenv("FTE_CLS_ROT")) : 3;      // tuning hook: 31 = no rotation
        q.cls_rot = cls_rot;
        hipLaunchKernelGGL(kern, dim3(mt * nt * p.ncls, 1), dim3(64 * WM * WN), lds_x, st, q);
    } else if (p.split_major > 0) hipLaunchKernelGGL(kern, dim3(mt * nt * splits), dim3(64 * WM * WN), lds_x, st, p);
    else hipLaunchKernelGGL(kern, dim3(mt * nt, splits), dim3(64 * WM * WN), lds_x, st, p);
    return hipGetLastError();
}

bool g_bf16 = false;      // igemm_set_bf16(): operand precision of every launch of the family
template <int BM, int BN, int WM, int WN, int AL, int BL, int EPI>
hipError_t launch_cfg(const IgemmParams& p, int splits, hipStream_t st) {
    if (p.src16) {                           // bf16 sources: forward / dgrad take NK weight packs, wgrad KM x KN
        if constexpr (!(AL == AL_MK && BL == BL_KN) && WM * WN == 4) return launch_cfg_p<BM, BN, WM, WN, AL, BL, EPI, 2>(p, splits, st);
        else return hipErrorInvalidValue;
    }
    if constexpr (WM * WN == 4) {
        if (g_bf16) return launch_cfg_p<BM, BN, WM, WN, AL, BL, EPI, 1>(p, splits, st);
    } else {
        if (g_bf16) return hipErrorInvalidValue;                  // 1 / 2-wave blocks exist for the fp32 products only
    }
    return launch_cfg_p<BM, BN, WM, WN, AL, BL, EPI, 0>(p, splits, st);
}

template <int AL, int BL, int EPI>
hipError_t launch_tile(const IgemmParams& p, int tile, int splits, hipStream_t st) {
    switch (tile) {
        case TILE_128x128: return launch_cfg<128, 128, 2, 2, AL, BL, EPI>(p, splits, st);
        case TILE_256x64:  return launch_cfg<256, 64, 4, 1, AL, BL, EPI>(p, splits, st);
        case TILE_128x64:  return launch_cfg<128, 64, 2, 2, AL, BL, EPI>(p, splits, st);
        case TILE_64x64:   return launch_cfg<64, 64, 2, 2, AL, BL, EPI>(p, splits, st);
        case TILE_64x64_W1:
            if constexpr (AL == AL_MK) return launch_cfg<64, 64, 1, 1, AL, BL, EPI>(p, splits, st);
            else return hipErrorInvalidValue;
        case TILE_64x64_W2:
            if constexpr (AL == AL_MK) return launch_cfg<64, 64, 2, 1, AL, BL, EPI>(p, splits, st);
            else return hipErrorInvalidValue;
        case TILE_192x64:
            // only the filter-gradient / dense-tn symbol is instantiated at this shape (M = 9 x 64 rows)
            if constexpr (AL == AL_KM) return launch_cfg<192, 64, 2, 2, AL, BL, EPI>(p, splits, st);
            else return hipErrorInvalidValue;
    }
    return hipErrorInvalidValue;
}

}  // namespace

// ---- launch records (bench.py's roofline leg) -----------------------------------------------------
// When enabled, every kernel launch of the family is bracketed by a HIP event pair on the launch
// stream and its algorithmic FLOPs (2 * rows * N * K of THAT launch) are noted, so that a caller can
// report FLOPs / duration per kernel symbol -- the same per-symbol average rocprofv3 --stats prints.
namespace {
struct ProfRec { int sig[5]; int mnk[3]; double flops, bytes; hipEvent_t e0, e1; char sym[96]; };
std::vector<ProfRec> g_prof;
bool g_prof_on = false;
char g_last_sym[96] = "";
void prof_clear() {
    for (auto& r : g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_prof.clear();
}
hipError_t dispatch(const IgemmParams& p, int al, int bl, int epi, int tile, int splits, hipStream_t st) {
    static const bool no16 = getenv("FTE_NO_IGEMM16") != nullptr;      // A/B hook: keep the register-staged BF = 2 kernels
    if (!no16 && igemm16_handles(p, al, bl, tile)) return igemm16_launch(p, epi, tile, splits, st);
    if (al == AL_MK && bl == BL_KN && epi == EPI_FWD) return launch_tile<AL_MK, BL_KN, EPI_FWD>(p, tile, splits, st);
    if (al == AL_MK && bl == BL_NK && epi == EPI_DGRAD) return launch_tile<AL_MK, BL_NK, EPI_DGRAD>(p, tile, splits, st);
    if (al == AL_MK && bl == BL_NK && epi == EPI_FWD) return launch_tile<AL_MK, BL_NK, EPI_FWD>(p, tile, splits, st);
    if (al == AL_KM && bl == BL_KN && epi == EPI_FWD) return launch_tile<AL_KM, BL_KN, EPI_FWD>(p, tile, splits, st);
    return hipErrorInvalidValue;
}
}  // namespace

hipError_t igemm_launch(const IgemmParams& p, int al, int bl, int epi, int tile, int splits, hipStream_t st) {
    // the loaders and the epilogue move 16 bytes per lane: every tensor the kernel touches must be 16-byte aligned and
    // its row pitch a multiple of 4 floats (torch allocations and the arenas' views are)
    const uintptr_t ptrs = (uintptr_t)p.A | (uintptr_t)p.B | (uintptr_t)p.Y | (uintptr_t)p.Z | (uintptr_t)p.R | (uintptr_t)p.ADD |
                           (uintptr_t)p.RAW | (uintptr_t)p.Zin | (uintptr_t)p.DZ | (uintptr_t)p.PW | (uintptr_t)p.bias |
                           (uintptr_t)p.R16 | (uintptr_t)p.ADD16 | (uintptr_t)p.Zin16 | (uintptr_t)p.Z16 | (uintptr_t)p.RAW16 |
                           (uintptr_t)p.Y16 | (uintptr_t)p.DZ16;
    if ((p.R16 || p.ADD16 || p.Zin16 || p.Z16 || p.RAW16) && !p.src16) return hipErrorInvalidValue;      // bf16 storage: the bf16-source kernels only
    if (epi == EPI_FWD ? (!p.Y && !p.Y16 && !p.PW) : (!p.DZ && !p.DZ16 && !p.PW)) return hipErrorInvalidValue;
    if (!p.src16 && (epi == EPI_FWD ? !p.Y : !p.DZ)) return hipErrorInvalidValue;
    const uintptr_t al_ptr = epi == EPI_FWD ? (uintptr_t)p.alpha : 0;      // read as float4 by the forward epilogue only
    if (((ptrs | al_ptr) & 15) || (p.c_ld & 3) || (p.a_ld & 3) || (p.b_ld & 3)) return hipErrorInvalidValue;
    if (p.M >= (1 << 24) || (al == AL_KM && p.K >= (1 << 24))) return hipErrorInvalidValue;      // fdiv() range (see the kernel)
    // k-contiguous A: the K loop walks whole 32-deep steps inside one tap (no partial-step row tests in the loop)
    if (al == AL_MK && ((p.a_KC % 32) || (p.kchunk % 32) || (p.K % 32))) return hipErrorInvalidValue;
    if (!g_prof_on) return dispatch(p, al, bl, epi, tile, splits, st);
    ProfRec r;
    r.sig[0] = al; r.sig[1] = bl; r.sig[2] = epi; r.sig[3] = tile; r.sig[4] = splits;
    const double rows = (double)(p.M - p.m_base) * (p.ncls > 1 ? p.ncls : 1);      // merged parity classes: M rows per class
    const double keff = p.ncls > 1 ? (double)(p.cls_tap0[p.ncls] * p.a_KC) / p.ncls : (double)p.K;
    r.flops = 2.0 * rows * (double)p.N * keff;
    r.mnk[0] = (int)rows; r.mnk[1] = p.N; r.mnk[2] = (int)keff;
    // ALGORITHMIC bytes of this launch: every operand tensor once, every result tensor once.  The A tensor is
    // counted on the launch that starts at row 0 (a big-tile main + small-tile tail pair reads it once between them).
    {
        const double esz = p.src16 ? 2.0 : 4.0;
        (void)esz;
        double b = (p.m_base == 0 ? (double)p.a_bytes : 0.0) + (double)p.b_bytes;
        const double tile = rows * (double)p.N * 4.0;
        if (p.PW) b += tile * splits;
        else if (epi == EPI_FWD) {
            if (p.Y) b += tile * splits;                      // Y (split-K: one slab per split)
            if (p.Z) b += tile;
            if (p.R) b += tile;
            if (p.Y16) b += tile / 2;
            if (p.Z16) b += tile / 2;
            if (p.R16) b += tile / 2;
        } else {
            if (p.DZ) b += tile;
            if (p.ADD) b += tile;
            if (p.Zin) b += tile;
            if (p.RAW) b += tile;
            if (p.DZ16) b += tile / 2;
            if (p.ADD16) b += tile / 2;
            if (p.Zin16) b += tile / 2;
            if (p.RAW16) b += tile / 2;
        }
        r.bytes = b;
    }
    hipError_t e = hipEventCreate(&r.e0);
    if (e != hipSuccess) return e;
    e = hipEventCreate(&r.e1);
    if (e != hipSuccess) return e;
    (void)hipEventRecord(r.e0, st);
    g_last_sym[0] = 0;
    e = dispatch(p, al, bl, epi, tile, splits, st);
    (void)hipEventRecord(r.e1, st);
    memcpy(r.sym, g_last_sym, sizeof(r.sym));
    g_prof.push_back(r);
    return e;
}

// launch records for kernels outside the family's dispatcher (wgrad16.hip): begin returns a handle (< 0: records are off)
int igemm_prof_begin(const int* sig5, int rows, int n, int k, double flops, double bytes, hipStream_t st) {
    if (!g_prof_on) return -1;
    ProfRec r;
    for (int i = 0; i < 5; ++i) r.sig[i] = sig5[i];
    r.mnk[0] = rows; r.mnk[1] = n; r.mnk[2] = k;
    r.flops = flops; r.bytes = bytes; r.sym[0] = 0;
    if (hipEventCreate(&r.e0) != hipSuccess) return -1;
    if (hipEventCreate(&r.e1) != hipSuccess) { (void)hipEventDestroy(r.e0); return -1; }
    (void)hipEventRecord(r.e0, st);
    g_prof.push_back(r);
    return (int)g_prof.size() - 1;
}
void igemm_prof_end(int handle, const char* sym, hipStream_t st) {
    if (handle < 0 || handle >= (int)g_prof.size()) return;
    (void)hipEventRecord(g_prof[handle].e1, st);
    snprintf(g_prof[handle].sym, sizeof(g_prof[handle].sym), "%s", sym);
}

hipError_t igemm_fixup(const IgemmParams& p, int epi, int tile, int splits, hipStream_t st) {
    if (epi == EPI_FWD) return fixup_tile<EPI_FWD>(p, tile, splits, st);
    return fixup_tile<EPI_DGRAD>(p, tile, splits, st);
}

void igemm_set_bf16(bool on) { g_bf16 = on; }
bool igemm_get_bf16() { return g_bf16; }

void igemm_prof_enable(bool on, bool clear) {
    if (on && clear) prof_clear();
    g_prof_on = on;
}
bool igemm_prof_on() { return g_prof_on; }
void igemm_note_symbol(const char* family, const int* targs, int ntargs) {
    int n = snprintf(g_last_sym, sizeof(g_last_sym), "%s<", family);
    for (int i = 0; i < ntargs && n < (int)sizeof(g_last_sym) - 16; ++i) n += snprintf(g_last_sym + n, sizeof(g_last_sym) - n, i ? ",%d" : "%d", targs[i]);
    snprintf(g_last_sym + n, sizeof(g_last_sym) - n, ">");
}
hipError_t igemm_prof_get_name(int i, char* buf, int buflen) {
    if (i < 0 || i >= (int)g_prof.size() || !buf || buflen <= 0) return hipErrorInvalidValue;
    snprintf(buf, (size_t)buflen, "%s", g_prof[i].sym);
    return hipSuccess;
}
int igemm_prof_count() { return (int)g_prof.size(); }
hipError_t igemm_prof_get_shape(int i, int* mnk, double* bytes) {
    if (i < 0 || i >= (int)g_prof.size()) return hipErrorInvalidValue;
    for (int k = 0; k < 3; ++k) mnk[k] = g_prof[i].mnk[k];
    *bytes = g_prof[i].bytes;
    return hipSuccess;
}
hipError_t igemm_prof_get(int i, int* sig, double* flops, float* ms) {
    if (i < 0 || i >= (int)g_prof.size()) return hipErrorInvalidValue;
    const ProfRec& r = g_prof[i];
    for (int k = 0; k < 5; ++k) sig[k] = r.sig[k];
    *flops = r.flops;
    return hipEventElapsedTime(ms, r.e0, r.e1);
}

namespace {
// resident blocks per CU the RUNTIME reports for a stream-K symbol (registers, LDS, wave slots of THIS device and compiler), queried once
// per symbol; -1: no device / query failed (the compiled figure is used)
template <int BM, int BN, int EPI>
int sk_runtime_occupancy() {
    static const int v = [] {
        int nb = 0;
        constexpr int AL = AL_MK, BL = EPI == EPI_FWD ? BL_KN : BL_NK;
        const size_t epi_b = (size_t)(BM + 4 * 32 * 36 + 2 * 2 * BN) * sizeof(float);
        const size_t loop_b = (FTE_SINGLE ? 1 : 2) * (size_t)(BM + BN) * igemm_dev::BK * sizeof(float);
        const size_t lds = epi_b > loop_b ? epi_b : loop_b;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(igemm_sk_kernel<BM, BN, 2, 2, AL, BL, EPI>), 256, lds);
        if (e != hipSuccess) { (void)hipGetLastError(); return -1; }
        return nb;
    }();
    return v;
}
}  // namespace

int igemm_num_cus() {
    static const int v = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount <= 0) {
            (void)hipGetLastError();
            return 256;                            // MI355X; a host without a device only sizes workspaces
        }
        return prop.multiProcessorCount;
    }();
    return v;
}

int igemm_sk_blocks_per_cu(int tile, int epi) {
    // Workers per CU of a stream-K launch = min(the figure the symbol is compiled for (SK_MINB), what the runtime reports as resident).
    // A finisher spins on flags of higher-numbered workers: planning more workers than are resident (a register count that moved
    // with the compiler -- round 5: six planned, five fit -- another part's CU / register file) serialises the launch into rounds
    // of spinning blocks, so the plan follows the device; 0 resident -> no stream-K for that shape (the one-block-per-tile plan).
    static const int env = getenv("FTE_SK_BPC") ? atoi(getenv("FTE_SK_BPC")) : 0;      // tuning hook: fewer workers per CU
    int v = 0, rt = -1;
    const bool fwd = epi == EPI_FWD;
    switch (tile) {
        case TILE_64x64:   v = SK_MINB(64, 64, epi);   rt = fwd ? sk_runtime_occupancy<64, 64, EPI_FWD>() : sk_runtime_occupancy<64, 64, EPI_DGRAD>(); break;
        case TILE_128x64:  v = SK_MINB(128, 64, epi);  rt = fwd ? sk_runtime_occupancy<128, 64, EPI_FWD>() : sk_runtime_occupancy<128, 64, EPI_DGRAD>(); break;
        case TILE_128x128: v = SK_MINB(128, 128, epi); rt = fwd ? sk_runtime_occupancy<128, 128, EPI_FWD>() : sk_runtime_occupancy<128, 128, EPI_DGRAD>(); break;
        default: return 0;
    }
    if (rt >= 0 && rt < v) v = rt;
    return env > 0 && env < v ? env : v;
}
size_t igemm_sk_ws_bytes(int tile, int workers) {
    int bm, bn;
    igemm_tile_dims(tile, &bm, &bn);
    return (size_t)workers * bm * bn * sizeof(float) + (((size_t)workers * 4 + 255) & ~(size_t)255);
}

void igemm_tile_dims(int tile, int* bm, int* bn) {
    switch (tile) {
        case TILE_128x128: *bm = 128; *bn = 128; break;
        case TILE_256x64:  *bm = 256; *bn = 64; break;
        case TILE_128x64:  *bm = 128; *bn = 64; break;
        case TILE_192x64:  *bm = 192; *bn = 64; break;
        default:           *bm = 64;  *bn = 64; break;
    }
}
