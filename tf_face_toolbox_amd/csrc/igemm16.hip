// igemm16.hip -- bf16 gathered-GEMM kernel with LDS-DMA operand staging for gfx950 (MI355X).
//
//   C[M,N] = sum_k A(m,k) * B(n,k),  A and B bf16 with k CONTIGUOUS, fp32 accumulate on v_mfma_f32_32x32x16_bf16:
//   conv forward : A = im2col rows of the bf16 activation copy (NHWC: k = channels of one tap), B = weights packed
//                  [tap][cout][cin]  (fte_pack_weights_bf16's transposed pack)
//   conv dgrad   : A = im2col rows of the bf16 dz copy, B = weights [tap][cin][cout] (the HWIO layout itself)
// i.e. the AL_MK x BL_NK cases of the igemm family with bf16 SOURCES (fte_conv2d_fwd16 / _dgrad16); filter gradients
// (both operands pixel-major: transposed reads) stay on igemm.hip's BF = 2 kernel.
//
// Why a second kernel: at the bf16 MFMA rate a 32-deep K-step is 2 MFMAs per accumulator block -- the register-staged loop of
// igemm.hip (global -> VGPR -> ds_write -> barrier per K-step) runs its load, LDS-write, MFMA and epilogue phases back to back
// ("a plain SUM").  Here, as the CDNA4 GEMM playbook prescribes for an MFMA-dense loop at ~1 block per CU:
//   * operands go global -> LDS directly (`buffer_load_dwordx4 ... offen lds`: no staging VGPRs, no ds_write pass);
//     out-of-image taps / rows beyond M use an offset beyond num_records, for which the DMA writes ZEROS (probe:
//     scripts/probes/lds_dma_oob.hip) -- padding costs nothing and needs no branch;
//   * BK = 64 bf16 = 128-byte operand rows; a ring of NST stages, NST - 1 K-steps in flight; ONE raw s_barrier per K-step
//     and a COUNTED s_waitcnt vmcnt (never 0 in the steady state), so the loads of the next stages stay in flight across
//     the barrier;
//   * the LDS image is lane-linear per DMA instruction (8 rows x 128 B per wave instruction), so the bank-conflict swizzle
//     is applied to the SOURCE address (which 16-byte k-chunk a lane fetches) and again on the fragment read:
//     chunk c of row r lives in slot c ^ ((r >> 1) & 7) -- conflict-free ds_read_b128 for the 32x32x16 operand map.
// The epilogue (bias / PReLU / residual, or the PReLU-gradient epilogue with column partials, LDS-transposed 16-byte stores,
// bf16 result copies) is igemm_dev.h's, shared with igemm.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "igemm_dev.h"

namespace {

using namespace igemm_dev;

constexpr int BK16 = 64;                 // bf16 elements per K-step
constexpr int ROWB = BK16 * 2;           // bytes per operand row in LDS

typedef __attribute__((address_space(3))) void lds_void;

// one LDS-DMA instruction: 16 bytes per lane from the buffer (offset voff + soff; beyond num_records -> zeros) to
// lds + 16 * lane.  (A named __device__ function: hipcc's host pass drops the enclosing kernel's stub when the builtin sits
// directly inside a lambda of the kernel.)
__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t& r, char* lds, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds, 16, voff, soff, 0, 0);
}

template <class T>
__device__ __forceinline__ void keep_alive(const T& v) {      // ablation builds: the value stays computed without being used
    asm volatile("" ::"v"(v));
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ABL (diagnostic builds only, FTE_IGEMM16_ABL): 0 = the kernel; 1 = no MFMAs, 2 = no DMA (stale LDS), 3 = no epilogue, 4 = no
// fragment reads (MFMAs on whatever the registers hold)
// PF = 1: the fragments of sub-step ks + 1 are read while the MFMAs of sub-step ks run (register double buffer), and the scalar
// loads of the next tile's tap offsets are taken before the barrier instead of between the fragment reads
// (body shared by two kernel symbols: igemm16_kernel, and igemm16_bn_kernel = the same with the BN-fusion epilogue compiled in)
template <int BM, int BN, int WM, int WN, int EPI, int NST, int MINW, int ABL, int PF, bool BNM>
__device__ __forceinline__ void igemm16_body(const IgemmParams& p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert((WM * WN == 4 || WM * WN == 8) && TM >= 1 && TN >= 1 && NST >= 2, "4 or 8 waves");
    constexpr int NW = WM * WN, RP = 8 * NW;          // a pass of the block's DMA instructions covers RP rows x 128 B
    constexpr int A_P = BM / RP, B_P = BN / RP;       // DMA instructions per thread and stage
    static_assert(BM % RP == 0 && BN % RP == 0, "tile rows per DMA pass");
    constexpr int L = A_P + B_P;
    constexpr int STAGE = (BM + BN) * ROWB;           // bytes per ring stage
    static_assert((NST - 2) * L <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) char smem16[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int li = lane & 31, lh = lane >> 5;

    // ---- tile coordinates (as igemm_kernel: XCD-aware bijective remap, n-tiles fastest, merged stride-2 dgrad classes) ----
    const int ntn = p.N / BN;
    int bid = blockIdx.x;
    const int split = blockIdx.y;
    {
        const int ntiles = gridDim.x;
        const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    int tap0 = 0, NT = p.a_NT, Kc = p.K, c_ph = p.c_ph, c_pw = p.c_pw, prow = p.prow0;
    if constexpr (EPI == EPI_DGRAD) {
        if (p.ncls > 1) {
            const int cls = bid % p.ncls;
            bid /= p.ncls;
            tap0 = p.cls_tap0[cls];
            NT = p.cls_tap0[cls + 1] - tap0;
            Kc = NT * p.a_KC;
            c_ph = p.cls_ph[cls]; c_pw = p.cls_pw[cls];
            prow += cls * p.cls_mtiles;
        }
    }
    const int mt = bid / ntn, nt_ = bid - mt * ntn;
    const int m0 = p.m_base + mt * BM, n0 = nt_ * BN;
    const int kbeg = split * p.kchunk;
    const int kend = min(Kc, kbeg + p.kchunk);
    const int nk = (kend - kbeg) / BK16;                   // the launcher guarantees whole 64-deep steps

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, p.b_bytes, 0x00020000);

    // ---- per-thread loader state: row (tid >> 3) + 32 i of the tile, LDS slot tid & 7 = source chunk slot ^ ((row >> 1) & 7) ----
    const int a_hw = p.a_OH * p.a_OW;
    const float r_ahw = 1.f / (float)a_hw, r_aow = 1.f / (float)p.a_OW;
    unsigned a_base[A_P];
    int a_mask[A_P];
#pragma unroll
    for (int i = 0; i < A_P; ++i) {
        const int r = (tid >> 3) + RP * i;
        const int m = m0 + r;
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
        const int chunk = (tid & 7) ^ ((r >> 1) & 7);
        a_base[i] = (unsigned)(base + (chunk << 3)) * 2u;
        a_mask[i] = mask;
    }
    unsigned b_base[B_P];
#pragma unroll
    for (int i = 0; i < B_P; ++i) {
        const int r = (tid >> 3) + RP * i;
        const int chunk = (tid & 7) ^ ((r >> 1) & 7);
        b_base[i] = (unsigned)((n0 + r) * p.b_ld + (chunk << 3)) * 2u;
    }

    // K-step sigma covers channel chunk sigma / NT of tap sigma % NT (chunk outer, tap inner: the taps of one 64-channel
    // chunk re-read nearly the same 128-byte row pieces -- L1 / L2 hits); kept incrementally for the step being ISSUED
    int itap = (kbeg / BK16) % NT, ikc = ((kbeg / BK16) / NT) * BK16;
    auto issue = [&](int stage) {
        char* As = smem16 + stage * STAGE + wid * 1024;
        char* Bs = As + BM * ROWB;
        const unsigned toff = (unsigned)((p.a_dh[tap0 + itap] * p.a_IW + p.a_dw[tap0 + itap]) * p.a_ld + ikc) * 2u;    // wave-uniform
        const unsigned boff = (unsigned)(p.b_tapoff[tap0 + itap] + ikc) * 2u;
#pragma unroll
        for (int i = 0; i < A_P; ++i)
            if constexpr (ABL != 2) dma16(rsrcA, As + i * (RP * ROWB), ((a_mask[i] >> itap) & 1) ? a_base[i] + toff : OOB, 0);
#pragma unroll
        for (int i = 0; i < B_P; ++i)
            if constexpr (ABL != 2) dma16(rsrcB, Bs + i * (RP * ROWB), b_base[i], boff);
        if (++itap == NT) { itap = 0; ikc += BK16; }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses: lane -> (row li of its 32-row block, k-half lh); chunk q = 2 ks + lh of the row sits in slot q ^ ((row >> 1) & 7)
    int a_row[TM], b_row[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a_row[i] = wm * (TM * 32) + i * 32 + li;
#pragma unroll
    for (int j = 0; j < TN; ++j) b_row[j] = wn * (TN * 32) + j * 32 + li;

    // One K-step: the 4 k16 sub-steps of the tile in `stage`, each = fragment reads + a quarter of the NEXT ring tile's DMA
    // instructions (into `fill`; an LDS-DMA piece costs the issuing wave ~100-150 cycles, so they are spread between the
    // MFMA groups instead of being issued in one burst ahead of them) + TM x TN MFMAs.
    auto kstep = [&](int stage, int fill, bool fillnext) {
        const char* As = smem16 + stage * STAGE;
        const char* Bs = As + BM * ROWB;
        char* Ad = smem16 + fill * STAGE + wid * 1024;
        char* Bd = Ad + BM * ROWB;
        unsigned toff = 0, boff = 0;
        if (fillnext) {
            toff = (unsigned)((p.a_dh[tap0 + itap] * p.a_IW + p.a_dw[tap0 + itap]) * p.a_ld + ikc) * 2u;    // wave-uniform
            boff = (unsigned)(p.b_tapoff[tap0 + itap] + ikc) * 2u;
        }
#pragma unroll
        for (int ks = 0; ks < BK16 / 16; ++ks) {
            bf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                if constexpr (ABL != 4) fa[i] = *reinterpret_cast<const bf16x8*>(As + a_row[i] * ROWB + (((2 * ks + lh) ^ ((a_row[i] >> 1) & 7)) << 4));
                else asm volatile("" : "=v"(fa[i]));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                if constexpr (ABL != 4) fb[j] = *reinterpret_cast<const bf16x8*>(Bs + b_row[j] * ROWB + (((2 * ks + lh) ^ ((b_row[j] >> 1) & 7)) << 4));
                else asm volatile("" : "=v"(fb[j]));
            if (fillnext) {
#pragma unroll
                for (int idx = 0; idx < L; ++idx) {
                    if (idx * 4 / L != ks) continue;
                    if constexpr (ABL != 2) {
                        if (idx < A_P) dma16(rsrcA, Ad + idx * (RP * ROWB), ((a_mask[idx < A_P ? idx : 0] >> itap) & 1) ? a_base[idx < A_P ? idx : 0] + toff : OOB, 0);
                        else dma16(rsrcB, Bd + (idx - A_P) * (RP * ROWB), b_base[idx >= A_P ? idx - A_P : 0], boff);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if constexpr (ABL != 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                    else { keep_alive(fa[i]); keep_alive(fb[j]); }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (fillnext) {
            if (++itap == NT) { itap = 0; ikc += BK16; }
        }
    };

    // PF = 1: the same K-step with the fragment reads one sub-step ahead of their MFMAs
    auto kstep_pf = [&](int stage, int fill, bool fillnext, unsigned toff, unsigned boff) {
        const char* As = smem16 + stage * STAGE;
        const char* Bs = As + BM * ROWB;
        char* Ad = smem16 + fill * STAGE + wid * 1024;
        char* Bd = Ad + BM * ROWB;
        bf16x8 fa[2][TM], fb[2][TN];
        auto ld = [&](int ks, int b) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[b][i] = *reinterpret_cast<const bf16x8*>(As + a_row[i] * ROWB + (((2 * ks + lh) ^ ((a_row[i] >> 1) & 7)) << 4));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[b][j] = *reinterpret_cast<const bf16x8*>(Bs + b_row[j] * ROWB + (((2 * ks + lh) ^ ((b_row[j] >> 1) & 7)) << 4));
        };
        ld(0, 0);
#pragma unroll
        for (int ks = 0; ks < BK16 / 16; ++ks) {
            if (ks + 1 < BK16 / 16) ld(ks + 1, (ks + 1) & 1);
            if (fillnext) {
#pragma unroll
                for (int idx = 0; idx < L; ++idx) {
                    if (idx * 4 / L != ks) continue;
                    if (idx < A_P) dma16(rsrcA, Ad + idx * (RP * ROWB), ((a_mask[idx < A_P ? idx : 0] >> itap) & 1) ? a_base[idx < A_P ? idx : 0] + toff : OOB, 0);
                    else dma16(rsrcB, Bd + (idx - A_P) * (RP * ROWB), b_base[idx >= A_P ? idx - A_P : 0], boff);
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][i], fb[ks & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (fillnext) {
            if (++itap == NT) { itap = 0; ikc += BK16; }
        }
    };

    // ---- ring: NST - 1 K-steps in flight; step t: wait for (this wave's part of) tile t, barrier (everybody's part has landed
    // AND everybody has finished reading tile t - 1, whose stage is the one refilled during this step), compute tile t while
    // issuing tile t + NST - 1 ----
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nk) issue(s);
    int stage = 0, fill = NST - 1;
    for (int t = 0; t < nk; ++t) {
        unsigned toff = 0, boff = 0;
        if constexpr (PF) {
            if (t + NST - 1 < nk) {
                toff = (unsigned)((p.a_dh[tap0 + itap] * p.a_IW + p.a_dw[tap0 + itap]) * p.a_ld + ikc) * 2u;    // wave-uniform
                boff = (unsigned)(p.b_tapoff[tap0 + itap] + ikc) * 2u;
            }
            asm volatile("" : "+s"(toff), "+s"(boff));               // the scalar loads complete here, under the wait for the tile
        }
        if (t + NST - 1 <= nk) wait_vmcnt<(NST - 2) * L>();          // steady state: the NST - 2 younger tiles stay in flight
        else wait_vmcnt<0>();                                        // last steps: fewer tiles are outstanding
        __builtin_amdgcn_s_barrier();
        if constexpr (PF) kstep_pf(stage, fill, t + NST - 1 < nk, toff, boff);
        else kstep(stage, fill, t + NST - 1 < nk);
        stage = stage + 1 == NST ? 0 : stage + 1;
        fill = fill + 1 == NST ? 0 : fill + 1;
    }
    wait_vmcnt<0>();
    __syncthreads();                       // the epilogue reuses the LDS
    if constexpr (ABL == 3) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) keep_alive(acc[i][j][r]);
        return;
    }
    igemm_epilogue<BM, BN, WM, WN, EPI, true, BNM>(p, acc, reinterpret_cast<float*>(smem16), bid, split, m0, n0, mt, c_ph, c_pw, prow);
}
template <int BM, int BN, int WM, int WN, int EPI, int NST, int MINW, int ABL = 0, int PF = 0>
__global__ __launch_bounds__(64 * WM * WN, MINW) void igemm16_kernel(const IgemmParams p) {
    igemm16_body<BM, BN, WM, WN, EPI, NST, MINW, ABL, PF, false>(p);
}
// conv -> BN pairs of the graph nets (igemm.h "BN fusion")
template <int BM, int BN, int WM, int WN, int EPI, int NST, int MINW>
__global__ __launch_bounds__(64 * WM * WN, MINW) void igemm16_bn_kernel(const IgemmParams p) {
    igemm16_body<BM, BN, WM, WN, EPI, NST, MINW, 0, 0, true>(p);
}

// ---- window variant (3x3, stride 1, TF-SAME; W <= 30): the A operand of a 64-channel chunk is fetched ONCE per tile -------------
// In linear pixel space the source row of (output row m, tap (dh, dw)) is m + dh * W + dw wherever it lies inside the image, so
// the nine taps of a chunk are row-shifted views of the BM + 2 W + 2 rows [m0 - W - 1, m0 + BM + W]: that window is DMA'd once
// per chunk (two buffers: the next chunk's window arrives one piece per K-step under the current chunk's taps), a tap's fragment
// is the row li + W + 1 + dh W + dw of it, and the image edges are a 9-bit mask per lane that zeroes the fragment.  igemm16_kernel
// re-fetches the A tile for every (tap, chunk): 16 of the 32 LDS-DMA pieces of a K-step; here 16 (B) + 24 / 9 (window).
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {           // s_waitcnt takes an immediate: n is wave-uniform and small
    switch (n) {
        case 0: wait_vmcnt<0>(); break;   case 1: wait_vmcnt<1>(); break;   case 2: wait_vmcnt<2>(); break;
        case 3: wait_vmcnt<3>(); break;   case 4: wait_vmcnt<4>(); break;   case 5: wait_vmcnt<5>(); break;
        case 6: wait_vmcnt<6>(); break;   case 7: wait_vmcnt<7>(); break;   case 8: wait_vmcnt<8>(); break;
        case 9: wait_vmcnt<9>(); break;   case 10: wait_vmcnt<10>(); break; case 11: wait_vmcnt<11>(); break;
        default: wait_vmcnt<12>(); break;
    }
}
// NSTB = stages of the B ring (NSTB - 1 K-steps of B in flight)
template <int BM, int BN, int WM, int WN, int EPI, int MINW, int NSTB = 2>
__global__ __launch_bounds__(64 * WM * WN, MINW) void igemm16w_kernel(const IgemmParams p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert((WM * WN == 4 || WM * WN == 8 || WM * WN == 16) && TM >= 1 && TN >= 1 && NSTB >= 2 && NSTB <= 4, "4, 8 or 16 waves");
    constexpr int NW = WM * WN, RP = 8 * NW, B_P = BN / RP, BSTAGE = BN * ROWB, MAXWP = (24 + NW - 1) / NW;
    static_assert(BN % RP == 0, "tile rows per DMA pass");
    extern __shared__ __attribute__((aligned(16))) char smem16[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int li = lane & 31, lh = lane >> 5;

    const int ntn = p.N / BN;
    int bid = blockIdx.x;
    const int split = blockIdx.y;
    {
        const int ntiles = gridDim.x;
        const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int NT = p.a_NT;
    const int mt = bid / ntn, nt_ = bid - mt * ntn;
    const int m0 = p.m_base + mt * BM, n0 = nt_ * BN;
    const int kbeg = split * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int nk = (kend - kbeg) / BK16;
    const int Wd = p.a_IW;
    const int NR = BM + 2 * Wd + 2;                     // window rows
    const int NPC = (NR + 7) >> 3;                      // 1-KiB DMA pieces per window
    const int WINB = NPC * 1024;
    char* const win = smem16;
    char* const bring = smem16 + 2 * WINB;

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, p.b_bytes, 0x00020000);

    // window pieces of this wave: wid, wid + 4, ...; a lane owns row 8 piece + (lane >> 3), LDS slot lane & 7 = source chunk ^ ((row >> 1) & 7)
    unsigned w_off[MAXWP];
#pragma unroll
    for (int i = 0; i < MAXWP; ++i) {
        const int r = (wid + NW * i) * 8 + (lane >> 3);
        const int g = m0 - Wd - 1 + r;
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        w_off[i] = (r < NR && g >= 0 && g < p.M) ? (unsigned)(g * p.a_ld + (chunk << 3)) * 2u : OOB;
    }
    unsigned b_base[B_P];
#pragma unroll
    for (int i = 0; i < B_P; ++i) {
        const int r = (tid >> 3) + RP * i;
        const int chunk = (tid & 7) ^ ((r >> 1) & 7);
        b_base[i] = (unsigned)((n0 + r) * p.b_ld + (chunk << 3)) * 2u;
    }
    auto issueW = [&](int buf, int chunk, int i) {
        if (wid + NW * i < NPC) dma16(rsrcA, win + buf * WINB + (wid + NW * i) * 1024, w_off[i], (unsigned)chunk * (BK16 * 2));
    };

    // fragment rows and their tap masks
    int a_row[TM], fmask[TM], b_row[TN];
    {
        const int a_hw = p.a_OH * p.a_OW;
        const float r_ahw = 1.f / (float)a_hw, r_aow = 1.f / (float)p.a_OW;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            a_row[i] = wm * (TM * 32) + i * 32 + li;
            const int m = m0 + a_row[i];
            int mask = 0;
            if (m < p.M) {
                const int n = fdiv(m, a_hw, r_ahw), rem = m - n * a_hw;
                const int oh = fdiv(rem, p.a_OW, r_aow), ow = rem - oh * p.a_OW;
                for (int t = 0; t < NT; ++t) {
                    const int ih = oh + p.a_dh[t], iw = ow + p.a_dw[t];
                    if (ih >= 0 && ih < p.a_IH && iw >= 0 && iw < p.a_IW) mask |= 1 << t;
                }
            }
            fmask[i] = mask;
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) b_row[j] = wn * (TN * 32) + j * 32 + li;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // K-step sigma = chunk * NT + tap (chunk outer, tap inner)
    const int s_first = kbeg / BK16;
    int ctap = s_first % NT, cchunk = s_first / NT;
    const int last_chunk = (s_first + nk - 1) / NT;
    int wnext = 0;                                       // pieces of the NEXT chunk's window issued so far (per wave)
    int itap = ctap, ichunk = cchunk;                    // the step whose B tile is issued next
    auto issueB = [&](int stage) {
        const unsigned boff = (unsigned)(p.b_tapoff[itap] + ichunk * BK16) * 2u;
#pragma unroll
        for (int i = 0; i < B_P; ++i) dma16(rsrcB, bring + stage * BSTAGE + wid * 1024 + i * (RP * ROWB), b_base[i], boff);
        if (++itap == NT) { itap = 0; ++ichunk; }
    };
    if (nk > 0) {
#pragma unroll
        for (int i = 0; i < MAXWP; ++i) issueW(cchunk & 1, cchunk, i);
#pragma unroll
        for (int s2 = 0; s2 < NSTB - 1; ++s2)
            if (s2 < nk) issueB(s2);
    }
    // DMAs retire in order.  Issued per step s, in this order: the B tile of step s + NSTB - 1, then (at most) one piece of the next
    // chunk's window.  Step t needs its own B tile (issued in step t - NSTB + 1) and every older piece; everything issued in steps
    // t - NSTB + 2 .. t - 1 -- and the window piece of step t - NSTB + 1 -- may stay in flight.
    int wq[NSTB];                                        // window pieces issued at the end of the last NSTB - 1 steps (wq[0] = oldest)
#pragma unroll
    for (int i = 0; i < NSTB; ++i) wq[i] = 0;
    int stage = 0, fill = NSTB - 1;
    for (int t = 0; t < nk; ++t) {
        {
            int allowed = 0;
#pragma unroll
            for (int i = 0; i < NSTB - 1; ++i) allowed += wq[i];
            const int btiles = min(NSTB - 2, nk - 1 - t);                  // younger B tiles in flight (fewer near the end)
            allowed += (btiles > 0 ? btiles : 0) * B_P;
            wait_vmcnt_dyn(allowed);
        }
        __builtin_amdgcn_s_barrier();
        const char* Aw = win + (cchunk & 1) * WINB;
        const char* Bs = bring + stage * BSTAGE;
        int ntap = ctap + 1, nchunk = cchunk;
        if (ntap == NT) { ntap = 0; ++nchunk; }
        const bool fillnext = t + NSTB - 1 < nk;
        char* Bd = bring + fill * BSTAGE + wid * 1024;
        const unsigned boff = fillnext ? (unsigned)(p.b_tapoff[itap] + ichunk * BK16) * 2u : 0u;
        const int offt = Wd + 1 + p.a_dh[ctap] * Wd + p.a_dw[ctap];          // wave-uniform
#pragma unroll
        for (int ks = 0; ks < BK16 / 16; ++ks) {
            bf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int wr = a_row[i] + offt;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(Aw + wr * ROWB + (((2 * ks + lh) ^ ((wr >> 1) & 7)) << 4));
                const bool ok = (fmask[i] >> ctap) & 1;
                const u32x4 raw = __builtin_bit_cast(u32x4, v);
                fa[i] = __builtin_bit_cast(bf16x8, ok ? raw : u32x4{0u, 0u, 0u, 0u});
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[j] = *reinterpret_cast<const bf16x8*>(Bs + b_row[j] * ROWB + (((2 * ks + lh) ^ ((b_row[j] >> 1) & 7)) << 4));
            if (fillnext) {
#pragma unroll
                for (int idx = 0; idx < B_P; ++idx)
                    if (idx * 4 / B_P == ks) dma16(rsrcB, Bd + idx * (RP * ROWB), b_base[idx], boff);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (fillnext && ++itap == NT) { itap = 0; ++ichunk; }
        // window of the next chunk: one piece per step, issued after this step's B pieces; whatever is left on the chunk's last step
        int wlast = 0;
        if (cchunk < last_chunk) {
            const int want = (ctap == NT - 1) ? MAXWP : min(MAXWP, wnext + 1);
            for (; wnext < want; ++wnext) {
                if (wid + NW * wnext < NPC) ++wlast;                 // pieces actually issued (a short window has fewer than MAXWP per wave)
#pragma unroll
                for (int i = 0; i < MAXWP; ++i)
                    if (i == wnext) issueW((cchunk + 1) & 1, cchunk + 1, i);
            }
        }
        if (ntap == 0) wnext = 0;
        ctap = ntap; cchunk = nchunk;
#pragma unroll
        for (int i = 0; i + 1 < NSTB - 1; ++i) wq[i] = wq[i + 1];
        wq[NSTB - 2] = wlast;
        stage = stage + 1 == NSTB ? 0 : stage + 1;
        fill = fill + 1 == NSTB ? 0 : fill + 1;
    }
    wait_vmcnt<0>();
    __syncthreads();
    igemm_epilogue<BM, BN, WM, WN, EPI, true>(p, acc, reinterpret_cast<float*>(smem16), bid, split, m0, n0, mt, p.c_ph, p.c_pw, p.prow0);
}

// ---- persistent variant ------------------------------------------------------------------------------------------------------
// Why: a launch of igemm16_kernel is rounds of co-resident blocks that run in phase -- every block of a round reaches its epilogue at
// about the same time, the epilogues' HBM traffic arrives as one burst per round (ablation: 0.03 ms of a 0.167-ms 14x14x256 forward,
// 0.08 of 0.22 at 28x28x128 = that traffic at HBM speed), a finished block holds its slot until its stores have drained, and its
// successor starts with an address prologue and a cold ring.  Here a block stays resident and walks its XCD's tiles:
//   * the ring never drains: the DMAs of the NEXT tile's first K-step are issued during the last K-step of the current one, so the
//     K-step sequence of a block is one uninterrupted stream across tiles;
//   * the epilogue's inputs (shortcut / skip gradient / previous z, bf16) are fetched into registers BEFORE the last K-step and its
//     stores are fire-and-forget: vmcnt retires in order, the next tile's first DMAs are OLDER than the stores, so step 0 of the next
//     tile waits with a counted vmcnt that leaves the stores in flight -- they drain under the next tile's K loop;
//   * no LDS in the epilogue (the ring is live): the MFMAs run with swapped operands, which leaves the accumulator as
//     lane = output row, registers = 4-column groups; v_permlane32_swap between the two half-waves gives every lane 8 consecutive
//     columns = 16 bytes of a bf16 row per load / store.  Row offsets live in registers; alpha / bias of the tile's columns in LDS.
//   * dalpha / dbias column partials (EPI_DGRAD): a halving butterfly over the 32 row-lanes (16 shuffles per 16 columns), then the
//     usual fixed-order sum over the row-waves through LDS.
// Restrictions (launch16p_ok): no split-K, no merged stride-2 classes, alpha period a multiple of 8.
__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
// ---- epilogue of the swapped-operand kernels (igemm16p / igemm16r / igemm16rw): accumulator lane = output row, registers = columns ----
// acc[i][j][r] of lane (li, lh): output row m0 + wm*TM*32 + i*32 + li (its offset: roff[i], < 0 beyond M), column n0 + wn*TN*32 + j*32 +
// (r & 3) + 8 (r >> 2) + 4 lh.  v_permlane32_swap between the half-waves gives every lane 8 consecutive columns -- 16 bytes of a bf16 row
// per load / store; no LDS patches.  These kernels only take bf16-STORAGE launches (launch16p_ok): the tensors of the epilogue are the
// four 16-bit pointers of EpiPtrs, fetched from the kernel arguments ONCE per kernel -- a `p.Z16` inside the unrolled blocks is a
// scalar load and a wait each time (stamped: 11-13k cycles per 256 x 128 tile, alone on the chip as in a full round, before this).
// colf = [alpha[BN], bias[BN]] of the tile's columns (LDS), red = [2][WM][BN] floats (LDS).  ein0 / ein1: the inputs fetched ahead.
// With EPI_DGRAD and PA the function holds ONE block barrier (every wave of the block must call it, or match it).  Returns the number
// of stores a wave issued (0 after the barrier form: count nothing).
// (the pointers carry the GLOBAL address space explicitly: a pointer rebuilt from the pinned integer is generic to the compiler, and
// generic means flat_load / flat_store -- which count on lgkmcnt as well, so that every wait for an LDS read also waits for the last
// store's round trip to HBM: the first version of this struct ran the epilogue at ~1000 cycles per 16-byte piece)
typedef __attribute__((address_space(1))) unsigned short g_u16;
typedef __attribute__((address_space(1))) const unsigned short g_cu16;
typedef __attribute__((address_space(1))) float g_f32;
typedef __attribute__((address_space(1))) u32x4 g_u32x4;
typedef __attribute__((address_space(1))) const u32x4 g_cu32x4;
struct EpiPtrs {
    g_u16* o0;                   // EPI_FWD: Z16 (may be null)     EPI_DGRAD: RAW16 (may be null)
    g_u16* o1;                   //          Y16                              DZ16
    g_cu16* i0;                  //          R16 (may be null)                ADD16 (may be null)
    g_cu16* i1;                  //          --                               Zin16 (null: no activation gradient)
    g_f32* PA; g_f32* PB;        // EPI_DGRAD: column partials (may be null)
    int has_bias, act, N, prow0, m_base, M;
    int dbg;                     // timing experiments (FTE_IGEMM16_DBG): 32 = no epilogue stores, 64 = no epilogue input loads, 128 = no epilogue
};
// The same tensors as buffer resources, built where they are used from the pinned pointers.  A null tensor or a row beyond M: offset
// EPI_OOB -- the load returns zeros, the store is dropped: the epilogue's memory instructions need no branch and no exec mask, so the
// wait-count pass counts them exactly, and inputs fetched AFTER earlier stores are waited for without draining those.
constexpr unsigned EPI_OOB = 0x80000000u;      // the resources cover 2 GiB (every tensor of these launches is smaller: set_bytes); this offset and its 16 bytes lie beyond
                                               // (0xfffffff0 with 2^32 - 1 records FAULTED on ragged last tiles: the range check's offset + size wraps)
template <class G>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t epi_rsrc(G* q) {
    // (constant record count: a `null ? 0 : ~0` select became a v_cndmask, the resource sat in vector registers and every load / store
    // of the epilogue was wrapped in a waterfall loop -- nullness goes into the OFFSET instead)
    const unsigned long long a = (unsigned long long)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, 0x80000000u, 0x00020000);
}
template <class G, class T>
__device__ __forceinline__ G* pin_sgpr(T* q) {                 // the pointer stays in scalar registers from here on
    unsigned long long v = reinterpret_cast<unsigned long long>(q);
    asm volatile("" : "+s"(v));
    return (G*)v;
}
template <int EPI>
__device__ __forceinline__ EpiPtrs epi_ptrs(const IgemmParams& p) {
    EpiPtrs e;
    if constexpr (EPI == EPI_FWD) {
        e.o0 = pin_sgpr<g_u16>(p.Z16); e.o1 = pin_sgpr<g_u16>(p.Y16); e.i0 = pin_sgpr<g_cu16>(p.R16); e.i1 = nullptr; e.PA = nullptr; e.PB = nullptr;
        e.has_bias = p.bias != nullptr; e.act = p.alpha != nullptr;
    } else {
        e.o0 = pin_sgpr<g_u16>(p.RAW16); e.o1 = pin_sgpr<g_u16>(p.DZ16); e.i0 = pin_sgpr<g_cu16>(p.ADD16); e.i1 = pin_sgpr<g_cu16>(p.Zin16);
        e.PA = pin_sgpr<g_f32>(p.PA); e.PB = pin_sgpr<g_f32>(p.PB);
        e.has_bias = 0; e.act = p.Zin16 != nullptr;
    }
    e.N = p.N; e.prow0 = p.prow0; e.m_base = p.m_base; e.M = p.M; e.dbg = p.ptiles_dbg;
    return e;
}
template <int BM, int BN, int WM, int WN, int EPI, bool WAITALL = true>
__device__ __forceinline__ int epilogue_rows(const EpiPtrs& ep, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], const int (&roff)[BM / WM / 32],
                                             u32x4 (&ein0)[BM / WM / 32][BN / WN / 32][2], u32x4 (&ein1)[BM / WM / 32][BN / WN / 32][2],
                                             const float* colf, float* red, int mt, int n0, int tid, int wm, int wn, int li, int lh,
                                             unsigned long long* est = nullptr, float* carry = nullptr, bool flush = true) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    if (est) est[0] = stamp_now();
    // Every fetched-ahead input is "used" here, in straight-line code BEFORE the first store: the compiler then waits for the loads
    // once, now.  Left to the first real use inside the conditional blocks below, its wait-count pass cannot tell how many stores were
    // issued since and writes vmcnt(0) before each piece -- every piece then waits for the stores of the piece before it to drain
    // (stamped: 11k cycles per 256 x 128 tile even alone on the chip).
    // (WAITALL = false: the caller fetched the inputs of the later row blocks AFTER the K loop; every memory instruction below is
    // unconditional, the pass counts the stores issued since and waits for those inputs with an exact vmcnt)
    if constexpr (WAITALL) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    asm volatile("" : "+v"(ein0[i][j][q]));
                    if constexpr (EPI == EPI_DGRAD) asm volatile("" : "+v"(ein1[i][j][q]));
                }
    }
    if (est) est[1] = stamp_now();
    // column partials of this wave (dalpha, dbias), one per 8-column piece.  `carry` (4 TN floats of the caller, zero at the start): the
    // sums run on across the tiles of a resident block and reach the partial rows only when `flush` is set
    float csa[TN][2], csb[TN][2];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q) { csa[j][q] = carry ? carry[2 * j + q] : 0.f; csb[j][q] = carry ? carry[2 * TN + 2 * j + q] : 0.f; }
    const bool act = ep.act;
    const __amdgpu_buffer_rsrc_t r_o0 = epi_rsrc(ep.o0), r_o1 = epi_rsrc(ep.o1), r_i0 = epi_rsrc(ep.i0), r_i1 = epi_rsrc(ep.i1);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int off = roff[i];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f32x16 a = acc[i][j];
            if (i == 0 && j == 0) asm volatile("s_nop 15\n\ts_nop 15");      // MFMA results -> VALU reads (the asm below hides the hazard from the compiler)
            // half-wave exchange: lower lanes end with columns 0-7 (a[0..7]) and 16-23 (a[8..15]) of the 32-column block,
            // upper lanes with 8-15 and 24-31
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // (inline asm: hipcc 7.2 folds the two results of __builtin_amdgcn_permlane32_swap into one once they
                    // are cast to float -- every column came out as column 0 of its block)
                    float x = a[8 * g + e], y = a[8 * g + 4 + e];
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
                    a[8 * g + e] = x;
                    a[8 * g + 4 + e] = y;
                }
            if (est && i == 0 && j == 0) est[2] = stamp_now();
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float sa8[8], sb8[8];
                if (est && i == 0 && j == 0 && q == 1) est[5] = stamp_now();
                const int cl = wn * (TN * 32) + j * 32 + 16 * q + 8 * lh;
                const int o = (off < 0 ? 0 : off) + n0 + cl;            // (32-bit: the resources cover 2 GiB)
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = a[8 * q + e];
                const f32x4 al0 = *reinterpret_cast<const f32x4*>(colf + cl), al1 = *reinterpret_cast<const f32x4*>(colf + cl + 4);
                const float al[8] = {al0[0], al0[1], al0[2], al0[3], al1[0], al1[1], al1[2], al1[3]};
                auto bf8 = [&](const u32x4& h, float (&dst)[8]) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { dst[2 * e] = __builtin_bit_cast(float, h[e] << 16); dst[2 * e + 1] = __builtin_bit_cast(float, h[e] & 0xffff0000u); }
                };
                auto st16 = [&](const __amdgpu_buffer_rsrc_t& dst, bool there, const float (&x)[8]) {
                    // (nontemporal stores measured: forward unchanged, data gradient 0.221 -> 0.243 ms at 28x28x128 -- its outputs are the
                    // next kernels' inputs)
                    // On the 256 x 128 tile the ADDRESS pattern of these stores is not what costs: the same pieces sent to 8 rows x 128
                    // contiguous bytes (wrong placement, same bytes) ran no faster (28x28x128 forward 0.168 -> 0.179 ms), while no stores at
                    // all (FTE_IGEMM16_DBG=32) gives 0.137, no input loads (=64) 0.143 of 0.158, no epilogue (=128) 0.127 -- the bytes of
                    // a tile leave in a burst at the end of its K loop, and the next tile's operand DMAs queue behind them.
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{pkbf(x[0], x[1]), pkbf(x[2], x[3]), pkbf(x[4], x[5]), pkbf(x[6], x[7])}, dst,
                                                           (off >= 0 && there && !(ep.dbg & 32)) ? (unsigned)o * 2u : EPI_OOB, 0, 0);
                };
                if constexpr (EPI == EPI_FWD) {
                    if (ep.has_bias) {
                        const f32x4 b0 = *reinterpret_cast<const f32x4*>(colf + BN + cl), b1 = *reinterpret_cast<const f32x4*>(colf + BN + cl + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
                    }
                    if (est && i == 0 && j == 0 && q == 0) est[3] = stamp_now();
                    st16(r_o0, ep.o0 != nullptr, v);
                    if (est && i == 0 && j == 0 && q == 0) est[4] = stamp_now();
                    if (act) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : al[e] * v[e];
                    }
                    float rs[8];
                    bf8(ein0[i][j][q], rs);                      // zeros when there is no shortcut
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += rs[e];
                    st16(r_o1, ep.o1 != nullptr, v);
                } else {
                    float ad[8], z[8];
                    bf8(ein0[i][j][q], ad);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += ad[e];
                    st16(r_o0, ep.o0 != nullptr, v);
                    if (act) {
                        bf8(ein1[i][j][q], z);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const bool in = off >= 0;
                            sa8[e] = in ? v[e] * fminf(z[e], 0.f) : 0.f;
                            v[e] *= prelu_slope(z[e], al[e]);
                            sb8[e] = in ? v[e] : 0.f;
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { sa8[e] = 0.f; sb8[e] = 0.f; }
                    }
                    st16(r_o1, ep.o1 != nullptr, v);
                    if (ep.PA && act) {
                        // halving butterfly over the 32 row-lanes, per 8-column piece (16 values at a time cost 32 live registers beside
                        // the accumulators and the fetched-ahead inputs): lane li ends with the sum of column (li >> 2) & 7 of the piece
                        auto fold = [&](float (&x)[8]) -> float {
                            float w4[4], w2[2], w1;
                            const bool h16 = li & 16, h8 = li & 8, h4 = li & 4;
#pragma unroll
                            for (int c = 0; c < 4; ++c) w4[c] = (h16 ? x[c + 4] : x[c]) + __shfl_xor(h16 ? x[c] : x[c + 4], 16);
#pragma unroll
                            for (int c = 0; c < 2; ++c) w2[c] = (h8 ? w4[c + 2] : w4[c]) + __shfl_xor(h8 ? w4[c] : w4[c + 2], 8);
                            w1 = (h4 ? w2[1] : w2[0]) + __shfl_xor(h4 ? w2[0] : w2[1], 4);
                            w1 += __shfl_xor(w1, 2);
                            return w1 + __shfl_xor(w1, 1);
                        };
                        csa[j][q] += fold(sa8);
                        csb[j][q] += fold(sb8);
                    }
                }
            }
        }
    }
    int nst = TM * TN * 2 * 2;                                         // stores issued per wave (those of a null tensor are issued and dropped)
    if constexpr (EPI == EPI_DGRAD) {
        if (ep.PA && !flush) {
            // A resident block whose next tile has the same columns keeps its sums in registers: this tile's partial rows are ZERO
            // (fire-and-forget stores; the rows' total is what the reduction reads).  The flush below costs a block barrier -- and
            // hipcc's __syncthreads waits for every store of the tile to drain first, i.e. nothing of the epilogue overlapped the
            // next tile's K loop (56x56x64 data gradient 0.306 ms against 0.203 for the forward of the same bytes).
            constexpr int NH = BM / 128;
            if (tid < BN * NH) {
                const int h = tid / BN, c = tid - h * BN;
                if (ep.m_base + (mt * NH + h) * 128 < ep.M) {
                    const long o = (long)(ep.prow0 + mt * NH + h) * ep.N + n0 + c;
                    ep.PA[o] = 0.f;
                    if (ep.PB) ep.PB[o] = 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q) { carry[2 * j + q] = csa[j][q]; carry[2 * TN + 2 * j + q] = csb[j][q]; }
        } else if (ep.PA) {    // column partials (dalpha, dbias) per 128 rows -- the planner's partial rows -- reduced later in a fixed order
            if (carry) {
#pragma unroll
                for (int j = 0; j < 4 * TN; ++j) carry[j] = 0.f;
            }
            if ((li & 3) == 0) {
                const int c8 = li >> 2;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int c = wn * (TN * 32) + j * 32 + 16 * q + 8 * lh + c8;
                        red[wm * BN + c] = csa[j][q];
                        red[(WM + wm) * BN + c] = csb[j][q];
                    }
            }
            __syncthreads();
            constexpr int NH = BM / 128, WH = WM / NH;          // 128-row parts of the tile, row-waves per part
            if (tid < BN * NH) {
                const int h = tid / BN, c = tid - h * BN;
                float sa = 0.f, sb = 0.f;
#pragma unroll
                for (int w = 0; w < WH; ++w) {
                    sa += red[(h * WH + w) * BN + c];
                    sb += red[(WM + h * WH + w) * BN + c];
                }
                if (ep.m_base + (mt * NH + h) * 128 < ep.M) {
                    const long o = (long)(ep.prow0 + mt * NH + h) * ep.N + n0 + c;
                    ep.PA[o] = sa;
                    if (ep.PB) ep.PB[o] = sb;
                }
            }
            nst = 0;          // the partials' stores are not counted: wait for everything
        }
    }
    return nst;
}

// ---- the same epilogue with ROW-COALESCED memory instructions (igemm16rw, STG > 0) ----------------------------------------------
// In epilogue_rows a 16-byte load / store has its 64 lanes on 64 different rows.  scripts/probes/store_patterns.hip: such an
// instruction costs a CU ~65 cycles (store) / ~73 (load) -- one per touched line -- against 19 / 44 when eight lanes share a 128-byte
// row; the 192 (forward) / 256 (data gradient) instructions of a 256 x 128 tile are the 10-12k / 18-20k cycles the stamped builds
// show for the epilogue of a 45-55k-cycle tile, with the MFMA pipe idle (every consumer wave is in its epilogue at the same time).
// Here the fp32 accumulators of a wave (32 TM rows x 64 columns, lane = row) go through LDS STG rows at a time -- masked
// ds_write_b128 of the 4-column register groups, 16-byte chunk ch of row r at chunk position ch ^ (r & 15): conflict-free both ways
// -- and come back with lane = (row r = lane >> 3 of eight, column chunk c = lane & 7 of eight columns): every bf16 load / store of
// the epilogue then covers 8 rows x 128 contiguous bytes, and the half-wave exchange is gone.  Unit k = rows 8k .. 8k + 7 of the wave.
// dalpha / dbias: a lane sums its eight columns over every unit (and over the tiles of a resident block, `carry`); the lanes of a
// column meet (xor 8, 16, 32) only when the partial rows are written.
// stg: this wave's STG x 256 bytes.  With KS = 2 they lie in the half of the B ring the last interval read (free until the loaders
// pass the next tile's first barrier; the caller holds a block barrier between the last K-step and this function).
template <int BM, int BN, int WM, int WN, int EPI, int STG>
__device__ __forceinline__ void epilogue_staged(const EpiPtrs& ep, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], char* stg, long base, int m_wave,
                                                u32x4 (&ein0)[BM / WM / 8], u32x4 (&ein1)[BM / WM / 8], const float* colf, float* red,
                                                int mt, int n0, int tid, int wm, int wn, int lane, float (&carry)[16], bool flush) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32, NU = TM * 4;
    static_assert(TN == 2 && (STG == 8 || STG == 16), "64-column slabs: eight lanes x eight columns per row");
    const int li = lane & 31, lh = lane >> 5, rr = lane >> 3, cq = lane & 7;
    // every fetched-ahead input "used" once, before the first store (see epilogue_rows)
#pragma unroll
    for (int k = 0; k < NU; ++k) {
        asm volatile("" : "+v"(ein0[k]));
        if constexpr (EPI == EPI_DGRAD) asm volatile("" : "+v"(ein1[k]));
    }
    const int cl = wn * 64 + 8 * cq;                   // this lane's first column within the tile
    const f32x4 al0 = *reinterpret_cast<const f32x4*>(colf + cl), al1 = *reinterpret_cast<const f32x4*>(colf + cl + 4);
    const float al[8] = {al0[0], al0[1], al0[2], al0[3], al1[0], al1[1], al1[2], al1[3]};
    float bi[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI == EPI_FWD) {
        if (ep.has_bias) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(colf + BN + cl), b1 = *reinterpret_cast<const f32x4*>(colf + BN + cl + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { bi[e] = b0[e]; bi[4 + e] = b1[e]; }
        }
    }
    const bool act = ep.act;
    auto bf8 = [&](const u32x4& h, float (&dst)[8]) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { dst[2 * e] = __builtin_bit_cast(float, h[e] << 16); dst[2 * e + 1] = __builtin_bit_cast(float, h[e] & 0xffff0000u); }
    };
#pragma unroll
    for (int r0 = 0; r0 < TM * 32; r0 += STG) {
        const int i = r0 / 32, lo = r0 % 32;
        if (li >= lo && li < lo + STG) {               // the rows of this pass: lane = row
            const int r = li - lo;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = j * 8 + 2 * g + lh;
                    *reinterpret_cast<f32x4*>(stg + r * 256 + ((ch ^ (r & 15)) << 4)) =
                        f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                }
        }
        // lanes exchange data through LDS: without the fence the compiler may keep a lane that did not write in this pass on the
        // value it read in the pass before (a thread's own view of memory it did not touch) -- it did, in the forward instantiation
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int u = 0; u < STG / 8; ++u) {
            const int k = r0 / 8 + u, r = 8 * u + rr;
            const f32x4 a = *reinterpret_cast<const f32x4*>(stg + r * 256 + (((2 * cq) ^ (r & 15)) << 4));
            const f32x4 b = *reinterpret_cast<const f32x4*>(stg + r * 256 + (((2 * cq + 1) ^ (r & 15)) << 4));
            float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
            const bool in = m_wave + 8 * k + rr < ep.M;
            const long o = base + (long)(8 * k) * ep.N;           // (c_ld = N for the layers this kernel takes)
            auto st16 = [&](g_u16* dst, const float (&x)[8]) __attribute__((always_inline)) {
                if (dst && in) *(g_u32x4*)(dst + o) = u32x4{pkbf(x[0], x[1]), pkbf(x[2], x[3]), pkbf(x[4], x[5]), pkbf(x[6], x[7])};
            };
            if constexpr (EPI == EPI_FWD) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += bi[e];
                st16(ep.o0, v);
                if (act) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : al[e] * v[e];
                }
                float rs[8];
                bf8(ein0[k], rs);                        // zeros when there is no shortcut
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += rs[e];
                st16(ep.o1, v);
            } else {
                float ad[8], z[8];
                bf8(ein0[k], ad);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += ad[e];
                st16(ep.o0, v);
                if (act) {
                    bf8(ein1[k], z);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        carry[e] += in ? v[e] * fminf(z[e], 0.f) : 0.f;
                        v[e] *= prelu_slope(z[e], al[e]);
                        carry[8 + e] += in ? v[e] : 0.f;
                    }
                }
                st16(ep.o1, v);
            }
            __builtin_amdgcn_sched_barrier(0);      // one unit at a time: scheduled across units, the unpacked inputs of all eight are live at once (94 spilled registers)
        }
    }
    if constexpr (EPI == EPI_DGRAD) {
        constexpr int NH = BM / 128, WH = WM / NH;          // 128-row parts of the tile, row-waves per part
        if (ep.PA && !flush) {
            // the block's next tile has the same columns: the sums stay in registers, this tile's partial rows are zero
            if (tid < BN * NH) {
                const int h = tid / BN, c = tid - h * BN;
                if (ep.m_base + (mt * NH + h) * 128 < ep.M) {
                    const long o = (long)(ep.prow0 + mt * NH + h) * ep.N + n0 + c;
                    ep.PA[o] = 0.f;
                    if (ep.PB) ep.PB[o] = 0.f;
                }
            }
        } else if (ep.PA) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float x = carry[e];
                x += __shfl_xor(x, 8);
                x += __shfl_xor(x, 16);
                x += __shfl_xor(x, 32);
                carry[e] = x;
            }
            if (lane < 8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    red[wm * BN + cl + e] = carry[e];
                    red[(WM + wm) * BN + cl + e] = carry[8 + e];
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) carry[e] = 0.f;
            __syncthreads();
            if (tid < BN * NH) {
                const int h = tid / BN, c = tid - h * BN;
                float sa = 0.f, sb = 0.f;
#pragma unroll
                for (int w = 0; w < WH; ++w) {
                    sa += red[(h * WH + w) * BN + c];
                    sb += red[(WM + h * WH + w) * BN + c];
                }
                if (ep.m_base + (mt * NH + h) * 128 < ep.M) {
                    const long o = (long)(ep.prow0 + mt * NH + h) * ep.N + n0 + c;
                    ep.PA[o] = sa;
                    if (ep.PB) ep.PB[o] = sb;
                }
            }
        }
    }
}

// a counted vmcnt whose count is only known at run time (wave-uniform): the largest listed count <= n (waiting for more is safe)
__device__ __forceinline__ void wait_vmcnt_upto(int n) {
    if (n >= 16) {
        if (n >= 48) wait_vmcnt<48>(); else if (n >= 40) wait_vmcnt<40>(); else if (n >= 32) wait_vmcnt<32>();
        else if (n >= 28) wait_vmcnt<28>(); else if (n >= 24) wait_vmcnt<24>(); else if (n >= 20) wait_vmcnt<20>(); else wait_vmcnt<16>();
    } else if (n >= 6) {
        if (n >= 14) wait_vmcnt<14>(); else if (n >= 12) wait_vmcnt<12>(); else if (n >= 10) wait_vmcnt<10>();
        else if (n >= 8) wait_vmcnt<8>(); else wait_vmcnt<6>();
    } else {
        if (n >= 4) wait_vmcnt<4>(); else if (n >= 3) wait_vmcnt<3>(); else if (n >= 2) wait_vmcnt<2>();
        else if (n >= 1) wait_vmcnt<1>(); else wait_vmcnt<0>();
    }
}

// NST = ring stages (NST - 1 K-steps in flight, across tile boundaries)
// DBG = 1 (diagnostic launches only, FTE_IGEMM16_STAMP): s_memtime stamps around the wait / barrier / body of every K-step
// PFD = sub-steps by which the fragment reads run ahead of their MFMAs (1..3; 3 = every fragment of the K-step is requested at
// its start): with two MFMAs per sub-step (wave tile 32 x 64) one sub-step of lookahead is 64 cycles, less than an LDS read takes
template <int BM, int BN, int WM, int WN, int EPI, int NST, int MINW, int PFD = 1, int DBG = 0>
__global__ __launch_bounds__(64 * WM * WN, MINW) void igemm16p_kernel(const IgemmParams p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int NW = WM * WN, RP = 8 * NW, A_P = BM / RP, B_P = BN / RP, L = A_P + B_P;
    static_assert(BM % RP == 0 && BN % RP == 0 && L <= 12 && NST >= 2 && NST <= 5, "tile rows per DMA pass");
    constexpr int STAGE = (BM + BN) * ROWB;
    constexpr int NQ = NST - 2;             // K-steps whose DMAs may stay in flight behind the awaited one
    constexpr int DSUB = NST == 2 ? 2 : 4;  // the DMAs of a K-step go out during its first DSUB sub-steps (two-stage ring: early, they are awaited next step)
    extern __shared__ __attribute__((aligned(16))) char smem16[];
    float* const colf = reinterpret_cast<float*>(smem16 + NST * STAGE);    // [2][BN]: alpha, bias of the tile's columns
    float* const red = colf + 2 * BN;                                      // [2][WM][BN]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int li = lane & 31, lh = lane >> 5;

    // ---- this block's tiles: XCD x (= blockIdx.x & 7: workgroups are dealt to the XCDs round-robin) owns a contiguous range of
    // the launch's tiles (n-tiles fastest), its blocks walk that range with stride gridDim.x / 8 ----
    const int ntn = p.N / BN;
    const int xcd = blockIdx.x & 7, per = gridDim.x >> 3;
    const int q8 = p.ptiles >> 3, r8 = p.ptiles & 7;
    const int xbase = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int xcnt = q8 + (xcd < r8 ? 1 : 0);
    int idx = blockIdx.x >> 3;
    if (idx >= xcnt) return;
    const int NT = p.a_NT, nk = p.K / BK16;

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, p.b_bytes, 0x00020000);

    // ---- loader state of one tile (as igemm16_kernel) ----
    const int a_hw = p.a_OH * p.a_OW;
    const float r_ahw = 1.f / (float)a_hw, r_aow = 1.f / (float)p.a_OW;
    unsigned a_base[A_P], b_base[B_P];
    int a_mask[A_P];
    auto setup = [&](int tile) {
        const int mt = tile / ntn, nt_ = tile - mt * ntn;
        const int m0 = p.m_base + mt * BM, n0 = nt_ * BN;
#pragma unroll
        for (int i = 0; i < A_P; ++i) {
            const int r = (tid >> 3) + RP * i;
            const int m = m0 + r;
            int base = 0, mask = 0;
            if (m < p.M) {
                const int n = fdiv(m, a_hw, r_ahw), rem = m - n * a_hw;
                const int oh = fdiv(rem, p.a_OW, r_aow), ow = rem - oh * p.a_OW;
                const int ih0 = oh * p.a_stride, iw0 = ow * p.a_stride;
                base = ((n * p.a_IH + ih0) * p.a_IW + iw0) * p.a_ld;
                for (int t = 0; t < NT; ++t) {
                    const int ih = ih0 + p.a_dh[t], iw = iw0 + p.a_dw[t];
                    if (ih >= 0 && ih < p.a_IH && iw >= 0 && iw < p.a_IW) mask |= 1 << t;
                }
            }
            const int chunk = (tid & 7) ^ ((r >> 1) & 7);
            a_base[i] = (unsigned)(base + (chunk << 3)) * 2u;
            a_mask[i] = mask;
        }
#pragma unroll
        for (int i = 0; i < B_P; ++i) {
            const int r = (tid >> 3) + RP * i;
            const int chunk = (tid & 7) ^ ((r >> 1) & 7);
            b_base[i] = (unsigned)((n0 + r) * p.b_ld + (chunk << 3)) * 2u;
        }
    };
    int itap = 0, ikc = 0;                 // the K-step being ISSUED (chunk outer, tap inner)
    // byte offsets of the taps, one per lane: a K-step's offsets come from v_readlane, not from three dependent scalar loads
    // (measured with the stamped build: ~400 cycles per K-step on the critical path between two barriers)
    int tapA = 0, tapB = 0;
    if (lane < NT) {
        tapA = ((p.a_dh[lane] * p.a_IW + p.a_dw[lane]) * p.a_ld) * 2;
        tapB = p.b_tapoff[lane] * 2;
    }
    const bool dbg_skipA = p.ptiles_dbg & 1;      // timing experiment (wrong results): A tiles fetched for tap 0 only
    const bool dbg_skipB = p.ptiles_dbg & 2;      // ... no B tiles after the first

    f32x16 acc[TM][TN];
    int a_row[TM], b_row[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a_row[i] = wm * (TM * 32) + i * 32 + li;
#pragma unroll
    for (int j = 0; j < TN; ++j) b_row[j] = wn * (TN * 32) + j * 32 + li;

    // One K-step: fragments read one sub-step ahead of their MFMAs, a quarter of the next ring tile's DMAs per sub-step.  The
    // MFMA operands are swapped (B rows first): accumulator lane = output ROW li of the 32x32 block, register r = column
    // (r & 3) + 8 (r >> 2) + 4 lh.
    auto kstep = [&](int stage, bool fillnext, unsigned toff, unsigned boff) {
        const char* As = smem16 + stage * STAGE;
        const char* Bs = As + BM * ROWB;
        char* Ad = smem16 + (stage == 0 ? NST - 1 : stage - 1) * STAGE + wid * 1024;      // the stage read in the previous K-step
        char* Bd = Ad + BM * ROWB;
        constexpr int NB = PFD + 1;
        bf16x8 fa[NB][TM], fb[NB][TN];
        auto ld = [&](int ks, int b) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[b][i] = *reinterpret_cast<const bf16x8*>(As + a_row[i] * ROWB + (((2 * ks + lh) ^ ((a_row[i] >> 1) & 7)) << 4));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[b][j] = *reinterpret_cast<const bf16x8*>(Bs + b_row[j] * ROWB + (((2 * ks + lh) ^ ((b_row[j] >> 1) & 7)) << 4));
        };
#pragma unroll
        for (int k0 = 0; k0 < PFD; ++k0) ld(k0, k0);
#pragma unroll
        for (int ks = 0; ks < BK16 / 16; ++ks) {
            if (ks + PFD < BK16 / 16) ld(ks + PFD, (ks + PFD) % NB);
            if (fillnext) {
#pragma unroll
                for (int d = 0; d < L; ++d) {
                    if (d * DSUB / L != ks) continue;
                    if (d < A_P) { if (!(dbg_skipA && itap != 0)) dma16(rsrcA, Ad + d * (RP * ROWB), ((a_mask[d < A_P ? d : 0] >> itap) & 1) ? a_base[d < A_P ? d : 0] + toff : OOB, 0); }
                    else if (!dbg_skipB) dma16(rsrcB, Bd + (d - A_P) * (RP * ROWB), b_base[d >= A_P ? d - A_P : 0], boff);
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks % NB][j], fa[ks % NB][i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (fillnext) {
            if (++itap == NT) { itap = 0; ikc += BK16; }
        }
    };

    // ---- first tile: loader state, the DMAs of its first NST - 1 K-steps (the launcher guarantees nk >= NST - 1) ----
    setup(xbase + idx);
#pragma unroll
    for (int s0 = 0; s0 < NST - 1; ++s0) {
        char* As = smem16 + s0 * STAGE + wid * 1024;
        char* Bs = As + BM * ROWB;
        const unsigned toff = (unsigned)(__builtin_amdgcn_readlane(tapA, itap) + ikc * 2);
        const unsigned boff = (unsigned)(__builtin_amdgcn_readlane(tapB, itap) + ikc * 2);
#pragma unroll
        for (int i = 0; i < A_P; ++i) dma16(rsrcA, As + i * (RP * ROWB), ((a_mask[i] >> itap) & 1) ? a_base[i] + toff : OOB, 0);
#pragma unroll
        for (int i = 0; i < B_P; ++i) dma16(rsrcB, Bs + i * (RP * ROWB), b_base[i], boff);
        if (++itap == NT) { itap = 0; ikc += BK16; }
    }
    const bool in16 = EPI == EPI_FWD ? p.R16 != nullptr : (p.ADD16 != nullptr || p.Zin16 != nullptr);
    const EpiPtrs ep = epi_ptrs<EPI>(p);
    const int n_ein = !in16 ? 0 : TM * TN * 2 * (EPI == EPI_FWD ? 1 : (p.ADD16 ? 1 : 0) + (p.Zin16 ? 1 : 0));
    int stage = 0;
    // vmcnt retires in issue order: a K-step's tile has landed once at most as many operations are outstanding as were issued
    // AFTER its DMAs.  qn[k] = vector-memory operations issued in each of the last NQ K-steps (oldest first); carry = the
    // previous tile's epilogue stores, younger than the awaited DMAs for the first NST - 1 steps of a tile.
    int qn[NQ > 0 ? NQ : 1];
#pragma unroll
    for (int k = 0; k < NQ; ++k) qn[k] = L;
    int carry = 0;
    int dbg_step = 0;
    int n0_colf = -1;

    for (;;) {
        const int tile = xbase + idx;
        const int mt = tile / ntn, nt_ = tile - mt * ntn;
        const int m0 = p.m_base + mt * BM, n0 = nt_ * BN;
        const int nidx = idx + per;
        const bool hasnext = nidx < xcnt;

        // output row of this lane per 32-row block
        int roff[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * (TM * 32) + i * 32 + li;
            int off = -1;
            if (m < p.M) {
                if (p.c_OH == 0) {
                    off = m * p.c_ld;
                } else {
                    const int hw = p.c_OH * p.c_OW;
                    const int n = fdiv(m, hw, 1.f / (float)hw), rem = m - n * hw;
                    const int oh = fdiv(rem, p.c_OW, 1.f / (float)p.c_OW), ow = rem - oh * p.c_OW;
                    off = ((n * p.c_FH + oh * p.c_step + p.c_ph) * p.c_FW + ow * p.c_step + p.c_pw) * p.c_ld;
                }
            }
            roff[i] = off;
        }
        // alpha / bias of the tile's columns -> LDS (read in the epilogue, nk barriers later; the previous tile's epilogue reads
        // are separated from this write by its closing barrier)
        // (only when the column tile changes: a load here waits, vmcnt being in order, for every store of the previous tile's epilogue)
        if (n0 != n0_colf) {
            if (tid < BN) {
                float al = 1.f, bi = 0.f;
                if constexpr (EPI == EPI_FWD) {
                    if (p.alpha) al = p.alpha[n0 + tid];
                    if (p.bias) bi = p.bias[n0 + tid];
                } else {
                    if (p.alpha) al = p.alpha[(n0 + tid) % p.amod];
                }
                colf[tid] = al;
                colf[BN + tid] = bi;
            }
            n0_colf = n0;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        u32x4 ein0[TM][TN][2], ein1[TM][TN][2];           // bf16 epilogue inputs, fetched before the last K-step
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q) { ein0[i][j][q] = u32x4{0u, 0u, 0u, 0u}; ein1[i][j][q] = u32x4{0u, 0u, 0u, 0u}; }

        for (int t = 0; t < nk; ++t) {
            const bool last = t == nk - 1;
            const bool fillnext = t + NST - 1 < nk || hasnext;           // the K-step NST - 1 ahead: this tile's, or the next tile's
            if (t + NST - 1 == nk && hasnext) { setup(xbase + nidx); itap = 0; ikc = 0; }
            const unsigned toff = (unsigned)(__builtin_amdgcn_readlane(tapA, itap) + ikc * 2);      // wave-uniform
            const unsigned boff = (unsigned)(__builtin_amdgcn_readlane(tapB, itap) + ikc * 2);
            unsigned long long st0 = 0, st1 = 0, st2 = 0;
            if constexpr (DBG) { __builtin_amdgcn_sched_barrier(0); st0 = stamp_now(); __builtin_amdgcn_sched_barrier(0); }
            {
                int allowed = t < NST - 1 ? carry : 0;
#pragma unroll
                for (int k = 0; k < NQ; ++k) allowed += qn[k];
                wait_vmcnt_upto(allowed);
            }
            if constexpr (DBG) { __builtin_amdgcn_sched_barrier(0); st1 = stamp_now(); __builtin_amdgcn_sched_barrier(0); }
            __builtin_amdgcn_s_barrier();
            if constexpr (DBG) { __builtin_amdgcn_sched_barrier(0); st2 = stamp_now(); __builtin_amdgcn_sched_barrier(0); }
            if (last && in16) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const long o = (long)(roff[i] < 0 ? 0 : roff[i]) + n0 + wn * (TN * 32) + j * 32 + 16 * q + 8 * lh;
                            if constexpr (EPI == EPI_FWD) {
                                if (ep.i0) ein0[i][j][q] = *(g_cu32x4*)(ep.i0 + o);
                            } else {
                                if (ep.i0) ein0[i][j][q] = *(g_cu32x4*)(ep.i0 + o);
                                if (ep.i1) ein1[i][j][q] = *(g_cu32x4*)(ep.i1 + o);
                            }
                        }
            }
            kstep(stage, fillnext, toff, boff);
            if constexpr (DBG) {
                __builtin_amdgcn_sched_barrier(0);
                const unsigned long long st3 = stamp_now();
                __builtin_amdgcn_sched_barrier(0);
                const int sb = blockIdx.x == 0 ? 0 : (blockIdx.x == 8 ? 1 : -1), sw = wid == 0 ? 0 : (wid == NW - 1 ? 1 : -1);
                if (sb >= 0 && sw >= 0 && dbg_step < 80 && lane == 0) {
                    unsigned long long* o = reinterpret_cast<unsigned long long*>(p.PW) + ((sb * 2 + sw) * 80 + dbg_step) * 4;
                    o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3;
                }
                ++dbg_step;
            }
            stage = stage + 1 == NST ? 0 : stage + 1;
#pragma unroll
            for (int k = 0; k + 1 < NQ; ++k) qn[k] = qn[k + 1];
            if (NQ > 0) qn[NQ > 0 ? NQ - 1 : 0] = (fillnext ? L : 0) + (last ? n_ein : 0);
        }

        // ---- epilogue (no LDS patches, no block barrier in the forward form) ----
        const int nst = epilogue_rows<BM, BN, WM, WN, EPI>(ep, acc, roff, ein0, ein1, colf, red, mt, n0, tid, wm, wn, li, lh);
        if (!hasnext) break;
        // colf is rewritten at the top of the next tile if its column tile differs: every wave must be done reading it
        {
            const int ntile = xbase + nidx;
            if ((ntile - (ntile / ntn) * ntn) * BN != n0) __syncthreads();
        }
        carry = nst;
        idx = nidx;
    }
}

template <int BM, int BN, int WM, int WN, int EPI, int NST, int MINW, int BPC, int PFD = 1>
hipError_t launch16p(const IgemmParams& p, hipStream_t st) {
    const int mt = (p.M - p.m_base + BM - 1) / BM, nt = p.N / BN;
    IgemmParams q = p;
    q.ptiles = mt * nt;
    static const int dbg = getenv("FTE_IGEMM16_DBG") ? atoi(getenv("FTE_IGEMM16_DBG")) : 0;
    q.ptiles_dbg = dbg;
    const size_t lds = (size_t)NST * (BM + BN) * ROWB + (size_t)(2 * BN + 2 * WM * BN) * sizeof(float);
    auto kern = igemm16p_kernel<BM, BN, WM, WN, EPI, NST, MINW, PFD>;
    if (igemm_prof_on()) { const int ta[9] = {BM, BN, WM, WN, EPI, NST, MINW, PFD, 0}; igemm_note_symbol("igemm16p_kernel", ta, 9); }
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
        cus = prop.multiProcessorCount;
    }
    const int per_xcd = (cus / 8) * BPC;                               // resident blocks per XCD
    const int need = (q.ptiles + 7) / 8;
    const int grid = 8 * (need < per_xcd ? need : per_xcd);
    static const bool stamps = getenv("FTE_IGEMM16_STAMP") != nullptr;
    if (stamps) {              // diagnostic: the stamped build of this configuration, its table on stderr
        auto dk = igemm16p_kernel<BM, BN, WM, WN, EPI, NST, MINW, PFD, 1>;
        static unsigned long long* buf = nullptr;
        const size_t nb = 2 * 2 * 80 * 4 * sizeof(unsigned long long);
        if (!buf) {
            if (hipMalloc(&buf, nb) != hipSuccess) return hipErrorOutOfMemory;
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        (void)hipMemsetAsync(buf, 0, nb, st);
        q.PW = reinterpret_cast<float*>(buf);
        hipLaunchKernelGGL(dk, dim3(grid), dim3(64 * WM * WN), lds, st, q);
        (void)hipStreamSynchronize(st);
        static unsigned long long host[2 * 2 * 80 * 4];
        (void)hipMemcpy(host, buf, nb, hipMemcpyDeviceToHost);
        const int nk = q.K / BK16;
        fprintf(stderr, "[stamp] igemm16p<%d,%d,%d,%d,%d,%d> M %d N %d K %d tiles %d grid %d\n", BM, BN, WM, WN, EPI, NST, q.M, q.N, q.K, q.ptiles, grid);
        for (int b = 0; b < 2; ++b)
            for (int w = 0; w < 2; ++w) {
                const unsigned long long* h = host + (b * 2 + w) * 80 * 4;
                double sw = 0, sb = 0, sk = 0, sp = 0; int n = 0;
                for (int i = 4; i < nk && i < 80; ++i) {
                    if (!h[i * 4]) break;
                    sw += (double)(h[i * 4 + 1] - h[i * 4]); sb += (double)(h[i * 4 + 2] - h[i * 4 + 1]); sk += (double)(h[i * 4 + 3] - h[i * 4 + 2]);
                    sp += (double)(h[i * 4] - h[(i - 1) * 4]); ++n;
                }
                if (n) fprintf(stderr, "[stamp]  block %d wave %d: per K-step (steps 4..%d of tile 0): period %.0f = vmcnt wait %.0f + barrier %.0f + body %.0f + rest  (s_memtime ticks)\n",
                               b ? 8 : 0, w ? WM * WN - 1 : 0, 3 + n, sp / n, sw / n, sb / n, sk / n);
                if (nk < 80 && h[nk * 4]) fprintf(stderr, "[stamp]   tile 0 -> tile 1: last body end to first wait start %.0f ticks, first wait %.0f, tile 0 start..end %.0f\n",
                                                  (double)(h[nk * 4] - h[(nk - 1) * 4 + 3]), (double)(h[nk * 4 + 1] - h[nk * 4]), (double)(h[(nk - 1) * 4 + 3] - h[0]));
            }
        return hipGetLastError();
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WM * WN), lds, st, q);
    return hipGetLastError();
}

// ---- ring variant with loader waves -----------------------------------------------------------------------------------------
// What the stamped builds of igemm16p say (FTE_IGEMM16_STAMP, 14x14x256 at batch 512): a K-step of a block lasts ~1850-2000 cycles for
// 256 cycles of MFMA per wave; the same with a four-stage ring, with one block per CU, with 44 % of the DMAs left out.  The MFMA pipe
// and the LDS-DMA path are far from their limits (scripts/probes/lds_dma_vs_read.hip: 116 GB/s of DMA per CU BESIDE 93 % of the MFMA
// peak, in different waves) -- what costs is a wave doing everything in turn: fragment reads it must wait for, two MFMAs, a DMA
// instruction that holds the wave while the address path takes it, a wait for the landing, a barrier.  So the roles are split:
//   * NLW loader waves only move tiles: per K-step each issues its share of the stage's 1-KiB pieces NST - 1 steps ahead (across tile
//     boundaries), waits with a counted vmcnt for the stage the consumers need next, and meets them at the ONE barrier of the step;
//   * WM x WN consumer waves (wave tile 64 x 64: four MFMAs per 16-deep sub-step, half the LDS fragment bytes per MFMA of the 32 x 64
//     tile) only read fragments (PFD sub-steps ahead) and issue MFMAs; their epilogue is epilogue_rows, during which the loaders
//     are already filling the ring for the next tile.
// One block per CU (a 256 x 128 tile's three stages are 144 KB), three waves per SIMD.
template <int BM, int BN, int WM, int WN, int EPI, int NST, int NLW, int PFD, int DBG = 0>
__global__ __launch_bounds__(64 * (WM * WN + NLW), 1) void igemm16r_kernel(const IgemmParams p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32, NCW = WM * WN;
    constexpr int KA = BM / 8 / NLW, KB = BN / 8 / NLW, PPL = KA + KB;      // 1-KiB pieces (8 rows x 128 B) per loader wave and K-step
    static_assert((BM / 8) % NLW == 0 && (BN / 8) % NLW == 0 && NST >= 3 && NST <= 4 && PFD >= 1 && PFD <= 3, "pieces per loader");
    constexpr int STAGE = (BM + BN) * ROWB;
    constexpr int NQ = NST - 2;
    extern __shared__ __attribute__((aligned(16))) char smem16[];
    float* const colf2 = reinterpret_cast<float*>(smem16 + NST * STAGE);   // [2][2][BN]: alpha, bias of the tile's columns, double-buffered
    float* const red = colf2 + 4 * BN;                                     // [2][WM][BN]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int ntn = p.N / BN;
    const int xcd = blockIdx.x & 7, per = gridDim.x >> 3;
    const int q8 = p.ptiles >> 3, r8 = p.ptiles & 7;
    const int xbase = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int xcnt = q8 + (xcd < r8 ? 1 : 0);
    int idx = blockIdx.x >> 3;
    if (idx >= xcnt) return;
    const int NT = p.a_NT, nk = p.K / BK16;
    const bool epi_barrier = EPI == EPI_DGRAD && p.PA != nullptr;
    unsigned long long* const stamps = DBG ? reinterpret_cast<unsigned long long*>(p.PW) : nullptr;
    int dbg_step = 0;

    if (wid >= NCW) {
        // =================================================== loader wave ===================================================
        const int lw = wid - NCW;
        constexpr unsigned OOB = 0x80000000u;
        const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, p.a_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, p.b_bytes, 0x00020000);
        const int a_hw = p.a_OH * p.a_OW;
        const float r_ahw = 1.f / (float)a_hw, r_aow = 1.f / (float)p.a_OW;
        int tapA = 0, tapB = 0;                 // byte offsets of the taps, one per lane (v_readlane instead of dependent scalar loads)
        if (lane < NT) {
            tapA = ((p.a_dh[lane] * p.a_IW + p.a_dw[lane]) * p.a_ld) * 2;
            tapB = p.b_tapoff[lane] * 2;
        }
        unsigned a_base[KA], b_base[KB];
        int a_mask[KA];
        // piece k of this wave = piece lw + NLW k of the stage: rows 8 (lw + NLW k) + (lane >> 3); k < KA: A rows, else B rows
        auto setup = [&](int tile) {
            const int mt = tile / ntn, nt_ = tile - mt * ntn;
            const int m0 = p.m_base + mt * BM, n0 = nt_ * BN;
#pragma unroll
            for (int k = 0; k < KA; ++k) {
                const int r = 8 * (lw + NLW * k) + (lane >> 3);
                const int m = m0 + r;
                int base = 0, mask = 0;
                if (m < p.M) {
                    const int n = fdiv(m, a_hw, r_ahw), rem = m - n * a_hw;
                    const int oh = fdiv(rem, p.a_OW, r_aow), ow = rem - oh * p.a_OW;
                    const int ih0 = oh * p.a_stride, iw0 = ow * p.a_stride;
                    base = ((n * p.a_IH + ih0) * p.a_IW + iw0) * p.a_ld;
                    for (int t = 0; t < NT; ++t) {
                        const int ih = ih0 + p.a_dh[t], iw = iw0 + p.a_dw[t];
                        if (ih >= 0 && ih < p.a_IH && iw >= 0 && iw < p.a_IW) mask |= 1 << t;
                    }
                }
                const int chunk = (lane & 7) ^ ((r >> 1) & 7);
                a_base[k] = (unsigned)(base + (chunk << 3)) * 2u;
                a_mask[k] = mask;
            }
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                const int r = 8 * (lw + NLW * k) + (lane >> 3);
                const int chunk = (lane & 7) ^ ((r >> 1) & 7);
                b_base[k] = (unsigned)((n0 + r) * p.b_ld + (chunk << 3)) * 2u;
            }
        };
        int itap = 0, ikc = 0;
        auto issue = [&](int stage) {
            char* As = smem16 + stage * STAGE + lw * 1024;
            char* Bs = As + BM * ROWB;
            const unsigned toff = (unsigned)(__builtin_amdgcn_readlane(tapA, itap) + ikc * 2);
            const unsigned boff = (unsigned)(__builtin_amdgcn_readlane(tapB, itap) + ikc * 2);
#pragma unroll
            for (int k = 0; k < KA; ++k) dma16(rsrcA, As + k * (NLW * 1024), ((a_mask[k] >> itap) & 1) ? a_base[k] + toff : OOB, 0);
#pragma unroll
            for (int k = 0; k < KB; ++k) dma16(rsrcB, Bs + k * (NLW * 1024), b_base[k], boff);
            if (++itap == NT) { itap = 0; ikc += BK16; }
        };
        setup(xbase + idx);
#pragma unroll
        for (int s0 = 0; s0 < NST - 1; ++s0) issue(s0);
        int qn[NQ];
#pragma unroll
        for (int k = 0; k < NQ; ++k) qn[k] = PPL;
        int stage = 0;
        for (;;) {
            const int nidx = idx + per;
            const bool hasnext = nidx < xcnt;
            for (int t = 0; t < nk; ++t) {
                const bool fillnext = t + NST - 1 < nk || hasnext;
                if (t + NST - 1 == nk && hasnext) { setup(xbase + nidx); itap = 0; ikc = 0; }
                unsigned long long st0 = 0, st1 = 0, st2 = 0;
                if constexpr (DBG) { __builtin_amdgcn_sched_barrier(0); st0 = stamp_now(); __builtin_amdgcn_sched_barrier(0); }
                {
                    int allowed = 0;
#pragma unroll
                    for (int k = 0; k < NQ; ++k) allowed += qn[k];
                    wait_vmcnt_upto(allowed);
                }
                if constexpr (DBG) { __builtin_amdgcn_sched_barrier(0); st1 = stamp_now(); __builtin_amdgcn_sched_barrier(0); }
                __builtin_amdgcn_s_barrier();
                if constexpr (DBG) { __builtin_amdgcn_sched_barrier(0); st2 = stamp_now(); __builtin_amdgcn_sched_barrier(0); }
                if (fillnext) issue(stage == 0 ? NST - 1 : stage - 1);                  // the stage the consumers read in the previous K-step
                if constexpr (DBG) {
                    __builtin_amdgcn_sched_barrier(0);
                    const unsigned long long st3 = stamp_now();
                    __builtin_amdgcn_sched_barrier(0);
                    if (blockIdx.x == 0 && lw == 0 && dbg_step < 80 && lane == 0) {
                        unsigned long long* o = stamps + (1 * 80 + dbg_step) * 4;
                        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3;
                    }
                    ++dbg_step;
                }
                stage = stage + 1 == NST ? 0 : stage + 1;
#pragma unroll
                for (int k = 0; k + 1 < NQ; ++k) qn[k] = qn[k + 1];
                qn[NQ - 1] = fillnext ? PPL : 0;
            }
            if (epi_barrier) __builtin_amdgcn_s_barrier();           // the consumers' epilogue holds one block barrier
            if (!hasnext) break;
            idx = nidx;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ===================================================== consumer wave =====================================================
    const int wm = wid / WN, wn = wid % WN;
    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc[TM][TN];
    int a_row[TM], b_row[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a_row[i] = wm * (TM * 32) + i * 32 + li;
#pragma unroll
    for (int j = 0; j < TN; ++j) b_row[j] = wn * (TN * 32) + j * 32 + li;
    auto kstep = [&](int stage) {
        const char* As = smem16 + stage * STAGE;
        const char* Bs = As + BM * ROWB;
        constexpr int NB = PFD + 1;
        bf16x8 fa[NB][TM], fb[NB][TN];
        auto ld = [&](int ks, int b) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[b][i] = *reinterpret_cast<const bf16x8*>(As + a_row[i] * ROWB + (((2 * ks + lh) ^ ((a_row[i] >> 1) & 7)) << 4));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[b][j] = *reinterpret_cast<const bf16x8*>(Bs + b_row[j] * ROWB + (((2 * ks + lh) ^ ((b_row[j] >> 1) & 7)) << 4));
        };
#pragma unroll
        for (int k0 = 0; k0 < PFD; ++k0) ld(k0, k0);
#pragma unroll
        for (int ks = 0; ks < BK16 / 16; ++ks) {
            if (ks + PFD < BK16 / 16) ld(ks + PFD, (ks + PFD) % NB);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks % NB][j], fa[ks % NB][i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const bool in16 = EPI == EPI_FWD ? p.R16 != nullptr : (p.ADD16 != nullptr || p.Zin16 != nullptr);
    const EpiPtrs ep = epi_ptrs<EPI>(p);
    int stage = 0, kt = 0, n0_colf = -1;
    for (;;) {
        const int tile = xbase + idx;
        const int mt = tile / ntn, nt_ = tile - mt * ntn;
        const int m0 = p.m_base + mt * BM, n0 = nt_ * BN;
        const int nidx = idx + per;
        const bool hasnext = nidx < xcnt;
        if (n0 != n0_colf) ++kt;                       // a new column tile: the other half of colf2
        float* const colf = colf2 + (kt & 1) * 2 * BN;
        int roff[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * (TM * 32) + i * 32 + li;
            int off = -1;
            if (m < p.M) {
                if (p.c_OH == 0) {
                    off = m * p.c_ld;
                } else {
                    const int hw = p.c_OH * p.c_OW;
                    const int n = fdiv(m, hw, 1.f / (float)hw), rem = m - n * hw;
                    const int oh = fdiv(rem, p.c_OW, 1.f / (float)p.c_OW), ow = rem - oh * p.c_OW;
                    off = ((n * p.c_FH + oh * p.c_step + p.c_ph) * p.c_FW + ow * p.c_step + p.c_pw) * p.c_ld;
                }
            }
            roff[i] = off;
        }
        // alpha / bias of the tile's columns -> this tile's half of colf2 (read in the epilogue, nk barriers later; the other half
        // may still be read by waves in the previous tile's epilogue)
        // (only when the column tile changes -- with an even block stride it never does: a load here waits, vmcnt being in order, for
        // every store of the previous tile's epilogue to drain)
        if (n0 != n0_colf) {
            for (int c = tid; c < BN; c += 64 * NCW) {
                float al = 1.f, bi = 0.f;
                if constexpr (EPI == EPI_FWD) {
                    if (p.alpha) al = p.alpha[n0 + c];
                    if (p.bias) bi = p.bias[n0 + c];
                } else {
                    if (p.alpha) al = p.alpha[(n0 + c) % p.amod];
                }
                colf[c] = al;
                colf[BN + c] = bi;
            }
            n0_colf = n0;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        u32x4 ein0[TM][TN][2], ein1[TM][TN][2];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q) { ein0[i][j][q] = u32x4{0u, 0u, 0u, 0u}; ein1[i][j][q] = u32x4{0u, 0u, 0u, 0u}; }

        for (int t = 0; t < nk; ++t) {
            unsigned long long st0 = 0, st1 = 0;
            if constexpr (DBG) { __builtin_amdgcn_sched_barrier(0); st0 = stamp_now(); __builtin_amdgcn_sched_barrier(0); }
            __builtin_amdgcn_s_barrier();
            if constexpr (DBG) { __builtin_amdgcn_sched_barrier(0); st1 = stamp_now(); __builtin_amdgcn_sched_barrier(0); }
            if (t == nk - 1 && in16) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const long o = (long)(roff[i] < 0 ? 0 : roff[i]) + n0 + wn * (TN * 32) + j * 32 + 16 * q + 8 * lh;
                            if constexpr (EPI == EPI_FWD) {
                                if (ep.i0) ein0[i][j][q] = *(g_cu32x4*)(ep.i0 + o);
                            } else {
                                if (ep.i0) ein0[i][j][q] = *(g_cu32x4*)(ep.i0 + o);
                                if (ep.i1) ein1[i][j][q] = *(g_cu32x4*)(ep.i1 + o);
                            }
                        }
            }
            kstep(stage);
            if constexpr (DBG) {
                __builtin_amdgcn_sched_barrier(0);
                const unsigned long long st2 = stamp_now();
                __builtin_amdgcn_sched_barrier(0);
                if (blockIdx.x == 0 && wid == 0 && dbg_step < 80 && lane == 0) {
                    unsigned long long* o = stamps + (0 * 80 + dbg_step) * 4;
                    o[0] = st0; o[1] = st1; o[2] = st2; o[3] = 0;
                }
                ++dbg_step;
            }
            stage = stage + 1 == NST ? 0 : stage + 1;
        }
        epilogue_rows<BM, BN, WM, WN, EPI>(ep, acc, roff, ein0, ein1, colf, red, mt, n0, tid, wm, wn, li, lh);
        if (!hasnext) break;
        idx = nidx;
    }
}

template <int BM, int BN, int WM, int WN, int EPI, int NST, int NLW, int PFD>
hipError_t launch16r(const IgemmParams& p, hipStream_t st) {
    const int mt = (p.M - p.m_base + BM - 1) / BM, nt = p.N / BN;
    IgemmParams q = p;
    q.ptiles = mt * nt;
    q.ptiles_dbg = 0;
    const size_t lds = (size_t)NST * (BM + BN) * ROWB + (size_t)(4 * BN + 2 * WM * BN) * sizeof(float);
    auto kern = igemm16r_kernel<BM, BN, WM, WN, EPI, NST, NLW, PFD>;
    if (igemm_prof_on()) { const int ta[9] = {BM, BN, WM, WN, EPI, NST, NLW, PFD, 0}; igemm_note_symbol("igemm16r_kernel", ta, 9); }
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
        cus = prop.multiProcessorCount;
    }
    const int per_xcd = cus / 8;                                       // one block per CU
    const int need = (q.ptiles + 7) / 8;
    const int grid = 8 * (need < per_xcd ? need : per_xcd);
    constexpr int THREADS = 64 * (WM * WN + NLW);
    static const bool stamps = getenv("FTE_IGEMM16_STAMP") != nullptr;
    if (stamps) {              // diagnostic: the stamped build, its table on stderr
        auto dk = igemm16r_kernel<BM, BN, WM, WN, EPI, NST, NLW, PFD, 1>;
        static unsigned long long* buf = nullptr;
        const size_t nb = 2 * 80 * 4 * sizeof(unsigned long long);
        if (!buf) {
            if (hipMalloc(&buf, nb) != hipSuccess) return hipErrorOutOfMemory;
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        (void)hipMemsetAsync(buf, 0, nb, st);
        q.PW = reinterpret_cast<float*>(buf);
        hipLaunchKernelGGL(dk, dim3(grid), dim3(THREADS), lds, st, q);
        (void)hipStreamSynchronize(st);
        static unsigned long long host[2 * 80 * 4];
        (void)hipMemcpy(host, buf, nb, hipMemcpyDeviceToHost);
        const int nk = q.K / BK16;
        fprintf(stderr, "[stamp] igemm16r<%d,%d,%d,%d,%d,%d,%d,%d> M %d N %d K %d tiles %d grid %d\n", BM, BN, WM, WN, EPI, NST, NLW, PFD, q.M, q.N, q.K, q.ptiles, grid);
        {
            const unsigned long long* h = host;
            double sb = 0, sk = 0, sp = 0; int n = 0;
            for (int i = 4; i < nk && i < 80; ++i) {
                if (!h[i * 4]) break;
                sb += (double)(h[i * 4 + 1] - h[i * 4]); sk += (double)(h[i * 4 + 2] - h[i * 4 + 1]); sp += (double)(h[i * 4] - h[(i - 1) * 4]); ++n;
            }
            if (n) fprintf(stderr, "[stamp]  consumer wave 0: per K-step period %.0f = barrier %.0f + body %.0f + rest\n", sp / n, sb / n, sk / n);
            if (nk < 80 && h[nk * 4]) fprintf(stderr, "[stamp]   tile 0 -> 1: last body end to next barrier entry %.0f ticks (epilogue), tile 0 K loop %.0f\n",
                                              (double)(h[nk * 4] - h[(nk - 1) * 4 + 2]), (double)(h[(nk - 1) * 4 + 2] - h[0]));
            h = host + 80 * 4;
            double sw = 0, sbb = 0, si = 0; sp = 0; n = 0;
            for (int i = 4; i < nk && i < 80; ++i) {
                if (!h[i * 4]) break;
                sw += (double)(h[i * 4 + 1] - h[i * 4]); sbb += (double)(h[i * 4 + 2] - h[i * 4 + 1]); si += (double)(h[i * 4 + 3] - h[i * 4 + 2]);
                sp += (double)(h[i * 4] - h[(i - 1) * 4]); ++n;
            }
            if (n) fprintf(stderr, "[stamp]  loader wave 0:   per K-step period %.0f = vmcnt wait %.0f + barrier %.0f + issue %.0f + rest\n", sp / n, sw / n, sbb / n, si / n);
        }
        return hipGetLastError();
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), lds, st, q);
    return hipGetLastError();
}

// ---- loader waves + A WINDOW (3x3, stride 1, TF-SAME) -------------------------------------------------------------------------
// igemm16r moves (BM + BN) x 128 bytes per K-step and its loaders are what the consumers wait for (stamped: 1000 of the 2000 cycles
// of a K-step at the barrier; 12 pieces per loader and step at the address path's ~90 cycles each).  For a 3x3 / stride-1 layer the
// nine taps of a 64-channel chunk read row-shifted views of the same pixels, so the A operand is fetched ONCE per chunk:
//   * pixels live in LDS at PADDED slots -- slot(img, y, x) = img (H+1)(W+1) + (y+1)(W+1) + x + 1: one zero slot ahead of every image
//     row and one zero row ahead of every image, which is all the padding the eight neighbours ever touch -- so tap (dh, dw) of an
//     output pixel is slot + dh (W+1) + dw for EVERY pixel: no edge masks in the MFMA loop (the zero slots are LDS-DMA'd from an
//     out-of-range offset: zeros, scripts/probes/lds_dma_oob.hip);
//   * the window of a tile = slots [slot(m0) - W - 2, slot(m0 + BM - 1) + W + 2], at most WCAP 8-slot pieces (48 KB), double-buffered:
//     the next chunk's window (or the next tile's first) arrives two pieces per loader and K-step under the current chunk;
//   * B (weights) keeps its three-stage ring, 16 KB per K-step: a K-step moves 16 + 48/9 KB instead of 48;
//   * vmcnt retires in order and the window pieces of chunk c + 1 are issued before the B pieces of its first K-step, so waiting
//     for a step's B tile also waits for its window.
// Fragment addresses: row s = slot - window start + tap shift, chunk q of it at slot q ^ ((s >> 1) & 7) (the writer's swizzle is keyed
// by the same window-relative slot); the four 16-deep sub-steps are a0 ^ (ks << 5).
// KS = K-steps per block barrier.  KS = 1: three B stages, the barrier of every K-step.  KS = 2 (stamped at KS = 1: of a 1700-cycle
// K-step the two consumer waves of a SIMD fill the MFMA pipe for ~1050, the rest is the barrier -- waiting for the slowest of twelve
// waves -- and loop overhead): four B stages, consumers run two K-steps (32 MFMAs per wave) between barriers, loaders fill the
// other two stages meanwhile; the window shrinks to the layers' real need (WCAP = 45) to make room in the 160 KB.
// STG = rows per pass of the row-coalesced epilogue (epilogue_staged; 0 = the register epilogue epilogue_rows)
template <int BM, int BN, int WM, int WN, int EPI, int NLW, int PFD, int WCAP, int KS = 1, int STG = 0, int DBG = 0>
__global__ __launch_bounds__(64 * (WM * WN + NLW), 1) void igemm16rw_kernel(const IgemmParams p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32, NCW = WM * WN, NSTB = KS == 1 ? 3 : 2 * KS;
    constexpr int KB = BN / 8 / NLW;                    // B pieces per loader wave and K-step
    constexpr int KW = (WCAP + NLW - 1) / NLW;          // window pieces (8 slots x 128 B) per loader wave; WCAP = the window's capacity
    constexpr int WPS = KS == 1 ? (KW + 5) / 6 : (KW + 2) / 3;      // window pieces a loader issues per barrier interval: all of them early in a chunk
    static_assert((BN / 8) % NLW == 0 && (KS == 1 || KS == 2) && (KS == 2 || (KW + WPS - 1) / WPS <= 9 - NSTB), "pieces per loader");
    constexpr int WINB = WCAP * 1024, BSTAGE = BN * ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem16[];
    char* const bring = smem16 + 2 * WINB;
    float* const colf2 = reinterpret_cast<float*>(bring + NSTB * BSTAGE);  // [2][2][BN]
    float* const red = colf2 + 4 * BN;                                     // [2][WM][BN]
    char* const stgfix = reinterpret_cast<char*>(red + 2 * WM * BN);       // STG > 0, KS = 1: [NCW][STG] rows of 256 bytes (KS = 2: in the B ring)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int ntn = p.N / BN;
    const int xcd = blockIdx.x & 7, per = gridDim.x >> 3;
    const int q8 = p.ptiles >> 3, r8 = p.ptiles & 7;
    const int xbase = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int xcnt = q8 + (xcd < r8 ? 1 : 0);
    int idx = blockIdx.x >> 3;
    if (idx >= xcnt) return;
    const int nk = p.K / BK16, nch = nk / 9;             // nine taps per 64-channel chunk
    // Blocks that have one tile less than the longest of their XCD start late, spread over most of a tile's time: the resident blocks
    // of a launch otherwise run in phase, every epilogue of a round hits HBM at once (stamped: 15k cycles per 256 x 128 tile = the
    // burst at HBM speed, MFMAs idle) -- staggered, the traffic of the many is spread under the K loops of the others, and the
    // launch still ends with the blocks that had the extra tile.
    if (!(p.ptiles_dbg & 8)) {
        const int mine = (xcnt - idx + per - 1) / per, longest = (xcnt + per - 1) / per;
        if (mine < longest) {
            const unsigned long long wait = (unsigned long long)(((idx * 37) % per) * (long)nk * 1500 / per);
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(16);
        }
    }
    const bool epi_barrier = EPI == EPI_DGRAD && p.PA != nullptr;
    unsigned long long* const stamps = DBG ? reinterpret_cast<unsigned long long*>(p.PW) : nullptr;
    int dbg_step = 0;

    // padded slot space
    const int H = p.a_IH, W = p.a_IW, PW1 = W + 1, IS = (H + 1) * PW1, HW = H * W;
    const float r_hw = 1.f / (float)HW, r_w = 1.f / (float)W, r_is = 1.f / (float)IS, r_pw1 = 1.f / (float)PW1;
    auto slot_of = [&](int m) {                         // pixel index -> padded slot
        const int n = fdiv(m, HW, r_hw), rem = m - n * HW;
        const int y = fdiv(rem, W, r_w), x = rem - y * W;
        return n * IS + (y + 1) * PW1 + x + 1;
    };

    if (wid >= NCW) {
        // =================================================== loader wave ===================================================
        const int lw = wid - NCW;
        if (!(p.ptiles_dbg & 4)) __builtin_amdgcn_s_setprio(3);          // few instructions, all on the critical path of the block
        constexpr unsigned OOB = 0x80000000u;
        const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, p.a_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, p.b_bytes, 0x00020000);
        int tapB = 0;
        if (lane < 9) tapB = p.b_tapoff[lane] * 2;
        unsigned w_off[KW], b_base[KB];
        const int apix = (int)(p.a_bytes / (unsigned)(p.a_ld * 2));
        int npc = 0;                                    // pieces of the current window (whole launch: <= WCAP, the launcher checks)
        auto setup_w = [&](int tile) {                  // window piece k of this wave = piece lw + NLW k: slots lo + 8 (lw + NLW k) + (lane >> 3)
            const int mt = tile / ntn;
            const int m0 = p.m_base + mt * BM;
            const int mlast = min(m0 + BM, p.M) - 1;
            const int lo = slot_of(m0) - PW1 - 1, hi = slot_of(mlast) + PW1 + 1;
            npc = (hi - lo + 8) >> 3;
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                const int srel = 8 * (lw + NLW * k) + (lane >> 3);
                const int S = lo + srel;
                unsigned off = OOB;
                if (S >= 0 && S <= hi) {
                    const int n = fdiv(S, IS, r_is), rem = S - n * IS;
                    const int r = fdiv(rem, PW1, r_pw1), c = rem - r * PW1;
                    const int m = n * HW + (r - 1) * W + (c - 1);
                    if (r >= 1 && c >= 1 && m < apix) {             // (the TENSOR's pixels: a main launch of a main + tail pair stops short of them)
                        const int chunk = (lane & 7) ^ ((srel >> 1) & 7);
                        off = (unsigned)(m * p.a_ld + (chunk << 3)) * 2u;
                    }
                }
                w_off[k] = off;
            }
        };
        auto setup_b = [&](int tile) {
            const int mt = tile / ntn, nt_ = tile - mt * ntn;
            const int n0 = nt_ * BN;
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                const int r = 8 * (lw + NLW * k) + (lane >> 3);
                const int chunk = (lane & 7) ^ ((r >> 1) & 7);
                b_base[k] = (unsigned)((n0 + r) * p.b_ld + (chunk << 3)) * 2u;
            }
        };
        int btap = 0, bkc = 0;                          // the K-step whose B tile is issued next
        auto issue_b = [&](int stage) {
            char* Bs = bring + stage * BSTAGE + lw * 1024;
            const unsigned boff = (unsigned)(__builtin_amdgcn_readlane(tapB, btap) + bkc * 2);
#pragma unroll
            for (int k = 0; k < KB; ++k) dma16(rsrcB, Bs + k * (NLW * 1024), b_base[k], boff);
            if (++btap == 9) { btap = 0; bkc += BK16; }
        };
        // window pieces [from, from + WPS) of this wave for channel chunk `kc` into buffer `buf`; returns how many were issued
        auto issue_w = [&](int buf, int kc, int from) -> int {
            int n = 0;
            char* const dst = smem16 + buf * WINB + lw * 1024;
            const unsigned so = (unsigned)kc * 2u;
            static_for<0, (KW + WPS - 1) / WPS>([&](auto G) {          // `from` is a multiple of WPS: one uniform branch per group
                constexpr int g = decltype(G)::value;
                if (from == g * WPS) {
#pragma unroll
                    for (int k = g * WPS; k < g * WPS + WPS && k < KW; ++k)
                        if (lw + NLW * k < npc) { dma16(rsrcA, dst + k * (NLW * 1024), w_off[k], so); ++n; }
                }
            });
            return n;
        };
        // prologue: the first tile's first window (all of it), then the B tiles of its first NSTB - 1 K-steps
        setup_w(xbase + idx);
        setup_b(xbase + idx);
        for (int f = 0; f < KW; f += WPS) issue_w(0, 0, f);
#pragma unroll
        for (int s0 = 0; s0 < (KS == 1 ? NSTB - 1 : KS); ++s0) issue_b(s0);
        if constexpr (KS == 2) {
            // two K-steps per barrier: interval I reads stages 2 (I & 1), 2 (I & 1) + 1; its B tiles were issued during interval I - 1
            // BEFORE that interval's window pieces, which are all that may stay in flight at the barrier.  The window of the chunk
            // after chunk c goes out in three intervals from the first interval that STARTS inside c (an interval that only ends in c
            // still reads the buffer being replaced), i.e. at least a whole interval before the B tiles of the interval that needs it.
            int wprev = 0, half = 0, gc = 0;
            for (;;) {
                const int nidx = idx + per;
                const bool hasnext = nidx < xcnt;
                int wfrom = KW, wbuf = 0, wkc = 0;
                bool wnext = false, wstart = false;
                for (int t = 0, tau = 0, ch = 0; t < nk; t += 2) {
                    // a chunk whose first K-step is this interval's first: its successor's window may start now; one that starts at the
                    // second K-step: next interval
                    if (tau == 0 || wstart) {
                        wfrom = 0; wbuf = (gc + 1) & 1; wstart = false;
                        if (ch + 1 < nch) { wnext = true; wkc = (ch + 1) * BK16; }
                        else if (hasnext) { wnext = true; wkc = 0; setup_w(xbase + nidx); }
                        else wnext = false;
                    }
                    const bool fillb = t + 2 < nk || hasnext;
                    if (t + 2 == nk && hasnext) { setup_b(xbase + nidx); btap = 0; bkc = 0; }
                    unsigned long long st0 = 0, st1 = 0, st2 = 0;
                    if constexpr (DBG) { __builtin_amdgcn_sched_barrier(0); st0 = stamp_now(); __builtin_amdgcn_sched_barrier(0); }
                    wait_vmcnt_upto(wprev);
                    if constexpr (DBG) { __builtin_amdgcn_sched_barrier(0); st1 = stamp_now(); __builtin_amdgcn_sched_barrier(0); }
                    __builtin_amdgcn_s_barrier();
                    if constexpr (DBG) { __builtin_amdgcn_sched_barrier(0); st2 = stamp_now(); __builtin_amdgcn_sched_barrier(0); }
                    int nw = 0;
                    if (fillb) { issue_b(2 * (half ^ 1)); issue_b(2 * (half ^ 1) + 1); }
                    if (wnext && wfrom < KW) { nw = issue_w(wbuf, wkc, wfrom); wfrom += WPS; }
                    if constexpr (DBG) {
                        __builtin_amdgcn_sched_barrier(0);
                        const unsigned long long st3 = stamp_now();
                        __builtin_amdgcn_sched_barrier(0);
                        if (blockIdx.x == 0 && lw == 0 && dbg_step < 70 && lane == 0) {
                            unsigned long long* o = stamps + (1 * 80 + dbg_step) * 4;
                            o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3;
                        }
                        ++dbg_step;
                    }
                    wprev = nw;
                    half ^= 1;
                    tau += 2;
                    if (tau >= 9) { tau -= 9; ++ch; ++gc; wstart = tau == 1; }      // tau == 1: the new chunk began at this interval's second K-step
                }
                if constexpr (STG > 0) __builtin_amdgcn_s_barrier();      // the consumers stage their epilogue in the ring half the last interval read
                if (epi_barrier && (!hasnext || (xbase + nidx) % ntn != (xbase + idx) % ntn || (p.ptiles_dbg & 16))) __builtin_amdgcn_s_barrier();      // the consumers' flush
                if (!hasnext) break;
                idx = nidx;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
        // vmcnt retires in issue order: B(t) was issued in step t - 2 BEFORE that step's window pieces; younger than it are those
        // window pieces and everything of step t - 1
        int w2 = 0, b1 = KB, w1 = 0;                   // window pieces of step t - 2, B and window pieces of step t - 1
        int stage = 0, gc = 0;                          // B ring stage of the current step; chunks so far (window buffer = gc & 1)
        for (;;) {
            const int nidx = idx + per;
            const bool hasnext = nidx < xcnt;
            int wfrom = 0, wbuf = 0, wkc = 0;          // the window being fetched under the current chunk
            bool wnext = false;
            for (int t = 0, tau = 0, ch = 0; t < nk; ++t) {
                if (tau == 0) {                         // a chunk starts: what arrives under it?
                    wfrom = 0; wbuf = (gc + 1) & 1;
                    if (ch + 1 < nch) { wnext = true; wkc = (ch + 1) * BK16; }
                    else if (hasnext) { wnext = true; wkc = 0; setup_w(xbase + nidx); }
                    else wnext = false;
                }
                const bool fillb = t + NSTB - 1 < nk || hasnext;
                if (t + NSTB - 1 == nk && hasnext) { setup_b(xbase + nidx); btap = 0; bkc = 0; }
                unsigned long long st0 = 0, st1 = 0, st2 = 0;
                if constexpr (DBG) { __builtin_amdgcn_sched_barrier(0); st0 = stamp_now(); __builtin_amdgcn_sched_barrier(0); }
                wait_vmcnt_upto(w2 + b1 + w1);
                if constexpr (DBG) { __builtin_amdgcn_sched_barrier(0); st1 = stamp_now(); __builtin_amdgcn_sched_barrier(0); }
                __builtin_amdgcn_s_barrier();
                if constexpr (DBG) { __builtin_amdgcn_sched_barrier(0); st2 = stamp_now(); __builtin_amdgcn_sched_barrier(0); }
                int nb = 0, nw = 0;
                if (fillb) { issue_b(stage == 0 ? NSTB - 1 : stage - 1); nb = KB; }
                if (wnext && wfrom < KW) { nw = issue_w(wbuf, wkc, wfrom); wfrom += WPS; }
                if constexpr (DBG) {
                    __builtin_amdgcn_sched_barrier(0);
                    const unsigned long long st3 = stamp_now();
                    __builtin_amdgcn_sched_barrier(0);
                    if (blockIdx.x == 0 && lw == 0 && dbg_step < 70 && lane == 0) {
                        unsigned long long* o = stamps + (1 * 80 + dbg_step) * 4;
                        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3;
                    }
                    ++dbg_step;
                }
                w2 = w1; b1 = nb; w1 = nw;
                stage = stage + 1 == NSTB ? 0 : stage + 1;
                if (++tau == 9) { tau = 0; ++ch; ++gc; }
            }
            if (epi_barrier && (!hasnext || (xbase + nidx) % ntn != (xbase + idx) % ntn || (p.ptiles_dbg & 16))) __builtin_amdgcn_s_barrier();   // the consumers' flush holds one block barrier
            if (!hasnext) break;
            idx = nidx;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ===================================================== consumer wave =====================================================
    const int wm = wid / WN, wn = wid % WN;
    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc[TM][TN];
    int b_row[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) b_row[j] = wn * (TN * 32) + j * 32 + li;
    int tshift = 0;                                     // window-row shifts of the taps, one per lane
    if (lane < 9) tshift = p.a_dh[lane] * PW1 + p.a_dw[lane] + PW1 + 1;      // >= 0: the window starts W + 2 slots ahead of the tile's first pixel
    int sl[TM];                                         // this lane's fragment rows: slot - slot(m0) (>= 0)
    auto kstep = [&](int stage, int wbuf, int tap) {
        const char* Bs = bring + stage * BSTAGE;
        const int sh = __builtin_amdgcn_readlane(tshift, tap);
        unsigned a0[TM];                                   // byte offsets from smem16 (kept as integers: the XOR below must not cost the LDS address space)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int s_ = sl[i] + sh;
            a0[i] = (unsigned)(wbuf * WINB + (s_ << 7) + (((lh ^ (s_ >> 1)) & 7) << 4));
        }
        constexpr int NB = PFD + 1;
        bf16x8 fa[NB][TM], fb[NB][TN];
        auto ld = [&](int ks, int b) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[b][i] = *reinterpret_cast<const bf16x8*>(smem16 + (a0[i] ^ (unsigned)(ks << 5)));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[b][j] = *reinterpret_cast<const bf16x8*>(Bs + b_row[j] * ROWB + (((2 * ks + lh) ^ ((b_row[j] >> 1) & 7)) << 4));
        };
#pragma unroll
        for (int k0 = 0; k0 < PFD; ++k0) ld(k0, k0);
#pragma unroll
        for (int ks = 0; ks < BK16 / 16; ++ks) {
            if (ks + PFD < BK16 / 16) ld(ks + PFD, (ks + PFD) % NB);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks % NB][j], fa[ks % NB][i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // KS = 2: the two K-steps of a barrier interval as ONE stream of eight 16-deep sub-steps -- the fragments of the second K-step are
    // fetched under the last MFMAs of the first (its stage and window have been complete since the interval's barrier); with one
    // kstep() per K-step every K-step began with the MFMA pipe waiting for its first fragment reads, both waves of a SIMD at once
    // (tried: B fragment addresses as lane constants with the stage as the read's immediate offset, the two stage pairs as two
    // instantiations of this lambda behind a uniform branch -- the 64 accumulators then live across the branch and 179 / 356 registers spill)
    auto kstep2 = [&](int st0, int wb0, int tap0, int st1, int wb1, int tap1) {
        const char* Bs[2] = {bring + st0 * BSTAGE, bring + st1 * BSTAGE};
        // (window addresses as integers of the LDS address space, base included once: with `smem16 + (a0 ^ ..)` every fragment read paid a
        // vector add for the base -- 16 of the 55 vector instructions beside the 32 MFMAs of an interval, and each costs the pipe issue
        // cycles; the dynamic LDS of this kernel starts at a multiple of 128, so the XOR of bits 5-6 commutes with the base)
        const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem16;
        const int sh[2] = {__builtin_amdgcn_readlane(tshift, tap0), __builtin_amdgcn_readlane(tshift, tap1)};
        const int wb[2] = {wb0, wb1};
        unsigned a0[2][TM];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                // row s_ at s_ << 7, its chunk lh at position (lh ^ (s_ >> 1)) & 7 -- written as shift, and, shift-or, xor-add (five
                // vector instructions per fragment row and K-step instead of seven); an LDS ADDRESS: the reads below add nothing
                const unsigned t1 = (unsigned)(sl[i] + sh[h]) << 3;
                a0[h][i] = (((t1 << 4) | (t1 & 0x70u)) ^ (unsigned)(lh << 4)) + (lds0 + (unsigned)(wb[h] * WINB));
            }
        constexpr int NB = PFD + 1, NSUB = 2 * (BK16 / 16);
        bf16x8 fa[NB][TM], fb[NB][TN];
        auto ld = [&](int u, int b) {
            const int h = u / (BK16 / 16), ks = u % (BK16 / 16);
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[b][i] = *(const __attribute__((address_space(3))) bf16x8*)(a0[h][i] ^ (unsigned)(ks << 5));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[b][j] = *reinterpret_cast<const bf16x8*>(Bs[h] + b_row[j] * ROWB + (((2 * ks + lh) ^ ((b_row[j] >> 1) & 7)) << 4));
        };
#pragma unroll
        for (int k0 = 0; k0 < PFD; ++k0) ld(k0, k0);
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
            if (u + PFD < NSUB) ld(u + PFD, (u + PFD) % NB);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[u % NB][j], fa[u % NB][i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const bool in16 = EPI == EPI_FWD ? p.R16 != nullptr : (p.ADD16 != nullptr || p.Zin16 != nullptr);
    const EpiPtrs ep = epi_ptrs<EPI>(p);
    int stage = 0, kt = 0, gc = 0, n0_colf = -1, kt_dbg = 0;
    float carry[STG > 0 ? 16 : 4 * TN];                // dalpha / dbias column sums carried across this block's tiles (epilogue_rows / _staged)
#pragma unroll
    for (int j = 0; j < (STG > 0 ? 16 : 4 * TN); ++j) carry[j] = 0.f;
    constexpr int NU = TM * 4;                         // STG > 0: units of eight rows per wave (lane = row lane >> 3, columns 8 (lane & 7) ..)
    const int rr = lane >> 3, cq = lane & 7;
    for (;;) {
        const int tile = xbase + idx;
        const int mt = tile / ntn, nt_ = tile - mt * ntn;
        const int m0 = p.m_base + mt * BM, n0 = nt_ * BN;
        const int nidx = idx + per;
        const bool hasnext = nidx < xcnt;
        if (n0 != n0_colf) ++kt;                       // a new column tile: the other half of colf2
        float* const colf = colf2 + (kt & 1) * 2 * BN;
        int roff[TM];
        const int s0_ = slot_of(m0);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * (TM * 32) + i * 32 + li;
            int off = -1;
            sl[i] = 0;
            if (m < p.M) {
                sl[i] = slot_of(m) - s0_;
                if (p.c_OH == 0) {
                    off = m * p.c_ld;
                } else {
                    const int hw = p.c_OH * p.c_OW;
                    const int n = fdiv(m, hw, 1.f / (float)hw), rem = m - n * hw;
                    const int oh = fdiv(rem, p.c_OW, 1.f / (float)p.c_OW), ow = rem - oh * p.c_OW;
                    off = ((n * p.c_FH + oh * p.c_step + p.c_ph) * p.c_FW + ow * p.c_step + p.c_pw) * p.c_ld;
                }
            }
            roff[i] = off;
        }
        // (only when the column tile changes -- with an even block stride it never does: a load here waits, vmcnt being in order, for
        // every store of the previous tile's epilogue to drain)
        if (n0 != n0_colf) {
            for (int c = tid; c < BN; c += 64 * NCW) {
                float al = 1.f, bi = 0.f;
                if constexpr (EPI == EPI_FWD) {
                    if (p.alpha) al = p.alpha[n0 + c];
                    if (p.bias) bi = p.bias[n0 + c];
                } else {
                    if (p.alpha) al = p.alpha[(n0 + c) % p.amod];
                }
                colf[c] = al;
                colf[BN + c] = bi;
            }
            n0_colf = n0;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        u32x4 ein0[TM][TN][2], ein1[TM][TN][2];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 2; ++q) { ein0[i][j][q] = u32x4{0u, 0u, 0u, 0u}; ein1[i][j][q] = u32x4{0u, 0u, 0u, 0u}; }
        u32x4 es0[NU], es1[NU];
#pragma unroll
        for (int k = 0; k < NU; ++k) { es0[k] = u32x4{0u, 0u, 0u, 0u}; es1[k] = u32x4{0u, 0u, 0u, 0u}; }
        const int m_wave = m0 + wm * (TM * 32);
        const long sbase = (long)(m_wave + rr) * p.c_ld + n0 + wn * (TN * 32) + 8 * cq;      // element offset of this lane's unit 0

        for (int t = 0, tau = 0; t < nk; ++t) {
            unsigned long long st0 = 0, st1 = 0;
            if constexpr (DBG) { __builtin_amdgcn_sched_barrier(0); st0 = stamp_now(); __builtin_amdgcn_sched_barrier(0); }
            if (KS == 1 || (t & 1) == 0) __builtin_amdgcn_s_barrier();
            if constexpr (DBG) { __builtin_amdgcn_sched_barrier(0); st1 = stamp_now(); __builtin_amdgcn_sched_barrier(0); }
            if (STG > 0 && t == nk - KS && in16) {
#pragma unroll
                for (int k = 0; k < NU; ++k) {
                    const long o = m_wave + 8 * k + rr < p.M ? sbase + (long)(8 * k) * p.c_ld : 8 * cq;
                    if (ep.i0) es0[k] = *(g_cu32x4*)(ep.i0 + o);
                }
            }
            // the inputs of row block 0 before the last barrier interval, those of the later row blocks right after the K loop: BOTH
            // inputs of the data gradient's first block are then fetched ahead in the registers one of them took for the whole tile
            if (STG == 0 && t == nk - KS && in16 && !(p.ptiles_dbg & 64)) {
                const __amdgpu_buffer_rsrc_t r_i0 = epi_rsrc(ep.i0), r_i1 = epi_rsrc(ep.i1);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const unsigned o = roff[0] < 0 ? EPI_OOB : (unsigned)(roff[0] + n0 + wn * (TN * 32) + j * 32 + 16 * q + 8 * lh) * 2u;
                        ein0[0][j][q] = __builtin_amdgcn_raw_buffer_load_b128(r_i0, ep.i0 ? o : EPI_OOB, 0, 0);
                        if constexpr (EPI == EPI_DGRAD) ein1[0][j][q] = __builtin_amdgcn_raw_buffer_load_b128(r_i1, ep.i1 ? o : EPI_OOB, 0, 0);
                    }
            }
            if constexpr (KS == 2) {
                if ((t & 1) == 0) {
                    const int tau1 = tau + 1 == 9 ? 0 : tau + 1, gc1 = tau + 1 == 9 ? gc + 1 : gc;
                    kstep2(stage, gc & 1, tau, stage + 1, gc1 & 1, tau1);       // (stage is even here: stage + 1 < NSTB)
                }
            } else {
                kstep(stage, gc & 1, tau);
            }
            if constexpr (DBG) {
                __builtin_amdgcn_sched_barrier(0);
                const unsigned long long st2 = stamp_now();
                __builtin_amdgcn_sched_barrier(0);
                if (blockIdx.x == 0 && wid == 0 && dbg_step < 80 && lane == 0) {
                    unsigned long long* o = stamps + (0 * 80 + dbg_step) * 4;
                    o[0] = st0; o[1] = st1; o[2] = st2; o[3] = 0;
                }
                ++dbg_step;
            }
            stage = stage + 1 == NSTB ? 0 : stage + 1;
            if (++tau == 9) { tau = 0; ++gc; }
        }
        if constexpr (EPI == EPI_DGRAD && STG > 0) {
            if (ep.i1) {
#pragma unroll
                for (int k = 0; k < NU; ++k) {
                    const long o = m_wave + 8 * k + rr < p.M ? sbase + (long)(8 * k) * p.c_ld : 8 * cq;
                    es1[k] = *(g_cu32x4*)(ep.i1 + o);
                }
            }
        }
        unsigned long long se0 = 0;
        if constexpr (DBG) { __builtin_amdgcn_sched_barrier(0); se0 = stamp_now(); __builtin_amdgcn_sched_barrier(0); }
        unsigned long long estv[6] = {0, 0, 0, 0, 0, 0};
        const bool flush = !hasnext || (xbase + nidx) % ntn != nt_ || (p.ptiles_dbg & 16);       // the block's next tile has other columns (or there is none)
        if constexpr (STG == 0) {
            // (before any store: vmcnt retires in order, a load issued behind the stores of row block 0 would be waited for together with
            // them -- measured: data gradients 3.23 -> 3.50 ms per step with the loads of block 1 between the stores)
            if (in16 && !(p.ptiles_dbg & 64)) {
                const __amdgpu_buffer_rsrc_t r_i0 = epi_rsrc(ep.i0), r_i1 = epi_rsrc(ep.i1);
#pragma unroll
                for (int i = 1; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const unsigned o = roff[i] < 0 ? EPI_OOB : (unsigned)(roff[i] + n0 + wn * (TN * 32) + j * 32 + 16 * q + 8 * lh) * 2u;
                            ein0[i][j][q] = __builtin_amdgcn_raw_buffer_load_b128(r_i0, ep.i0 ? o : EPI_OOB, 0, 0);
                            if constexpr (EPI == EPI_DGRAD) ein1[i][j][q] = __builtin_amdgcn_raw_buffer_load_b128(r_i1, ep.i1 ? o : EPI_OOB, 0, 0);
                        }
            }
        }
        if constexpr (STG > 0) {
            char* stg;
            if constexpr (KS == 2) {
                // the ring half of the last interval (the loaders refill it after the NEXT tile's first barrier); every consumer must
                // have read its last B fragments first: one block barrier, matched by the loaders
                __builtin_amdgcn_s_barrier();
                stg = bring + ((stage + 2) & 3) * BSTAGE + wid * (STG * 256);
                static_assert(KS != 2 || NCW * STG * 256 <= 2 * BSTAGE, "staging fits the ring half");
            } else {
                stg = stgfix + wid * (STG * 256);
            }
            epilogue_staged<BM, BN, WM, WN, EPI, STG>(ep, acc, stg, sbase, m_wave, es0, es1, colf, red, mt, n0, tid, wm, wn, lane, carry, flush);
        } else {
            if (!(p.ptiles_dbg & 128) || flush)
                epilogue_rows<BM, BN, WM, WN, EPI, false>(ep, acc, roff, ein0, ein1, colf, red, mt, n0, tid, wm, wn, li, lh, DBG ? estv : nullptr, carry, flush);
        }
        if constexpr (DBG) {
            if (blockIdx.x == 0 && wid == 0 && lane == 0 && kt_dbg == 1) {
                unsigned long long* o = stamps + (1 * 80 + 76) * 4;
                o[0] = estv[1] - estv[0]; o[1] = estv[2] - estv[1]; o[2] = estv[3] - estv[2]; o[3] = estv[4] - estv[3];
                o[4] = estv[5] - estv[4]; o[5] = 7;
            }
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long se1 = stamp_now();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long se2 = stamp_now();
            __builtin_amdgcn_sched_barrier(0);
            if (blockIdx.x == 0 && wid == 0 && lane == 0 && kt_dbg < 4) {
                unsigned long long* o = stamps + (1 * 80 + 70 + kt_dbg) * 4;       // rows 70.. of the loader's table (the loader stops at 80 steps)
                o[0] = se0; o[1] = se1; o[2] = se2; o[3] = 1;
            }
            ++kt_dbg;
        }
        if (!hasnext) break;
        idx = nidx;
    }
}

// the window kernel's layers: 3x3 taps within one pixel, stride 1, output grid = image; every tile's padded window within WCAP pieces
static bool launch16rw_ok(const IgemmParams& p, int BM, int wcap) {
    if (p.a_NT != 9 || p.a_stride != 1 || p.a_OH != p.a_IH || p.a_OW != p.a_IW || p.a_KC % BK16 || p.K != 9 * p.a_KC) return false;
    for (int t = 0; t < 9; ++t)
        if (p.a_dh[t] < -1 || p.a_dh[t] > 1 || p.a_dw[t] < -1 || p.a_dw[t] > 1) return false;
    if (p.m_base != 0 || p.c_OH != 0 || p.c_ld != p.N) return false;      // (rows stored at m * N: what the row-coalesced epilogue assumes)
    // widest window: BM pixels spread over padded rows / images, plus a padded row and a slot on either side
    const long H = p.a_IH, W = p.a_IW, hw = H * W;
    const long imgx = (BM + hw - 1) / hw, rowx = (BM - 1) / W + 1;       // image / row boundaries BM consecutive pixels can cross
    const long span = (BM - 1) + rowx + imgx * (W + 1) + 2 * (W + 1) + 3;
    return (span + 7) / 8 <= wcap;
}

template <int BM, int BN, int WM, int WN, int EPI, int NLW, int PFD, int WCAP, int KS = 1, int STG = 0>
hipError_t launch16rw(const IgemmParams& p, hipStream_t st) {
    const int mt = (p.M - p.m_base + BM - 1) / BM, nt = p.N / BN;
    IgemmParams q = p;
    q.ptiles = mt * nt;
    static const int dbg = getenv("FTE_IGEMM16_DBG") ? atoi(getenv("FTE_IGEMM16_DBG")) : 0;
    q.ptiles_dbg = dbg;
    const size_t lds = (size_t)2 * WCAP * 1024 + (size_t)(KS == 1 ? 3 : 2 * KS) * BN * ROWB + (size_t)(4 * BN + 2 * WM * BN) * sizeof(float) +
                       (size_t)(STG > 0 && KS == 1 ? WM * WN * STG * 256 : 0);
    static_assert(2 * WCAP * 1024 + (KS == 1 ? 3 : 2 * KS) * BN * ROWB + (4 * BN + 2 * WM * BN) * 4 + (STG > 0 && KS == 1 ? WM * WN * STG * 256 : 0) <= 160 * 1024, "LDS");
    auto kern = igemm16rw_kernel<BM, BN, WM, WN, EPI, NLW, PFD, WCAP, KS, STG>;
    if (igemm_prof_on()) { const int ta[11] = {BM, BN, WM, WN, EPI, NLW, PFD, WCAP, KS, STG, 0}; igemm_note_symbol("igemm16rw_kernel", ta, 11); }
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
        cus = prop.multiProcessorCount;
    }
    const int per_xcd = cus / 8;
    const int need = (q.ptiles + 7) / 8;
    const int grid = 8 * (need < per_xcd ? need : per_xcd);
    constexpr int THREADS = 64 * (WM * WN + NLW);
    static const bool stamps = getenv("FTE_IGEMM16_STAMP") != nullptr;
    if (stamps) {
        auto dk = igemm16rw_kernel<BM, BN, WM, WN, EPI, NLW, PFD, WCAP, KS, STG, 1>;
        static unsigned long long* buf = nullptr;
        const size_t nb = 2 * 80 * 4 * sizeof(unsigned long long);
        if (!buf) {
            if (hipMalloc(&buf, nb) != hipSuccess) return hipErrorOutOfMemory;
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        (void)hipMemsetAsync(buf, 0, nb, st);
        q.PW = reinterpret_cast<float*>(buf);
        hipLaunchKernelGGL(dk, dim3(grid), dim3(THREADS), lds, st, q);
        (void)hipStreamSynchronize(st);
        static unsigned long long host[2 * 80 * 4];
        (void)hipMemcpy(host, buf, nb, hipMemcpyDeviceToHost);
        const int nk = q.K / BK16;
        fprintf(stderr, "[stamp] igemm16rw<%d,%d,%d,%d,%d,%d,%d> M %d N %d K %d tiles %d grid %d\n", BM, BN, WM, WN, EPI, NLW, PFD, q.M, q.N, q.K, q.ptiles, grid);
        {
            const unsigned long long* h = host;
            double sb = 0, sk = 0, sp = 0; int n = 0;
            for (int i = 4; i < nk && i < 80; ++i) {
                if (!h[i * 4]) break;
                sb += (double)(h[i * 4 + 1] - h[i * 4]); sk += (double)(h[i * 4 + 2] - h[i * 4 + 1]); sp += (double)(h[i * 4] - h[(i - 1) * 4]); ++n;
            }
            if (n) fprintf(stderr, "[stamp]  consumer wave 0: per K-step period %.0f = barrier %.0f + body %.0f + rest\n", sp / n, sb / n, sk / n);
            if (nk < 80 && h[nk * 4]) fprintf(stderr, "[stamp]   tile 0 -> 1: last body end to next barrier entry %.0f ticks (epilogue), tile 0 K loop %.0f\n",
                                              (double)(h[nk * 4] - h[(nk - 1) * 4 + 2]), (double)(h[(nk - 1) * 4 + 2] - h[0]));
            h = host + 80 * 4;
            double sw = 0, sbb = 0, si = 0; sp = 0; n = 0;
            for (int i = 4; i < nk && i < 80; ++i) {
                if (!h[i * 4]) break;
                sw += (double)(h[i * 4 + 1] - h[i * 4]); sbb += (double)(h[i * 4 + 2] - h[i * 4 + 1]); si += (double)(h[i * 4 + 3] - h[i * 4 + 2]);
                sp += (double)(h[i * 4] - h[(i - 1) * 4]); ++n;
            }
            if (n) fprintf(stderr, "[stamp]  loader wave 0:   per K-step period %.0f = vmcnt wait %.0f + barrier %.0f + issue %.0f + rest\n", sp / n, sw / n, sbb / n, si / n);
            for (int k = 0; k < 4; ++k) {
                const unsigned long long* e = host + (80 + 70 + k) * 4;
                if (e[3] == 1) fprintf(stderr, "[stamp]  consumer wave 0, tile %d: epilogue_rows %.0f ticks, then %.0f until its stores have drained\n", k,
                                       (double)(e[1] - e[0]), (double)(e[2] - e[1]));
            }
            {
                const unsigned long long* e = host + (80 + 76) * 4;
                if (e[5] == 7) fprintf(stderr, "[stamp]   inside (tile 1): wait for the fetched-ahead inputs %.0f, to the first block %.0f, exchange + bias to the first store %.0f, "
                                               "first store %.0f, activation + shortcut + second store %.0f\n", (double)e[0], (double)e[1], (double)e[2], (double)e[3], (double)e[4]);
            }
        }
        return hipGetLastError();
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), lds, st, q);
    return hipGetLastError();
}

// what the persistent kernel takes: whole-K launches of the 128-row tiles without merged dgrad classes
static bool launch16p_ok(const IgemmParams& p, int epi, int tile, int splits) {
    if (splits != 1 || p.PW || p.ncls > 1 || p.split_major > 0) return false;
    if (tile != TILE_128x128 && tile != TILE_128x64) return false;
    if (p.kchunk < p.K || p.K % BK16 || p.K / BK16 < 4 || p.a_KC % BK16 || p.a_NT > 9) return false;
    if (epi == EPI_DGRAD && p.alpha && p.amod % 8) return false;
    if (p.c_ld % 8 || p.N % 8) return false;
    // bf16-STORAGE launches only: with fp32 tensors in the epilogue (the operand-copies mode, the last layer's fp32 z / y) the row-per-lane
    // form moves 32 bytes per lane and tensor and loses to the LDS-transposed epilogue (SphereNet bf16 mode 14.54 -> 15.18 ms with it)
    if (epi == EPI_FWD ? (p.Z || p.Y || p.R) : (p.RAW || p.DZ || p.ADD || p.Zin)) return false;
    return true;
}

template <int BM, int BN, int WM, int WN, int EPI, int MINW, int NSTB = 2>
hipError_t launch16w(const IgemmParams& p, int splits, hipStream_t st) {
    const int mt = (p.M - p.m_base + BM - 1) / BM, nt = p.N / BN;
    const int npc = (BM + 2 * p.a_IW + 2 + 7) / 8;
    const size_t ring = (size_t)2 * npc * 1024 + (size_t)NSTB * BN * ROWB;
    const size_t epi = (size_t)(BM + WM * WN * 32 * 36 + 2 * WM * BN) * sizeof(float);
    const size_t lds = ring > epi ? ring : epi;
    auto kern = igemm16w_kernel<BM, BN, WM, WN, EPI, MINW, NSTB>;
    if (igemm_prof_on()) { const int ta[7] = {BM, BN, WM, WN, EPI, MINW, NSTB}; igemm_note_symbol("igemm16w_kernel", ta, 7); }
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(mt * nt, splits), dim3(64 * WM * WN), lds, st, p);
    return hipGetLastError();
}

template <int BM, int BN, int WM, int WN, int EPI, int NST, int MINW, int ABL = 0, int PF = 0>
hipError_t launch16(const IgemmParams& p, int splits, hipStream_t st) {
    const int mt = (p.M - p.m_base + BM - 1) / BM, nt = p.N / BN;
    const size_t ring = (size_t)NST * (BM + BN) * ROWB;
    const size_t epi = (size_t)(BM + WM * WN * 32 * 36 + 2 * WM * BN) * sizeof(float);
    const size_t lds = ring > epi ? ring : epi;
    auto kern = igemm16_kernel<BM, BN, WM, WN, EPI, NST, MINW, ABL, PF>;
    if (igemm_prof_on()) { const int ta[9] = {BM, BN, WM, WN, EPI, NST, MINW, ABL, PF}; igemm_note_symbol("igemm16_kernel", ta, 9); }
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (p.ncls > 1) {
        IgemmParams q = p;
        q.cls_tiles = mt * nt; q.cls_mtiles = mt;
        hipLaunchKernelGGL(kern, dim3(mt * nt * p.ncls, 1), dim3(64 * WM * WN), lds, st, q);
    } else {
        hipLaunchKernelGGL(kern, dim3(mt * nt, splits), dim3(64 * WM * WN), lds, st, p);
    }
    return hipGetLastError();
}

template <int BM, int BN, int WM, int WN, int EPI, int NST, int MINW>
hipError_t launch16bn(const IgemmParams& p, hipStream_t st) {
    const int mt = (p.M - p.m_base + BM - 1) / BM, nt = p.N / BN;
    const size_t ring = (size_t)NST * (BM + BN) * ROWB;
    const size_t epi = (size_t)(BM + WM * WN * 32 * 36 + 2 * WM * BN) * sizeof(float);
    const size_t lds = ring > epi ? ring : epi;
    auto kern = igemm16_bn_kernel<BM, BN, WM, WN, EPI, NST, MINW>;
    if (igemm_prof_on()) { const int ta[7] = {BM, BN, WM, WN, EPI, NST, MINW}; igemm_note_symbol("igemm16_bn_kernel", ta, 7); }
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (p.ncls > 1) {
        IgemmParams q = p;
        q.cls_tiles = mt * nt; q.cls_mtiles = mt;
        hipLaunchKernelGGL(kern, dim3(mt * nt * p.ncls, 1), dim3(64 * WM * WN), lds, st, q);
    } else {
        hipLaunchKernelGGL(kern, dim3(mt * nt, 1), dim3(64 * WM * WN), lds, st, p);
    }
    return hipGetLastError();
}

}  // namespace

bool igemm16_handles(const IgemmParams& p, int al, int bl, int tile) {
    if (!p.src16 || al != AL_MK || bl != BL_NK) return false;
    if (p.a_KC % BK16 || p.kchunk % BK16 || p.split_major > 0) return false;
    // N = 64 layers (TILE_128x64; bound by their epilogue's HBM traffic, not by operand staging): with four waves per block this kernel
    // lost to the register-staged one (0.51 vs 0.49 ms forward, 0.69 vs 0.65 dgrad on 56x56x64 at batch 512); with eight waves
    // (three blocks per CU = six waves per SIMD) it wins: SphereNet bf16 step 16.04 -> 15.82 ms.  FTE_IGEMM16_NARROW=0 keeps them there.
    static const bool narrow = !(getenv("FTE_IGEMM16_NARROW") && atoi(getenv("FTE_IGEMM16_NARROW")) == 0);
    return tile == TILE_128x128 || (narrow && tile == TILE_128x64);
}

hipError_t igemm16_launch(const IgemmParams& p, int epi, int tile, int splits, hipStream_t st) {
    if (p.SP || p.bn_mu) {
        // conv -> BN pairs: the per-tile kernel with the shared (LDS-transposed) epilogue, whose BNM form carries the statistics /
        // the BN mask and sums; the resident kernels' register epilogues do not (yet)
        if (splits != 1 || p.PW) return hipErrorInvalidValue;
        // Launches with at most a block or two per CU (the 14x14 / 7x7 / 4x4 layers of a 128-image shard) are paced by the bytes their K
        // loop keeps in flight: one K-step per block on the two-stage ring = 24-32 KB per block, 1.5 us per step.  A three-stage ring
        // (two steps in flight, still two blocks per CU) runs them in 0.6x the time.  FTE_BN16_NST=2 restores the two-stage ring.
        static const int nst_env = getenv("FTE_BN16_NST") ? atoi(getenv("FTE_BN16_NST")) : 3;
        // ... while the launch has at most ONE tile per CU.  From 257 to 768 tiles the two-stage ring puts every tile on the chip in one
        // round (up to four blocks per CU) where the 96 KB of the three-stage ring admit one block per CU and need two or three: conv +
        // statistics at 128 images, us, three / two stages: 14x14 1024->256 (392 tiles) 37.0 / 31.4, 7x7 2048->1024 (392) 52.9 / 44.3,
        // 7x7 2048->512 (196) 29.4 / 35.2; ResNeXt-50 step 6.55 -> 6.47 ms (round 5; the threshold was 768: profiles/r5_notes.md 12)
        static const int nst_tiles = getenv("FTE_BN16_NST_TILES") ? atoi(getenv("FTE_BN16_NST_TILES")) : 256;
        const long ntl = (long)((p.M - p.m_base + 127) / 128) * (p.N / (tile == TILE_128x128 ? 128 : 64)) * (p.ncls > 1 ? p.ncls : 1);
        const bool deep = nst_env >= 3 && ntl <= nst_tiles && p.K / BK16 >= 6;
        if (tile == TILE_128x128) {
            if (deep) {
                if (epi == EPI_FWD) return launch16bn<128, 128, 4, 2, EPI_FWD, 3, 2>(p, st);
                return launch16bn<128, 128, 4, 2, EPI_DGRAD, 3, 2>(p, st);
            }
            if (epi == EPI_FWD) return launch16bn<128, 128, 4, 2, EPI_FWD, 2, 4>(p, st);
            return launch16bn<128, 128, 4, 2, EPI_DGRAD, 2, 4>(p, st);
        }
        if (deep) {
            if (epi == EPI_FWD) return launch16bn<128, 64, 4, 2, EPI_FWD, 3, 4>(p, st);
            return launch16bn<128, 64, 4, 2, EPI_DGRAD, 3, 4>(p, st);
        }
        if (epi == EPI_FWD) return launch16bn<128, 64, 4, 2, EPI_FWD, 2, 6>(p, st);
        return launch16bn<128, 64, 4, 2, EPI_DGRAD, 2, 4>(p, st);      // (four waves per SIMD: at six the BN inputs spill 63 registers)
    }
    static const int abl = getenv("FTE_IGEMM16_ABL") ? atoi(getenv("FTE_IGEMM16_ABL")) : 0;      // diagnostic: see the kernel's ABL
    if (abl && tile == TILE_128x128 && epi == EPI_FWD) {
        if (abl == 1) return launch16<128, 128, 2, 2, EPI_FWD, 4, 1, 1>(p, splits, st);
        if (abl == 2) return launch16<128, 128, 2, 2, EPI_FWD, 4, 1, 2>(p, splits, st);
        if (abl == 3) return launch16<128, 128, 2, 2, EPI_FWD, 4, 1, 3>(p, splits, st);
        // the default configuration (two-stage ring, two blocks per CU) with the same ablations: 11, 12, 13; 10 = that kernel itself
        if (abl == 10) return launch16<128, 128, 2, 2, EPI_FWD, 2, 2, 0>(p, splits, st);
        if (abl == 11) return launch16<128, 128, 2, 2, EPI_FWD, 2, 2, 1>(p, splits, st);
        if (abl == 12) return launch16<128, 128, 2, 2, EPI_FWD, 2, 2, 2>(p, splits, st);
        if (abl == 13) return launch16<128, 128, 2, 2, EPI_FWD, 2, 2, 3>(p, splits, st);
    }
    // Measured on MI355X at batch 512 (forward, ms: 14x14x256 / 28x28x128 / 7x7x512): 4-stage ring at one block per CU 0.30 / 0.41 /
    // 0.29; 2-stage ring at two blocks per CU 0.24 / 0.30 / 0.23 (default); 256x128 tile, 8 waves, 3 stages 0.24 / 0.31 / 0.28;
    // the register-staged BF = 2 kernel 0.25 / 0.32 / 0.26.  Two co-resident blocks hide each other's prologue, epilogue and
    // DMA latency better than a deeper ring of one block does.
    // The window kernel (3x3 / stride 1 / W <= 30: A fetched once per 64-channel chunk instead of once per tap) is an OPTION, not the
    // default.  Measured on SphereNet at batch 512, bf16 mode, one stream, ms per step (per-tap kernel -> window kernel):
    //   four waves per block   17.19 -> 16.81   (16 + 24/9 DMA pieces per K-step instead of 32: the piece count is not the bound;
    //                                            with vmcnt(0) at every step -- the next chunk's window piece forced to land within one
    //                                            K-step -- nothing at all)
    //   three / four B stages  20.5 / 20.9      (96 / 112 KB of LDS = ONE block per CU: deeper look-ahead loses to fewer waves)
    //   eight waves (4 x 2)    15.77 -> 15.75   two blocks per CU = four waves per SIMD; 2 x 4: 16.1; sixteen waves: 17.3
    // i.e. these launches are paced by how many waves a SIMD has to switch between while one waits on its LDS reads / barrier, and
    // eight waves per block give the per-tap kernel the same gain.  FTE_IGEMM16_WIN = 1 (4 waves), 8, 9 (2 x 4), 16, 3 / 4 (B stages).
    static const int winmode = getenv("FTE_IGEMM16_WIN") ? atoi(getenv("FTE_IGEMM16_WIN")) : 0;
    bool near = true;                                   // every tap within one pixel of the output position (3x3, pad 1)
    for (int t = 0; t < p.a_NT && t < 9; ++t) near = near && p.a_dh[t] >= -1 && p.a_dh[t] <= 1 && p.a_dw[t] >= -1 && p.a_dw[t] <= 1;
    if (winmode && tile == TILE_128x128 && p.a_NT == 9 && near && p.a_stride == 1 && p.ncls <= 1 && p.a_IW <= 30 && p.a_IW >= 2 &&
        p.a_OH == p.a_IH && p.a_OW == p.a_IW) {
        if (winmode == 8) {          // eight waves (4 x 2, wave tile 32 x 64), two blocks per CU: four waves per SIMD
            if (epi == EPI_FWD) return launch16w<128, 128, 4, 2, EPI_FWD, 4>(p, splits, st);
            return launch16w<128, 128, 4, 2, EPI_DGRAD, 4>(p, splits, st);
        }
        if (winmode == 9) {          // eight waves as 2 x 4 (wave tile 64 x 32)
            if (epi == EPI_FWD) return launch16w<128, 128, 2, 4, EPI_FWD, 4>(p, splits, st);
            return launch16w<128, 128, 2, 4, EPI_DGRAD, 4>(p, splits, st);
        }
        if (winmode == 16) {         // sixteen waves (4 x 4, wave tile 32 x 32), two blocks per CU: eight waves per SIMD
            if (epi == EPI_FWD) return launch16w<128, 128, 4, 4, EPI_FWD, 8>(p, splits, st);
            return launch16w<128, 128, 4, 4, EPI_DGRAD, 8>(p, splits, st);
        }
        if (winmode == 3) {          // three B stages (two K-steps of B in flight), 96 KB: one block per CU
            if (epi == EPI_FWD) return launch16w<128, 128, 2, 2, EPI_FWD, 1, 3>(p, splits, st);
            return launch16w<128, 128, 2, 2, EPI_DGRAD, 1, 3>(p, splits, st);
        }
        if (winmode == 4) {          // four B stages, 112 KB: one block per CU
            if (epi == EPI_FWD) return launch16w<128, 128, 2, 2, EPI_FWD, 1, 4>(p, splits, st);
            return launch16w<128, 128, 2, 2, EPI_DGRAD, 1, 4>(p, splits, st);
        }
        if (epi == EPI_FWD) return launch16w<128, 128, 2, 2, EPI_FWD, 2>(p, splits, st);
        return launch16w<128, 128, 2, 2, EPI_DGRAD, 2>(p, splits, st);
    }
    // Resident kernels for bf16-STORAGE launches.  FTE_IGEMM16_PERSIST: 1 (default) = the window kernel igemm16rw for
    // every 3x3 / stride-1 layer, igemm16p for the other eligible forward launches and the N = 64 data gradients; 0 = the per-tile
    // kernels below; 14 = igemm16p for every eligible launch; 22 = the window kernel with one K-step per barrier; 10 = igemm16r (loader
    // waves, no window).  Measured and dropped (same table): igemm16p with a four-stage ring at one block per CU (0.234 vs 0.160 ms,
    // 14x14x256 forward), a 256x128 tile without loader waves (0.209), fragment reads two / three sub-steps ahead at 128 registers
    // (spills: 0.168 / 0.170), four waves of 64x64 (0.168); the 128x128 data gradient on igemm16p (0.214 vs 0.202 per-tile).
    // Pointwise launches with at most a block or two per CU (the 1x1 data gradients of the 14x14 / 7x7 / 4x4 stages at a 128-image
    // shard): the three-stage ring, as for the conv -> BN launches above.  FTE_IGEMM16_DEEP=0 turns it off.
    {
        static const int deep_env = getenv("FTE_IGEMM16_DEEP") ? atoi(getenv("FTE_IGEMM16_DEEP")) : 1;
        static const int deep_tiles = getenv("FTE_IGEMM16_DEEP_TILES") ? atoi(getenv("FTE_IGEMM16_DEEP_TILES")) : 256;      // (one tile per CU at most: see the conv -> BN launches above)
        const long ntl = (long)((p.M - p.m_base + 127) / 128) * (p.N / (tile == TILE_128x128 ? 128 : 64)) * (p.ncls > 1 ? p.ncls : 1);
        if (deep_env && splits == 1 && !p.PW && p.a_NT == 1 && ntl <= deep_tiles && p.kchunk >= p.K && p.K / BK16 >= 6) {
            if (tile == TILE_128x128) {
                if (epi == EPI_FWD) return launch16<128, 128, 4, 2, EPI_FWD, 3, 2>(p, splits, st);
                return launch16<128, 128, 4, 2, EPI_DGRAD, 3, 2>(p, splits, st);
            }
            if (tile == TILE_128x64) {
                if (epi == EPI_FWD) return launch16<128, 64, 4, 2, EPI_FWD, 3, 4>(p, splits, st);
                return launch16<128, 64, 4, 2, EPI_DGRAD, 3, 4>(p, splits, st);
            }
        }
    }
    static const int pers = getenv("FTE_IGEMM16_PERSIST") ? atoi(getenv("FTE_IGEMM16_PERSIST")) : 1;
    static const int stg_env = getenv("FTE_IGEMM16_STG") ? atoi(getenv("FTE_IGEMM16_STG")) : 1;      // A/B hook: 0 = the register epilogue everywhere
    const bool stg = stg_env != 0, stg2 = stg_env == 2;
    if (pers && launch16p_ok(p, epi, tile, splits)) {
        const bool win = pers == 1 || pers == 22;
        if (tile == TILE_128x128) {
            // batch 512, ms on one box, per-tile kernel -> igemm16rw: forward 28x28x128 0.220 -> 0.155, 14x14x256 0.161 -> 0.138, 7x7x512
            // 0.185 -> 0.126; data gradient 0.293 -> 0.225, 0.198 -> 0.177, 0.215 -> 0.146
            // two K-steps per barrier, their fragments one stream (kstep2) with one sub-step fetched ahead: on one box, four alternating
            // runs each, the per-step totals of the 28x28 / 14x14 / 7x7 layers 2.163 -> 2.141 ms forward, 2.840 -> 2.793 data gradient
            // (two sub-steps ahead spills 13 / 25 registers there)
            if (pers == 1 && (p.K / BK16) % 2 == 0 && launch16rw_ok(p, 256, 45)) {
                // (the row-coalesced epilogue through the free half of the B ring: SLOWER here -- 28x28x128 forward 0.157 -> 0.180 ms,
                // data gradient 0.220 -> 0.294 (94 spilled registers beside 64 accumulators and two fetched-ahead inputs), 14x14x256
                // 0.140 -> 0.146 / 0.175 -> 0.211; the stamped forward epilogue stays at 11-13k cycles: the wait for the inputs and
                // ~600 VALU instructions per wave, two waves per SIMD, are what is left of it.  FTE_IGEMM16_STG=2 selects it)
                if (stg2) {
                    if (epi == EPI_FWD) return launch16rw<256, 128, 4, 2, EPI_FWD, 4, 2, 45, 2, 16>(p, st);
                    return launch16rw<256, 128, 4, 2, EPI_DGRAD, 4, 2, 45, 2, 16>(p, st);
                }
                if (epi == EPI_FWD) return launch16rw<256, 128, 4, 2, EPI_FWD, 4, 1, 45, 2>(p, st);
                return launch16rw<256, 128, 4, 2, EPI_DGRAD, 4, 1, 45, 2>(p, st);
            }
            if (win && launch16rw_ok(p, 256, 48)) {
                if (epi == EPI_FWD) return launch16rw<256, 128, 4, 2, EPI_FWD, 4, 2, 48>(p, st);
                return launch16rw<256, 128, 4, 2, EPI_DGRAD, 4, 2, 48>(p, st);
            }
            if (pers == 10 && p.K / BK16 >= 4 && (p.M - p.m_base) >= 256) {      // loader waves + 64 x 64 consumers, no window
                if (epi == EPI_FWD) return launch16r<256, 128, 4, 2, EPI_FWD, 3, 4, 2>(p, st);
                return launch16r<256, 128, 4, 2, EPI_DGRAD, 3, 4, 2>(p, st);
            }
            if (epi == EPI_FWD) return launch16p<128, 128, 4, 2, EPI_FWD, 2, 4, 2>(p, st);
            if (pers == 14) return launch16p<128, 128, 4, 2, EPI_DGRAD, 2, 4, 2>(p, st);
            // (a 128 x 128 data gradient the window kernel does not take -- stride 2, 1x1 -- stays on the per-tile kernel)
        } else {
            // N = 64: 256 x 64 tile, consumers 64 x 32 (56x56x64 at batch 512: forward 0.42 -> 0.37 (igemm16p) -> 0.24 ms, data
            // gradient 0.62 -> 0.51 -> 0.42)
            if (win && launch16rw_ok(p, 256, 56)) {
                // eight row-waves of 32 x 64 (whole 128-byte rows per wave) and the row-coalesced epilogue (epilogue_staged): 56x56x64 at
                // batch 512 forward 0.2415 -> 0.2250 ms, data gradient 0.381 -> 0.338
                if (stg && pers == 1) {
                    if (epi == EPI_FWD) return launch16rw<256, 64, 8, 1, EPI_FWD, 4, 2, 56, 1, 8>(p, st);
                    return launch16rw<256, 64, 8, 1, EPI_DGRAD, 4, 2, 56, 1, 8>(p, st);
                }
                if (epi == EPI_FWD) return launch16rw<256, 64, 4, 2, EPI_FWD, 4, 2, 56>(p, st);
                return launch16rw<256, 64, 4, 2, EPI_DGRAD, 4, 2, 56>(p, st);
            }
            if (epi == EPI_FWD) return launch16p<128, 64, 4, 2, EPI_FWD, 2, 6, 3>(p, st);
            return launch16p<128, 64, 4, 2, EPI_DGRAD, 2, 6, 3>(p, st);
        }
    }
    // cfg 4 = cfg 1 with eight waves per block: SphereNet bf16 step 16.57 -> 15.77 ms (one stream), 16.0 -> 15.5 (two streams)
    static const int cfg = getenv("FTE_IGEMM16_CFG") ? atoi(getenv("FTE_IGEMM16_CFG")) : 4;      // tuning hook
    if (abl >= 20 && abl <= 24 && tile == TILE_128x128 && epi == EPI_FWD) {      // ablations of the default configuration
        if (abl == 20) return launch16<128, 128, 4, 2, EPI_FWD, 2, 4, 0>(p, splits, st);
        if (abl == 21) return launch16<128, 128, 4, 2, EPI_FWD, 2, 4, 1>(p, splits, st);
        if (abl == 22) return launch16<128, 128, 4, 2, EPI_FWD, 2, 4, 2>(p, splits, st);
        if (abl == 23) return launch16<128, 128, 4, 2, EPI_FWD, 2, 4, 3>(p, splits, st);
        if (abl == 24) return launch16<128, 128, 4, 2, EPI_FWD, 2, 4, 4>(p, splits, st);
    }
    if (cfg == 5 && tile == TILE_128x128) {          // four waves (2 x 2, wave tile 64 x 64), fragments read one sub-step ahead
        if (epi == EPI_FWD) return launch16<128, 128, 2, 2, EPI_FWD, 2, 2, 0, 1>(p, splits, st);
        return launch16<128, 128, 2, 2, EPI_DGRAD, 2, 2, 0, 1>(p, splits, st);
    }
    if (cfg == 6 && tile == TILE_128x128) {          // eight waves (4 x 2), fragments read one sub-step ahead
        if (epi == EPI_FWD) return launch16<128, 128, 4, 2, EPI_FWD, 2, 4, 0, 1>(p, splits, st);
        return launch16<128, 128, 4, 2, EPI_DGRAD, 2, 4, 0, 1>(p, splits, st);
    }
    if (cfg == 4 && tile == TILE_128x128) {          // as cfg 1 with eight waves per block (4 x 2): four waves per SIMD
        if (epi == EPI_FWD) return launch16<128, 128, 4, 2, EPI_FWD, 2, 4>(p, splits, st);
        return launch16<128, 128, 4, 2, EPI_DGRAD, 2, 4>(p, splits, st);
    }
    if (cfg == 1 && tile == TILE_128x128) {          // 2-stage ring, two blocks per CU
        if (epi == EPI_FWD) return launch16<128, 128, 2, 2, EPI_FWD, 2, 2>(p, splits, st);
        return launch16<128, 128, 2, 2, EPI_DGRAD, 2, 2>(p, splits, st);
    }
    if (cfg == 2 && tile == TILE_128x128 && epi == EPI_FWD && !p.PW && (p.M - p.m_base) >= 256) {          // 256x128, eight waves (two per SIMD), one block per CU
        // forward only: the dgrad epilogue's column partials are numbered by the PLANNED tile's rows (api.hip)
        if (epi == EPI_FWD) return launch16<256, 128, 4, 2, EPI_FWD, 3, 2>(p, splits, st);
    }
    if (cfg == 3 && tile == TILE_128x128 && epi == EPI_FWD && !p.PW && (p.M - p.m_base) >= 256 && p.N % 256 == 0) {
        // 256x256, eight waves of 64 x 128 (two per SIMD), one block per CU: half the staged bytes per FLOP of the 128x128 tile.
        // Measured (forward, batch 512): 14x14x512->512 (784 tiles) 0.729 -> 0.618 ms = 767 TFLOP/s, but the net's own layers are
        // too small for it -- 14x14x256 is 392 tiles on 256 CUs (0.239 -> 0.243 ms), 7x7x512 196 tiles (0.244 -> 0.287 ms).
        return launch16<256, 256, 4, 2, EPI_FWD, 2, 2>(p, splits, st);
    }
    if (tile == TILE_128x128) {
        if (epi == EPI_FWD) return launch16<128, 128, 2, 2, EPI_FWD, 4, 1>(p, splits, st);
        return launch16<128, 128, 2, 2, EPI_DGRAD, 4, 1>(p, splits, st);
    }
    if (tile == TILE_128x64) {
        static const int nmode = getenv("FTE_IGEMM16_NARROW") ? atoi(getenv("FTE_IGEMM16_NARROW")) : 8;      // 1: four waves, 3-stage ring; 4: four waves, 2-stage
        if (nmode == 8) {            // eight waves (4 x 2, wave tile 32 x 32), 2-stage ring of 24 KB stages: three blocks per CU
            if (epi == EPI_FWD) return launch16<128, 64, 4, 2, EPI_FWD, 2, 6>(p, splits, st);
            return launch16<128, 64, 4, 2, EPI_DGRAD, 2, 6>(p, splits, st);
        }
        if (nmode == 4) {            // four waves, 2-stage ring, three blocks per CU
            if (epi == EPI_FWD) return launch16<128, 64, 2, 2, EPI_FWD, 2, 3>(p, splits, st);
            return launch16<128, 64, 2, 2, EPI_DGRAD, 2, 3>(p, splits, st);
        }
        if (epi == EPI_FWD) return launch16<128, 64, 2, 2, EPI_FWD, 3, 2>(p, splits, st);
        return launch16<128, 64, 2, 2, EPI_DGRAD, 3, 2>(p, splits, st);
    }
    return hipErrorInvalidValue;
}
