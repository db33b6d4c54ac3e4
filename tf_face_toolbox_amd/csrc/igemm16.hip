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

// ABL (diagnostic builds only, FTE_IGEMM16_ABL): 0 = the kernel; 1 = no MFMAs, 2 = no DMA (stale LDS), 3 = no epilogue
template <int BM, int BN, int WM, int WN, int EPI, int NST, int MINW, int ABL = 0>
__global__ __launch_bounds__(64 * WM * WN, MINW) void igemm16_kernel(const IgemmParams p) {
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
                fa[i] = *reinterpret_cast<const bf16x8*>(As + a_row[i] * ROWB + (((2 * ks + lh) ^ ((a_row[i] >> 1) & 7)) << 4));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[j] = *reinterpret_cast<const bf16x8*>(Bs + b_row[j] * ROWB + (((2 * ks + lh) ^ ((b_row[j] >> 1) & 7)) << 4));
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

    // ---- ring: NST - 1 K-steps in flight; step t: wait for (this wave's part of) tile t, barrier (everybody's part has landed
    // AND everybody has finished reading tile t - 1, whose stage is the one refilled during this step), compute tile t while
    // issuing tile t + NST - 1 ----
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nk) issue(s);
    int stage = 0, fill = NST - 1;
    for (int t = 0; t < nk; ++t) {
        if (t + NST - 1 <= nk) wait_vmcnt<(NST - 2) * L>();          // steady state: the NST - 2 younger tiles stay in flight
        else wait_vmcnt<0>();                                        // last steps: fewer tiles are outstanding
        __builtin_amdgcn_s_barrier();
        kstep(stage, fill, t + NST - 1 < nk);
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
    igemm_epilogue<BM, BN, WM, WN, EPI, true>(p, acc, reinterpret_cast<float*>(smem16), bid, split, m0, n0, mt, c_ph, c_pw, prow);
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

template <int BM, int BN, int WM, int WN, int EPI, int NST, int MINW, int ABL = 0>
hipError_t launch16(const IgemmParams& p, int splits, hipStream_t st) {
    const int mt = (p.M - p.m_base + BM - 1) / BM, nt = p.N / BN;
    const size_t ring = (size_t)NST * (BM + BN) * ROWB;
    const size_t epi = (size_t)(BM + WM * WN * 32 * 36 + 2 * WM * BN) * sizeof(float);
    const size_t lds = ring > epi ? ring : epi;
    auto kern = igemm16_kernel<BM, BN, WM, WN, EPI, NST, MINW, ABL>;
    if (igemm_prof_on()) { const int ta[8] = {BM, BN, WM, WN, EPI, NST, MINW, ABL}; igemm_note_symbol("igemm16_kernel", ta, 8); }
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
    // cfg 4 = cfg 1 with eight waves per block: SphereNet bf16 step 16.57 -> 15.77 ms (one stream), 16.0 -> 15.5 (two streams)
    static const int cfg = getenv("FTE_IGEMM16_CFG") ? atoi(getenv("FTE_IGEMM16_CFG")) : 4;      // tuning hook
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
