// wgrad16.hip -- filter gradient of a 3x3 / stride-1 conv from bf16 x and dz (bf16 storage mode), resident blocks, LDS-DMA staging.
//
//   dW[tap][cin][cout] = sum over output pixels p of  x[p + tap][cin] * dz[p][cout]                (fp32 accumulate)
//
// The register-staged filter gradient of igemm.hip (BF = 2) runs one (tap, 128 cin) x 128 cout tile per block: every tap of a layer
// re-reads the same dz rows and the same x rows from L2 (64 FLOP per staged byte), global -> VGPR -> LDS, 0.25 of the bf16 peak.
// Here a block owns (CT cin) x (BN cout) x ALL NINE TAPS for one range of the reduction, and the reduction index is the PADDED
// SLOT of igemm16rw's window (slot(img, y, x) = img (H+1)(W+1) + (y+1)(W+1) + x + 1; pad slots are zero rows, LDS-DMA'd from an
// out-of-range offset), so that tap (dh, dw) is the same dz rows against x rows shifted by dh (W+1) + dw:
//   * a K-piece is 64 slots: dz tile [64 slots][BN] and x window [64 + 2 (W+1) + 2 slots][CT], both row-contiguous in memory and
//     DMA'd to LDS AS THEY ARE (1-KiB pieces); the 32x32x16 operands (8 consecutive slots of one channel per lane) come back through
//     ds_read_b64_tr_b16 (the lane map of igemm.hip's frag_tr); 9 * (CT / 32) + BN / 32 fragments feed 9 * (CT / 32) * (BN / 32)
//     MFMAs per 16 slots -- 214 FLOP per staged byte;
//   * twelve waves, each 3 taps x 2 column blocks (six 32x32 accumulators), 5 fragments per 6 MFMAs; every wave also issues its share
//     of the next-but-one K-piece's DMA pieces (the traffic is light: ~40 pieces per 864 MFMAs); three stages, one barrier per K-piece;
//   * split over the slot range (S ranges -> S partial slabs, summed in fixed order by the caller's reduction as before).
// The zero rows cost (H+1)(W+1) / (H W) - 1 extra MFMA work (15 % at 14x14, 7 % at 28x28).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "igemm_dev.h"
#include "wgrad16.h"

#ifndef WG_PFD
#define WG_PFD 1
#endif
#ifndef WG_STAMP
#define WG_STAMP 0          // diagnostic build (scripts/build_variant-style): s_memtime stamps per K-piece, table on stderr
#endif

namespace {

using namespace igemm_dev;

typedef __attribute__((address_space(3))) void lds_void;
typedef short s16x4t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t& r, char* lds, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_vmcnt_upto(int n) {
    if (n >= 6) wait_vmcnt<6>(); else if (n >= 5) wait_vmcnt<5>(); else
    if (n >= 4) wait_vmcnt<4>(); else if (n >= 3) wait_vmcnt<3>(); else if (n >= 2) wait_vmcnt<2>(); else if (n >= 1) wait_vmcnt<1>(); else wait_vmcnt<0>();
}

constexpr int KP = 64;                 // slots per K-piece
constexpr int NWAVES = 12;

// CT = input channels per block (32 | 64), BN = output channels per block; (CT / 32) * (BN / 32) = 8 (72 accumulator blocks, six per
// wave) or 4 (the 64 -> 64 layers: 36 blocks, three per wave); XCAP = window rows (>= 64 + 2 (W + 1) + 2)
template <int CT, int BN, int NST, int XCAP>
__global__ __launch_bounds__(64 * NWAVES, 1) void wgrad16_kernel(const Wgrad16Params p) {
    constexpr int NCB = CT / 32, NNB = BN / 32;
    constexpr int NB2 = NCB * NNB / 4;                           // output-channel blocks per wave
    static_assert((NB2 == 1 || NB2 == 2) && NST >= 3, "36 or 72 accumulator blocks over twelve waves");
    constexpr int DZROW = BN * 2, XROW = CT * 2;                 // bytes per row of the two LDS images
    constexpr int DZ_RPP = 1024 / DZROW, X_RPP = 1024 / XROW;    // rows per 1-KiB DMA piece
    constexpr int DZ_PIECES = KP / DZ_RPP;
    constexpr int X_PIECES = XCAP / X_RPP;
    constexpr int PIECES = DZ_PIECES + X_PIECES;
    constexpr int MAXP = (PIECES + NWAVES - 1) / NWAVES;         // DMA pieces per wave and K-piece
    constexpr int STAGE = KP * DZROW + XCAP * XROW;
    static_assert(MAXP <= KP / 16, "one DMA piece per 16-deep step");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tg = wid >> 2, wq = wid & 3;                       // taps 3 tg .. 3 tg + 2
    const int cb = wq / (4 / NCB), ng = wq % (4 / NCB);          // input-channel block, pair of output-channel blocks
    const int nb0 = ng * NB2;

    // ---- block -> (channel tile pair, slot range); the blocks of one slot range share an XCD (b & 7) where S allows ----
    const int npairs = (p.cin / CT) * (p.cout / BN);
    int split, pair;
    if (p.S >= 8) { split = (blockIdx.x & 7) + 8 * ((blockIdx.x >> 3) / npairs); pair = (blockIdx.x >> 3) % npairs; }
    else { split = blockIdx.x % p.S; pair = blockIdx.x / p.S; }
    const int c0 = (pair / (p.cout / BN)) * CT, n0 = (pair % (p.cout / BN)) * BN;
    const int H = p.H, W = p.W, PW1 = W + 1, IS = (H + 1) * PW1, HW = H * W;
    const int KT = p.n * IS;                                     // padded slots of the whole tensor
    const int kbeg = split * p.kper;
    const int kend = min(kbeg + p.kper, (KT + KP - 1) / KP * KP);
    const int np = (kend - kbeg) / KP;
    float* const out = p.out + (long)split * p.slab;

    // everything the per-K-piece address decode reads, pinned in scalar registers: left to the compiler, kernel arguments used inside
    // the loop are re-fetched with s_load + a wait every time (the decode was a quarter of the kernel: 0.175 ms with it, 0.122 without)
    unsigned long long magic_is = p.magic_is;
    int cin_ = p.cin, cout_ = p.cout, dbg_ = p.dbg;
    asm volatile("" : "+s"(magic_is), "+s"(cin_), "+s"(cout_), "+s"(dbg_));
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcD = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.dz), 0, p.dz_bytes, 0x00020000);

    // Address decode of a DMA piece.  Its slots are consecutive and its first slot S0 is wave-uniform: (image, slot within the image)
    // of S0 come from ONE scalar division (host-made reciprocal); a lane adds its row d < 16 and looks the pixel up in an LDS table --
    // tab[s], s < IS + 16: pixel (within the image; + HW beyond it, i.e. in the next image) of padded slot s, a large negative number
    // for a pad slot -- so a piece costs a lane ~9 vector instructions.  (Before the table: two divisions' worth of wraps and compares
    // per lane, ~25 instructions per piece; every vector instruction of these waves costs the MFMA pipe issue cycles, and the decode
    // was a quarter of the kernel: 0.167 ms with it, 0.122 without, 14x14x256.)  The table reads are inline assembly: the compiler
    // writes vmcnt(0) in front of an LDS read that follows an LDS-DMA it cannot tell apart -- a wait for every DMA in flight.
    int* const tab = reinterpret_cast<int*>(smem + NST * STAGE);
    for (int s_ = tid; s_ < IS + 16; s_ += 64 * NWAVES) {
        const int rem = s_ >= IS ? s_ - IS : s_;
        const int r = rem / PW1, c = rem - r * PW1;
        tab[s_] = (r >= 1 && c >= 1) ? (r - 1) * W + (c - 1) + (s_ >= IS ? HW : 0) : -(1 << 30);
    }
    __syncthreads();
    // this wave's pieces of a K-piece: piece ids wid, wid + 12, ...; ids < DZ_PIECES are dz rows, then the x window rows actually read.
    // Per piece, constant over the K-pieces: kind (0 none, 1 dz, 2 x), first slot relative to the K-piece's, destination inside the
    // stage, this lane's row within the piece and its channel offset (source chunk swizzle included)
    int pkind[MAXP], prel[MAXP], pdst[MAXP], pd[MAXP], pcst[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int j = wid + NWAVES * i;
        pkind[i] = 0; prel[i] = 0; pdst[i] = 0; pd[i] = 0; pcst[i] = 0;
        if (j < DZ_PIECES) {
            const int d = lane / (DZROW / 16), pos = lane % (DZROW / 16);
            const int k = j * DZ_RPP + d;
            const int chunk = NNB >= 4 ? pos ^ ((k & 3) << 2) : pos ^ (((k >> 1) & 1) << 2);      // 128-byte rows: two 64-byte segments
            pkind[i] = 1; prel[i] = j * DZ_RPP; pdst[i] = j * 1024; pd[i] = d; pcst[i] = n0 + chunk * 8;
        } else if (j < PIECES && (j - DZ_PIECES) * X_RPP < KP + 2 * PW1 + 2) {
            const int jx = j - DZ_PIECES;
            const int d = lane / (XROW / 16), pos = lane % (XROW / 16);
            const int r = jx * X_RPP + d;
            const int chunk = NCB == 1 ? pos : pos ^ (((r >> 1) & 1) << 2);
            pkind[i] = 2; prel[i] = jx * X_RPP - PW1 - 1; pdst[i] = KP * DZROW + jx * 1024; pd[i] = d; pcst[i] = c0 + chunk * 8;
        }
    }
    auto issue_range = [&](int t, int st, int i0, int i1) __attribute__((always_inline)) -> int {
        int n = 0;
        if (dbg_ & 16) return 0;                      // timing experiment: no address decode, no DMA
        int tv[MAXP], mb[MAXP], su[MAXP];
        unsigned ta[MAXP];
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            tv[i] = 0; mb[i] = 0; su[i] = 0; ta[i] = (unsigned)(NST * STAGE);
            if (i < i0 || i >= i1 || pkind[i] == 0) continue;
            const int Su = __builtin_amdgcn_readfirstlane(kbeg + t * KP + prel[i]);
            const int neg = Su < 0 ? 1 : 0;                                  // a window that starts ahead of the tensor (> -IS): one image up
            const int S = Su + neg * IS;
            const int img = (int)(((unsigned long long)(unsigned)S * magic_is) >> 40) - neg;          // scalar unit
            const int rem0 = S - (img + neg) * IS;
            mb[i] = img * HW; su[i] = Su;
            ta[i] = (unsigned)(NST * STAGE) + (unsigned)(rem0 + pd[i]) * 4u;
        }
        // reads and their wait in ONE statement: the compiler takes an asm's outputs for complete and copied them to other registers
        // right behind a lone ds_read (an absent piece reads tab[0])
        static_assert(MAXP >= 1 && MAXP <= 4, "table reads");
        if constexpr (MAXP == 4)
            asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %5\n\tds_read_b32 %2, %6\n\tds_read_b32 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(tv[0]), "=&v"(tv[1]), "=&v"(tv[2]), "=&v"(tv[3]) : "v"(ta[0]), "v"(ta[1]), "v"(ta[2]), "v"(ta[3]) : "memory");
        else if constexpr (MAXP == 3)
            asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %4\n\tds_read_b32 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(tv[0]), "=&v"(tv[1]), "=&v"(tv[2]) : "v"(ta[0]), "v"(ta[1]), "v"(ta[2]) : "memory");
        else if constexpr (MAXP == 2)
            asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(tv[0]), "=&v"(tv[1]) : "v"(ta[0]), "v"(ta[1]) : "memory");
        else
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(tv[0]) : "v"(ta[0]) : "memory");
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            if (i < i0 || i >= i1 || pkind[i] == 0) continue;
            if ((dbg_ & 4) && pkind[i] == 2) continue;      // timing experiments (wrong results): no x pieces / no dz pieces
            if ((dbg_ & 8) && pkind[i] == 1) continue;
            const int m = mb[i] + tv[i];                                    // pixel; negative for a pad slot or a slot ahead of the tensor
            const bool ok = m >= 0 && su[i] + pd[i] < KT;
            const unsigned voff = ok ? (unsigned)(__mul24(m, pkind[i] == 2 ? cin_ : cout_) + pcst[i]) * 2u : OOB;
            dma16(pkind[i] == 2 ? rsrcX : rsrcD, smem + st * STAGE + pdst[i], voff, 0);
            ++n;
        }
        return n;
    };
    auto issue = [&](int t, int st) __attribute__((always_inline)) -> int { return issue_range(t, st, 0, MAXP); };
    // Stamped: a wave's DMA phase (address decode + up to four LDS-DMA instructions) is 1300-1800 cycles, its MFMAs 770 of pipe time;
    // with every wave doing MFMAs first and DMAs last, the last of a SIMD's three waves exposed its whole DMA phase at the end of each
    // K-piece (period 4900 for 2300 cycles of MFMA).  The three tap groups (one wave of each per SIMD) take turns instead: group 1
    // issues its DMAs before its MFMAs, group 0 after, group 2 half and half.
    const int npre = (dbg_ & 2) ? 0 : (tg == 1 ? MAXP : tg == 2 ? MAXP / 2 : 0);

    f32x16 acc[3][NB2];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < NB2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // transposing-read lane map (igemm.hip frag_tr): of each 16 lanes, lane 4 q + pp addresses k-row q, columns 4 pp .. 4 pp + 3
    const int g4 = lane >> 4, idx = lane & 15;
    const int kk = 8 * (g4 >> 1) + (idx >> 2);                   // k-row within a 16-deep step (second read: + 4)
    const int mm = 16 * (g4 & 1) + 4 * (idx & 3);                // column within a 32-wide block
    // byte offsets inside a stage, for sub-step 0: dz fragments of the two column blocks, x fragments of the three taps
    unsigned dz_off[NB2], x_off[3];
#pragma unroll
    for (int b = 0; b < NB2; ++b) {
        const int col = (nb0 + b) * 32 + mm;
        dz_off[b] = (unsigned)(kk * DZROW + (((col >> 3) ^ (NNB >= 4 ? (kk & 3) << 2 : ((kk >> 1) & 1) << 2)) << 4) + (col & 7) * 2);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int tap = 3 * tg + a;
        // tap = 3 (dh + 1) + (dw + 1) (the HWIO order); NOT p.dh[tap]: a dynamically indexed kernel-argument array makes the compiler
        // copy the whole parameter block to scratch and reload fields from there -- inside the K loop, each reload with a vmcnt(0)
        const int r0 = (tap / 3) * PW1 + tap % 3 + kk;                       // window row of this lane's first k
        const int col = cb * 32 + mm;
        if constexpr (NCB == 1) x_off[a] = (unsigned)(KP * DZROW + r0 * XROW + col * 2);
        else x_off[a] = (unsigned)(KP * DZROW + r0 * XROW + (((col >> 3) ^ (((r0 >> 1) & 1) << 2)) << 4) + (col & 7) * 2);
    }
    // (CT = 64: the x swizzle is keyed by bit 1 of the window row; rows r0 + 16 h keep it, row + 4 keeps it: one offset per tap)
    // (addresses as integers of the LDS address space: `base` = stage + this lane's fragment offset, once per K-piece; the sub-step and
    // the second k-half are compile-time and end in the read's immediate offset -- through `smem + off` every read paid a vector add,
    // 33 beside the 24 MFMAs of a K-piece, and each vector instruction of these waves costs the MFMA pipe issue cycles)
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    auto tr8 = [&](unsigned base, int imm, int rowbytes) __attribute__((always_inline)) -> bf16x8 {
        const s16x4t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4t*)(base + (unsigned)imm));
        const s16x4t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4t*)(base + (unsigned)(imm + 4 * rowbytes)));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    // ---- ring: the DMAs of K-piece t + 2 go out after the barrier of K-piece t ----
    int n1 = 0;                                                   // pieces this wave issued for the K-piece after the awaited one
    if (np > 0) issue(0, 0);
    if (np > 1) n1 = issue(1, 1);
    int st = 0;
    for (int t = 0; t < np; ++t) {
#if WG_STAMP
        unsigned long long s0_ = stamp_now();
#endif
        wait_vmcnt_upto(t + 1 < np ? n1 : 0);
#if WG_STAMP
        unsigned long long s1_ = stamp_now();
#endif
        __builtin_amdgcn_s_barrier();
#if WG_STAMP
        unsigned long long s2_ = stamp_now();
#endif
        const bool fill = t + 2 < np;
        const int fst = st + 2 >= NST ? st + 2 - NST : st + 2;
        int nfill = 0;
        if (fill) nfill = issue_range(t + 2, fst, 0, npre);
        const unsigned sb = (unsigned)(st * STAGE);
        // fragments single-buffered (the registers of a 16-deep step are free once its six MFMAs are issued; the two other waves of the
        // SIMD cover the read latency): a second set does not fit beside six accumulators in 168 registers -- and a spill reloaded
        // inside this loop brings a vmcnt(0) with it, i.e. a wait for every DMA in flight
        constexpr int NB_ = WG_PFD + 1;
        bf16x8 fx[NB_][3], fd[NB_][NB2];
        unsigned vd[NB2], vx[3];
#pragma unroll
        for (int b = 0; b < NB2; ++b) vd[b] = lds0 + sb + dz_off[b];
#pragma unroll
        for (int a = 0; a < 3; ++a) vx[a] = lds0 + sb + x_off[a];
        auto ld = [&](int h, int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int b = 0; b < NB2; ++b) fd[buf][b] = tr8(vd[b], h * 16 * DZROW, DZROW);
#pragma unroll
            for (int a = 0; a < 3; ++a) fx[buf][a] = tr8(vx[a], h * 16 * XROW, XROW);
        };
        if (WG_PFD) ld(0, 0);
#pragma unroll
        for (int h = 0; h < KP / 16; ++h) {
            if (WG_PFD == 0) ld(h, 0);
            else if (h + 1 < KP / 16) ld(h + 1, (h + 1) & 1);
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < NB2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fx[WG_PFD ? (h & 1) : 0][a], fd[WG_PFD ? (h & 1) : 0][b], acc[a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#if WG_STAMP
        unsigned long long s3_ = stamp_now();
#endif
        if (fill) nfill += issue_range(t + 2, fst, npre, MAXP);
#if WG_STAMP
        {
            unsigned long long s4_ = stamp_now();
            if (blockIdx.x == 0 && (wid == 0 || wid == NWAVES - 1) && lane == 0 && t < 60) {
                unsigned long long* o = p.stamps + ((wid ? 1 : 0) * 60 + t) * 5;
                o[0] = s0_; o[1] = s1_; o[2] = s2_; o[3] = s3_; o[4] = s4_;
            }
        }
#endif
        if (fill) n1 = nfill;
        st = st + 1 == NST ? 0 : st + 1;
    }

    // ---- partial slab: rows (tap, cin), 32 consecutive output channels per half-wave store ----
    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int tap = 3 * tg + a;
#pragma unroll
        for (int b = 0; b < NB2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                out[((long)tap * p.cin + c0 + cb * 32 + row) * p.cout + n0 + (nb0 + b) * 32 + li] = acc[a][b][r];
            }
    }
}

template <int CT, int BN, int XCAP>
hipError_t launch(const Wgrad16Params& p, hipStream_t st) {
    constexpr int NST = 3;
    constexpr int TABCAP = (XCAP >= 192 ? 57 * 57 : 31 * 31) + 16;      // slot table: the widest layer of this configuration (wgrad16_plan)
    if ((p.H + 1) * (p.W + 1) + 16 > TABCAP) return hipErrorInvalidValue;
    const size_t lds = (size_t)NST * (KP * BN * 2 + XCAP * CT * 2) + 1024 + (size_t)TABCAP * sizeof(int);
    auto kern = wgrad16_kernel<CT, BN, NST, XCAP>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int npairs = (p.cin / CT) * (p.cout / BN);
#if WG_STAMP
    {
        static unsigned long long* buf = nullptr;
        const size_t nb = 2 * 60 * 5 * sizeof(unsigned long long);
        if (!buf && hipMalloc(&buf, nb) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemsetAsync(buf, 0, nb, st);
        Wgrad16Params q = p;
        q.stamps = buf;
        hipLaunchKernelGGL(kern, dim3(p.S * npairs), dim3(64 * NWAVES), lds, st, q);
        (void)hipStreamSynchronize(st);
        static unsigned long long h[2 * 60 * 5];
        (void)hipMemcpy(h, buf, nb, hipMemcpyDeviceToHost);
        fprintf(stderr, "[stamp] wgrad16<%d,%d> %dx%d cin %d cout %d S %d kper %d\n", CT, BN, p.H, p.W, p.cin, p.cout, p.S, p.kper);
        for (int w = 0; w < 2; ++w) {
            double a = 0, b = 0, c = 0, d = 0, per = 0; int n = 0;
            for (int t = 4; t < 50; ++t) {
                const unsigned long long* r = h + (w * 60 + t) * 5;
                if (!r[0] || !h[(w * 60 + t - 1) * 5]) break;
                a += (double)(r[1] - r[0]); b += (double)(r[2] - r[1]); c += (double)(r[3] - r[2]); d += (double)(r[4] - r[3]);
                per += (double)(r[0] - h[(w * 60 + t - 1) * 5]); ++n;
            }
            if (n) fprintf(stderr, "[stamp]  wave %d: per K-piece period %.0f = vmcnt wait %.0f + barrier %.0f + reads and MFMAs %.0f + DMA issue %.0f + rest\n",
                           w ? NWAVES - 1 : 0, per / n, a / n, b / n, c / n, d / n);
        }
        return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL(kern, dim3(p.S * npairs), dim3(64 * NWAVES), lds, st, p);
    return hipGetLastError();
}


// ---- 1x1 convs, stride 1 (the BN nets' bottlenecks): dW[cin][cout] = sum over pixels p of x[p][cin] dz[p][cout] -------------------
// The same staging without taps, slots or address decode: a K-piece is 64 PIXELS, x tile [64][CT] and dz tile [64][BN] LDS-DMA'd as they
// lie in memory (rows of 256 or 512 bytes; 16-byte chunk c of row k at c ^ ((k & 3) << 2), the swizzle on the DMA's source chunk), the
// operands back through ds_read_b64_tr_b16, eight waves of (CT / WM) x (BN / WN) accumulators, three stages, one barrier per K-piece,
// ranges of the pixel axis -> partial slabs.  The per-tile kernel (igemm.hip, BF = 2) ran these at ~6 % of the bf16 peak (46 us for a
// launch whose operands are 77 MB): 17 vector instructions per MFMA in its loop.
template <int CT, int BN, int WM, int WN>
__global__ __launch_bounds__(512, 1) void wgrad16p_kernel(const Wgrad16Params p) {
    constexpr int TM = CT / WM / 32, TN = BN / WN / 32, NST = 3, NW = 8;
    static_assert(WM * WN == NW && CT >= 64 && BN >= 64 && TM >= 1 && TN >= 1, "eight waves; rows of 128 bytes or more");
    constexpr int XROW = CT * 2, DROW = BN * 2;
    constexpr int X_RPP = 1024 / XROW, D_RPP = 1024 / DROW;       // rows per 1-KiB DMA piece
    constexpr int X_PIECES = KP / X_RPP, D_PIECES = KP / D_RPP, PIECES = X_PIECES + D_PIECES;
    constexpr int MAXP = (PIECES + NW - 1) / NW;
    static_assert(MAXP <= 6, "wait_vmcnt_upto");
    constexpr int STAGE = KP * (XROW + DROW);
    // swizzle key of row k (XOR-ed into the 16-byte chunk index): four consecutive rows of >= 256 bytes share their banks -> two bits of
    // k move the chunk by 64 bytes; rows of 128 bytes pair up -> one bit (the transposing read takes rows k, k+1, k+2, k+3 of 16 lanes)
    auto xkey = [](int k) { return XROW >= 256 ? (k & 3) << 2 : ((k >> 1) & 1) << 2; };
    auto dkey = [](int k) { return DROW >= 256 ? (k & 3) << 2 : ((k >> 1) & 1) << 2; };
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int npairs = (p.cin / CT) * (p.cout / BN);
    int split, pair;
    if (p.S >= 8) { split = (blockIdx.x & 7) + 8 * ((blockIdx.x >> 3) / npairs); pair = (blockIdx.x >> 3) % npairs; }
    else { split = blockIdx.x % p.S; pair = blockIdx.x / p.S; }
    const int c0 = (pair / (p.cout / BN)) * CT, n0 = (pair % (p.cout / BN)) * BN;
    const int KT = p.n * p.H * p.W;                              // pixels
    const int kbeg = split * p.kper;
    const int kend = min(kbeg + p.kper, (KT + KP - 1) / KP * KP);
    const int np = kend > kbeg ? (kend - kbeg) / KP : 0;
    float* const out = p.out + (long)split * p.slab;
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcD = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.dz), 0, p.dz_bytes, 0x00020000);
    // this wave's DMA pieces of a K-piece (ids wid, wid + 8, ...; the first X_PIECES are x rows): constant over the K-pieces are the
    // destination, this lane's row within the K-piece and its source offset at K-piece 0; a K-piece later the source is KP rows on
    int pkind[MAXP], pdst[MAXP], prow[MAXP];
    unsigned pv0[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int j = wid + NW * i;
        pkind[i] = 0; pdst[i] = 0; prow[i] = 0; pv0[i] = 0;
        if (j < X_PIECES) {
            const int row = j * X_RPP + lane / (XROW / 16), pos = lane % (XROW / 16);
            pkind[i] = 2; pdst[i] = j * 1024; prow[i] = row;
            pv0[i] = (unsigned)((kbeg + row) * p.cin + c0 + ((pos ^ xkey(row)) << 3)) * 2u;
        } else if (j < PIECES) {
            const int jd = j - X_PIECES;
            const int row = jd * D_RPP + lane / (DROW / 16), pos = lane % (DROW / 16);
            pkind[i] = 1; pdst[i] = KP * XROW + jd * 1024; prow[i] = row;
            pv0[i] = (unsigned)((kbeg + row) * p.cout + n0 + ((pos ^ dkey(row)) << 3)) * 2u;
        }
    }
    const unsigned xstep = (unsigned)(KP * p.cin) * 2u, dstep = (unsigned)(KP * p.cout) * 2u;
    auto issue = [&](int t, int st) __attribute__((always_inline)) -> int {
        int n = 0;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            if (pkind[i] == 0) continue;
            const bool ok = kbeg + t * KP + prow[i] < KT;
            const unsigned voff = ok ? pv0[i] + (unsigned)t * (pkind[i] == 2 ? xstep : dstep) : OOB;
            dma16(pkind[i] == 2 ? rsrcX : rsrcD, smem + st * STAGE + pdst[i], voff, 0);
            ++n;
        }
        return n;
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // transposing-read lane map (see wgrad16_kernel)
    const int g4 = lane >> 4, idx = lane & 15;
    const int kk = 8 * (g4 >> 1) + (idx >> 2);
    const int mm = 16 * (g4 & 1) + 4 * (idx & 3);
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    unsigned x_off[TM], d_off[TN];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        const int col = wm * (TM * 32) + a * 32 + mm;
        x_off[a] = lds0 + (unsigned)(kk * XROW + (((col >> 3) ^ xkey(kk)) << 4) + (col & 7) * 2);
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int col = wn * (TN * 32) + b * 32 + mm;
        d_off[b] = lds0 + (unsigned)(KP * XROW + kk * DROW + (((col >> 3) ^ dkey(kk)) << 4) + (col & 7) * 2);
    }
    auto tr8 = [&](unsigned base, int imm, int rowbytes) __attribute__((always_inline)) -> bf16x8 {
        const s16x4t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4t*)(base + (unsigned)imm));
        const s16x4t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4t*)(base + (unsigned)(imm + 4 * rowbytes)));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    int n1 = 0;
    if (np > 0) issue(0, 0);
    if (np > 1) n1 = issue(1, 1);
    int st = 0;
    for (int t = 0; t < np; ++t) {
        wait_vmcnt_upto(t + 1 < np ? n1 : 0);
        __builtin_amdgcn_s_barrier();
        const unsigned sb = (unsigned)(st * STAGE);
        unsigned vx[TM], vd[TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) vx[a] = sb + x_off[a];
#pragma unroll
        for (int b = 0; b < TN; ++b) vd[b] = sb + d_off[b];
        bf16x8 fx[2][TM], fd[2][TN];
        auto ld = [&](int h, int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int a = 0; a < TM; ++a) fx[buf][a] = tr8(vx[a], h * 16 * XROW, XROW);
#pragma unroll
            for (int b = 0; b < TN; ++b) fd[buf][b] = tr8(vd[b], h * 16 * DROW, DROW);
        };
        ld(0, 0);
#pragma unroll
        for (int h = 0; h < KP / 16; ++h) {
            if (h + 1 < KP / 16) ld(h + 1, (h + 1) & 1);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fx[h & 1][a], fd[h & 1][b], acc[a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (t + 2 < np) n1 = issue(t + 2, st + 2 >= NST ? st + 2 - NST : st + 2);
        st = st + 1 == NST ? 0 : st + 1;
    }
    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                out[(long)(c0 + wm * (TM * 32) + a * 32 + row) * p.cout + n0 + wn * (TN * 32) + b * 32 + li] = acc[a][b][r];
            }
}

template <int CT, int BN, int WM, int WN>
hipError_t launch_p(const Wgrad16Params& p, hipStream_t st) {
    const size_t lds = (size_t)3 * KP * (CT * 2 + BN * 2);
    auto kern = wgrad16p_kernel<CT, BN, WM, WN>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int npairs = (p.cin / CT) * (p.cout / BN);
    hipLaunchKernelGGL(kern, dim3(p.S * npairs), dim3(512), lds, st, p);
    return hipGetLastError();
}

}  // namespace

bool wgrad16_plan(int n, int h, int w, int cin, int cout, Wgrad16Params* p, int* cfg) {
    static const bool on = !(getenv("FTE_WGRAD16_RESIDENT") && atoi(getenv("FTE_WGRAD16_RESIDENT")) == 0);
    if (!on || w > 62 || w < 7 || h < 7) return false;
    int ct, bn;
    if (w <= 30 && cout % 256 == 0 && cin % 32 == 0) { ct = 32; bn = 256; *cfg = 0; }
    else if (w <= 30 && cout % 128 == 0 && cin % 64 == 0) { ct = 64; bn = 128; *cfg = 1; }
    else if (cout % 64 == 0 && cin % 64 == 0) { ct = 64; bn = 64; *cfg = 2; }             // the 64 -> 64 layers (56x56): 192 window rows
    else return false;
    if ((long)(h + 1) * (w + 1) > (*cfg == 2 ? 57 * 57 : 31 * 31)) return false;      // the kernel's slot table (launch: TABCAP)
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
        cus = prop.multiProcessorCount;
    }
    const int npairs = (cin / ct) * (cout / bn);
    int S = cus / npairs;
    if (S >= 8) S &= ~7; else if (S >= 4) S = 4; else if (S >= 2) S = 2; else S = 1;
    const long KT = (long)n * (h + 1) * (w + 1);
    const long pieces = (KT + KP - 1) / KP;
    // at least `minp` K-pieces per block: S is LOWERED until that holds (round 4 returned to the per-tile kernel's plan instead, which put
    // the BN nets' 14x14 / 7x7 3x3 layers at a 128-image shard -- 225 / 64 pieces -- on the register-staged 64x64 kernel: 44-46 us each)
    static const long minp = getenv("FTE_WGRAD16_MINP") ? atol(getenv("FTE_WGRAD16_MINP")) : 8;
    static const bool lower = !(getenv("FTE_WGRAD16_LOWER_S") && atoi(getenv("FTE_WGRAD16_LOWER_S")) == 0);
    if (KT >= (1L << 24)) return false;
    if (pieces < minp * S) {
        if (!lower) return false;
        S = (int)(pieces / minp);
        if (S >= 8) S &= ~7; else if (S >= 4) S = 4; else if (S >= 2) S = 2; else S = 1;
        if (pieces < 2) return false;
    }
    p->kper = (int)((pieces + S - 1) / S) * KP;
    p->S = S;                                                    // (a last range may come out short or empty: it writes a zero slab)
    p->n = n; p->H = h; p->W = w; p->cin = cin; p->cout = cout;
    static const int dbg = getenv("FTE_WGRAD16_DBG") ? atoi(getenv("FTE_WGRAD16_DBG")) : 0;
    p->dbg = dbg;
    // q = (s * magic) >> 40 is floor(s / d) for s < 2^24, d < 2^16
    p->magic_is = ((1ull << 40) / (unsigned long long)((h + 1) * (w + 1))) + 1;
    p->magic_pw1 = ((1ull << 40) / (unsigned long long)(w + 1)) + 1;
    p->slab = 9L * cin * cout;
    return true;
}

hipError_t wgrad16_launch(const Wgrad16Params& p, int cfg, hipStream_t st) {
    return cfg == 0 ? launch<32, 256, 128>(p, st) : cfg == 1 ? launch<64, 128, 128>(p, st) : launch<64, 64, 192>(p, st);
}

// 1x1 / stride 1: cfg 0: 128 cin x 256 cout per block, 1: 256 x 128, 2: 128 x 128
bool wgrad16p_plan(int n, int h, int w, int cin, int cout, Wgrad16Params* p, int* cfg) {
    static const bool on = !(getenv("FTE_WGRAD16_POINTWISE") && atoi(getenv("FTE_WGRAD16_POINTWISE")) == 0);
    if (!on) return false;
    int ct, bn;
    if (cin % 128 == 0 && cout % 256 == 0) { ct = 128; bn = 256; *cfg = 0; }
    else if (cin % 256 == 0 && cout % 128 == 0) { ct = 256; bn = 128; *cfg = 1; }
    else if (cin % 128 == 0 && cout % 128 == 0) { ct = 128; bn = 128; *cfg = 2; }
    else if (cin % 64 == 0 && cout % 256 == 0) { ct = 64; bn = 256; *cfg = 3; }
    else if (cin % 256 == 0 && cout % 64 == 0) { ct = 256; bn = 64; *cfg = 4; }
    else if (cin % 64 == 0 && cout % 128 == 0) { ct = 64; bn = 128; *cfg = 5; }
    else if (cin % 128 == 0 && cout % 64 == 0) { ct = 128; bn = 64; *cfg = 6; }
    else return false;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
        cus = prop.multiProcessorCount;
    }
    const int npairs = (cin / ct) * (cout / bn);
    const long KT = (long)n * h * w;
    const long pieces = (KT + KP - 1) / KP;
    int S = cus / npairs;
    if (S >= 8) S &= ~7; else if (S >= 4) S = 4; else if (S >= 2) S = 2; else S = 1;
    while (S > 1 && pieces < 4L * S) S >>= 1;                    // at least four K-pieces per block
    if (pieces < 4 || KT * (cin > cout ? cin : cout) * 2 >= (1L << 31)) return false;
    p->kper = (int)((pieces + S - 1) / S) * KP;
    p->S = S;
    p->n = n; p->H = h; p->W = w; p->cin = cin; p->cout = cout;
    p->slab = (long)cin * cout;
    p->dbg = 0; p->stamps = nullptr; p->magic_is = 0; p->magic_pw1 = 0;
    return true;
}

hipError_t wgrad16p_launch(const Wgrad16Params& p, int cfg, hipStream_t st) {
    switch (cfg) {
    case 0: return launch_p<128, 256, 2, 4>(p, st);
    case 1: return launch_p<256, 128, 4, 2>(p, st);
    case 2: return launch_p<128, 128, 2, 4>(p, st);
    case 3: return launch_p<64, 256, 1, 8>(p, st);
    case 4: return launch_p<256, 64, 8, 1>(p, st);
    case 5: return launch_p<64, 128, 2, 4>(p, st);
    default: return launch_p<128, 64, 4, 2>(p, st);
    }
}
