// wino.hip -- Winograd F(2x2, 3x3) for the stride-1 3x3 convolutions, fp32, gfx950 (MI355X).
//
// The reference enables cuDNN's Winograd algorithm for every run (train.py:260, TF_ENABLE_WINOGRAD_NONFUSED=1); 16 of
// SphereNet-20's 20 convs are stride-1 3x3 resBlock layers (nets/sphere.py:38-45,58-70) = 90.5 % of the MACs.  On a chip whose fp32
// matrix rate is 1/16 of its bf16 rate (157 TFLOP/s, `v_mfma_f32_32x32x2_f32`) and whose 3x3 layers sit 5-40x above the fp32 ridge,
// 2.25x fewer multiplies for a few hundred MB of transform traffic is the MI355X-first trade.
//
//   forward        y  = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A      d = 4x4 input patch of a 2x2 output tile, g = 3x3 filter
//   data gradient  the same with d = dz patches and the filters rotated by 180 degrees, channels swapped
//   filter gradient dw = A'^T [ sum_tiles (G' e G'^T) (.) (B^T d B) ] A'   F(3x3, 2x2): e = 2x2 tile of dz, the SAME B^T d B of x
//
// Pieces (all fp32, exact transforms: the matrices hold 0, +-1, +-1/2 only):
//   wino_tiles_kernel<0/1>   HBM-bound: x (or dz) -> V = B^T d B per tile, or dz -> U' = G' e G'^T, written in the "pack" layout
//                            (wino.h): one contiguous 32 KiB slab per (64 tiles, 8 channels) = the LDS image of one K-step
//   wino_filter_kernel       G g G^T of every filter, same pack layout with rows = output channels (4 MB for 256 x 256)
//   wino_mm_kernel           the 16 products [tiles x K] x [K x N] of ONE 64-tile x 64-channel block in one workgroup: 8 waves, each
//                            8 of the 16 t-planes of a 32 x 32 sub-tile (128 accumulator registers), operands global -> LDS by
//                            LDS-DMA (no VGPR staging, no address arithmetic: slabs are contiguous), two 64 KiB stages, one barrier per
//                            K-step of 8 channels; the OUTPUT TRANSFORM runs on the accumulators (rows i of A^T M A split between
//                            the two t-half waves, two floats per position exchanged through LDS) and feeds the same fused
//                            epilogues as igemm_dev.h (bias / PReLU / shortcut; PReLU gradient + dalpha / dbias partial rows)
//   wino_wgrad_kernel        16 products [cin x tiles] x [tiles x cout]: one resident workgroup per CU (64 x 64 x 16 t block of the
//                            result, an equal share of the tiles), partial slabs -> wino_wgrad_finish_kernel (ordered sum over the
//                            shares + A'^T . A')
// Everything is deterministic: fixed summation orders, no atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "igemm.h"
#include "igemm_dev.h"
#include "wino.h"

#ifdef FTE_WINO_STAMP
// DIAGNOSTIC BUILD ONLY (scripts/dev/build_wino_stamp.sh -> variants/libfte_wstamp.so; never the product library): every resident block of
// wino_mm_kernel stamps s_memtime (shader clock) / s_memrealtime (100 MHz) at its tile boundaries into a buffer nothing else reads.
__device__ unsigned long long* g_wino_stamp = nullptr;
extern "C" int fte_debug_set_wino_stamp(void* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wino_stamp), &buf, sizeof(buf));
}
#endif

namespace {

using namespace igemm_dev;

typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t& r, char* lds, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds, 16, voff, soff, 0, 0);
}

constexpr unsigned OOB = 0x80000000u;
#ifndef WMM_ABL
#define WMM_ABL 0      // diagnostic builds only (scripts/dev/build_wino_stamp.sh, EXTRA=-DWMM_ABL=n): 1 = no DMA in the K loop, 2 = no fragment reads, 4 = no mid-step barrier (wrong results, right timing)
#endif
constexpr int SLAB_F = 16 * 64 * 8;            // floats per (row block, K-step) slab
constexpr int SLAB_B = SLAB_F * 4;             // 32 KiB

// ---------------------------------------------------------------------------------------------------------------------------------
// tile transforms (HBM-bound).  Block = 64 tiles x 32 channels (one 128-byte line of every pixel it touches), 512 threads; thread =
// (tile, channel quad of 8), a wave = 8 tiles x 8 quads: loads are whole 128-byte pixel pieces, stores 256 contiguous bytes per (K-step,
// plane).  Planes are formed and stored a row at a time: 68 registers, so that a block (2 waves per SIMD) fits on a CU BESIDE a resident
// block of the filter-gradient kernel (2 waves x 176 registers per SIMD) -- which is how the data-gradient side's transforms run
// under the MFMA-bound kernel of the other stream (nets/sphere.py _body_walk: one stream 34.85 ms per step, two 33.9-34.1).
// TPB = 256: half a row block (32 tiles) per block, one wave per SIMD -- 72 registers per SIMD, what is left beside a resident block of
// the FORWARD product (2 x 216): the forward walk's two half shards (nets/sphere.py backbone) run one half's transform under the other's product.
template <int MODE, int TPB>
__global__ __launch_bounds__(TPB) void wino_tiles_kernel(const float* __restrict__ x, float* __restrict__ pack, int H, int W, int C,
                                                         int TH, int TW, long M, int MB, unsigned x_bytes) {
    constexpr int P = MODE == 0 ? 4 : 2;
    constexpr int SUB = 512 / TPB;             // blocks per row block
    const int tid = threadIdx.x;
    const int c4 = tid & 7;
    // consecutive block ids go round the 8 XCDs: an XCD takes a CONTIGUOUS range of row blocks (neighbours share halo rows in its L2)
    const int per = (MB + 7) >> 3;
    const int slot = blockIdx.x >> 3;
    const int mb = (blockIdx.x & 7) * per + slot / SUB, KS = C >> 3;
    if (mb >= MB) return;
    const int r = (slot % SUB) * (64 / SUB) + (tid >> 3);
    const long m = (long)mb * 64 + r;
    const int tpi = TH * TW;
    int n = 0, ty = 0, tx = 0;
    const bool mv = m < M;
    if (mv) { n = (int)(m / tpi); const int rem = (int)(m - (long)n * tpi); ty = rem / TW; tx = rem - ty * TW; }
    const int y0 = 2 * ty - (MODE == 0 ? 1 : 0), x0 = 2 * tx - (MODE == 0 ? 1 : 0);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, x_bytes, 0x00020000);
    const unsigned choff = (unsigned)(blockIdx.y * 32 + c4 * 4) * 4u;
    f32x4 d[P][P];
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int yy = y0 + i, xx = x0 + j;
            const bool ok = mv && yy >= 0 && yy < H && xx >= 0 && xx < W;
            d[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (unsigned)(((n * H + yy) * W + xx) * C) * 4u + choff : OOB, 0, 0));
        }
    const int ks = blockIdx.y * 4 + (c4 >> 1), sw = (r >> 4) & 1;
    float* o = pack + (((size_t)mb * KS + ks) * 16 * 64 + r) * 8 + (((c4 & 1) ^ sw) << 2);
    if constexpr (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {          // row i of B^T d, then (B^T d) B: planes 4 i .. 4 i + 3
            f32x4 w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                w[j] = i == 0 ? d[0][j] - d[2][j] : i == 1 ? d[1][j] + d[2][j] : i == 2 ? d[2][j] - d[1][j] : d[1][j] - d[3][j];
            *reinterpret_cast<f32x4*>(o + (size_t)(4 * i + 0) * 512) = w[0] - w[2];
            *reinterpret_cast<f32x4*>(o + (size_t)(4 * i + 1) * 512) = w[1] + w[2];
            *reinterpret_cast<f32x4*>(o + (size_t)(4 * i + 2) * 512) = w[2] - w[1];
            *reinterpret_cast<f32x4*>(o + (size_t)(4 * i + 3) * 512) = w[1] - w[3];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 u[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
                u[j] = i == 0 ? d[0][j] : i == 1 ? 0.5f * (d[0][j] + d[1][j]) : i == 2 ? 0.5f * (d[0][j] - d[1][j]) : d[1][j];
            *reinterpret_cast<f32x4*>(o + (size_t)(4 * i + 0) * 512) = u[0];
            *reinterpret_cast<f32x4*>(o + (size_t)(4 * i + 1) * 512) = 0.5f * (u[0] + u[1]);
            *reinterpret_cast<f32x4*>(o + (size_t)(4 * i + 2) * 512) = 0.5f * (u[0] - u[1]);
            *reinterpret_cast<f32x4*>(o + (size_t)(4 * i + 3) * 512) = u[1];
        }
    }
}

// filter transform: one thread per (k, row) pair.  dgrad = 0: g[kh][kw] = w[kh][kw][k][row]; dgrad = 1: g[kh][kw] = w[2-kh][2-kw][row][k]
__global__ __launch_bounds__(256) void wino_filter_kernel(const float* __restrict__ w, float* __restrict__ pack, int cin, int cout, int dgrad) {
    const int K = dgrad ? cout : cin, NR = dgrad ? cin : cout;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= K * NR) return;
    int k, row;
    if (dgrad) { k = idx % K; row = idx / K; }
    else { row = idx % NR; k = idx / NR; }
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
            g[a][b] = dgrad ? w[(((2 - a) * 3 + (2 - b)) * (size_t)cin + row) * cout + k] : w[((a * 3 + b) * (size_t)cin + k) * cout + row];
    float p[4][3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        p[0][b] = g[0][b];
        p[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
        p[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
        p[3][b] = g[2][b];
    }
    const int KS = K >> 3, rb = row >> 6, r = row & 63, c8 = k & 7;
    float* o = pack + ((((size_t)rb * KS + (k >> 3)) * 16) * 64 + r) * 8 + (((c8 >> 2) ^ ((r >> 4) & 1)) << 2) + (c8 & 3);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o[(size_t)(4 * i + 0) * 512] = p[i][0];
        o[(size_t)(4 * i + 1) * 512] = 0.5f * (p[i][0] + p[i][1] + p[i][2]);
        o[(size_t)(4 * i + 2) * 512] = 0.5f * (p[i][0] - p[i][1] + p[i][2]);
        o[(size_t)(4 * i + 3) * 512] = p[i][2];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// wino_mm_kernel: block = 64 tiles x 64 output channels x 16 t.  Wave (q, mi, ni) = wid bits 0, 1, 2: t-planes [8q, 8q+8) of tile rows
// [32 mi, +32) x channels [32 ni, +32).  MFMA operand map (32x32x2): a = V[t][tile li][k], b = U[t][channel li][k] with k = 4 lh + j for
// the j-th MFMA of a K-step (one ds_read_b128 per operand and t feeds four MFMAs); C rows (registers) = tiles, lanes = channels.
// NWN = channel halves of the block: 2 = the 64 x 64 block on 8 waves; 1 = HALF tiles, 64 tiles x 32 channels on 4 waves, one per SIMD
// (a lone wave issues its 64-cycle MFMAs back to back: ~2600 cycles per K-step, a half tile takes 0.55 of a tile's loop) -- for launches
// that are less than half a round of whole tiles (the 7x7x512 layers at a 64-image shard: 128 tiles on 256 CUs -> 256 half tiles, 0.126
// -> 0.087 ms).  Measured and NOT used for the last, partly filled round of a many-round launch: a launch of its own costs ~11 us of
// ramp and epilogue beside its loop, as much as the resident blocks' last round saves (14x14x256 at 512 images: 0.482 -> 0.479 ms);
// four extra loader waves in the half-tile block changed nothing (r6_notes.md 7).
template <int EPI, int NWN>
__global__ __launch_bounds__(256 * NWN, 1) void wino_mm_kernel(const WinoMMParams p, int NBX, int GRP, int vid0, int nvirt) {
    constexpr int NW = 4 * NWN;                    // waves per block
    constexpr int BSTG = NWN * (SLAB_B / 2);       // bytes of a stage's filter image (NWN = 1: the block's 32 rows of the 64-row slab)
    constexpr int STG = SLAB_B + BSTG;             // bytes per stage
    extern __shared__ __attribute__((aligned(16))) char wsm[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = wid & 1, mi = (wid >> 1) & 1, ni = NWN == 2 ? wid >> 2 : 0;
    const int hb = NWN == 2 ? 0 : (int)(blockIdx.x & 1);      // NWN = 1: which channel half of the column block
    const int li = lane & 31, lh = lane >> 5;
    const int KS = p.K >> 3, NB = p.N >> 6;
    // ---- virtual block id -> (mb, nb): consecutive ids go round the 8 XCDs; an XCD keeps NBX column blocks (their filter slabs stay in
    // its L2) and walks its own class of row blocks, the NBX column blocks of a row block side by side (they share the A slabs in L2).
    // The launch is ONE resident block per CU; block b takes the virtual ids b, b + grid, ... (grid % 8 == 0: same XCD every round).
    auto decode = [&](int vid, int& mb, int& nb) -> bool {
        if (GRP > 0) {
            const int xcd = vid & 7, slot = vid >> 3;
            const int grp = xcd % GRP, cls = xcd / GRP, C = 8 / GRP;
            mb = (slot / NBX) * C + cls;
            nb = grp * NBX + slot % NBX;
        } else {
            mb = vid / NB; nb = vid - mb * NB;
        }
        return mb < p.g.MB;
    };
    // the block's tiles: NWN = 2: the valid ids vid0 + b, + grid, ... below nvirt; NWN = 1: ONE tile, the (b >> 1)-th valid id of [vid0, nvirt)
    auto next_valid = [&](int vid, int& mb, int& nb) -> int {      // first valid id >= vid of this block's sequence, or nvirt
        if constexpr (NWN == 2) {
            for (; vid < nvirt; vid += (int)gridDim.x)
                if (decode(vid, mb, nb)) return vid;
        } else if (vid == vid0) {
            int c = 0;
            for (int v = vid0; v < nvirt; ++v)
                if (decode(v, mb, nb)) {
                    if (c == (int)(blockIdx.x >> 1)) return v;
                    ++c;
                }
        }
        return nvirt;
    };
    const unsigned voff = (unsigned)lane * 16u;
    auto issue = [&](int mb, int nb, int s, int stage) {
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.V + (size_t)mb * KS * SLAB_F), 0, (unsigned)KS * SLAB_B, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.U + (size_t)nb * KS * SLAB_F), 0, (unsigned)KS * SLAB_B, 0x00020000);
        char* base = wsm + stage * STG;
#pragma unroll
        for (int i = 0; i < 32 / NW; ++i) {
            const int piece = wid + NW * i;
            dma16(rsA, base + piece * 1024, voff, (unsigned)(s * SLAB_B + piece * 1024));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {             // the filter image: 32 pieces, or the 16 of this block's channel half (plane t: piece 2 t + hb)
            const int piece = wid + NW * i;
            dma16(rsB, base + SLAB_B + piece * 1024, voff, (unsigned)(s * SLAB_B + (NWN == 2 ? piece : 2 * piece + hb) * 1024));
        }
    };
    constexpr int BPL = NWN * 1024;               // bytes per plane of the filter image
    const int a_off = ((8 * q) * 64 + 32 * mi + li) * 32 + ((lh ^ ((li >> 4) & 1)) << 4);      // bytes
    const int b_off = (8 * q) * BPL + (32 * ni + li) * 32 + ((lh ^ ((li >> 4) & 1)) << 4);
    float* xb = reinterpret_cast<float*>(wsm + STG);                  // stage 1, free after the last K-step: [NW waves][32][64 lanes]
    int* rowpix = reinterpret_cast<int*>(wsm + 2 * STG);              // [64] pixel index of output (n, 2 ty, 2 tx), or -1
    int* rowflag = rowpix + 64;                                       // bit 0: row 2 ty + 1 inside, bit 1: column 2 tx + 1 inside
    float* red = reinterpret_cast<float*>(rowflag + 64);              // [2][NW waves][32] column partials of the data-gradient epilogue

    // ---- software pipeline of the K loop (a K-step = 8 channels = 32 MFMAs per wave, in two halves of four t-planes) ---------------
    //   half 0: MFMAs on fragment set X (planes 0-3 of step g) | payload: fragment reads of planes 4-7 of step g -> set Y
    //   wait for Y and for this wave's DMA pieces of step g + 1, ONE barrier: every wave has its last fragments of step g's stage in
    //           registers (the stage is free) and step g + 1 is complete in the other stage
    //   half 1: MFMAs on Y | payload: fragment reads of planes 0-3 of step g + 1 -> X, then the 8 DMA pieces of step g + 2 into the
    //           stage just freed
    // so the MFMAs behind the barrier have their operands in registers already and the LDS-DMA issue (~60 cycles a piece) and the
    // fragment round trips sit in MFMA shadows (first form: barrier, 8 DMA issues, 16 reads, then 32 MFMAs -- 15 % of every K-step
    // was the matrix pipe waiting for that head).  The pipeline runs ACROSS tiles: behind a tile's last step come the next tile's
    // steps 0 and 1, so the epilogue runs with step 0's first fragments in registers and its stage untouched; step 1's DMA is held back
    // until the epilogue's exchange buffer (stage 1) has been read.
    int mb = 0, nb = 0;
    int vid = next_valid(NWN == 2 ? vid0 + (int)blockIdx.x : vid0, mb, nb);
    if (vid >= nvirt) return;
    f32x4 xa[4], xb4[4], ya[4], yb[4];
    auto read_frags = [&](int stage, int half, f32x4 (&fa)[4], f32x4 (&fb)[4], int t) {
        const char* As = wsm + stage * STG;
        fa[t] = *reinterpret_cast<const f32x4*>(As + a_off + (4 * half + t) * 2048);
        fb[t] = *reinterpret_cast<const f32x4*>(As + SLAB_B + b_off + (4 * half + t) * BPL);
    };
    auto dma_piece = [&](const __amdgpu_buffer_rsrc_t& rsA, const __amdgpu_buffer_rsrc_t& rsB, int s, int stage, int i) {
        // piece i of this wave's list for one K-step: NWN = 2: 8 (A and filter pieces alternating); NWN = 1: 12 (8 of A, then 4 filter pieces)
        char* base = wsm + stage * STG;
        if constexpr (NWN == 2) {
            const int piece = wid + 8 * (i >> 1);
            if (i & 1) dma16(rsB, base + SLAB_B + piece * 1024, voff, (unsigned)(s * SLAB_B + piece * 1024));
            else dma16(rsA, base + piece * 1024, voff, (unsigned)(s * SLAB_B + piece * 1024));
        } else {
            const int piece = wid + 4 * (i & 7);
            if (i >= 8) dma16(rsB, base + SLAB_B + piece * 1024, voff, (unsigned)(s * SLAB_B + (2 * piece + hb) * 1024));
            else dma16(rsA, base + piece * 1024, voff, (unsigned)(s * SLAB_B + piece * 1024));
        }
    };
    constexpr int NDMA = NWN == 2 ? 8 : 12;       // DMA pieces per wave and K-step
    issue(mb, nb, 0, 0);
    issue(mb, nb, 1, 1);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int t = 0; t < 4; ++t) read_frags(0, 0, xa, xb4, t);
    bool first = true;
    while (vid < nvirt) {
        int mb2 = 0, nb2 = 0;
        const int vid2 = NWN == 2 ? next_valid(vid + (int)gridDim.x, mb2, nb2) : nvirt;
        const bool has_next = vid2 < nvirt;
#ifdef FTE_WINO_STAMP
        unsigned long long st0 = 0, sr0 = 0, st1 = 0;
        if (g_wino_stamp) { st0 = __builtin_amdgcn_s_memtime(); sr0 = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_s_waitcnt(0xC07F); }
#endif
        f32x16 acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        for (int s = 0; s < KS; ++s) {
            const int st = s & 1;
            // ---- half 0 ----
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, 16>([&](auto ic) {
                constexpr int idx = decltype(ic)::value, j = idx >> 2, t = idx & 3;
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[t][j], xb4[t][j], acc[t], 0, 0, 0);
                if constexpr (idx < 4 && !(WMM_ABL & 2)) read_frags(st, 1, ya, yb, idx);
                __builtin_amdgcn_sched_barrier(0);
            });
            // this wave's pieces of step g + 1 have landed.  In a tile's first step behind an epilogue they are OLDER than the epilogue's
            // 16 stores (vector-memory operations retire in issue order): vmcnt(16) leaves the stores draining under the MFMAs
            if constexpr (WMM_ABL & 4) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else if (s == 0 && !first) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // ---- half 1 ----
            // step g + 2: of this tile, or step 0 / 1 of the block's next tile (step 1 of the next tile waits for the epilogue)
            const bool same = s + 2 < KS;
            const bool dma = same || (has_next && s + 2 == KS);
            const int dmb = same ? mb : mb2, dnb = same ? nb : nb2, ds = same ? s + 2 : 0;
            const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.V + (size_t)dmb * KS * SLAB_F), 0, dma ? (unsigned)KS * SLAB_B : 0u, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.U + (size_t)dnb * KS * SLAB_F), 0, dma ? (unsigned)KS * SLAB_B : 0u, 0x00020000);
            if (s == 0 && tid < 64) {          // (every wave has left the previous tile's epilogue: the row table may change)
                const int m = mb * 64 + tid;               // (tiles < 2^31: the output tensor is below 2 GiB)
                int pix = -1, fl = 0;
                if (m < (int)p.g.M) {
                    const int tpi = p.g.th * p.g.tw;
                    const int n = m / tpi, rem = m - n * tpi;
                    const int ty = rem / p.g.tw, tx = rem - ty * p.g.tw;
                    pix = (n * p.g.h + 2 * ty) * p.g.w + 2 * tx;
                    fl = ((2 * ty + 1 < p.g.h) ? 1 : 0) | ((2 * tx + 1 < p.g.w) ? 2 : 0);
                }
                rowpix[tid] = pix; rowflag[tid] = fl;
            }
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, 16>([&](auto ic) {
                constexpr int idx = decltype(ic)::value, j = idx >> 2, t = idx & 3;
                acc[4 + t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ya[t][j], yb[t][j], acc[4 + t], 0, 0, 0);
                // (the fragments of the step after the block's very last one are read from a stage nobody refills: unused)
                if constexpr (idx < 4 && !(WMM_ABL & 2)) read_frags(st ^ 1, 0, xa, xb4, idx);
                if constexpr (idx >= 4 && idx < 4 + NDMA && !(WMM_ABL & 1)) dma_piece(rsA, rsB, ds, st, idx - 4);      // (no step g + 2: descriptors of zero records)
                __builtin_amdgcn_sched_barrier(0);
            });
        }
#ifdef FTE_WINO_STAMP
        if (g_wino_stamp) { st1 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); }
#endif
        // stage 1 (the last K-step's) becomes the epilogue's exchange buffer.  vmcnt(0): the last K-step's DMA slots ran on descriptors of
        // zero records, and an out-of-range LDS-DMA load still WRITES its zeros -- into stage 1, possibly after the exchange values (seen
        // as a rare wrong output of the half-tile kernel, whose last DMA slot is the loop's last instruction)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

        // ---- output transform Y = A^T M A, A^T = [[1,1,1,0],[0,1,-1,-1]], M[i][j] = plane t = 4 i + j.  This wave holds rows i = 2q, 2q+1.
        // P[i][b] = (M A)[i][b];  Y[0][b] = (P0 + P1) + P2,  Y[1][b] = P1 + (-P2 - P3): the q = 0 wave finishes output row 0 and receives
        // P2, the q = 1 wave finishes output row 1 and receives P1 (two floats per position through LDS).
        // (everything the epilogue derives from the lane id is recomputed per tile from an opaque copy: hoisted out of the tile loop it
        // stayed live across the K loop -- 40 spilled registers)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int li_e = lane_e & 31, lh_e = lane_e >> 5;
        float keep[16][2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p00 = acc[0][r] + acc[1][r] + acc[2][r], p01 = acc[1][r] - acc[2][r] - acc[3][r];
            const float p10 = acc[4][r] + acc[5][r] + acc[6][r], p11 = acc[5][r] - acc[6][r] - acc[7][r];
            float s0, s1;
            if (q == 0) { keep[r][0] = p00 + p10; keep[r][1] = p01 + p11; s0 = p10; s1 = p11; }
            else { keep[r][0] = -p00 - p10; keep[r][1] = -p01 - p11; s0 = p00; s1 = p01; }
            xb[(wid * 32 + 2 * r) * 64 + lane_e] = s0;
            xb[(wid * 32 + 2 * r + 1) * 64 + lane_e] = s1;
        }
        __builtin_amdgcn_sched_barrier(0);       // the accumulators are dead from here on: the loads below must not be hoisted above
        // ---- the epilogue's global traffic moves 16 bytes per lane in a TRANSPOSED lane map (lane = (row prw of 8, channel quad pc4): 8
        // rows x 128 B per wave instruction; the accumulator map, lane = channel, gives 4-byte accesses -- 96 of them per lane, store-issue
        // bound: 27 k cycles per tile when measured).  Its inputs are ALL requested here, before anything waits for one.  Raw buffer
        // accesses: an output outside the image (odd sizes, rows beyond M) or an absent tensor (descriptor of zero records) is an
        // out-of-range offset -- loads return 0, stores are dropped -- so the epilogue has no branches.
        // pass (b, h2, j): output column b, tile row 32 mi + 16 h2 + prw + 8 j
        const int prw = lane_e >> 3, pc4 = lane_e & 7;
        const int ch0 = nb * 64 + 32 * (NWN == 2 ? ni : hb) + 4 * pc4;
        unsigned off[2][2][2];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int rr = 32 * mi + 16 * h2 + prw + 8 * j;
                const int pix = rowpix[rr], fl = rowflag[rr];
                const bool rowok = pix >= 0 && (q == 0 || (fl & 1));
                const unsigned o = (unsigned)((pix + q * p.g.w) * p.N + ch0) * 4u;
                off[0][h2][j] = rowok ? o : OOB;
                off[1][h2][j] = (rowok && (fl & 2)) ? o + (unsigned)p.N * 4u : OOB;
            }
        const unsigned tbytes = (unsigned)((size_t)p.g.n * p.g.h * p.g.w * p.N * 4);
        auto rsrc_of = [&](const float* q_) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q_), 0, q_ ? tbytes : 0u, 0x00020000); };
        auto ld4 = [&](const __amdgpu_buffer_rsrc_t& r_, unsigned o) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_, o, 0, 0)); };
        auto st4 = [&](const __amdgpu_buffer_rsrc_t& r_, unsigned o, const f32x4& v) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r_, o, 0, 0); };
        f32x4 in0[2][2][2], in1[2][2][2];
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, al4 = {1.f, 1.f, 1.f, 1.f};
        bool act = false;
        if constexpr (EPI == EPI_FWD) {
            const __amdgpu_buffer_rsrc_t rsR = rsrc_of(p.R);
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                    for (int j = 0; j < 2; ++j) in0[b][h2][j] = ld4(rsR, off[b][h2][j]);
            if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + ch0);
            act = p.alpha != nullptr;
            if (act) al4 = *reinterpret_cast<const f32x4*>(p.alpha + ch0);
        } else {
            const __amdgpu_buffer_rsrc_t rsA = rsrc_of(p.ADD), rsZ = rsrc_of(p.Zin);
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                    for (int j = 0; j < 2; ++j) { in0[b][h2][j] = ld4(rsA, off[b][h2][j]); in1[b][h2][j] = ld4(rsZ, off[b][h2][j]); }
            act = p.Zin != nullptr;
            if (act) al4 = *reinterpret_cast<const f32x4*>(p.alpha + ch0 % p.amod);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int b = 0; b < 2; ++b) keep[r][b] += xb[((wid ^ 1) * 32 + 2 * r + b) * 64 + lane_e];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");           // every wave has its partner's values: stage 1 may be refilled
        if (has_next) issue(mb2, nb2, 1, 1);
        // ---- accumulator map -> transposed map through a wave-private 16 x 36 float patch, four passes ----
        float* patch = reinterpret_cast<float*>(wsm + 2 * STG + 2560) + wid * (16 * 36);
        f32x4 sa4 = {0.f, 0.f, 0.f, 0.f}, sb4 = {0.f, 0.f, 0.f, 0.f};
        const __amdgpu_buffer_rsrc_t rsO0 = rsrc_of(EPI == EPI_FWD ? p.Z : p.RAW), rsO1 = rsrc_of(EPI == EPI_FWD ? p.Y : p.DZ);
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8) patch[((r8 & 3) + 8 * (r8 >> 2) + 4 * lh_e) * 36 + li_e] = keep[8 * h2 + r8][b];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(patch + (prw + 8 * j) * 36 + 4 * pc4);
                    const unsigned o = off[b][h2][j];
                    if constexpr (EPI == EPI_FWD) {
                        v += bias4;
                        st4(rsO0, o, v);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : al4[e] * v[e];          // (no activation: al = 1)
                        v += in0[b][h2][j];
                        st4(rsO1, o, v);
                    } else {
                        v += in0[b][h2][j];
                        st4(rsO0, o, v);
                        const bool in = o != OOB;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float z = act ? in1[b][h2][j][e] : 1.f;        // (no PReLU below: slope 1, no dalpha term)
                            sa4[e] += in ? v[e] * fminf(z, 0.f) : 0.f;
                            v[e] *= prelu_slope(z, al4[e]);
                            sb4[e] += in ? v[e] : 0.f;
                        }
                        st4(rsO1, o, v);
                    }
                }
            }
        if constexpr (EPI == EPI_DGRAD) {
            if (p.PA) {
                // column partials of the block: a wave's lanes = 8 rows x 8 channel quads; the four (q, mi) waves of a channel half in wave order
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a_ = sa4[e], b_ = sb4[e];
                    a_ += __shfl_xor(a_, 8); b_ += __shfl_xor(b_, 8);
                    a_ += __shfl_xor(a_, 16); b_ += __shfl_xor(b_, 16);
                    a_ += __shfl_xor(a_, 32); b_ += __shfl_xor(b_, 32);
                    if (prw == 0) { red[wid * 32 + 4 * pc4 + e] = a_; red[(NW + wid) * 32 + 4 * pc4 + e] = b_; }
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (tid < 32 * NWN) {
                    const int n2 = tid >> 5, l2 = tid & 31;
                    float a = 0.f, b = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { a += red[(4 * n2 + k) * 32 + l2]; b += red[(NW + 4 * n2 + k) * 32 + l2]; }
                    const size_t o = (size_t)mb * p.N + nb * 64 + (NWN == 2 ? tid : 32 * hb + tid);
                    p.PA[o] = a;
                    if (p.PB) p.PB[o] = b;
                }
            }
        }
#ifdef FTE_WINO_STAMP
        if (g_wino_stamp && tid == 0) {
            const unsigned long long t2 = __builtin_amdgcn_s_memtime(), r2 = __builtin_amdgcn_s_memrealtime();
            unsigned long long* o = g_wino_stamp + 8 * (size_t)vid;
            o[0] = st1 - st0; o[1] = t2 - st1; o[2] = t2 - st0; o[3] = r2 - sr0; o[4] = sr0; o[5] = (unsigned long long)KS; o[6] = blockIdx.x; o[7] = r2;
        }
#endif
        // planes 0-3 of the next tile's first step (its stage landed before the epilogue; read again here rather than carried through the
        // epilogue in 32 registers)
        if (has_next) {
#pragma unroll
            for (int t = 0; t < 4; ++t) read_frags(0, 0, xa, xb4, t);
        }
        vid = vid2; mb = mb2; nb = nb2;
        first = false;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// wino_wgrad_kernel: 256 resident blocks.  Block = (share s of the tiles, 64 cin x 64 cout block of the 16 planes); wave w: planes 2w,
// 2w + 1 of the whole block.  K-step = 8 tiles (one octet o of a 64-tile row block).
//   A = V (the forward pass's B^T d B pack of x): the 8 channel steps of the block's cin block, 256 contiguous bytes per (channel step,
//       plane), by LDS-DMA;
//   B = U' = G' e G'^T of the 2x2 dz tiles, G' = [[1,0],[.5,.5],[.5,-.5],[0,1]]: computed HERE from dz (tiles do not overlap: 8 KB of dz
//       per K-step instead of 32 KB of a U' pack that a separate pass would first write to HBM) by waves 0-3, one (tile, channel pair)
//       per lane and K-step: 4 eight-byte loads one step ahead, ~50 vector instructions, 16 ds_write_b64.
// LDS image of both: [8 channel steps][16 planes][8 tiles][8 channels], channel steps 4128 bytes apart (+32: the strided ds_read_b32
// fragments -- a = V[t][tile 4 lh + j][cin li] -- then hit 32 different banks).
#ifndef WG_ABL
#define WG_ABL 0      // diagnostic builds only (scripts/dev/build_wino_stamp.sh with EXTRA=-DWG_ABL=n): 1 = no DMA, 2 = no U' work, 4 = no fragment reads
#endif
constexpr int WG_KSTRIDE = 4096 + 32;
constexpr int WG_OP = 8 * WG_KSTRIDE;          // bytes per operand image
__global__ __launch_bounds__(512, 1) void wino_wgrad_kernel(const float* __restrict__ V, const float* __restrict__ dz, float* __restrict__ slabs,
                                                            int cin, int cout, WinoGeom g, int S, unsigned dz_bytes) {
    extern __shared__ __attribute__((aligned(16))) char wsm[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int KSa = cin >> 3, CB = cin >> 6, NB2 = cout >> 6, P = CB * NB2, MB = g.MB;
    // the blocks of one share sit on one XCD (ids 8 apart): its 64-tile slabs leave HBM once and are shared through that L2
    const int L = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int split = L / P, pair = L - split * P;
    const int cb = pair / NB2, nb = pair - cb * NB2;
    const int mb0 = (int)((long)split * MB / S), mb1 = (int)((long)(split + 1) * MB / S);
    const int nsteps = (mb1 - mb0) * 8;
    const unsigned voff = (unsigned)((lane >> 4) * 2048 + (lane & 15) * 16);
    // ---- the dz side: wave w forms U' of tile w of every K-step's 8, lane = channel of the block's 64.  The tile's pixel coordinates
    // are wave-uniform (scalar registers, advanced by 8 tiles per step: no division, no vector instruction); per lane and step 4 dword
    // loads one step ahead, 12 adds (the 1/2 factors of G' are folded into the finish kernel: exact, powers of two), 16 ds_write_b32.
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dz), 0, dz_bytes, 0x00020000);
    const int q8 = 8 / g.tw, r8 = 8 - q8 * g.tw;
    int pn, pty, ptx;                           // tile of the NEXT load_dz call
    {
        const int m = mb0 * 64 + wid, tpi = g.th * g.tw;
        pn = m / tpi;
        const int rem = m - pn * tpi;
        pty = rem / g.tw; ptx = rem - pty * g.tw;
    }
    int ps = 0;                                 // its K-step
    float e[2][2];
    auto load_dz = [&]() {                      // the 2x2 tile (zeros outside the image / beyond the share), then advance by one K-step
        const bool mv = ps < nsteps && pn < g.n;
        const int base = ((pn * g.h + 2 * pty) * g.w + 2 * ptx) * cout + nb * 64;
#pragma unroll
        for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
            for (int b_ = 0; b_ < 2; ++b_) {
                const bool ok = mv && 2 * pty + a_ < g.h && 2 * ptx + b_ < g.w;
                // (the range check covers the VECTOR offset only: an absent pixel is an out-of-range vector offset, the wave-uniform pixel
                // offset rides in the scalar one)
                const unsigned so = ok ? (unsigned)(base + (a_ * g.w + b_) * cout) * 4u : 0u;
                e[a_][b_] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsD, ok ? (unsigned)lane * 4u : OOB, so, 0));
            }
        ++ps;
        ptx += r8; pty += q8;
        if (ptx >= g.tw) { ptx -= g.tw; ++pty; }
        while (pty >= g.th) { pty -= g.th; ++pn; }
    };
    const unsigned up_base = (unsigned)(WG_OP + (lane >> 3) * WG_KSTRIDE + wid * 32 + (lane & 7) * 4);
    auto store_up_row = [&](int stage, int i) {      // planes 4 i .. 4 i + 3 of the (unscaled) U' of the tile in `e` -> the stage's B image
        const float c0 = i == 0 ? e[0][0] : i == 1 ? e[0][0] + e[1][0] : i == 2 ? e[0][0] - e[1][0] : e[1][0];
        const float c1 = i == 0 ? e[0][1] : i == 1 ? e[0][1] + e[1][1] : i == 2 ? e[0][1] - e[1][1] : e[1][1];
        char* base = wsm + stage * (2 * WG_OP) + up_base + (4 * i) * 256;
        *reinterpret_cast<float*>(base) = c0;
        *reinterpret_cast<float*>(base + 256) = c0 + c1;
        *reinterpret_cast<float*>(base + 512) = c0 - c1;
        *reinterpret_cast<float*>(base + 768) = c1;
    };
    auto store_up = [&](int stage) {
#pragma unroll
        for (int i = 0; i < 4; ++i) store_up_row(stage, i);
    };
    // wave w = planes 2w, 2w + 1 of the WHOLE 64 x 64 block (2 x 2 MFMA blocks per plane): every fragment feeds two MFMAs -- half the
    // ds_read_b32 traffic of a (8 planes, 32 x 32) wave tile, which ran the loop LDS-bound (6100 cycles per K-step for 4096 of MFMA)
    f32x16 acc[2][2][2];                       // [plane][ci block][co block]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.f;
    // fragment base addresses (bytes).  A: the pack holds the 16-byte halves of a row swapped for rows 16-31 / 48-63 of a row block
    // (two phases of K-steps); B: written here, plain
    const unsigned lbase = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char*)wsm;
    // (four plain registers, selected by wave-uniform conditions: as an indexed array the bases went to scratch memory and every K-step
    // began with a scratch load and a vmcnt(0))
    const unsigned a_lo = lbase + (unsigned)((li >> 3) * WG_KSTRIDE + (2 * wid) * 256 + lh * 128 + ((li & 7) << 2));
    const unsigned a_hi = lbase + (unsigned)((li >> 3) * WG_KSTRIDE + (2 * wid) * 256 + lh * 128 + (((li & 7) ^ 4) << 2));
    const unsigned b_b = lbase + (unsigned)(WG_OP + (li >> 3) * WG_KSTRIDE + (2 * wid) * 256 + lh * 128 + ((li & 7) << 2));
    // ---- software pipeline (as wino_mm_kernel): a K-step = 32 MFMAs per wave in two halves (k pairs j = 0, 1 | j = 2, 3) ----
    //   half 0: MFMAs on fragment set X (j = 0, 1 of step g) | payload: reads of set Y (j = 2, 3 of step g)
    //   wait for Y, for this wave's pieces of step g + 1 and its U' writes, ONE barrier (step g's stage is free, step g + 1 complete)
    //   half 1: MFMAs on Y | payload: reads of X of step g + 1, the 4 DMA pieces of step g + 2 into the freed stage, U' of this
    //           wave's tile of step g + 2 (its dz loads were requested a step ago), then the loads of step g + 3
    float xa[2][2][2], xb[2][2][2], ya[2][2][2], yb[2][2][2];      // [plane][block][j in the half]
    auto read_frag = [&](unsigned abase, unsigned bbase, int half, float (&fa)[2][2][2], float (&fb)[2][2][2], int idx) {      // idx 0..7: (t, i, jj)
        const int t = idx >> 2, i = (idx >> 1) & 1, jj = idx & 1;
        const unsigned so = (unsigned)(i * 4 * WG_KSTRIDE + t * 256 + (2 * half + jj) * 32);       // (an immediate of the read)
        fa[t][i][jj] = *(const __attribute__((address_space(3))) float*)(unsigned long long)(abase + so);
        fb[t][i][jj] = *(const __attribute__((address_space(3))) float*)(unsigned long long)(bbase + so);
    };
    auto a_base = [&](int s_) { return ((((s_ & 7) >> 1) & 1) ? a_hi : a_lo) + (unsigned)((s_ & 1) * (2 * WG_OP)); };
    auto b_base = [&](int s_) { return b_b + (unsigned)((s_ & 1) * (2 * WG_OP)); };
    auto dma_piece = [&](int s_, int i) {         // piece i (0..3) of this wave for step s_ (a descriptor of zero records beyond the share)
        const int mb = mb0 + (s_ >> 3), o = s_ & 7;
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(V + (size_t)(s_ < nsteps ? mb : mb0) * KSa * SLAB_F), 0,
                                                                             s_ < nsteps ? (unsigned)KSa * SLAB_B : 0u, 0x00020000);
        const int piece = wid + 8 * i, ks = piece >> 2, t4 = piece & 3;
        dma16(rsA, wsm + (s_ & 1) * (2 * WG_OP) + ks * WG_KSTRIDE + t4 * 1024, voff, (unsigned)(((8 * cb + ks) * 16 + 4 * t4) * 2048 + o * 256));
    };
    if (nsteps > 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { dma_piece(0, i); dma_piece(1, i); }
        load_dz(); store_up(0);
        load_dz(); store_up(1);
        load_dz();                     // step 2
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (the 4 loads of step 2 stay in flight)
        const unsigned pa = a_base(0), pb = b_base(0);
#pragma unroll
        for (int idx = 0; idx < 8; ++idx) read_frag(pa, pb, 0, xa, xb, idx);
    }
    for (int s = 0; s < nsteps; ++s) {
        const unsigned ca = a_base(s), cbb = b_base(s), na = a_base(s + 1), nbb = b_base(s + 1);
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 16>([&](auto ic) {
            constexpr int idx = decltype(ic)::value, jj = idx >> 3, t = (idx >> 2) & 1, i = (idx >> 1) & 1, k = idx & 1;
            acc[t][i][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[t][i][jj], xb[t][k][jj], acc[t][i][k], 0, 0, 0);
            if constexpr (idx < 8 && !(WG_ABL & 4)) read_frag(ca, cbb, 1, ya, yb, idx);
            __builtin_amdgcn_sched_barrier(0);
        });
        // this wave's pieces of step s + 1 landed (the 4 dz loads of step s + 2 behind them stay in flight), its U' writes are done
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 16>([&](auto ic) {
            constexpr int idx = decltype(ic)::value, jj = idx >> 3, t = (idx >> 2) & 1, i = (idx >> 1) & 1, k = idx & 1;
            acc[t][i][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(ya[t][i][jj], yb[t][k][jj], acc[t][i][k], 0, 0, 0);
            if constexpr (idx < 8 && !(WG_ABL & 4)) read_frag(na, nbb, 0, xa, xb, idx);
            if constexpr (idx >= 8 && idx < 12 && !(WG_ABL & 1)) dma_piece(s + 2, idx - 8);
            if constexpr (idx >= 12 && !(WG_ABL & 2)) {                                    // U' of step s + 2 (a row of planes per slot), dz of step s + 3
                store_up_row(s & 1, idx - 12);
                if constexpr (idx == 15) load_dz();
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    // (the last two steps' DMA slots ran on descriptors of zero records: such a load still writes zeros to LDS -- let them land before
    // the block can end and its LDS go to another block)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // raw partial planes -> slabs[split][t][cin][cout]
    const size_t plane = (size_t)cin * cout;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float* o = slabs + ((size_t)split * 16 + 2 * wid + t) * plane + (size_t)(cb * 64 + 32 * i + 4 * lh) * cout + nb * 64 + 32 * k + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[(size_t)((r & 3) + 8 * (r >> 2)) * cout] = acc[t][i][k][r];
            }
}

// dw[kh][kw][ci][co] = A'^T (sum over shares, in share order) A',  A'^T = [[1,1,1,0],[0,1,-1,0],[0,1,1,-1]]
__global__ __launch_bounds__(256) void wino_wgrad_finish_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int S, long plane) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= plane) return;
    float m[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) m[t] = 0.f;
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int t = 0; t < 16; ++t) m[t] += slabs[((size_t)s * 16 + t) * plane + idx];
    // the 1/2 factors of G' = diag(1, 1/2, 1/2, 1) [[1,0],[1,1],[1,-1],[0,1]], left out of the kernel's U' (exact: powers of two)
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const float sc = (((t >> 2) == 1 || (t >> 2) == 2) ? 0.5f : 1.f) * (((t & 3) == 1 || (t & 3) == 2) ? 0.5f : 1.f);
        m[t] *= sc;
    }
    float pr[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        pr[0][j] = m[0 + j] + m[4 + j] + m[8 + j];
        pr[1][j] = m[4 + j] - m[8 + j];
        pr[2][j] = m[4 + j] + m[8 + j] - m[12 + j];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        dw[(size_t)(a * 3 + 0) * plane + idx] = pr[a][0] + pr[a][1] + pr[a][2];
        dw[(size_t)(a * 3 + 1) * plane + idx] = pr[a][1] - pr[a][2];
        dw[(size_t)(a * 3 + 2) * plane + idx] = pr[a][1] + pr[a][2] - pr[a][3];
    }
}

int g_algo = -1;

}  // namespace

WinoGeom wino_geom(int n, int h, int w) {
    WinoGeom g;
    g.n = n; g.h = h; g.w = w; g.th = (h + 1) / 2; g.tw = (w + 1) / 2;
    g.M = (long)n * g.th * g.tw;
    g.MB = (int)((g.M + 63) / 64);
    return g;
}

hipError_t wino_transform_tiles(const float* x, float* pack, int n, int h, int w, int c, int mode, hipStream_t st, bool small) {
    if (c % 32) return hipErrorInvalidValue;
    const WinoGeom g = wino_geom(n, h, w);
    const size_t xb = (size_t)n * h * w * c * 4;
    if (xb >= ((size_t)1 << 31)) return hipErrorInvalidValue;
    const dim3 grid(8 * ((g.MB + 7) / 8) * (small ? 2 : 1), c / 32);
    if (mode == 0 && small) hipLaunchKernelGGL((wino_tiles_kernel<0, 256>), grid, dim3(256), 0, st, x, pack, h, w, c, g.th, g.tw, g.M, g.MB, (unsigned)xb);
    else if (mode == 0) hipLaunchKernelGGL((wino_tiles_kernel<0, 512>), grid, dim3(512), 0, st, x, pack, h, w, c, g.th, g.tw, g.M, g.MB, (unsigned)xb);
    else hipLaunchKernelGGL((wino_tiles_kernel<1, 512>), grid, dim3(512), 0, st, x, pack, h, w, c, g.th, g.tw, g.M, g.MB, (unsigned)xb);
    return hipGetLastError();
}

hipError_t wino_transform_filter(const float* w, float* pack, int cin, int cout, int dgrad, hipStream_t st) {
    if (cin % 64 || cout % 64) return hipErrorInvalidValue;
    const int total = cin * cout;
    hipLaunchKernelGGL(wino_filter_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, pack, cin, cout, dgrad);
    return hipGetLastError();
}

namespace {
template <int EPI, int NWN>
hipError_t wino_mm_launch(const WinoMMParams& p, int NBX, int GRP, int vid0, int vid1, int grid, int tiles, hipStream_t st) {
    constexpr int NW = 4 * NWN;
    const size_t lds = 2 * ((size_t)SLAB_B + NWN * (SLAB_B / 2)) + 2560 + (size_t)NW * 16 * 36 * 4;      // two stages, row table + column partials, transpose patches
    static std::atomic<unsigned> attr{0u};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned bit = 1u << (dev & 31);
    if (!(attr.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wino_mm_kernel<EPI, NWN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr.fetch_or(bit, std::memory_order_release);
    }
    const int sig[5] = {AL_MK, BL_NK, EPI, 8, 1};                    // tile id 8: the Winograd product (bench.py TILES)
    const double frac = (double)tiles / ((double)p.g.MB * (p.N / 64));
    const double flops = 2.0 * 16.0 * (double)tiles * 64.0 * 64.0 * p.K;      // MFMA FLOPs the launch EXECUTES (padded row blocks included)
    const double bytes = frac * ((double)wino_pack_floats(p.g.M, p.K) * 4.0 + 16.0 * p.N * p.K * 4.0 + (double)p.g.n * p.g.h * p.g.w * p.N * 4.0 *
                                 (EPI == EPI_FWD ? (1 + (p.Z ? 1 : 0) + (p.R ? 1 : 0)) : (1 + (p.ADD ? 1 : 0) + (p.RAW ? 1 : 0) + (p.Zin ? 1 : 0))));
    const int h = igemm_prof_begin(sig, (int)((long)tiles * 64 / (p.N / 64)), p.N, 16 * p.K, flops, bytes, st);      // rows = tiles of the launch
    hipLaunchKernelGGL((wino_mm_kernel<EPI, NWN>), dim3(grid), dim3(256 * NWN), lds, st, p, NBX, GRP, vid0, vid1);
    igemm_prof_end(h, EPI == EPI_FWD ? (NWN == 2 ? "wino_mm_kernel<0,2>" : "wino_mm_kernel<0,1>") : (NWN == 2 ? "wino_mm_kernel<1,2>" : "wino_mm_kernel<1,1>"), st);
    return hipGetLastError();
}
}  // namespace

hipError_t wino_mm(const WinoMMParams& p, int epi, hipStream_t st) {
    if (p.K % 64 || p.N % 64 || p.g.MB <= 0) return hipErrorInvalidValue;
    if ((size_t)p.g.n * p.g.h * p.g.w * p.N >= ((size_t)1 << 31)) return hipErrorInvalidValue;
    const int NB = p.N / 64;
    int NBX = NB < 2 ? NB : 2, GRP = NB / NBX;
    static const int plain = getenv("FTE_WINO_PLAIN_ORDER") ? atoi(getenv("FTE_WINO_PLAIN_ORDER")) : 0;      // A/B hook: no XCD-aware block order
    int nvirt;
    if (plain || NB % NBX || 8 % GRP) { GRP = 0; nvirt = p.g.MB * NB; }
    else { const int C = 8 / GRP; nvirt = 8 * ((p.g.MB + C - 1) / C) * NBX; }
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorInvalidDevice;
        cus = prop.multiProcessorCount > 8 ? prop.multiProcessorCount / 8 * 8 : 8;
    }
    const int grid = nvirt < cus ? (nvirt + 7) / 8 * 8 : cus;      // one resident block per CU (a multiple of 8: a block keeps its XCD)
    // A launch of at most half a round of whole tiles runs as HALF tiles (wino_mm_kernel<EPI, 1>): 7x7x512 at 64 images: 256 half tiles
    // instead of 128 tiles.  FTE_WINO_HALF_TILES=2 (A/B hook) also sends the last, partly filled round of a many-round launch there as a
    // launch of its own (measured: no gain, see the kernel).  (the same id -> tile map as the kernel's decode)
    static const int halves = getenv("FTE_WINO_HALF_TILES") ? atoi(getenv("FTE_WINO_HALF_TILES")) : 1;      // A/B hook
    auto valid = [&](int vid) {
        if (GRP > 0) { const int xcd = vid & 7, slot = vid >> 3, cls = xcd / GRP, C = 8 / GRP; return (slot / NBX) * C + cls < p.g.MB; }
        return vid / NB < p.g.MB;
    };
    const int rounds = (nvirt + grid - 1) / grid, last = grid * (rounds - 1);
    int L = 0;
    for (int v = last; v < nvirt; ++v) L += valid(v) ? 1 : 0;
    const int total = p.g.MB * NB;
    int tail_base = nvirt;
    if (halves && L > 0 && 2 * L <= cus && (rounds == 1 || halves == 2)) tail_base = last;
    hipError_t e = hipSuccess;
    if (tail_base > 0) {
        e = epi == EPI_FWD ? wino_mm_launch<EPI_FWD, 2>(p, NBX, GRP, 0, tail_base, grid, total - (tail_base < nvirt ? L : 0), st)
                           : wino_mm_launch<EPI_DGRAD, 2>(p, NBX, GRP, 0, tail_base, grid, total - (tail_base < nvirt ? L : 0), st);
        if (e != hipSuccess) return e;
    }
    if (tail_base < nvirt)
        e = epi == EPI_FWD ? wino_mm_launch<EPI_FWD, 1>(p, NBX, GRP, tail_base, nvirt, 2 * L, L, st)
                           : wino_mm_launch<EPI_DGRAD, 1>(p, NBX, GRP, tail_base, nvirt, 2 * L, L, st);
    return e;
}

int wino_wgrad_splits(int cin, int cout) {
    if (cin % 64 || cout % 64) return 0;
    const int P = (cin / 64) * (cout / 64);
    if (P > 256 || 256 % P) return 0;
    return 256 / P;
}

hipError_t wino_wgrad(const float* V, const float* dz, float* slabs, float* dw, const WinoGeom& g, int cin, int cout, hipStream_t st) {
    const int S = wino_wgrad_splits(cin, cout);
    if (!S) return hipErrorInvalidValue;
    const size_t db = (size_t)g.n * g.h * g.w * cout * 4;
    if (db >= ((size_t)1 << 31)) return hipErrorInvalidValue;      // buffer range
    const size_t lds = 4 * (size_t)WG_OP;
    static std::atomic<unsigned> attr{0u};               // per device (bit), as wino_mm_launch's
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned bit = 1u << (dev & 31);
    if (!(attr.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wino_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr.fetch_or(bit, std::memory_order_release);
    }
    const int sig[5] = {AL_KM, BL_KN, EPI_FWD, 8, S};
    const double flops = 2.0 * 16.0 * (double)g.MB * 64.0 * cin * cout;
    const double bytes = (double)wino_pack_floats(g.M, cin) * 4.0 + (double)db + (double)S * 16.0 * cin * cout * 4.0;
    const int h = igemm_prof_begin(sig, 16 * cin, cout, (int)(g.MB * 64), flops, bytes, st);
    hipLaunchKernelGGL(wino_wgrad_kernel, dim3(256), dim3(512), lds, st, V, dz, slabs, cin, cout, g, S, (unsigned)db);
    igemm_prof_end(h, "wino_wgrad_kernel", st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const long plane = (long)cin * cout;
    hipLaunchKernelGGL(wino_wgrad_finish_kernel, dim3((unsigned)((plane + 255) / 256)), dim3(256), 0, st, slabs, dw, S, plane);
    return hipGetLastError();
}

void wino_set_algo(int a) { g_algo = a; }
int wino_get_algo() {
    if (g_algo < 0) {
        const char* e = getenv("FTE_CONV_ALGO");
        g_algo = 2;
        if (e) {
            if (!strcmp(e, "direct")) g_algo = 0;
            else if (!strcmp(e, "winograd")) g_algo = 1;
            else if (!strcmp(e, "auto")) g_algo = 2;
        }
    }
    return g_algo;
}
