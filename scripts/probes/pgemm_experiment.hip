// EXPERIMENT, NOT PART OF libfte.so (round 2).  Result on MI355X, batch 512, fp32, against igemm.hip's 64x64 kernel:
//   14x14x128->128: 39-45 us vs 42;  7x7x256->256: 42-43 vs 39;  4x4x512->512: 42 vs 42;  ShuffleNet-v2 / ResNet-50 / ResNeXt-50 steps: equal.
// With every global load and store switched off (FTE_PGEMM_ABL=3) the same launches still take 34-38 us against 21-27 us of MFMA
// time: the small-K 1x1 convolutions are NOT bound by per-tile prologues or by HBM but by the K-step itself (one barrier + LDS
// write/read turn-around per sixteen dependent 64-cycle MFMAs at 4 waves per SIMD).  Two accumulator chains, a start stagger per
// SIMD wave slot and forced 3/4/5 blocks per CU changed nothing.  Kept for the record; to build it, add it to build.sh's list.
//
// pgemm.hip -- persistent fp32 MFMA GEMM for the 1x1 stride-1 convolutions (and every other plain row-major product):
//     C[M][N] = A[M][K] * B (+ ADD),   B given as [K][N] (forward: HWIO weights) or [N][K] (dgrad: the same weights, read transposed).
//
// Why a second kernel.  The 1x1 convs of ShuffleNet-v2 / ResNet have K = 64..512: a 64x64 tile is 2..16 K-steps of 32.  In
// igemm.hip every tile is its own block -- prologue (addresses, first loads: an HBM round trip), a handful of K-steps, epilogue --
// and all ~2000 resident blocks of a launch start together, so the chip alternates between everyone waiting for HBM and everyone
// multiplying: 14x14x128->128 at batch 512 ran 42 us against 21 us of MFMA time and 17 us of HBM time.  Here a block is
// PERSISTENT: it walks its share of the tiles as one flattened (tile, K-step) sequence, the operand loads run two steps ahead of
// the MFMAs ACROSS tile boundaries (registers for step s+2, LDS for step s+1), and a finished tile is stored straight from the
// accumulators (32 lanes x 4 B = one 128-B line per row and instruction, no LDS transpose, no barrier) while the next tile's
// operands are already landing.  One barrier per K-step, no vector ALU addressing inside a tile.
//
// Tiles: 64 x 64, four waves of 32 x 32 (one v_mfma_f32_32x32x2_f32 accumulator block each).  Small tiles keep the per-CU
// tile counts even (a persistent grid cannot rebalance): 3136 tiles on 1024 blocks = 12.25 per CU.
// Block ids are dealt to the XCDs round-robin by the hardware; the remap below gives each XCD a contiguous range of
// tiles (n fastest), so the N/64 tiles that share A rows run on one XCD at the same time and A leaves HBM once.
#include "../../tf_face_toolbox_amd/csrc/igemm_dev.h"
#include "pgemm_experiment.h"

namespace {

using namespace igemm_dev;

#ifndef FTE_PGEMM_CHAINS
#define FTE_PGEMM_CHAINS 1
#endif
template <int BL, bool HAS_ADD>
__global__ __launch_bounds__(256, 4) void pgemm_kernel(const PgemmParams p) {
    constexpr int BM = 64, BN = 64;
    constexpr int STAGE = (BM + BN) * BK;
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
    constexpr unsigned OOB = 0x80000000u;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int li = lane & 31, lh = lane >> 5;

    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, p.b_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, p.c_bytes, 0x00020000);

    // logical block id: XCD x (= blockIdx & 7) owns ids [x*G/8, (x+1)*G/8)
    const int G = gridDim.x;
    const int lb = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int ntn = p.N / BN;
    const int ntiles = p.tiles;
    const int ksteps = p.K / BK;
    const int mine = lb < ntiles ? (ntiles - lb + G - 1) / G : 0;
    const int S = mine * ksteps;                     // flattened steps of this block
    if (S == 0) return;

    // ---- load cursor: (tile, k) of the next step to fetch -----------------------------------------------------------
    int l_tile = lb, l_k = 0;
    unsigned a_off[2], b_off[2];                     // per-thread byte offsets of the cursor's tile (k = 0)
    // (a tile index past the end gives out-of-range offsets: the fetches of the last two steps run on unconditionally and
    // return zeros nobody reads -- a conditional fetch makes hipcc wait for vmcnt(0) before every LDS write, which cancels
    // the two-step lead of the loads)
    auto tile_offsets = [&](int tile) {
        const int mt = tile / ntn, nt = tile - mt * ntn;
        const int m0 = mt * BM, n0 = nt * BN;
        const bool live = tile < ntiles && !(p.abl & 2);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + (tid >> 3) + 32 * i;
            a_off[i] = (live && m < p.M) ? (unsigned)(m * p.lda + ((tid & 7) << 2)) * 4u : OOB;
            if constexpr (BL == BL_KN) b_off[i] = live ? (unsigned)(((tid >> 4) + 16 * i) * p.ldb + n0 + ((tid & 15) << 2)) * 4u : OOB;
            else b_off[i] = live ? (unsigned)((n0 + (tid >> 3) + 32 * i) * p.ldb + ((tid & 7) << 2)) * 4u : OOB;
        }
    };
    tile_offsets(l_tile);
    auto ldg = [&](const __amdgpu_buffer_rsrc_t& r, unsigned voff, unsigned soff) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    };
    auto fetch = [&](f32x4 (&ra)[2], f32x4 (&rb)[2]) {          // issue the loads of the cursor's step, then advance it
        const unsigned ka = (unsigned)l_k * 4u;                                             // wave-uniform: scalar offsets
        const unsigned kb = BL == BL_KN ? (unsigned)(l_k * p.ldb) * 4u : (unsigned)l_k * 4u;
#pragma unroll
        for (int i = 0; i < 2; ++i) { ra[i] = ldg(rsrcA, a_off[i], ka); rb[i] = ldg(rsrcB, b_off[i], kb); }
        l_k += BK;
        if (l_k == p.K) {
            l_k = 0;
            l_tile += G;
            tile_offsets(l_tile);
        }
    };
    auto stash = [&](int stage, const f32x4 (&ra)[2], const f32x4 (&rb)[2]) {
        float* As = smem + stage * STAGE;
        float* Bs = As + BM * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = (tid >> 3) + 32 * i;
            *reinterpret_cast<f32x4*>(As + r * BK + swz(r, tid & 7)) = ra[i];
            if constexpr (BL == BL_KN) *reinterpret_cast<f32x4*>(Bs + ((tid >> 4) + 16 * i) * BN + ((tid & 15) << 2)) = rb[i];
            else *reinterpret_cast<f32x4*>(Bs + r * BK + swz(r, tid & 7)) = rb[i];
        }
    };

    f32x16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }

    const int row = wm * 32 + li, col = wn * 32 + li;
    auto frags = [&](const float* As, const float* Bs, int u, f32x4& fa, f32x4& fb) {
        fa = *reinterpret_cast<const f32x4*>(As + row * BK + swz(row, 2 * u + lh));
        if constexpr (BL == BL_KN) {
#pragma unroll
            for (int t = 0; t < 4; ++t) fb[t] = Bs[(8 * u + 4 * lh + t) * BN + col];
        } else {
            fb = *reinterpret_cast<const f32x4*>(Bs + col * BK + swz(col, 2 * u + lh));
        }
    };
    auto compute = [&](int stage) {
        const float* As = smem + stage * STAGE;
        const float* Bs = As + BM * BK;
        f32x4 fa[2], fb[2];
        frags(As, Bs, 0, fa[0], fb[0]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u + 1 < 4) frags(As, Bs, u + 1, fa[(u + 1) & 1], fb[(u + 1) & 1]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (FTE_PGEMM_CHAINS == 2 && (t & 1)) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u & 1][t], fb[u & 1][t], acc2, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u & 1][t], fb[u & 1][t], acc, 0, 0, 0);
            }
        }
    };
    // ---- compute cursor ------------------------------------------------------------------------------------------
    int c_tile = lb, c_k = 0;
    auto finish = [&]() {                            // called after every compute: store the tile when its last K-step is in
        c_k += BK;
        if (c_k != p.K) return;
        c_k = 0;
        const int mt = c_tile / ntn, nt = c_tile - mt * ntn;
        c_tile += G;
        const int n = nt * BN + col;
        const int mbase = mt * BM + wm * 32 + 4 * lh;
        float add[16];
        if constexpr (HAS_ADD) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mbase + (r & 3) + 8 * (r >> 2);
                add[r] = m < p.M ? p.ADD[(long)m * p.ldc + n] : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {            // branch-free: rows past M get an out-of-range offset (the store is dropped)
            const int m = mbase + (r & 3) + 8 * (r >> 2);
            const unsigned o = (m < p.M && !(p.abl & 1)) ? (unsigned)(m * p.ldc + n) * 4u : OOB;
            float v = acc[r];
            if (FTE_PGEMM_CHAINS == 2) { v += acc2[r]; acc2[r] = 0.f; }
            if constexpr (HAS_ADD) v += add[r];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrcC, o, 0, 0);
            acc[r] = 0.f;
        }
    };

    // ---- software pipeline: registers hold step s+2 (set s&1) and s+1, LDS[(s+1)&1] gets step s+1 during step s ---------
    f32x4 ra0[2], rb0[2], ra1[2], rb1[2];
    fetch(ra0, rb0);                                 // step 0
    fetch(ra1, rb1);                                 // step 1
    // De-phase the co-resident blocks.  All blocks of a persistent grid start together and advance at the same rate, so the
    // (up to) four waves that share a SIMD -- one per block -- reach their barrier, LDS writes and fragment reads TOGETHER
    // and the matrix pipe idles while they do (pure MFMA + LDS + barrier, no memory traffic: 54 % of the MFMA rate).
    // The wave slot on its SIMD (HW_ID.wave_id) differs between exactly those waves: block-wide, wave 0's slot sets a start
    // delay of slot x p.stagger x 64 cycles.
    if (p.stagger > 0) {
        __shared__ int s_slot;
        if (tid == 0) s_slot = (int)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 7;      // HW_REG_HW_ID[3:0] = wave_id
        __syncthreads();
        const int n = __builtin_amdgcn_readfirstlane(s_slot) * p.stagger;
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
    }
    stash(0, ra0, rb0);
    __syncthreads();
    int s = 0;
    while (true) {
        // even step: set 0 is free (step s is in LDS[0]); it takes step s+2.  Set 1 (step s+1) goes to LDS[1] after the MFMAs,
        // BEFORE the tile's stores are issued: the wait in front of the LDS write then counts only the four loads behind it.
        fetch(ra0, rb0);
        compute(0);
        stash(1, ra1, rb1);
        finish();
        if (++s == S) break;
        __syncthreads();
        // odd step
        fetch(ra1, rb1);
        compute(1);
        stash(0, ra0, rb0);
        finish();
        if (++s == S) break;
        __syncthreads();
    }
}

}  // namespace

bool pgemm_handles(const PgemmParams& p) {
    static const bool off = getenv("FTE_NO_PGEMM") != nullptr;         // A/B hook
    static const int maxk = getenv("FTE_PGEMM_MAXK") ? atoi(getenv("FTE_PGEMM_MAXK")) : 1024;
    return !off && p.K % 32 == 0 && p.K >= 32 && p.K <= maxk && p.N % 64 == 0 && p.lda % 4 == 0 && p.ldb % 4 == 0 && p.M > 0 &&
           (size_t)p.M * p.lda * 4 < ((size_t)1 << 31) && (size_t)p.M * p.ldc * 4 < ((size_t)1 << 31);
}

hipError_t pgemm_launch(PgemmParams p, int bl, hipStream_t st) {
    const int mt = (p.M + 63) / 64, nt = p.N / 64;
    p.tiles = mt * nt;
    p.c_bytes = (unsigned)((size_t)p.M * p.ldc * 4);
    static const int abl = getenv("FTE_PGEMM_ABL") ? atoi(getenv("FTE_PGEMM_ABL")) : 0;      // ablation hook: 1 no stores, 2 no loads
    p.abl = abl;
    static const int stagger = getenv("FTE_PGEMM_STAGGER") ? atoi(getenv("FTE_PGEMM_STAGGER")) : 0;
    p.stagger = stagger;
    // A persistent grid cannot rebalance: with G blocks the launch lasts ceil(tiles / G) tiles.  Take the round count the
    // full chip (256 CUs x bpc blocks) needs and then the SMALLEST grid that still finishes in that many rounds -- every
    // block gets the same number of tiles (7x7x256->256 at batch 512: 1568 tiles = 2 rounds on 1024 blocks, of which the
    // second is half empty; 784 blocks do 2 tiles each with fewer waves sharing each SIMD).  Multiple of 8: XCD remap.
    static const int bpc = getenv("FTE_PGEMM_BPC") ? atoi(getenv("FTE_PGEMM_BPC")) : 4;
    const int cap = 256 * bpc;
    const int rounds = (p.tiles + cap - 1) / cap;
    int grid = (p.tiles + rounds - 1) / rounds;
    grid = (grid + 7) / 8 * 8;
    static const int ldspad = getenv("FTE_PGEMM_LDSPAD") ? atoi(getenv("FTE_PGEMM_LDSPAD")) : 0;
#define FTE_PG(BL_, ADD_) hipLaunchKernelGGL((pgemm_kernel<BL_, ADD_>), dim3(grid), dim3(256), ldspad, st, p)
    if (bl == BL_KN) { if (p.ADD) FTE_PG(BL_KN, true); else FTE_PG(BL_KN, false); }
    else { if (p.ADD) FTE_PG(BL_NK, true); else FTE_PG(BL_NK, false); }
#undef FTE_PG
    return hipGetLastError();
}
