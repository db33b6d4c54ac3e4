// api.hip -- extern "C" entry points of libfte.so (declared in include/fte.h).
// Host-side only: shape checks, operand-gather descriptors for the igemm kernel
// family, tile / split-K selection, ordered reductions.  Never allocates, never
// synchronises; every launch goes to the caller's stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <initializer_list>

#include "../../include/fte.h"
#include "igemm.h"
#include "wgrad16.h"
#include "pw16.h"
#include "kernels.h"
#include "layers.h"
#include "wino.h"

namespace {

struct Pads { int out, before; };
inline Pads same_pads(int in, int k, int stride) {
    const int out = (in + stride - 1) / stride;
    int total = (out - 1) * stride + k - in;
    if (total < 0) total = 0;
    return {out, total / 2};
}

inline int rc(hipError_t e) { return e == hipSuccess ? FTE_OK : (int)e; }

inline long tiles_of(int tile, long M, long N) {
    int bm, bn;
    igemm_tile_dims(tile, &bm, &bn);
    return ((M + bm - 1) / bm) * (N / bn);
}

// the bf16 tile preferences apply in the bf16-operand mode and inside the bf16-source (fte_*16) entry points
thread_local bool g_plan16 = false;      // (thread_local: a query on one host thread must not re-plan a launch on another)
struct Plan16 { bool prev; Plan16() : prev(g_plan16) { g_plan16 = true; } ~Plan16() { g_plan16 = prev; } };
// workspace queries plan for BOTH operand modes (both_modes): a thread-local override, never a flip of the process-global mode that a
// launch on another host thread is reading (-1: follow fte_set_mfma_dtype)
thread_local int g_plan_mode = -1;
struct PlanMode { int prev; explicit PlanMode(bool bf16) : prev(g_plan_mode) { g_plan_mode = bf16 ? 1 : 0; } ~PlanMode() { g_plan_mode = prev; } };
inline bool plan_bf16() { return g_plan16 || (g_plan_mode >= 0 ? g_plan_mode == 1 : igemm_get_bf16()); }
// Largest tile that still gives the chip >= 1.5 blocks per CU; else the smallest legal one.
inline int pick_tile(long M, long N) {
    const bool b16 = plan_bf16();
    static const long want32 = getenv("FTE_PICK_WANT") ? atol(getenv("FTE_PICK_WANT")) : 384;      // (measured again in round 2 on the BN nets: 384 / 800 / 1200 / always-smallest are within 0.5 %)
    // bf16 operands: the 128-row tiles run on the LDS-DMA kernels (igemm16.hip), the 64 x 64 tile on the register-staged one, which is
    // the slower kernel by more than the half-empty chip costs the larger tiles (FTE_PICK_WANT16 sweep in DESIGN.md)
    static const long want16 = getenv("FTE_PICK_WANT16") ? atol(getenv("FTE_PICK_WANT16")) : 120;
    const long want = b16 ? want16 : want32;
    int cands[3];
    int nc = 0;
    if (N % 128 == 0) { cands[nc++] = TILE_128x128; cands[nc++] = TILE_128x64; cands[nc++] = TILE_64x64; }
    else { cands[nc++] = TILE_256x64; cands[nc++] = TILE_128x64; cands[nc++] = TILE_64x64; }
    for (int i = 0; i < nc; ++i)
        if (tiles_of(cands[i], M, N) >= want) return cands[i];
    return cands[nc - 1];
}

// Resident blocks the planner assumes: 256 CUs x 6 blocks of the 64x64 tile (16 KiB of LDS, <= 80 VGPRs).
// fp32 MFMAs are 64 cycles each, so operand reuse is not what limits these kernels -- latency hiding is: on
// MI355X at batch 512 the conv time per step is 60.7 ms with 128x128 tiles at 2 blocks/CU, 56.2 ms at 3 blocks/CU
// (single LDS stage) and 53.4 ms with 64x64 tiles at 6 blocks/CU.
constexpr long SLOTS_BIG = 768;       // 128x128 / 128x64 tiles (wgrad, dense split-K): 3 blocks per CU
constexpr long SLOTS = 1536;

// Row plan of an M x N output with reduction length K.  T big tiles on SLOTS resident blocks run in
// ceil(T/SLOTS) rounds, so a launch of 1568 tiles pays for 4 rounds while doing 3.06 rounds of work.  The
// plan keeps whole rounds in a big-tile MAIN launch (epilogue fused in the kernel) and runs the leftover
// tiles -- or ALL tiles when there is less than one round of them (small per-GPU shards) -- as a TAIL:
//   mode 2 (needs workspace): the same big tiles with split-K, P = SLOTS / tiles splits, raw partial
//          tiles to the workspace, then igemm_fixup sums them and applies the epilogue;
//   mode 1 (no workspace / P < 2): 64x64 tiles (4x the blocks, 1/4 of the work each).
//   mode 3 (needs workspace): STREAM-K -- one launch of resident workers that share the (tile, K-step) space equally (igemm.hip
//          "stream-K"); main_tile / main_mtiles describe its tile grid (the partial rows of dalpha / dbias are per tile row as ever).
struct RowPlan {
    int main_tile; long main_rows, main_mtiles;
    int tail_mode, tail_tile; long tail_mtiles; int tail_splits, tail_kchunk; size_t pw_bytes;
    int sk_workers;
};
#define NUM_CU ((long)igemm_num_cus())      // (was a constant 256: the planner now follows the device)
// ... and a storage-only launch (bf16 tensors in the epilogue, no fp32 outputs): the persistent kernels of igemm16.hip take it
thread_local bool g_plan_s16 = false;
struct PlanS16 { bool prev; explicit PlanS16(bool on) : prev(g_plan_s16) { g_plan_s16 = on; } ~PlanS16() { g_plan_s16 = prev; } };

// ... and the "BN fusion" launches (fte_conv2d_bn_fwd, fte_conv2d_dgrad_bn): no split-K (the statistics / BN sums come from the
// epilogue of the launch that holds the whole reduction) and the per-tile kernels, whose shared epilogue carries them
thread_local bool g_plan_bn = false;
struct PlanBn { bool prev; PlanBn() : prev(g_plan_bn) { g_plan_bn = true; } ~PlanBn() { g_plan_bn = prev; } };

// Stream-K plan of an fp32 forward / data-gradient launch, or false.  The one-block-per-tile schedule finishes a launch of T tiles on
// S = CUs x blocks-per-CU slots in ceil-ish rounds: what the chip loses is (rounds up - rounds) of the LAST round, which matters when
// there are only one or two (the 64- and 128-image shards of SphereNet: 14x14x256 at 64 images is 1.53 tiles of 128x64 per CU).
inline bool plan_sk(long M, long N, long K, int epi, RowPlan* out) {
    static const int mode = getenv("FTE_SK") ? atoi(getenv("FTE_SK")) : 1;          // 0: off; 1: the rule below; 2: every eligible launch (tests, A/B)
    static const int env_tile = getenv("FTE_SK_TILE") ? atoi(getenv("FTE_SK_TILE")) : -1;
    static const int env_tile_d = getenv("FTE_SK_TILE_DGRAD") ? atoi(getenv("FTE_SK_TILE_DGRAD")) : env_tile;
    static const int ops = getenv("FTE_SK_OPS") ? atoi(getenv("FTE_SK_OPS")) : 3;      // A/B hook: 1 = forward only, 2 = data gradient only
    if (!mode || K % 32 || N % 64 || !(ops & (epi == EPI_FWD ? 1 : 2))) return false;
    const long ksteps = K / 32;
    // measured on MI355X (profiles/r5_notes.md): forward -- 64x64 tiles, six workers per CU; data gradient (more registers: the
    // PReLU / partial-sum epilogue) -- 128x64 tiles, four per CU
    const int want = epi == EPI_FWD ? env_tile : env_tile_d;
    int tile = want >= 0 ? want : (epi == EPI_FWD ? TILE_64x64 : TILE_128x64);
    if (tile == TILE_128x128 && N % 128) tile = TILE_128x64;
    const int bpc = igemm_sk_blocks_per_cu(tile, epi);
    if (bpc <= 0) return false;
    int bm, bn;
    igemm_tile_dims(tile, &bm, &bn);
    const long MT = (M + bm - 1) / bm, T = MT * (N / bn), iters = T * ksteps;
    long W = NUM_CU * bpc;
    if (iters >= (1L << 31)) return false;
    if (mode == 1) {
        // The rule, in units of 128x64 tiles per CU (the one-block-per-tile plans' tile at these sizes): stream-K wins up to ~3 tiles
        // per CU (14x14x256 at 64 images, 1.53 per CU: forward 152 -> 127 us, data gradient 160 -> 138; at 128 images, 3.06: 248 ->
        // 228 / 255 -> 245), is even at ~6 (28x28x128 at 128 images) and loses at 12 (56x56x64: 304 -> 321); and every worker should
        // have a few K-steps of its own.
        const double per_cu = (double)(((M + 127) / 128) * (N / 64)) / NUM_CU;
        static const double max_per_cu = getenv("FTE_SK_MAX_PER_CU") ? atof(getenv("FTE_SK_MAX_PER_CU")) : 4.0;
        static const long min_steps = getenv("FTE_SK_MIN_STEPS") ? atol(getenv("FTE_SK_MIN_STEPS")) : 12;      // K-steps per worker
        if (per_cu > max_per_cu) return false;
        if (iters / W < min_steps) return false;
    }
    if (W > iters) W = iters;
    memset(out, 0, sizeof(*out));
    out->main_tile = tile; out->main_rows = M; out->main_mtiles = MT;
    out->tail_mode = 3; out->sk_workers = (int)W;
    out->pw_bytes = igemm_sk_ws_bytes(tile, (int)W);
    return true;
}

inline RowPlan plan_rows(long M, long N, long K, bool allow_pw, bool small_only = false, int epi = -1) {
    RowPlan r;
    memset(&r, 0, sizeof(r));
    if (g_plan_bn) allow_pw = false;
    if (epi >= 0 && allow_pw && !small_only && !plan_bf16() && plan_sk(M, N, K, epi, &r)) return r;
    static const int narrow_tile = getenv("FTE_NARROW_TILE") ? atoi(getenv("FTE_NARROW_TILE")) : TILE_64x64;   // N = 64: measured on MI355X
    // 64x64 beats 128x64 beats 256x64 (fwd 83 / 82 / 75 TF, dgrad 80 / 76 / 65): with only 18 K-steps per tile the
    // layer lives on co-resident blocks hiding each other's prologue / epilogue, not on operand reuse.
    static const int wide_tile = getenv("FTE_WIDE_TILE") ? atoi(getenv("FTE_WIDE_TILE")) : TILE_64x64;         // measured: see below
    // bf16-operand mode: the loop is instruction-issue bound (17 VALU + 20 SALU per MFMA on the 64x64 tile), so the tile
    // with 4x the MFMAs per staged operand wins where N allows it and the launch still fills the chip (batch 512: forward
    // 6.8 -> 5.8 ms, dgrad 8.1 -> 7.5 ms per step; at batch 64 the big tile loses: 3.7 -> 4.7 ms);
    // the short-K stride-2 dgrad classes stay on 64x64 (`small_only`)
    static const bool wide_env = getenv("FTE_WIDE_TILE") != nullptr;
    static const long fills16 = getenv("FTE_FILLS16") ? atol(getenv("FTE_FILLS16")) : 384;       // half a round: measured 768 / 384 / 190 / 120 on the bf16s nets
    const bool fills = ((M + 127) / 128) * (N / 128) >= fills16;        // at least one round of the big tile's slots
    const int wide = (!wide_env && plan_bf16() && !small_only && fills) ? TILE_128x128 : wide_tile;
    static const bool narrow_env = getenv("FTE_NARROW_TILE") != nullptr;
    const bool fills_n = ((M + 127) / 128) * (N / 64) >= fills16;
    const int narrow = (!narrow_env && plan_bf16() && !small_only && fills_n) ? TILE_128x64 : narrow_tile;   // N = 64 layers: +3 %
    // fp32, very tall outputs (>= 4096 tiles of 128 x 64: the 56x56 and 28x28 layers at batch 512): 128 x 64 tiles -- two
    // accumulator blocks per wave, 32 MFMAs per barrier instead of 16 -- measured after the K-order / priority work of round 2:
    // 28x28x128->128 forward 123.5 -> 130.1, dgrad 120.3 -> 127.3 TFLOP/s, 56x56x64->64 91.2 -> 93.3 / 94.8 -> 97.1; at 14x14 and
    // below (fewer tiles than ~4 rounds of the big tile's slots) the 64 x 64 tile stays ahead
    // (not for K < 576: with two K-steps per tile the bigger tile only lengthens the epilogue -- 28x28x64->256 dgrad 122 vs 92 us)
    const bool tall = !wide_env && !narrow_env && !plan_bf16() && !small_only && K >= 576 && ((M + 127) / 128) * (N / 64) >= 4096;
    const int big = tall ? TILE_128x64 : ((N % 128 == 0) ? wide : narrow);
    int bm, bn;
    igemm_tile_dims(big, &bm, &bn);
    const long ntn = N / bn, MT = (M + bm - 1) / bm, T = MT * ntn, ksteps = (K + 31) / 32;
    r.main_tile = big;
    long tail_rows = 0;
    // bf16-STORAGE launches (fte_conv2d_{fwd,dgrad}_s16 without fp32 outputs) with at least a round of the window kernel's 256-row
    // tiles stay ONE unsplit launch: resident loader / consumer blocks (igemm16rw) beat split-K + fix-up there -- 7x7x512 at batch 512:
    // forward 0.186 -> 0.134 ms, data gradient 0.214 -> 0.148.  FTE_PLAN_UNSPLIT16=0 restores the split plan.
    static const bool unsplit16 = !(getenv("FTE_PLAN_UNSPLIT16") && atoi(getenv("FTE_PLAN_UNSPLIT16")) == 0);
    if (unsplit16 && g_plan_s16 && !small_only) {
        const int t16 = N % 128 == 0 ? TILE_128x128 : TILE_128x64;
        const long m16 = (M + 127) / 128, n16 = (M + 127) / 128 * (N / (N % 128 == 0 ? 128 : 64));
        if (n16 >= 512 && n16 < SLOTS) { r.main_tile = t16; r.main_rows = M; r.main_mtiles = m16; return r; }
    }
    if (T >= SLOTS) {
        const long full = T / SLOTS * SLOTS;
        // A leftover fraction of a round used to go to a separate small-tile TAIL launch (mode 1 below).  Measured again at the end of
        // round 2 it no longer pays: blocks of a many-round launch drift apart, the last partial round is absorbed, and the tail
        // launch's own ramp costs more -- SphereNet step 49.30 -> 49.16 ms fp32, 17.15 -> 16.78 ms in the bf16 mode (whose main
        // launches are 4x shorter).  FTE_TAIL_SPLIT=1 brings the tail back (A/B hook).
        static const bool tail_split = getenv("FTE_TAIL_SPLIT") != nullptr;
        if ((bm == 64 && bn == 64) || T - full == 0 || T - full >= SLOTS * 4 / 5 || !tail_split) {
            r.main_rows = M; r.main_mtiles = MT;
            return r;
        }
        r.main_mtiles = full / ntn;
        r.main_rows = r.main_mtiles * bm;
        tail_rows = M - r.main_rows;
    } else {
        tail_rows = M;                                            // less than one round in total
    }
    const long tmt = (tail_rows + bm - 1) / bm, R = tmt * ntn;
    long P = (SLOTS + R / 2) / R;
    // every split keeps >= 32 K-steps (K = 1024): the 4x4x512->512 1x1 convs of ShuffleNet's last stage (1024 tiles, P = 2) ran
    // 58 us split + fixed up against 43 us in one launch, and with 16..32 K-steps per split the 1x1 convs of the ResNet family
    // lose too (SE-ResNet-50 at 128 per GPU +1.1 % without those splits); the 3x3 layers of SphereNet at 64 images per GPU
    // (K = 2304 / 4608, 36 K-steps per split) kept theirs while the step ran on one stream (48 cost them 1.4 %).  With the filter
    // gradients on a second stream (nets/sphere.py) a half-filled launch is no longer alone on the chip and the 2-way splits of the
    // K = 2304 layers stop paying once the unsplit launch has a block for every CU: 48 steps per split there, one GPU, images/s:
    // 32 per GPU 6.99 k -> 7.15 k, 64: 8.40 k -> 8.50 k (bf16 mode 16.2 k -> 17.5 k), 128 / 256 and the BN nets unchanged.
    static const int min_steps_env = getenv("FTE_SPLIT_MINSTEPS") ? atoi(getenv("FTE_SPLIT_MINSTEPS")) : 0;      // tuning hook
    const int min_steps = min_steps_env > 0 ? min_steps_env : (R >= 256 ? 48 : 32);
    if (P > ksteps / min_steps) P = ksteps / min_steps;
    // Split-K pays only when there is NO whole round (small per-GPU shards): measured on MI355X at batch
    // 512 a 32-tile tail split 16 ways is slower than the 64x64 tail (which costs ~3 % of the kernel).
    // P = 2 already pays (batch 64, 14x14x256: 96 -> 101 TFLOP/s forward, 89 -> 95
    // dgrad; batch 128, 7x7x512: 100 -> 111); P = 1 means the tiles fill the chip on their own.
    static const int min_p = getenv("FTE_MIN_SPLITP") ? atoi(getenv("FTE_MIN_SPLITP")) : 2;     // tuning hook
    // bf16 plans: an unsplit launch on the LDS-DMA kernels (three-stage ring below 768 tiles) instead of split-K + fix-up on the
    // register-staged 64x64 kernel, whenever the 128-row tiles give pick_tile's minimum.  FTE_PLAN16_NOSPLIT=0: the split plan.
    static const bool nosplit16 = !(getenv("FTE_PLAN16_NOSPLIT") && atoi(getenv("FTE_PLAN16_NOSPLIT")) == 0);
    if (nosplit16 && plan_bf16() && !small_only && T < SLOTS) {
        const int t16 = pick_tile(M, N);
        if (t16 == TILE_128x128 || t16 == TILE_128x64) {
            int tbm, tbn;
            igemm_tile_dims(t16, &tbm, &tbn);
            r.main_tile = t16; r.main_rows = M; r.main_mtiles = (M + tbm - 1) / tbm;
            return r;
        }
    }
    if (allow_pw && P >= min_p && T < SLOTS) {
        r.tail_mode = 2; r.tail_tile = big; r.tail_mtiles = tmt * FIXUP_CHUNKS;   // partial rows: one per (tile row, chunk)
        r.tail_kchunk = (int)((ksteps + P - 1) / P * 32);
        r.tail_splits = (int)((K + r.tail_kchunk - 1) / r.tail_kchunk);
        r.pw_bytes = (size_t)r.tail_splits * R * bm * bn * sizeof(float);
        return r;
    }
    if (T >= SLOTS) {
        r.tail_mode = 1; r.tail_tile = TILE_64x64; r.tail_mtiles = (tail_rows + 63) / 64;
        return r;
    }
    r.main_tile = pick_tile(M, N);                                // single launch with a smaller tile
    igemm_tile_dims(r.main_tile, &bm, &bn);
    r.main_rows = M; r.main_mtiles = (M + bm - 1) / bm;
    return r;
}

// split-K plan: pick the split count whose tiles*splits fills whole rounds of SLOTS best
// (>= 8 K-steps per split; ties go to fewer splits = less slab traffic)
inline void plan_splits(long tiles, int K, int* splits, int* kchunk, bool prefer8 = false) {
    static const int mink = getenv("FTE_SPLIT_MINK") ? atoi(getenv("FTE_SPLIT_MINK")) : 256;      // tuning hook: shortest K range per split
    const int maxs = K / mink > 0 ? K / mink : 1;
    const long SLOTS = SLOTS_BIG;
    static const int max_rounds = getenv("FTE_SPLIT_ROUNDS") ? atoi(getenv("FTE_SPLIT_ROUNDS")) : 3;      // tuning hook
    static const int min_rounds = getenv("FTE_SPLIT_MINROUNDS") ? atoi(getenv("FTE_SPLIT_MINROUNDS")) : 1;
    long lo = (min_rounds * SLOTS + tiles - 1) / tiles, hi = (max_rounds * SLOTS + tiles - 1) / tiles;
    if (lo < 1) lo = 1;
    if (hi > maxs) hi = maxs;
    if (lo > hi) lo = hi;
    double best = -1.0;
    int bs = 1;
    for (long sI = lo; sI <= hi; ++sI) {
        const int kc = (int)(((K + sI - 1) / sI + 31) / 32 * 32);
        const long sp = (K + kc - 1) / kc;
        const double rounds = (double)(tiles * sp) / SLOTS;
        double eff = rounds / (double)(long)(rounds + 0.999999);
        if (prefer8 && sp % 8 == 0) eff += 0.05;     // split-major placement: one K range per XCD
        if (eff > best + 0.02) { best = eff; bs = (int)sI; }
    }
    const int kc = ((K + bs - 1) / bs + 31) / 32 * 32;
    *kchunk = kc;
    *splits = (K + kc - 1) / kc;
}

inline void zero_params(IgemmParams* p) { memset(p, 0, sizeof(*p)); }

inline void plain_a(IgemmParams* p, const float* a, int ld, int kc) {
    p->A = a; p->a_OH = 1; p->a_OW = 1; p->a_IH = 1; p->a_IW = 1; p->a_stride = 1;
    p->a_ld = ld; p->a_KC = kc; p->a_NT = 1;
}

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }
constexpr size_t SCRATCH_BYTES = (size_t)REDUCE_SCRATCH_FLOATS * sizeof(float);

// operand sizes for the buffer-load range check; tensors must stay below 2 GiB (offsets are 32-bit,
// 0x80000000 is the out-of-range marker)
inline bool set_bytes(IgemmParams* p, size_t a_elems, size_t b_elems, size_t esize = 4) {
    const size_t lim = (size_t)1 << 31;
    if (a_elems * esize >= lim || b_elems * esize >= lim) return false;
    p->a_bytes = (unsigned)(a_elems * esize);
    p->b_bytes = (unsigned)(b_elems * esize);
    return true;
}

// main (+ tail) launches of one row-tiled op; PA/PB partial rows are numbered main first, then tail
inline hipError_t launch_rows(IgemmParams p, const RowPlan& rp, int al, int bl, int epi, long prow0, float* pw, hipStream_t st) {
    const int M = p.M;
    hipError_t e = hipSuccess;
    if (rp.tail_mode == 3) {                      // stream-K: slabs first, the flag words behind them
        int bm, bn;
        igemm_tile_dims(rp.main_tile, &bm, &bn);
        const long T = rp.main_mtiles * (p.N / bn), iters = T * (p.K / 32);
        p.m_base = 0; p.prow0 = (int)prow0;
        p.sk_workers = rp.sk_workers; p.sk_base = (int)(iters / rp.sk_workers); p.sk_rem = (int)(iters % rp.sk_workers);
        p.SKW = pw; p.SKF = reinterpret_cast<unsigned*>(pw + (size_t)rp.sk_workers * bm * bn);
        return igemm_launch(p, al, bl, epi, rp.main_tile, 1, st);
    }
    if (rp.main_rows > 0) {
        p.m_base = 0; p.M = (int)rp.main_rows; p.prow0 = (int)prow0;
        e = igemm_launch(p, al, bl, epi, rp.main_tile, 1, st);
        if (e != hipSuccess) return e;
    }
    if (rp.tail_mode == 0) return e;
    p.m_base = (int)rp.main_rows; p.M = M; p.prow0 = (int)(prow0 + rp.main_mtiles);
    if (rp.tail_mode == 1) return igemm_launch(p, al, bl, epi, rp.tail_tile, 1, st);
    p.PW = pw; p.kchunk = rp.tail_kchunk;
    e = igemm_launch(p, al, bl, epi, rp.tail_tile, rp.tail_splits, st);
    if (e != hipSuccess) return e;
    return igemm_fixup(p, epi, rp.tail_tile, rp.tail_splits, st);
}

// the tile plan depends on the MFMA dtype (fte_set_mfma_dtype): workspace queries answer for BOTH modes, so that a buffer
// sized once stays valid when the mode is switched
template <class F>
size_t both_modes(F f) {
    size_t a, b, c;
    { PlanMode m(false); a = f(); }
    { PlanMode m(true); b = f(); }
    {                      // ... and the bf16-STORAGE plan of the *_s16 entry points (one unsplit launch of 128-row tiles where it applies)
        Plan16 guard;
        PlanS16 storage(true);
        c = f();
    }
    return a > b ? (a > c ? a : c) : (b > c ? b : c);
}

// ---- Winograd F(2x2, 3x3) plan (wino.hip) ---------------------------------------------------------------------------------------
// The stride-1 3x3 layers of the BN-free fp32 path (nets/sphere.py:41-42) may run as 16 products of 1/2.25 of the MACs.  FTE_CONV_ALGO /
// fte_set_conv_algo: direct = never, winograd = wherever the kernels exist (channels % 64, fp32 operands), auto = the measured rule:
// layers of >= FTE_WINO_MIN_C channels (default 128: SphereNet's stages 2-4) take it for all three products; 64-channel layers per FTE_WINO_OPS64 -- below that the transform traffic (16 floats per tile and channel, in
// and out of HBM) outweighs the saved MFMA time.
inline int wino_min_c() {
    static const int v = getenv("FTE_WINO_MIN_C") ? atoi(getenv("FTE_WINO_MIN_C")) : 128;
    return v;
}
// forward-side tile transforms in 256-thread blocks (wino.h; the default: 72 registers per SIMD fit beside a resident block of the forward
// product, so one half shard's transform runs under the other's product -- nets/sphere.py backbone; alone 34.72 -> 34.58 ms per step at 512
// images, with the half shards 34.36 -> 34.21).  FTE_WINO_FWD_TPB=512: the 512-thread blocks of the data-gradient side.
inline bool wino_fwd_small_tiles() {
    static const int tpb = getenv("FTE_WINO_FWD_TPB") ? atoi(getenv("FTE_WINO_FWD_TPB")) : 256;
    return tpb != 512;
}
inline bool wino_shape_ok(int n, int h, int wd, int cin, int cout, int ksize, int stride) {
    if (ksize != 3 || stride != 1 || cin % 64 || cout % 64 || n <= 0 || h < 2 || wd < 2) return false;
    const size_t lim = (size_t)1 << 31;
    return (size_t)n * h * wd * cin * 4 < lim && (size_t)n * h * wd * cout * 4 < lim;
}
// op: 0 forward, 1 data gradient, 2 filter gradient.  wino_sized: the current switch would pick the algorithm for this layer in the
// fp32 mode (what the workspace queries size for); wino_wanted: ... and this launch is an fp32, BN-free one
inline bool wino_sized(int n, int h, int wd, int cin, int cout, int ksize, int stride, int op) {
    const int algo = wino_get_algo();
    if (algo == 0 || !wino_shape_ok(n, h, wd, cin, cout, ksize, stride)) return false;
    if (op == 2 && !wino_wgrad_splits(cin, cout)) return false;
    if (algo == 1) return true;
    static const int ops = getenv("FTE_WINO_OPS") ? atoi(getenv("FTE_WINO_OPS")) : 7;      // A/B hook: bit per op
    // 64-channel layers (SphereNet's stage 1, 56x56): the tile packs are 16 floats per tile and channel -- 1.6 GB at 512 images, 0.4 ms of
    // HBM time per transform against ~0.5 ms saved in a product.  Measured (one box, same call): forward + filter gradient (they share
    // one transform) 35.60 -> 34.96 ms per step, the data gradient on top 34.98: it stays direct.  FTE_WINO_OPS64: bit per op for that class
    static const int ops64 = getenv("FTE_WINO_OPS64") ? atoi(getenv("FTE_WINO_OPS64")) : 5;
    const int cmin = cin < cout ? cin : cout;
    if (cmin < wino_min_c()) return cmin >= 64 && (ops64 & (1 << op)) != 0;
    return (ops & (1 << op)) != 0;
}
inline bool wino_wanted(int n, int h, int wd, int cin, int cout, int ksize, int stride, int op) {
    return !plan_bf16() && !g_plan_bn && wino_sized(n, h, wd, cin, cout, ksize, stride, op);
}
// every tensor the Winograd kernels touch moves 16 bytes per lane, like the direct path's (igemm_launch): 16-byte aligned or an error code
inline bool aligned16(std::initializer_list<const void*> ptrs) {
    uintptr_t v = 0;
    for (const void* q : ptrs) v |= (uintptr_t)q;
    return (v & 15) == 0;
}
struct WinoWs { size_t v_off, u_off, slab_off, total; };
// workspace layout behind `head` bytes the caller keeps for itself: [V pack | U pack (| slabs)]
inline WinoWs wino_ws(int n, int h, int wd, int cin, int cout, int op, size_t head) {
    const WinoGeom g = wino_geom(n, h, wd);
    WinoWs w;
    w.v_off = align_up(head);
    if (op == 0) {            // V(x): cin channels; U: filters
        w.u_off = w.v_off + align_up(wino_pack_floats(g.M, cin) * 4);
        w.slab_off = w.total = w.u_off + align_up((size_t)16 * cin * cout * 4);
    } else if (op == 1) {     // V(dz): cout channels; U: rotated filters
        w.u_off = w.v_off + align_up(wino_pack_floats(g.M, cout) * 4);
        w.slab_off = w.total = w.u_off + align_up((size_t)16 * cin * cout * 4);
    } else {                  // V(x), slabs (U' of dz is computed inside the kernel)
        w.u_off = w.v_off + align_up(wino_pack_floats(g.M, cin) * 4);
        w.slab_off = w.u_off;
        w.total = w.slab_off + align_up((size_t)wino_wgrad_splits(cin, cout) * 16 * cin * cout * 4);
    }
    return w;
}
}  // namespace

extern "C" {

#ifndef FTE_SRC_SHA
#define FTE_SRC_SHA "unstamped"
#endif
// "... src:<hash of csrc/*.hip, *.h at build time>" (csrc/build.sh): bench.py and __graft_entry__.build() compare it with the sources'
const char* fte_version(void) { return "fte 0.2 gfx950 fp32-mfma src:" FTE_SRC_SHA; }

int fte_to_bf16(const float* x, uint16_t* y, long n, void* stream) {
    if (!x || !y || n <= 0 || n % 4) return FTE_EINVAL;
    return rc(k_to_bf16(x, y, n, (hipStream_t)stream));
}
int fte_pack_weights_bf16(const float* w, uint16_t* w16, uint16_t* w16t, int ksize, int cin, int cout, void* stream) {
    if (!w || (!w16 && !w16t) || ksize <= 0 || cin <= 0 || cout <= 0) return FTE_EINVAL;
    return rc(k_pack_weights_bf16(w, w16, w16t, ksize * ksize, cin, cout, (hipStream_t)stream));
}

int fte_pack_weights_bf16_table(const float* params, uint16_t* dst, const int32_t* table, int nconv, long total, int transposed, void* stream) {
    if (!params || !dst || !table || nconv <= 0 || nconv > 64 || total <= 0 || total % 4) return FTE_EINVAL;
    return rc(k_pack_weights_table(params, dst, table, nconv, total, transposed, (hipStream_t)stream));
}

int fte_set_mfma_dtype(int dtype) {
    if (dtype != FTE_MFMA_F32 && dtype != FTE_MFMA_BF16) return FTE_EINVAL;
    igemm_set_bf16(dtype == FTE_MFMA_BF16);
    return FTE_OK;
}
int fte_get_mfma_dtype(void) { return igemm_get_bf16() ? FTE_MFMA_BF16 : FTE_MFMA_F32; }

int fte_set_conv_algo(int algo) {
    if (algo < FTE_CONV_DIRECT || algo > FTE_CONV_AUTO) return FTE_EINVAL;
    wino_set_algo(algo);
    return FTE_OK;
}
int fte_get_conv_algo(void) { return wino_get_algo(); }

int fte_prof_enable(int on) { igemm_prof_enable(on != 0, on == 1); return FTE_OK; }
int fte_prof_count(void) { return igemm_prof_count(); }
int fte_prof_get_shape(int i, int* mnk, double* bytes) {
    if (!mnk || !bytes) return FTE_EINVAL;
    return rc(igemm_prof_get_shape(i, mnk, bytes));
}
int fte_prof_get_name(int i, char* buf, int buflen) {
    if (!buf || buflen <= 0) return FTE_EINVAL;
    return rc(igemm_prof_get_name(i, buf, buflen));
}
int fte_prof_get(int i, int* sig, double* flops, float* ms) {
    if (!sig || !flops || !ms) return FTE_EINVAL;
    return rc(igemm_prof_get(i, sig, flops, ms));
}

// ------------------------------------------------------------------------------------------------
size_t fte_conv2d_fwd_ws_bytes(int n, int h, int wd, int cin, int cout, int ksize, int stride) {
    if (n <= 0 || cin <= 0 || cin % 32 || cout <= 0 || cout % 64) return 0;
    const Pads ph = same_pads(h, ksize, stride), pw = same_pads(wd, ksize, stride);
    size_t need = both_modes([&] { return plan_rows((long)n * ph.out * pw.out, cout, (long)ksize * ksize * cin, true, false, EPI_FWD).pw_bytes; });
    if (wino_sized(n, h, wd, cin, cout, ksize, stride, 0)) {
        const size_t wn = wino_ws(n, h, wd, cin, cout, 0, 0).total;
        if (wn > need) need = wn;
    }
    return need;
}

// x / w are bf16 copies (x16 [n,h,wd,cin]; w16t [k*k][cout][cin], fte_pack_weights_bf16) when `src16`
static int conv2d_fwd_impl(const void* x, const void* w, bool src16, const float* bias, const float* alpha, const float* res,
                           float* z, float* y, uint16_t* y16, int n, int h, int wd, int cin, int cout, int ksize, int stride,
                           void* ws, size_t ws_bytes, void* stream, const uint16_t* res16 = nullptr, uint16_t* z16 = nullptr,
                           float* stat_part = nullptr, int* stat_rows = nullptr, float* vpack = nullptr) {
    if (!x || !w || (!y && !(src16 && y16)) || n <= 0 || cin % 32 || cout % 64 || (stride != 1 && stride != 2) || (ksize != 1 && ksize != 3))
        return FTE_EINVAL;
    const Pads ph = same_pads(h, ksize, stride), pw = same_pads(wd, ksize, stride);
    IgemmParams p;
    zero_params(&p);
    p.M = n * ph.out * pw.out; p.N = cout; p.K = ksize * ksize * cin; p.kchunk = p.K;
    p.a_OH = ph.out; p.a_OW = pw.out; p.a_IH = h; p.a_IW = wd; p.a_stride = stride;
    p.a_ld = cin; p.a_KC = cin; p.a_NT = ksize * ksize;
    for (int r = 0; r < ksize; ++r)
        for (int s = 0; s < ksize; ++s) { p.a_dh[r * ksize + s] = r - ph.before; p.a_dw[r * ksize + s] = s - pw.before; }
    p.A = (const float*)x;
    p.B = (const float*)w;
    p.c_ld = cout;
    p.Y = y; p.Z = z; p.R = res; p.bias = bias; p.alpha = alpha; p.Y16 = y16;
    p.R16 = res16; p.Z16 = z16;
    int bl = BL_KN;
    if (src16) {                                     // transposed pack: rows = output channels, k-contiguous per tap
        p.src16 = 1; p.b_ld = cin; bl = BL_NK;
        for (int t = 0; t < ksize * ksize; ++t) p.b_tapoff[t] = t * cout * cin;
    } else {
        p.b_ld = cout;
    }
    if (!set_bytes(&p, (size_t)n * h * wd * cin, (size_t)ksize * ksize * cin * cout, src16 ? 2 : 4)) return FTE_EINVAL;
    if (!src16 && !stat_part && y && wino_wanted(n, h, wd, cin, cout, ksize, stride, 0)) {
        WinoWs wl = wino_ws(n, h, wd, cin, cout, 0, 0);
        if (vpack) { wl.u_off = 0; wl.total = align_up((size_t)16 * cin * cout * 4); }      // the caller keeps V (fte_conv3x3_fwd_keep): ws holds the filters only
        if (ws && ws_bytes >= wl.total) {          // Winograd F(2x2,3x3): filter transform, tile transform, 16 products + output transform + epilogue
            if (!aligned16({x, w, bias, alpha, res, z, y, ws, vpack})) return FTE_EINVAL;
            float* V = vpack ? vpack : (float*)((char*)ws + wl.v_off);
            float* U = (float*)((char*)ws + wl.u_off);
            hipError_t e = wino_transform_filter((const float*)w, U, cin, cout, 0, (hipStream_t)stream);
            if (e != hipSuccess) return (int)e;
            e = wino_transform_tiles((const float*)x, V, n, h, wd, cin, 0, (hipStream_t)stream, wino_fwd_small_tiles());
            if (e != hipSuccess) return (int)e;
            WinoMMParams q;
            memset(&q, 0, sizeof(q));
            q.V = V; q.U = U; q.K = cin; q.N = cout; q.g = wino_geom(n, h, wd);
            q.Y = y; q.Z = z; q.R = res; q.bias = bias; q.alpha = alpha;
            return rc(wino_mm(q, EPI_FWD, (hipStream_t)stream));
        }
    }
    if (vpack) return FTE_EWORKSPACE;              // a kept V was asked for and the Winograd path did not run: never silently
    RowPlan rp = plan_rows(p.M, p.N, p.K, ws != nullptr, false, EPI_FWD);
    if (rp.tail_mode >= 2 && ws_bytes < rp.pw_bytes) rp = plan_rows(p.M, p.N, p.K, false);   // no room: small-tile tail
    if (stat_part) {                                 // "BN fusion": one statistics partial row per tile row of the launch(es)
        if (rp.tail_mode == 2) return FTE_EINVAL;    // (plan_rows never splits under PlanBn)
        p.SP = stat_part;
        *stat_rows = (int)(rp.main_mtiles + rp.tail_mtiles);
    }
    return rc(launch_rows(p, rp, AL_MK, bl, EPI_FWD, 0, (float*)ws, (hipStream_t)stream));
}
int fte_conv2d_fwd(const float* x, const float* w, const float* bias, const float* alpha, const float* res,
                   float* z, float* y, int n, int h, int wd, int cin, int cout, int ksize, int stride,
                   void* ws, size_t ws_bytes, void* stream) {
    return conv2d_fwd_impl(x, w, false, bias, alpha, res, z, y, nullptr, n, h, wd, cin, cout, ksize, stride, ws, ws_bytes, stream);
}
int fte_conv2d_fwd16(const uint16_t* x16, const uint16_t* w16t, const float* bias, const float* alpha, const float* res,
                     float* z, float* y, uint16_t* y16, int n, int h, int wd, int cin, int cout, int ksize, int stride,
                     void* ws, size_t ws_bytes, void* stream) {
    Plan16 guard;
    return conv2d_fwd_impl(x16, w16t, true, bias, alpha, res, z, y, y16, n, h, wd, cin, cout, ksize, stride, ws, ws_bytes, stream);
}

int fte_conv2d_fwd_s16(const uint16_t* x16, const uint16_t* w16t, const float* bias, const float* alpha, const uint16_t* res16,
                       uint16_t* z16, uint16_t* y16, float* z32, float* y32, int n, int h, int wd, int cin, int cout, int ksize, int stride,
                       void* ws, size_t ws_bytes, void* stream) {
    if (!y16 || (res16 && ((uintptr_t)res16 & 15))) return FTE_EINVAL;
    Plan16 guard;
    PlanS16 storage(!z32 && !y32);
    return conv2d_fwd_impl(x16, w16t, true, bias, alpha, nullptr, z32, y32, y16, n, h, wd, cin, cout, ksize, stride, ws, ws_bytes, stream, res16, z16);
}

size_t fte_conv3x3_fwd_ws_bytes(int n, int h, int wd, int cin, int cout, int stride) {
    return fte_conv2d_fwd_ws_bytes(n, h, wd, cin, cout, 3, stride);
}
int fte_conv3x3_fwd(const float* x, const float* w, const float* bias, const float* alpha, const float* res,
                    float* z, float* y, int n, int h, int wd, int cin, int cout, int stride,
                    void* ws, size_t ws_bytes, void* stream) {
    return fte_conv2d_fwd(x, w, bias, alpha, res, z, y, n, h, wd, cin, cout, 3, stride, ws, ws_bytes, stream);
}
int fte_conv3x3_fwd_keep(const float* x, const float* w, const float* bias, const float* alpha, const float* res,
                         float* z, float* y, int n, int h, int wd, int cin, int cout, int stride, float* vpack,
                         void* ws, size_t ws_bytes, void* stream) {
    if (vpack && ((uintptr_t)vpack & 15)) return FTE_EINVAL;
    return conv2d_fwd_impl(x, w, false, bias, alpha, res, z, y, nullptr, n, h, wd, cin, cout, 3, stride, ws, ws_bytes, stream,
                           nullptr, nullptr, nullptr, nullptr, vpack);
}
int fte_conv3x3_algo(int n, int h, int wd, int cin, int cout, int stride, int op) {
    if (op < 0 || op > 2) return FTE_EINVAL;
    return (!igemm_get_bf16() && wino_sized(n, h, wd, cin, cout, 3, stride, op)) ? FTE_CONV_WINOGRAD : FTE_CONV_DIRECT;
}
size_t fte_wino_pack_bytes(int n, int h, int wd, int c) {
    if (n <= 0 || h < 2 || wd < 2 || c <= 0 || c % 64) return 0;
    return wino_pack_floats(wino_geom(n, h, wd).M, c) * sizeof(float);
}

// ------------------------------------------------------------------------------------------------
namespace {
struct DgradClass { int ph, pw, hq, wq, ntap, dh[9], dw[9], wt[9]; RowPlan rp; long mtiles; };
int dgrad_classes(int n, int h, int wd, int cin, int cout, int ksize, int stride, DgradClass* cls) {
    const Pads ph = same_pads(h, ksize, stride), pw = same_pads(wd, ksize, stride);
    int nc = 0;
    for (int a = 0; a < stride; ++a)
        for (int b = 0; b < stride; ++b) {
            DgradClass& c = cls[nc];
            c.ph = a; c.pw = b;
            c.hq = (h - a + stride - 1) / stride;
            c.wq = (wd - b + stride - 1) / stride;
            c.ntap = 0;
            if (c.hq <= 0 || c.wq <= 0) continue;
            for (int r = 0; r < ksize; ++r) {
                if ((a + ph.before - r) % stride) continue;
                for (int s = 0; s < ksize; ++s) {
                    if ((b + pw.before - s) % stride) continue;
                    // hi = ho*stride + r - pt  with hi = hq*stride + a  ->  ho = hq + (a + pt - r)/stride
                    c.dh[c.ntap] = (a + ph.before - r) / stride;
                    c.dw[c.ntap] = (b + pw.before - s) / stride;
                    c.wt[c.ntap] = r * ksize + s;
                    ++c.ntap;
                }
            }
            const long M = (long)n * c.hq * c.wq;
            // FTE_DGRAD_CLS_BIG=1 (A/B hook): let the bf16 plans give the parity classes the 128-row tiles of the LDS-DMA kernels
            static const bool cls_big = getenv("FTE_DGRAD_CLS_BIG") && atoi(getenv("FTE_DGRAD_CLS_BIG")) == 1;
            c.rp = plan_rows(M, cin, (long)c.ntap * cout, true, stride != 1 && !cls_big, EPI_DGRAD);
            c.mtiles = c.rp.main_mtiles + c.rp.tail_mtiles;
            ++nc;
        }
    return nc;
}
// stride-2 3x3 with even image sides: the four classes have identical row grids and 4 / 2 / 2 / 1 taps
// ... and only while one class alone cannot fill the chip (fewer 64x64 tiles than resident slots: the small per-GPU
// shards).  Measured on MI355X, 28x28x128 <- 256 at batch 512: four launches 0.55 ms, the merged one 0.75 ms with the SAME
// instruction counts but half the resident waves (PMC); at batch 64 the merged launch is the faster one (36 -> 50 TFLOP/s).
// FTE_DGRAD_MERGED16=1 (A/B hook): in the bf16 plans the four classes always share ONE launch on the LDS-DMA kernels' 128-row tiles
static bool dgrad_merged16() {
    static const bool on = getenv("FTE_DGRAD_MERGED16") && atoi(getenv("FTE_DGRAD_MERGED16")) == 1;
    return on && plan_bf16();
}
bool dgrad_mergeable(const DgradClass* cls, int nc, int n, int cin) {
    static const char* mode = getenv("FTE_DGRAD_CLASSES");      // tuning hook: "split" | "merged"
    if (nc != 4 || (mode && mode[0] == 's')) return false;
    for (int i = 0; i < nc; ++i)
        if (cls[i].hq != cls[0].hq || cls[i].wq != cls[0].wq || cls[i].ntap < 1) return false;
    if ((mode && mode[0] == 'm') || dgrad_merged16()) return true;
    const long tiles = (((long)n * cls[0].hq * cls[0].wq + 63) / 64) * (cin / 64);
    return tiles < SLOTS;
}
static int dgrad_merged_tile(int cin) {
    if (!dgrad_merged16()) return TILE_64x64;
    return cin % 128 == 0 ? TILE_128x128 : TILE_128x64;
}
// partial rows (one per tile row and class) the merged launch writes for dalpha / dbias
long dgrad_merged_rows(const DgradClass* cls, int nc, int n, int cin) {
    if (!dgrad_mergeable(cls, nc, n, cin)) return 0;
    int bm, bn;
    igemm_tile_dims(dgrad_merged_tile(cin), &bm, &bn);
    return (long)nc * (((long)n * cls[0].hq * cls[0].wq + bm - 1) / bm);
}
// tile of the merged launch and the class order (most taps first)
int dgrad_merged_plan(const DgradClass* cls, int nc, int n, int cin, int* order) {
    for (int i = 0; i < nc; ++i) order[i] = i;
    for (int i = 0; i < nc; ++i)
        for (int j = i + 1; j < nc; ++j)
            if (cls[order[j]].ntap > cls[order[i]].ntap) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
    (void)n;
    return dgrad_merged_tile(cin);
}
}  // namespace

size_t fte_conv2d_dgrad_ws_bytes(int n, int h, int wd, int cin, int cout, int ksize, int stride) {
    if (n <= 0 || cin <= 0 || cin % 64 || cout % 32) return 0;      // shapes fte_conv2d_dgrad rejects need no workspace
    size_t need = both_modes([&] {
        DgradClass cls[4];
        const int nc = dgrad_classes(n, h, wd, cin, cout, ksize, stride, cls);
        long rows = 0;
        size_t pw = 0;
        for (int i = 0; i < nc; ++i) { rows += cls[i].mtiles; if (cls[i].rp.pw_bytes > pw) pw = cls[i].rp.pw_bytes; }
        const long merged_rows = dgrad_merged_rows(cls, nc, n, cin);
        if (merged_rows > rows) rows = merged_rows;
        // 64-row tiles give the most partial rows; the layout below (two halves, then scratch, then split-K tiles) is
        // sized for the larger of the two modes in each part
        return 2 * align_up((size_t)rows * cin * sizeof(float)) + SCRATCH_BYTES + pw;
    });
    if (wino_sized(n, h, wd, cin, cout, ksize, stride, 1)) {
        const WinoGeom g = wino_geom(n, h, wd);
        const size_t wn = wino_ws(n, h, wd, cin, cout, 1, 2 * align_up((size_t)g.MB * cin * sizeof(float)) + SCRATCH_BYTES).total;
        if (wn > need) need = wn;
    }
    return need;
}

// the BN layer a "BN fusion" data gradient lands on (fte_conv2d_dgrad_bn): see igemm.h
struct DgradBn {
    const void* zbn; const void* ybn;                 // the BN's input z; its (add + ReLU'd) output when the mask comes from there
    const float* gamma; const float* mean; const float* rstd; const float* scale; const float* shift;
    float* dgamma; float* dbeta; float* coef;
};
// dz / w are bf16 copies (dz16 [n,ho,wo,cout]; w16 [k*k][cin][cout], the HWIO layout) when `src16`
static int conv2d_dgrad_impl(const void* dz, const void* w, bool src16, const float* addin, const float* zprev,
                             const float* alpha_prev, float* raw, float* dzprev, uint16_t* dzprev16, float* dalpha_prev, float* dbias_prev,
                             int n, int h, int wd, int cin, int cout, int ksize, int stride, void* ws, size_t ws_bytes, void* stream,
                             const uint16_t* addin16 = nullptr, const uint16_t* zprev16 = nullptr, uint16_t* raw16 = nullptr,
                             const DgradBn* bn = nullptr) {
    const float* zx = nullptr;
    const uint16_t* zx16 = nullptr;
    if (bn) {          // mask tensor -> the Zin slot, the BN input -> Zx when the mask comes from the output
        const void* msk = (!bn->scale && bn->ybn) ? bn->ybn : bn->zbn;
        const void* zb = (!bn->scale && bn->ybn) ? bn->zbn : nullptr;
        if (src16) { zprev16 = (const uint16_t*)msk; zx16 = (const uint16_t*)zb; }
        else { zprev = (const float*)msk; zx = (const float*)zb; }
        alpha_prev = bn->gamma;                       // (never read by the BN epilogue; keeps the PReLU argument check below quiet)
        dalpha_prev = bn->dgamma; dbias_prev = bn->dbeta;
    }
    if (zprev16 && !zprev) zprev = reinterpret_cast<const float*>(zprev16);      // "has a PReLU mask" below; the kernels read p.Zin16
    const bool z16only = zprev16 != nullptr;
    if (!dz || !w || (!dzprev && !(src16 && dzprev16)) || n <= 0 || cout % 32 || cin % 64 || (stride != 1 && stride != 2) || (ksize != 1 && ksize != 3)) return FTE_EINVAL;
    if (zprev && !alpha_prev) return FTE_EINVAL;
    const Pads pho = same_pads(h, ksize, stride), pwo = same_pads(wd, ksize, stride);
    if (!src16 && !bn && dzprev && !dzprev16 && !addin16 && !zprev16 && !raw16 && wino_wanted(n, h, wd, cin, cout, ksize, stride, 1)) {
        // Winograd: the forward algorithm on dz with the rotated, channel-swapped filters; one partial row of dalpha / dbias per row block
        const WinoGeom g = wino_geom(n, h, wd);
        const size_t half_w = align_up((size_t)g.MB * cin * sizeof(float));
        const WinoWs wl = wino_ws(n, h, wd, cin, cout, 1, 2 * half_w + SCRATCH_BYTES);
        if (ws && ws_bytes >= wl.total) {
            if (!aligned16({dz, w, addin, zprev, alpha_prev, raw, dzprev, ws})) return FTE_EINVAL;
            const bool part = zprev && (dalpha_prev || dbias_prev);
            float* V = (float*)((char*)ws + wl.v_off);
            float* U = (float*)((char*)ws + wl.u_off);
            hipError_t e = wino_transform_filter((const float*)w, U, cin, cout, 1, (hipStream_t)stream);
            if (e != hipSuccess) return (int)e;
            e = wino_transform_tiles((const float*)dz, V, n, h, wd, cout, 0, (hipStream_t)stream);
            if (e != hipSuccess) return (int)e;
            WinoMMParams q;
            memset(&q, 0, sizeof(q));
            q.V = V; q.U = U; q.K = cout; q.N = cin; q.g = g;
            q.ADD = addin; q.RAW = raw; q.Zin = zprev; q.alpha = alpha_prev; q.amod = cin; q.DZ = dzprev;
            q.PA = part ? (float*)ws : nullptr; q.PB = part ? (float*)((char*)ws + half_w) : nullptr;
            e = wino_mm(q, EPI_DGRAD, (hipStream_t)stream);
            if (e != hipSuccess) return (int)e;
            if (part) {
                float* scratch = (float*)((char*)ws + 2 * half_w);
                if (dalpha_prev && dbias_prev) return rc(k_reduce_rows2(q.PA, dalpha_prev, q.PB, dbias_prev, nullptr, 1, g.MB, cin, 1, 1.f, scratch, (hipStream_t)stream));
                if (dalpha_prev) return rc(k_reduce_rows(q.PA, dalpha_prev, nullptr, 1, g.MB, cin, 1, 1.f, scratch, (hipStream_t)stream));
                return rc(k_reduce_rows(q.PB, dbias_prev, nullptr, 1, g.MB, cin, 1, 1.f, scratch, (hipStream_t)stream));
            }
            return FTE_OK;
        }
    }
    DgradClass cls[4];
    const int nc = dgrad_classes(n, h, wd, cin, cout, ksize, stride, cls);
    const bool merged = dgrad_mergeable(cls, nc, n, cin);
    long rows = 0;
    for (int i = 0; i < nc; ++i) rows += cls[i].mtiles;
    if (merged) rows = dgrad_merged_rows(cls, nc, n, cin);
    const bool want_part = zprev && (dalpha_prev || dbias_prev);
    const size_t half = align_up((size_t)rows * cin * sizeof(float));
    size_t pw_need = 0;
    for (int i = 0; i < nc && !merged; ++i) if (cls[i].rp.pw_bytes > pw_need) pw_need = cls[i].rp.pw_bytes;
    const size_t fixed = 2 * half + SCRATCH_BYTES;
    if ((want_part || pw_need) && (!ws || ws_bytes < fixed + pw_need)) return FTE_EWORKSPACE;
    float* scratch = want_part ? (float*)((char*)ws + 2 * half) : nullptr;
    float* pwbuf = pw_need ? (float*)((char*)ws + fixed) : nullptr;
    float* PA = want_part ? (float*)ws : nullptr;
    float* PB = want_part ? (float*)((char*)ws + half) : nullptr;
    if (merged) {
        // one launch for all parity classes (each had its own until now: at 36-60 % of the rate of a stride-1 layer,
        // the 1-tap class being a K = cout GEMM that cannot fill the chip on its own)
        int order[4];
        const int tile = dgrad_merged_plan(cls, nc, n, cin, order);
        IgemmParams p;
        zero_params(&p);
        const DgradClass& c0 = cls[0];
        p.M = n * c0.hq * c0.wq; p.N = cin; p.a_KC = cout;
        p.A = (const float*)dz; p.a_OH = c0.hq; p.a_OW = c0.wq; p.a_IH = pho.out; p.a_IW = pwo.out; p.a_stride = 1; p.a_ld = cout;
        p.ncls = nc;
        int t = 0;
        for (int k = 0; k < nc; ++k) {
            const DgradClass& c = cls[order[k]];
            p.cls_tap0[k] = t; p.cls_ph[k] = c.ph; p.cls_pw[k] = c.pw;
            for (int j = 0; j < c.ntap; ++j, ++t) { p.a_dh[t] = c.dh[j]; p.a_dw[t] = c.dw[j]; p.b_tapoff[t] = c.wt[j] * cin * cout; }
        }
        p.cls_tap0[nc] = t;
        p.a_NT = t; p.K = t * cout; p.kchunk = p.K;
        p.B = (const float*)w; p.b_ld = cout;
        p.c_OH = c0.hq; p.c_OW = c0.wq; p.c_FH = h; p.c_FW = wd; p.c_step = stride; p.c_ld = cin;
        p.ADD = addin; p.RAW = raw; p.Zin = z16only ? nullptr : zprev; p.alpha = alpha_prev; p.amod = cin; p.DZ = dzprev;
        p.ADD16 = addin16; p.Zin16 = zprev16; p.RAW16 = raw16;
        p.PA = PA; p.PB = PB; p.prow0 = 0;
        if (bn) { p.bn_mu = bn->mean; p.bn_rs = bn->rstd; p.bn_sc = bn->scale; p.bn_sh = bn->shift; p.Zx = zx; p.Zx16 = zx16; }
        p.src16 = src16 ? 1 : 0; p.DZ16 = dzprev16;
        if (!set_bytes(&p, (size_t)n * pho.out * pwo.out * cout, (size_t)ksize * ksize * cin * cout, src16 ? 2 : 4)) return FTE_EINVAL;
        hipError_t e = igemm_launch(p, AL_MK, BL_NK, EPI_DGRAD, tile, 1, (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    } else {
    long prow = 0;
    for (int i = 0; i < nc; ++i) {
        const DgradClass& c = cls[i];
        IgemmParams p;
        zero_params(&p);
        p.M = n * c.hq * c.wq; p.N = cin; p.K = c.ntap * cout; p.kchunk = p.K;
        p.A = (const float*)dz; p.a_OH = c.hq; p.a_OW = c.wq; p.a_IH = pho.out; p.a_IW = pwo.out; p.a_stride = 1;
        p.a_ld = cout; p.a_KC = cout; p.a_NT = c.ntap;
        for (int t = 0; t < c.ntap; ++t) {
            p.a_dh[t] = c.dh[t]; p.a_dw[t] = c.dw[t];
            p.b_tapoff[t] = c.wt[t] * cin * cout;
        }
        p.B = (const float*)w; p.b_ld = cout;
        p.src16 = src16 ? 1 : 0; p.DZ16 = dzprev16;
        if (stride == 1) { p.c_OH = 0; }
        else { p.c_OH = c.hq; p.c_OW = c.wq; p.c_FH = h; p.c_FW = wd; p.c_step = stride; p.c_ph = c.ph; p.c_pw = c.pw; }
        p.c_ld = cin;
        p.ADD = addin; p.RAW = raw; p.Zin = z16only ? nullptr : zprev; p.alpha = alpha_prev; p.amod = cin; p.DZ = dzprev;
        p.ADD16 = addin16; p.Zin16 = zprev16; p.RAW16 = raw16;
        p.PA = PA; p.PB = PB;
        if (bn) { p.bn_mu = bn->mean; p.bn_rs = bn->rstd; p.bn_sc = bn->scale; p.bn_sh = bn->shift; p.Zx = zx; p.Zx16 = zx16; }
        if (!set_bytes(&p, (size_t)n * pho.out * pwo.out * cout, (size_t)ksize * ksize * cin * cout, src16 ? 2 : 4)) return FTE_EINVAL;
        hipError_t e = launch_rows(p, c.rp, AL_MK, BL_NK, EPI_DGRAD, prow, pwbuf, (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
        prow += c.mtiles;
    }
    }
    if (bn)            // PA = sum g * xhat, PB = sum g per partial row -> dgamma, dbeta, the coefficients of dz = A g + B z + C0
        return rc(l_bn_bwd_finalize(PB, PA, cin, (int)rows, (long)n * h * wd, cin, bn->gamma, bn->mean, bn->rstd, bn->dgamma, bn->dbeta,
                                    bn->coef, (hipStream_t)stream));
    if (want_part) {
        if (dalpha_prev && dbias_prev) {
            hipError_t e = k_reduce_rows2(PA, dalpha_prev, PB, dbias_prev, nullptr, 1, rows, cin, 1, 1.f, scratch, (hipStream_t)stream);
            if (e != hipSuccess) return (int)e;
        } else {
            if (dalpha_prev) { hipError_t e = k_reduce_rows(PA, dalpha_prev, nullptr, 1, rows, cin, 1, 1.f, scratch, (hipStream_t)stream); if (e != hipSuccess) return (int)e; }
            if (dbias_prev) { hipError_t e = k_reduce_rows(PB, dbias_prev, nullptr, 1, rows, cin, 1, 1.f, scratch, (hipStream_t)stream); if (e != hipSuccess) return (int)e; }
        }
    }
    return FTE_OK;
}

int fte_conv2d_dgrad(const float* dz, const float* w, const float* addin, const float* zprev,
                     const float* alpha_prev, float* raw, float* dzprev, float* dalpha_prev, float* dbias_prev,
                     int n, int h, int wd, int cin, int cout, int ksize, int stride, void* ws, size_t ws_bytes, void* stream) {
    return conv2d_dgrad_impl(dz, w, false, addin, zprev, alpha_prev, raw, dzprev, nullptr, dalpha_prev, dbias_prev,
                             n, h, wd, cin, cout, ksize, stride, ws, ws_bytes, stream);
}
int fte_conv2d_dgrad16(const uint16_t* dz16, const uint16_t* w16, const float* addin, const float* zprev,
                       const float* alpha_prev, float* raw, float* dzprev, uint16_t* dzprev16, float* dalpha_prev, float* dbias_prev,
                       int n, int h, int wd, int cin, int cout, int ksize, int stride, void* ws, size_t ws_bytes, void* stream) {
    Plan16 guard;
    return conv2d_dgrad_impl(dz16, w16, true, addin, zprev, alpha_prev, raw, dzprev, dzprev16, dalpha_prev, dbias_prev,
                             n, h, wd, cin, cout, ksize, stride, ws, ws_bytes, stream);
}

int fte_conv2d_dgrad_s16(const uint16_t* dz16, const uint16_t* w16, const uint16_t* addin16, const uint16_t* zprev16,
                         const float* alpha_prev, uint16_t* raw16, uint16_t* dzprev16, float* dalpha_prev, float* dbias_prev,
                         int n, int h, int wd, int cin, int cout, int ksize, int stride, void* ws, size_t ws_bytes, void* stream) {
    if (!dzprev16) return FTE_EINVAL;
    Plan16 guard;
    PlanS16 storage(true);
    return conv2d_dgrad_impl(dz16, w16, true, nullptr, nullptr, alpha_prev, nullptr, nullptr, dzprev16, dalpha_prev, dbias_prev,
                             n, h, wd, cin, cout, ksize, stride, ws, ws_bytes, stream, addin16, zprev16, raw16);
}

size_t fte_conv3x3_dgrad_ws_bytes(int n, int h, int wd, int cin, int cout, int stride) {
    return fte_conv2d_dgrad_ws_bytes(n, h, wd, cin, cout, 3, stride);
}
int fte_conv3x3_dgrad(const float* dz, const float* w, const float* addin, const float* zprev,
                      const float* alpha_prev, float* raw, float* dzprev, float* dalpha_prev, float* dbias_prev,
                      int n, int h, int wd, int cin, int cout, int stride, void* ws, size_t ws_bytes, void* stream) {
    return fte_conv2d_dgrad(dz, w, addin, zprev, alpha_prev, raw, dzprev, dalpha_prev, dbias_prev,
                            n, h, wd, cin, cout, 3, stride, ws, ws_bytes, stream);
}

// ------------------------------------------------------------------------------------------------
namespace {
inline bool wgrad_split_major() {
    static const bool on = !getenv("FTE_WGRAD_SPLIT_MAJOR") || atoi(getenv("FTE_WGRAD_SPLIT_MAJOR")) != 0;      // tuning hook: 0 = 2-D grid
    return on;
}
void wgrad_plan(int n, int h, int wd, int cin, int cout, int ksize, int stride, int* tile, int* splits, int* kchunk, int* K) {
    const Pads ph = same_pads(h, ksize, stride), pw = same_pads(wd, ksize, stride);
    *K = n * ph.out * pw.out;
    const long M = (long)ksize * ksize * cin;
    static const int wg_tile = getenv("FTE_WGRAD_TILE") ? atoi(getenv("FTE_WGRAD_TILE")) : -1;      // tuning hook
    // N = 64 layers: M = 9 * 64 = 576 is 4.5 tiles of 128 rows (a tenth of the MFMAs multiply zero padding: the stage-1
    // filter gradient ran at 68 % of peak with its MFMA pipe 76 % busy) but exactly 3 tiles of 192
    const int narrow = (M % 192 == 0 && M % 128 != 0) ? TILE_192x64 : TILE_128x64;
    // a filter gradient of at most 256 x 256 (the 1x1 convs of the ShuffleNet stages) is 1-4 tiles of 128 x 128: every block is
    // then one of ~400 K ranges of 256 pixels -- 64 x 64 tiles give 4x the blocks over K ranges 4x as long for the same
    // slab bytes (14x14x128->128 at batch 512: 65 -> 51 us, 7x7x256->256: 52 -> 44 us with the slab reduction)
    const bool tiny = ksize == 1 && M * cout <= 256L * 256;
    *tile = wg_tile >= 0 ? wg_tile : (tiny ? TILE_64x64 : (cout % 128 == 0) ? TILE_128x128 : narrow);
    // Block placement: split-major (wgrad_split_major(), ON by default since round 2; FTE_WGRAD_SPLIT_MAJOR=0 gives the 2-D grid
    // back): the tiles of one pixel range sit 8 block ids apart -- same XCD, same L2 -- so x / dz leave HBM once per pixel range
    // instead of once per tile (stage-1 symbol 2844 -> 904 MB per launch, 128x128 symbol 1923 -> 523 MB) at unchanged kernel
    // time.  `prefer8` (split counts that are multiples of 8) stays off in plan_splits: the kernels are MFMA-bound and the
    // rounder split count cost more than the L2 locality gave (18.4 -> 18.9 ms per step when first measured).
    plan_splits(tiles_of(*tile, M, cout), *K, splits, kchunk, false);
    // Small per-GPU shards: when filling the chip with big tiles leaves every block fewer than 1024 pixels of K (32 K-steps), take
    // 64 x 64 tiles instead -- 4x the tiles, a quarter of the splits, K ranges 4x as long and a quarter of the slab traffic.
    // SphereNet step, one GPU: batch 32 +5.9 %, 64 +3.2 %, 128 +1.3 %, 256 +-0, 512 -1.1 % (where this rule does not fire).
    if (wg_tile < 0 && *tile != TILE_64x64 && *splits > 1 && *kchunk < 1024) {
        *tile = TILE_64x64;
        plan_splits(tiles_of(*tile, M, cout), *K, splits, kchunk, false);
    }
}
}  // namespace

size_t fte_conv2d_wgrad_ws_bytes(int n, int h, int wd, int cin, int cout, int ksize, int stride) {
    if (n <= 0 || cin <= 0 || cin % 4 || cout <= 0 || cout % 64) return 0;
    int tile, splits, kchunk, K;
    wgrad_plan(n, h, wd, cin, cout, ksize, stride, &tile, &splits, &kchunk, &K);
    size_t need = splits > 1 ? (size_t)splits * ksize * ksize * cin * cout * sizeof(float) : 0;
    if (ksize == 3 && stride == 1) {                 // the resident kernel's slot-range slabs (bf16 sources; wgrad16.hip)
        Wgrad16Params g;
        int cfg = 0;
        if (wgrad16_plan(n, h, wd, cin, cout, &g, &cfg) && g.S > 1) {
            const size_t gneed = (size_t)g.S * g.slab * sizeof(float);
            if (gneed > need) need = gneed;
        }
    }
    if (ksize == 1 && stride == 1) {                 // ... and the pointwise kernel's pixel-range slabs
        Wgrad16Params g;
        int cfg = 0;
        if (wgrad16p_plan(n, h, wd, cin, cout, &g, &cfg) && g.S > 1) {
            const size_t gneed = (size_t)g.S * g.slab * sizeof(float);
            if (gneed > need) need = gneed;
        }
    }
    if (wino_sized(n, h, wd, cin, cout, ksize, stride, 2)) {
        const size_t wn = wino_ws(n, h, wd, cin, cout, 2, 0).total;
        if (wn > need) need = wn;
    }
    return need + SCRATCH_BYTES;
}
size_t fte_conv3x3_wgrad_ws_bytes(int n, int h, int wd, int cin, int cout, int stride) {
    return fte_conv2d_wgrad_ws_bytes(n, h, wd, cin, cout, 3, stride);
}

static int conv2d_wgrad_impl(const void* x, const void* dz, bool src16, float* dw, int n, int h, int wd, int cin, int cout, int ksize, int stride,
                             void* ws, size_t ws_bytes, void* stream, const float* vpack = nullptr) {
    if (src16 && (cin % 8 || cout % 8)) return FTE_EINVAL;
    if (!x || !dz || !dw || n <= 0 || cin % 4 || cout % 64 || (stride != 1 && stride != 2) || (ksize != 1 && ksize != 3)) return FTE_EINVAL;
    const Pads ph = same_pads(h, ksize, stride), pw = same_pads(wd, ksize, stride);
    {       // tensors of 2 GiB or more (32-bit buffer offsets): an argument error, reported before any workspace question
        const size_t lim = (size_t)1 << 31, es = src16 ? 2 : 4;
        if ((size_t)n * h * wd * cin * es >= lim || (size_t)n * ph.out * pw.out * cout * es >= lim) return FTE_EINVAL;
    }
    int tile, splits, kchunk, K;
    wgrad_plan(n, h, wd, cin, cout, ksize, stride, &tile, &splits, &kchunk, &K);
    if (!src16 && wino_wanted(n, h, wd, cin, cout, ksize, stride, 2)) {
        // Winograd F(3x3, 2x2): V = B^T d B of x, 16 products over the tiles with U' = G' e G'^T of dz formed in the kernel, A'^T . A' of the sums
        WinoWs wl = wino_ws(n, h, wd, cin, cout, 2, 0);
        if (vpack) {                               // V = B^T d B of x kept by the forward pass (fte_conv3x3_fwd_keep): ws holds U' and the slabs
            const size_t vsz = wl.u_off - wl.v_off;
            wl.u_off -= vsz; wl.slab_off -= vsz; wl.total -= vsz;
        }
        if (ws && ws_bytes >= wl.total) {
            if (!aligned16({x, dz, dw, ws, vpack})) return FTE_EINVAL;
            const float* V = vpack ? vpack : (float*)((char*)ws + wl.v_off);
            float* slabs = (float*)((char*)ws + wl.slab_off);
            hipError_t e = hipSuccess;
            if (!vpack) e = wino_transform_tiles((const float*)x, (float*)((char*)ws + wl.v_off), n, h, wd, cin, 0, (hipStream_t)stream);
            if (e != hipSuccess) return (int)e;
            return rc(wino_wgrad(V, (const float*)dz, slabs, dw, wino_geom(n, h, wd), cin, cout, (hipStream_t)stream));
        }
    }
    if (vpack) return FTE_EWORKSPACE;              // a kept V was handed over and the Winograd path did not run: never silently
    if (src16 && ksize == 3 && stride == 1) {       // the resident kernel (wgrad16.hip): all nine taps per block, LDS-DMA, slot-range splits
        Wgrad16Params g;
        int cfg = 0;
        if (wgrad16_plan(n, h, wd, cin, cout, &g, &cfg)) {
            const size_t gneed = g.S > 1 ? (size_t)g.S * g.slab * sizeof(float) : 0;
            const size_t xb = (size_t)n * h * wd * cin * 2, db = (size_t)n * h * wd * cout * 2;
            if ((!gneed || (ws && ws_bytes >= gneed)) && xb < (1ull << 31) && db < (1ull << 31)) {
                g.x = (const unsigned short*)x; g.dz = (const unsigned short*)dz;
                g.out = g.S > 1 ? (float*)ws : dw;
                g.x_bytes = (unsigned)xb; g.dz_bytes = (unsigned)db;
                for (int r = 0; r < 3; ++r)
                    for (int s2 = 0; s2 < 3; ++s2) { g.dh[r * 3 + s2] = r - 1; g.dw[r * 3 + s2] = s2 - 1; }
                const int sig[5] = {AL_KM, BL_KN, EPI_FWD, 7, g.S};              // tile id 7: the resident kernel (bench.py TILES)
                const double rows = 9.0 * cin;
                const int h = igemm_prof_begin(sig, (int)rows, cout, K, 2.0 * rows * cout * (double)K, (double)xb + (double)db + (double)g.S * g.slab * 4.0,
                                               (hipStream_t)stream);
                hipError_t e = wgrad16_launch(g, cfg, (hipStream_t)stream);
                igemm_prof_end(h, cfg == 0 ? "wgrad16_kernel<32,256,3,128>" : cfg == 1 ? "wgrad16_kernel<64,128,3,128>" : "wgrad16_kernel<64,64,3,192>", (hipStream_t)stream);
                if (e != hipSuccess) return (int)e;
                if (g.S > 1) return rc(k_reduce_rows((const float*)ws, dw, nullptr, 1, g.S, g.slab, 1, 1.f, nullptr, (hipStream_t)stream));
                return FTE_OK;
            }
        }
    }
    if (src16 && ksize == 1 && stride == 1) {       // 1x1 convs: the pointwise resident kernel (wgrad16.hip: wgrad16p_kernel)
        Wgrad16Params g;
        int cfg = 0;
        if (wgrad16p_plan(n, h, wd, cin, cout, &g, &cfg)) {
            const size_t gneed = g.S > 1 ? (size_t)g.S * g.slab * sizeof(float) : 0;
            const size_t xb = (size_t)n * h * wd * cin * 2, db = (size_t)n * h * wd * cout * 2;
            if ((!gneed || (ws && ws_bytes >= gneed)) && xb < (1ull << 31) && db < (1ull << 31)) {
                g.x = (const unsigned short*)x; g.dz = (const unsigned short*)dz;
                g.out = g.S > 1 ? (float*)ws : dw;
                g.x_bytes = (unsigned)xb; g.dz_bytes = (unsigned)db;
                const int sig[5] = {AL_KM, BL_KN, EPI_FWD, 7, g.S};
                const int hr = igemm_prof_begin(sig, cin, cout, K, 2.0 * cin * cout * (double)K, (double)xb + (double)db + (double)g.S * g.slab * 4.0,
                                                (hipStream_t)stream);
                hipError_t e = wgrad16p_launch(g, cfg, (hipStream_t)stream);
                static const char* const psym[7] = {"wgrad16p_kernel<128,256,2,4>", "wgrad16p_kernel<256,128,4,2>", "wgrad16p_kernel<128,128,2,4>", "wgrad16p_kernel<64,256,1,8>",
                                                   "wgrad16p_kernel<256,64,8,1>", "wgrad16p_kernel<64,128,2,4>", "wgrad16p_kernel<128,64,4,2>"};
                igemm_prof_end(hr, psym[cfg], (hipStream_t)stream);
                if (e != hipSuccess) return (int)e;
                if (g.S > 1) return rc(k_reduce_rows((const float*)ws, dw, nullptr, 1, g.S, g.slab, 1, 1.f, nullptr, (hipStream_t)stream));
                return FTE_OK;
            }
        }
    }
    const size_t need = splits > 1 ? (size_t)splits * ksize * ksize * cin * cout * sizeof(float) : 0;
    if (need && (!ws || ws_bytes < need)) return FTE_EWORKSPACE;
    IgemmParams p;
    zero_params(&p);
    p.M = ksize * ksize * cin; p.N = cout; p.K = K; p.kchunk = kchunk;
    p.A = (const float*)x; p.a_OH = ph.out; p.a_OW = pw.out; p.a_IH = h; p.a_IW = wd; p.a_stride = stride;
    p.a_ld = cin; p.a_KC = cin; p.a_NT = ksize * ksize;
    for (int r = 0; r < ksize; ++r)
        for (int s = 0; s < ksize; ++s) { p.a_dh[r * ksize + s] = r - ph.before; p.a_dw[r * ksize + s] = s - pw.before; }
    p.B = (const float*)dz; p.b_ld = cout;
    p.c_ld = cout;
    p.slab = (long)p.M * p.N;
    p.Y = splits > 1 ? (float*)ws : dw;
    p.split_major = (wgrad_split_major() && splits > 1) ? splits : 0;
    p.src16 = src16 ? 1 : 0;
    if (!set_bytes(&p, (size_t)n * h * wd * cin, (size_t)K * cout, src16 ? 2 : 4)) return FTE_EINVAL;
    hipError_t e = igemm_launch(p, AL_KM, BL_KN, EPI_FWD, tile, splits, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    if (splits > 1) return rc(k_reduce_rows((const float*)ws, dw, nullptr, 1, splits, p.slab, 1, 1.f, nullptr, (hipStream_t)stream));
    return FTE_OK;
}
int fte_conv2d_wgrad(const float* x, const float* dz, float* dw, int n, int h, int wd, int cin, int cout, int ksize, int stride,
                     void* ws, size_t ws_bytes, void* stream) {
    return conv2d_wgrad_impl(x, dz, false, dw, n, h, wd, cin, cout, ksize, stride, ws, ws_bytes, stream);
}
int fte_conv2d_wgrad16(const uint16_t* x16, const uint16_t* dz16, float* dw, int n, int h, int wd, int cin, int cout, int ksize, int stride,
                       void* ws, size_t ws_bytes, void* stream) {
    return conv2d_wgrad_impl(x16, dz16, true, dw, n, h, wd, cin, cout, ksize, stride, ws, ws_bytes, stream);
}
int fte_conv3x3_wgrad(const float* x, const float* dz, float* dw, int n, int h, int wd, int cin, int cout, int stride,
                      void* ws, size_t ws_bytes, void* stream) {
    return fte_conv2d_wgrad(x, dz, dw, n, h, wd, cin, cout, 3, stride, ws, ws_bytes, stream);
}
int fte_conv3x3_wgrad_kept(const float* x, const float* dz, float* dw, int n, int h, int wd, int cin, int cout, int stride,
                           const float* vpack, void* ws, size_t ws_bytes, void* stream) {
    if (vpack && ((uintptr_t)vpack & 15)) return FTE_EINVAL;
    return conv2d_wgrad_impl(x, dz, false, dw, n, h, wd, cin, cout, 3, stride, ws, ws_bytes, stream, vpack);
}

// ------------------------------------------------------------------------------------------------
int fte_conv3x3_first_fwd(const float* x, const float* w, const float* bias, const float* alpha, float* z, float* y,
                          int n, int h, int wd, int cin, int cout, int stride, void* stream) {
    if (!x || !w || !y || (cout != 64 && cout != 32) || (cin != 1 && cin != 3)) return FTE_EINVAL;
    const Pads ph = same_pads(h, 3, stride), pw = same_pads(wd, 3, stride);
    return rc(k_conv_first_fwd(x, w, bias, alpha, z, y, nullptr, nullptr, n, h, wd, cin, cout, ph.out, pw.out, stride, ph.before, pw.before, (hipStream_t)stream));
}
int fte_conv3x3_first_fwd_s16(const float* x, const float* w, const float* bias, const float* alpha, uint16_t* z16, uint16_t* y16,
                              int n, int h, int wd, int cin, int cout, int stride, void* stream) {
    if (!x || !w || !y16 || (cout != 64 && cout != 32) || (cin != 1 && cin != 3)) return FTE_EINVAL;
    const Pads ph = same_pads(h, 3, stride), pw = same_pads(wd, 3, stride);
    return rc(k_conv_first_fwd(x, w, bias, alpha, nullptr, nullptr, z16, y16, n, h, wd, cin, cout, ph.out, pw.out, stride, ph.before, pw.before, (hipStream_t)stream));
}

size_t fte_conv3x3_first_wgrad_ws_bytes(int n, int h, int wd, int cin, int cout, int stride) {
    const Pads ph = same_pads(h, 3, stride), pw = same_pads(wd, 3, stride);
    return align_up((size_t)k_conv_first_wgrad_blocks((long)n * ph.out * pw.out) * 9 * cin * cout * sizeof(float)) + SCRATCH_BYTES;
}

static int conv_first_wgrad_impl(const float* x, const float* dz, const uint16_t* dz16, float* dw, int n, int h, int wd, int cin, int cout, int stride,
                                 void* ws, size_t ws_bytes, void* stream) {
    if (!x || (!dz && !dz16) || !dw || (cout != 64 && cout != 32) || (cin != 1 && cin != 3)) return FTE_EINVAL;
    const Pads ph = same_pads(h, 3, stride), pw = same_pads(wd, 3, stride);
    const int blocks = k_conv_first_wgrad_blocks((long)n * ph.out * pw.out);
    const size_t need = align_up((size_t)blocks * 9 * cin * cout * sizeof(float));
    if (!ws || ws_bytes < need + SCRATCH_BYTES) return FTE_EWORKSPACE;
    hipError_t e = k_conv_first_wgrad(x, dz, dz16, (float*)ws, n, h, wd, cin, cout, ph.out, pw.out, stride, ph.before, pw.before, blocks, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    return rc(k_reduce_rows((const float*)ws, dw, nullptr, 1, blocks, 9L * cin * cout, 1, 1.f, (float*)((char*)ws + need), (hipStream_t)stream));
}
int fte_conv3x3_first_wgrad(const float* x, const float* dz, float* dw, int n, int h, int wd, int cin, int cout, int stride,
                            void* ws, size_t ws_bytes, void* stream) {
    return conv_first_wgrad_impl(x, dz, nullptr, dw, n, h, wd, cin, cout, stride, ws, ws_bytes, stream);
}
int fte_conv3x3_first_wgrad_s16(const float* x, const uint16_t* dz16, float* dw, int n, int h, int wd, int cin, int cout, int stride,
                                void* ws, size_t ws_bytes, void* stream) {
    return conv_first_wgrad_impl(x, nullptr, dz16, dw, n, h, wd, cin, cout, stride, ws, ws_bytes, stream);
}

// ------------------------------------------------------------------------------------------------
// dense
namespace {
// split-K plan of a dense product with `rows` x `cols` outputs and reduction length `red`
size_t dense_slab_bytes(int tile, long rows, long cols, int red) {
    int splits = 1, kchunk = red;
    if (tiles_of(tile, rows, cols) < 256) plan_splits(tiles_of(tile, rows, cols), red, &splits, &kchunk);
    return splits > 1 ? (size_t)splits * rows * cols * sizeof(float) : 0;
}
inline int tn_tile(long k, long n) {
    return (n % 128 == 0) ? (tiles_of(TILE_128x128, k, n) >= 384 ? TILE_128x128 : TILE_128x64) : TILE_128x64;
}
}  // namespace

size_t fte_gemm_ws_bytes(int m, int n, int k) {
    // max over nn (m x n, red k), nt (m x k, red n; or its dalpha partials), tn (k x n, red m)
    size_t need = 0, v;
    if (n % 64 == 0) { v = dense_slab_bytes(pick_tile(m, n), m, n, k); if (v > need) need = v; }
    if (k % 64 == 0) {
        v = dense_slab_bytes(pick_tile(m, k), m, k, n); if (v > need) need = v;
        v = align_up((size_t)((m + 63) / 64) * k * sizeof(float)) + SCRATCH_BYTES; if (v > need) need = v;
    }
    if (n % 64 == 0) { v = dense_slab_bytes(tn_tile(k, n), k, n, m); if (v > need) need = v; }
    return need;
}

static int gemm_nn_impl(const float* x, const float* w, const float* bias, float* y, int m, int n, int k, int act,
                        void* ws, size_t ws_bytes, void* stream) {
    if (!x || !w || !y || m <= 0 || n % 64 || k % 32 || act < 0 || act > 2) return FTE_EINVAL;
    IgemmParams p;
    zero_params(&p);
    p.M = m; p.N = n; p.K = k;
    plain_a(&p, x, k, k);
    p.B = w; p.b_ld = n; p.c_ld = n;
    if (!set_bytes(&p, (size_t)m * k, (size_t)k * n)) return FTE_EINVAL;
    const int tile = pick_tile(m, n);
    int splits = 1, kchunk = k;
    if (tiles_of(tile, m, n) < 256) plan_splits(tiles_of(tile, m, n), k, &splits, &kchunk);
    p.kchunk = kchunk;
    p.slab = (long)m * n;
    if (splits > 1) {
        if (!ws || ws_bytes < (size_t)splits * p.slab * sizeof(float)) return FTE_EWORKSPACE;
        p.Y = (float*)ws;
        hipError_t e = igemm_launch(p, AL_MK, BL_KN, EPI_FWD, tile, splits, (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
        // (the activation rides in the slab reduction's epilogue: no separate pass over y)
        return rc(k_reduce_rows((const float*)ws, y, bias, n, splits, p.slab, 1, 1.f, nullptr, (hipStream_t)stream, act));
    }
    p.Y = y; p.bias = bias;
    hipError_t e = igemm_launch(p, AL_MK, BL_KN, EPI_FWD, tile, 1, (hipStream_t)stream);
    if (e != hipSuccess || !act) return rc(e);
    return rc(l_act_fwd(y, y, (long)m * n, act - 1, (hipStream_t)stream));
}
int fte_gemm_nn(const float* x, const float* w, const float* bias, float* y, int m, int n, int k,
                void* ws, size_t ws_bytes, void* stream) {
    return gemm_nn_impl(x, w, bias, y, m, n, k, 0, ws, ws_bytes, stream);
}
int fte_gemm_nn_act(const float* x, const float* w, const float* bias, float* y, int m, int n, int k, int act,
                    void* ws, size_t ws_bytes, void* stream) {
    return gemm_nn_impl(x, w, bias, y, m, n, k, act, ws, ws_bytes, stream);
}

int fte_dense_small(const float* a, const float* w, const float* bias, const float* mask, float* out, int m, int n, int k,
                    int trans_w, int act, void* stream) {
    if (!a || !w || !out || m <= 0 || n <= 0 || n % 32 || k <= 0 || k % 128 || act < 0 || act > 2 || (trans_w & ~1)) return FTE_EINVAL;
    return rc(k_dense_small(a, w, bias, mask, out, m, n, k, trans_w != 0, act, plan_bf16(), (hipStream_t)stream));
}

int fte_gemm_nt(const float* dy, const float* w, const float* zprev, const float* alpha_prev, int amod,
                float* raw, float* dx, float* dalpha_prev, int m, int n, int k, void* ws, size_t ws_bytes, void* stream) {
    // dx[m,k] = dy[m,n] @ w[k,n]^T : GEMM with rows m, cols k, reduction n
    if (!dy || !w || !dx || m <= 0 || k % 64 || n % 32) return FTE_EINVAL;
    if (zprev && (!alpha_prev || amod <= 0 || k % amod)) return FTE_EINVAL;
    IgemmParams p;
    zero_params(&p);
    p.M = m; p.N = k; p.K = n;
    plain_a(&p, dy, n, n);
    p.B = w; p.b_ld = n; p.c_ld = k;
    if (!set_bytes(&p, (size_t)m * n, (size_t)k * n)) return FTE_EINVAL;
    const int tile = pick_tile(m, k);
    if (!zprev && !raw) {     // plain product, split-K allowed
        int splits = 1, kchunk = n;
        if (tiles_of(tile, m, k) < 256) plan_splits(tiles_of(tile, m, k), n, &splits, &kchunk);
        p.kchunk = kchunk;
        p.slab = (long)m * k;
        if (splits > 1) {
            if (!ws || ws_bytes < (size_t)splits * p.slab * sizeof(float)) return FTE_EWORKSPACE;
            p.Y = (float*)ws;
            hipError_t e = igemm_launch(p, AL_MK, BL_NK, EPI_FWD, tile, splits, (hipStream_t)stream);
            if (e != hipSuccess) return (int)e;
            return rc(k_reduce_rows((const float*)ws, dx, nullptr, 1, splits, p.slab, 1, 1.f, nullptr, (hipStream_t)stream));
        }
        p.Y = dx;
        return rc(igemm_launch(p, AL_MK, BL_NK, EPI_FWD, tile, 1, (hipStream_t)stream));
    }
    p.kchunk = n;
    int bm, bn;
    igemm_tile_dims(tile, &bm, &bn);
    const long mtiles = ((long)m + bm - 1) / bm;
    const bool want_part = zprev && dalpha_prev;
    const size_t part_bytes = align_up((size_t)mtiles * k * sizeof(float));
    if (want_part && (!ws || ws_bytes < part_bytes + SCRATCH_BYTES)) return FTE_EWORKSPACE;
    p.RAW = raw; p.Zin = zprev; p.alpha = alpha_prev; p.amod = zprev ? amod : 1; p.DZ = dx;
    p.PA = want_part ? (float*)ws : nullptr;
    hipError_t e = igemm_launch(p, AL_MK, BL_NK, EPI_DGRAD, tile, 1, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    if (want_part) return rc(k_reduce_rows((const float*)ws, dalpha_prev, nullptr, 1, mtiles, k, k / amod, 1.f,
                                           (float*)((char*)ws + part_bytes), (hipStream_t)stream));
    return FTE_OK;
}

int fte_gemm_tn(const float* x, const float* dy, float* dw, int m, int n, int k, void* ws, size_t ws_bytes, void* stream) {
    // dw[k,n] = x[m,k]^T @ dy[m,n] : GEMM with rows k, cols n, reduction m
    if (!x || !dy || !dw || m <= 0 || n % 64 || k % 4) return FTE_EINVAL;
    IgemmParams p;
    zero_params(&p);
    p.M = k; p.N = n; p.K = m;
    plain_a(&p, x, k, k);
    p.B = dy; p.b_ld = n; p.c_ld = n;
    if (!set_bytes(&p, (size_t)m * k, (size_t)m * n)) return FTE_EINVAL;
    const int tile = tn_tile(k, n);
    int splits = 1, kchunk = (m + 31) / 32 * 32;
    if (tiles_of(tile, k, n) < 256) plan_splits(tiles_of(tile, k, n), m, &splits, &kchunk);
    p.kchunk = kchunk;
    p.slab = (long)k * n;
    if (splits > 1) {
        if (!ws || ws_bytes < (size_t)splits * p.slab * sizeof(float)) return FTE_EWORKSPACE;
        p.Y = (float*)ws;
        hipError_t e = igemm_launch(p, AL_KM, BL_KN, EPI_FWD, tile, splits, (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
        return rc(k_reduce_rows((const float*)ws, dw, nullptr, 1, splits, p.slab, 1, 1.f, nullptr, (hipStream_t)stream));
    }
    p.Y = dw;
    return rc(igemm_launch(p, AL_KM, BL_KN, EPI_FWD, tile, 1, (hipStream_t)stream));
}

// ------------------------------------------------------------------------------------------------
// loss heads
int fte_softmax_ce_fwd_bwd(const float* logits, const int32_t* labels, float* loss_rows, float* dlogits,
                           int n, int c, int ld, float grad_scale, void* stream) {
    if (!logits || !labels || !loss_rows || !dlogits || n <= 0 || c <= 0 || ld < c) return FTE_EINVAL;
    return rc(k_softmax_ce(logits, labels, loss_rows, dlogits, n, c, ld, grad_scale, (hipStream_t)stream));
}
int fte_focal_loss_fwd_bwd(const float* logits, const int32_t* labels, float* loss_rows, float* dlogits,
                           int n, int c, int ld, float gamma, float alpha, float grad_scale, void* stream) {
    if (!logits || !labels || !loss_rows || !dlogits || n <= 0 || c <= 0 || ld < c || alpha < 1.f) return FTE_EINVAL;
    return rc(k_focal_loss(logits, labels, loss_rows, dlogits, n, c, ld, gamma, alpha, grad_scale, (hipStream_t)stream));
}
int fte_asoftmax_fwd_bwd(const float* s, const float* xn, const float* wn, const int32_t* labels, float lambda,
                         float* f, float* loss_rows, float* G, float* rowcoef, int n, int c, int ld,
                         float grad_scale, void* stream) {
    if (!s || !xn || !wn || !labels || !loss_rows || !G || !rowcoef || n <= 0 || c <= 0 || ld < c) return FTE_EINVAL;
    return rc(k_asoftmax(s, xn, wn, labels, lambda, f, loss_rows, G, rowcoef, n, c, ld, grad_scale, (hipStream_t)stream));
}
int fte_asoftmax_colcoef(const float* G, const float* s, const float* wn, float* colcoef, int n, int c, int ld, void* stream) {
    if (!G || !s || !wn || !colcoef) return FTE_EINVAL;
    return rc(k_asoftmax_colcoef(G, s, wn, colcoef, n, c, ld, (hipStream_t)stream));
}
int fte_row_norms(const float* a, float* out, int rows, int cols, int ld, void* stream) {
    if (!a || !out || rows <= 0) return FTE_EINVAL;
    return rc(k_row_norms(a, out, rows, cols, ld, (hipStream_t)stream));
}
int fte_col_norms(const float* a, float* out, int rows, int cols, int ld, void* stream) {
    if (!a || !out || rows <= 0) return FTE_EINVAL;
    return rc(k_col_norms(a, out, rows, cols, ld, (hipStream_t)stream));
}
int fte_flip_width(const float* x, float* y, int n, int h, int wd, int c, void* stream) {
    if (!x || !y || x == y || n <= 0 || h <= 0 || wd <= 0 || c <= 0) return FTE_EINVAL;
    if ((c % 4 == 0) && (((uintptr_t)x | (uintptr_t)y) & 15)) return FTE_EINVAL;
    return rc(k_flip_w(x, y, (long)n * h, wd, c, (hipStream_t)stream));
}
int fte_axpby(float a, const float* x, float b, const float* y, float* out, long n, void* stream) {
    if (!x || !y || !out || n <= 0) return FTE_EINVAL;
    return rc(k_axpby(a, x, b, y, out, n, (hipStream_t)stream));
}
int fte_add_scaled_rows_cols(float* a, const float* b, const float* rcf, const float* cc, int rows, int cols, int ld, void* stream) {
    if (!a || !b || rows <= 0) return FTE_EINVAL;
    return rc(k_add_scaled(a, b, rcf, cc, rows, cols, ld, (hipStream_t)stream));
}
int fte_center_loss_fwd_bwd_update(const float* feat, const int32_t* labels, float* centers, float* loss_rows, float* dfeat,
                                   int n, int d, int num_classes, float alpha, float grad_scale, void* ws, size_t ws_bytes, void* stream) {
    if (!feat || !labels || !centers || !loss_rows || !dfeat || n <= 0 || d <= 0 || num_classes <= 0) return FTE_EINVAL;
    if (!ws || ws_bytes < (size_t)n * d * sizeof(float)) return FTE_EWORKSPACE;
    return rc(k_center_loss(feat, labels, centers, loss_rows, dfeat, n, d, num_classes, alpha, grad_scale, (float*)ws, (hipStream_t)stream));
}
int fte_center_scatter_update(const float* diff, const int32_t* labels, float* centers, int n, int d, int num_classes, float alpha, void* stream) {
    if (!diff || !labels || !centers || n <= 0 || d <= 0 || num_classes <= 0) return FTE_EINVAL;
    return rc(k_center_update(diff, labels, centers, n, d, num_classes, alpha, (hipStream_t)stream));
}
int fte_batch_hard_triplet_fwd_bwd(const float* feat, const int32_t* labels, float margin, int soft_margin, float loss_weight,
                                   float* loss_rows, float* dfeat, int n, int d, void* ws, size_t ws_bytes, void* stream) {
    if (!feat || !labels || !loss_rows || !dfeat || n <= 0) return FTE_EINVAL;
    if (!ws || ws_bytes < (size_t)2 * n * n * sizeof(float)) return FTE_EWORKSPACE;
    return rc(k_triplet(feat, labels, margin, soft_margin != 0, loss_weight, loss_rows, dfeat, n, d, (float*)ws, (hipStream_t)stream));
}

// ------------------------------------------------------------------------------------------------
// reductions / optimizers
int fte_reduce_rows(const float* in, float* out, const float* bias, int bmod, long rows, long cols, int fold, float scale, void* stream) {
    if (!in || !out || rows <= 0 || cols <= 0 || fold <= 0 || cols % fold) return FTE_EINVAL;
    return rc(k_reduce_rows(in, out, bias, bmod > 0 ? bmod : 1, rows, cols, fold, scale, nullptr, (hipStream_t)stream));
}
int fte_sumsq(const float* a, long n, float scale, float* out, void* ws, size_t ws_bytes, void* stream) {
    if (!a || !out || ((uintptr_t)a & 15)) return FTE_EINVAL;
    if (!ws || ws_bytes < 1024 * sizeof(float)) return FTE_EWORKSPACE;
    return rc(k_sum(a, n, scale, out, (float*)ws, true, (hipStream_t)stream));
}
int fte_sum(const float* a, long n, float scale, float* out, void* ws, size_t ws_bytes, void* stream) {
    if (!a || !out || ((uintptr_t)a & 15)) return FTE_EINVAL;
    if (!ws || ws_bytes < 1024 * sizeof(float)) return FTE_EWORKSPACE;
    return rc(k_sum(a, n, scale, out, (float*)ws, false, (hipStream_t)stream));
}
int fte_momentum_update(float* w, float* acc, const float* g, long n, float lr, float mom, float wd, float gscale, void* stream) {
    if (!w || !acc || !g || n <= 0 || (((uintptr_t)w | (uintptr_t)acc | (uintptr_t)g) & 15)) return FTE_EINVAL;
    return rc(k_momentum(w, acc, g, n, lr, mom, wd, gscale, (hipStream_t)stream));
}
int fte_adam_update(float* w, float* m, float* v, const float* g, long n, float lr, float b1, float b2, float eps,
                    float wd, float gscale, int t, void* stream) {
    if (!w || !m || !v || !g || n <= 0 || t < 1) return FTE_EINVAL;
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t));
    return rc(k_adam(w, m, v, g, n, (float)lr_t, b1, b2, eps, wd, gscale, (hipStream_t)stream));
}

// ------------------------------------------------------------------------------------------------
// layers of the BN / pooling nets
// split partials (3 per channel: n, mean, M2) + the group partials of the in-launch finalize (one per 16 splits) + the backward coefficients
size_t fte_bn_ws_bytes(int c) { return ((size_t)BN_MAX_SPLITS * 3 * c + (size_t)(BN_MAX_SPLITS / 16) * 3 * c + 3 * (size_t)c) * sizeof(float); }

int fte_bn_train_fwd(const float* z, const float* gamma, const float* beta, const float* res, float* y,
                     float* mean, float* rstd, float* scale, float* shift, float* moving_mean, float* moving_var,
                     long rows, int c, float eps, float decay, int relu, void* ws, size_t ws_bytes, void* stream) {
    if (!z || !gamma || !beta || !y || !mean || !rstd || !scale || !shift || rows <= 0 || c % 4) return FTE_EINVAL;
    if (!ws || ws_bytes < fte_bn_ws_bytes(c)) return FTE_EWORKSPACE;
    hipError_t e = l_bn_train_stats(z, gamma, beta, rows, c, eps, decay, mean, rstd, scale, shift, moving_mean, moving_var,
                                    (float*)ws, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    return rc(l_bn_apply(z, scale, shift, res, y, rows, c, relu, (hipStream_t)stream));
}
int fte_bn_infer_fwd(const float* z, const float* gamma, const float* beta, const float* moving_mean, const float* moving_var,
                     const float* res, float* y, float* scale, float* shift, long rows, int c, float eps, int relu, void* stream) {
    if (!z || !gamma || !beta || !moving_mean || !moving_var || !y || !scale || !shift || rows <= 0 || c % 4) return FTE_EINVAL;
    hipError_t e = l_bn_infer_coef(gamma, beta, moving_mean, moving_var, eps, c, scale, shift, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    return rc(l_bn_apply(z, scale, shift, res, y, rows, c, relu, (hipStream_t)stream));
}
int fte_bn_train_bwd(const float* dy, const float* ymask, const float* z, const float* gamma, const float* mean,
                     const float* rstd, float* dz, float* dgamma, float* dbeta, long rows, int c,
                     void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !z || !gamma || !mean || !rstd || !dz || !dgamma || !dbeta || rows <= 0 || c % 4) return FTE_EINVAL;
    if (!ws || ws_bytes < fte_bn_ws_bytes(c)) return FTE_EWORKSPACE;
    return rc(l_bn_bwd(dy, ymask, z, gamma, mean, rstd, nullptr, nullptr, nullptr, dz, dgamma, dbeta, rows, c, (float*)ws, (hipStream_t)stream));
}
int fte_bn_train_bwd_res(const float* dy, const float* y, const float* z, const float* gamma, const float* mean, const float* rstd,
                         float* g_out, float* dz, float* dgamma, float* dbeta, long rows, int c,
                         void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !y || !z || !gamma || !mean || !rstd || !g_out || !dz || !dgamma || !dbeta || rows <= 0 || c % 4) return FTE_EINVAL;
    if (!ws || ws_bytes < fte_bn_ws_bytes(c)) return FTE_EWORKSPACE;
    return rc(l_bn_bwd(dy, y, z, gamma, mean, rstd, nullptr, nullptr, g_out, dz, dgamma, dbeta, rows, c, (float*)ws, (hipStream_t)stream));
}
int fte_bn_train_bwd_zmask(const float* dy, const float* z, const float* gamma, const float* mean, const float* rstd,
                           const float* scale, const float* shift, float* dz, float* dgamma, float* dbeta, long rows, int c,
                           void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !z || !gamma || !mean || !rstd || !scale || !shift || !dz || !dgamma || !dbeta || rows <= 0 || c % 4) return FTE_EINVAL;
    if (!ws || ws_bytes < fte_bn_ws_bytes(c)) return FTE_EWORKSPACE;
    return rc(l_bn_bwd(dy, nullptr, z, gamma, mean, rstd, scale, shift, nullptr, dz, dgamma, dbeta, rows, c, (float*)ws, (hipStream_t)stream));
}
int fte_bn_train_stats(const float* z, const float* gamma, const float* beta, float* mean, float* rstd, float* scale, float* shift,
                       float* moving_mean, float* moving_var, long rows, int c, float eps, float decay,
                       void* ws, size_t ws_bytes, void* stream) {
    if (!z || !gamma || !beta || !mean || !rstd || !scale || !shift || rows <= 0 || c % 4) return FTE_EINVAL;
    if (!ws || ws_bytes < fte_bn_ws_bytes(c)) return FTE_EWORKSPACE;
    return rc(l_bn_train_stats(z, gamma, beta, rows, c, eps, decay, mean, rstd, scale, shift, moving_mean, moving_var,
                               (float*)ws, (hipStream_t)stream));
}
int fte_bn_infer_coef(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var,
                      float* scale, float* shift, int c, float eps, void* stream) {
    if (!gamma || !beta || !moving_mean || !moving_var || !scale || !shift || c <= 0) return FTE_EINVAL;
    return rc(l_bn_infer_coef(gamma, beta, moving_mean, moving_var, eps, c, scale, shift, (hipStream_t)stream));
}
int fte_relu_bwd(const float* dy, const float* y, float* g, long n, void* stream) {
    if (!dy || !y || !g || n <= 0 || n % 4) return FTE_EINVAL;
    return rc(l_relu_bwd(dy, y, g, n, (hipStream_t)stream));
}
int fte_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int n, int h, int wd, int c, void* stream) {
    if (!x || !y || !idx || n <= 0 || c % 4) return FTE_EINVAL;
    const Pads ph = same_pads(h, 3, 2), pw = same_pads(wd, 3, 2);
    return rc(l_maxpool_fwd(x, y, idx, n, h, wd, c, ph.out, pw.out, ph.before, pw.before, (hipStream_t)stream));
}
int fte_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, float* dx, int n, int h, int wd, int c, void* stream) {
    if (!dy || !idx || !dx || n <= 0 || c % 4) return FTE_EINVAL;
    const Pads ph = same_pads(h, 3, 2), pw = same_pads(wd, 3, 2);
    return rc(l_maxpool_bwd(dy, idx, dx, n, h, wd, c, ph.out, pw.out, ph.before, pw.before, (hipStream_t)stream));
}
int fte_gap_fwd(const float* x, float* y, int n, int hw, int c, void* stream) {
    if (!x || !y || n <= 0 || hw <= 0 || c <= 0 || c % 4) return FTE_EINVAL;
    return rc(l_gap_fwd(x, y, n, hw, c, (hipStream_t)stream));
}
int fte_gap_bwd(const float* dy, float* dx, int n, int hw, int c, void* stream) {
    if (!dy || !dx || n <= 0 || hw <= 0) return FTE_EINVAL;
    return rc(l_gap_bwd(dy, dx, n, hw, c, (hipStream_t)stream));
}
// ---- bf16 STORAGE twins of the BN-net layers (fte.h): flags bit 0 (FTE_S16_Z) = z / dz are bf16, bit 1 (FTE_S16_A) = y / shortcut /
// dy / masked gradient are bf16; channel counts of the float4 layouts only (c % 4 == 0, c >= 32)
static inline const float* f32p(const void* p) { return reinterpret_cast<const float*>(p); }
static inline float* f32p(void* p) { return reinterpret_cast<float*>(p); }
int fte_bn_train_fwd_s16(const void* z, const float* gamma, const float* beta, const void* res, void* y,
                         float* mean, float* rstd, float* scale, float* shift, float* moving_mean, float* moving_var,
                         long rows, int c, float eps, float decay, int relu, int flags, void* ws, size_t ws_bytes, void* stream) {
    if (!z || !gamma || !beta || !y || !mean || !rstd || !scale || !shift || rows <= 0 || c % 4 || c < 32 || (flags & ~3)) return FTE_EINVAL;
    if (!ws || ws_bytes < fte_bn_ws_bytes(c)) return FTE_EWORKSPACE;
    hipError_t e = l_bn_train_stats(f32p(z), gamma, beta, rows, c, eps, decay, mean, rstd, scale, shift, moving_mean, moving_var,
                                    (float*)ws, (hipStream_t)stream, flags);
    if (e != hipSuccess) return (int)e;
    return rc(l_bn_apply(f32p(z), scale, shift, f32p(res), f32p(y), rows, c, relu, (hipStream_t)stream, flags));
}
int fte_bn_infer_fwd_s16(const void* z, const float* gamma, const float* beta, const float* moving_mean, const float* moving_var,
                         const void* res, void* y, float* scale, float* shift, long rows, int c, float eps, int relu, int flags, void* stream) {
    if (!z || !gamma || !beta || !moving_mean || !moving_var || !y || !scale || !shift || rows <= 0 || c % 4 || (flags & ~3)) return FTE_EINVAL;
    hipError_t e = l_bn_infer_coef(gamma, beta, moving_mean, moving_var, eps, c, scale, shift, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    return rc(l_bn_apply(f32p(z), scale, shift, f32p(res), f32p(y), rows, c, relu, (hipStream_t)stream, flags));
}
int fte_bn_train_bwd_s16(const void* dy, const void* y, const void* z, const float* gamma, const float* mean, const float* rstd,
                         const float* scale, const float* shift, void* g_out, void* dz, float* dgamma, float* dbeta,
                         long rows, int c, int flags, void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !z || !gamma || !mean || !rstd || !dz || !dgamma || !dbeta || rows <= 0 || c % 4 || c < 32 || (flags & ~3)) return FTE_EINVAL;
    if ((g_out && !y) || ((scale == nullptr) != (shift == nullptr)) || (scale && (g_out || y))) return FTE_EINVAL;
    if (!ws || ws_bytes < fte_bn_ws_bytes(c)) return FTE_EWORKSPACE;
    return rc(l_bn_bwd(f32p(dy), f32p(y), f32p(z), gamma, mean, rstd, scale, shift, f32p(g_out), f32p(dz), dgamma, dbeta, rows, c,
                       (float*)ws, (hipStream_t)stream, flags));
}
// ---- "BN fusion": conv -> BN pairs of the graph nets (fte.h) -----------------------------------------------------------------------
size_t fte_conv2d_bn_fwd_ws_bytes(int n, int h, int wd, int cin, int cout, int ksize, int stride) {
    if (n <= 0 || cin <= 0 || cout <= 0 || (stride != 1 && stride != 2)) return 0;
    const Pads ph = same_pads(h, ksize, stride), pw = same_pads(wd, ksize, stride);
    const size_t rows = ((size_t)n * ph.out * pw.out + 63) / 64 + 2;      // at most one partial row per 64 output rows
    return align_up(rows * 3 * cout * sizeof(float));
}
int fte_conv2d_bn_fwd_folds(int n, int h, int wd, int cin, int cout, int ksize, int stride, int s16) {
    Pw16Params pw;
    return s16 == 1 && ksize == 1 && stride == 1 && n > 0 && h > 0 && wd > 0 && pw16_plan((long)n * h * wd, cin, cout, PW_EPI_STATS, &pw) ? 1 : 0;
}
int fte_conv2d_bn_fwd(const void* x, const void* w, void* z, const float* gamma, const float* beta, float* mean, float* rstd,
                      float* scale, float* shift, float* moving_mean, float* moving_var, float eps, float decay,
                      const float* in_scale, const float* in_shift, void* y_side,
                      int n, int h, int wd, int cin, int cout, int ksize, int stride, int s16, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !w || !z || !gamma || !beta || !mean || !rstd || !scale || !shift || (s16 & ~1) || ((moving_mean == nullptr) != (moving_var == nullptr)))
        return FTE_EINVAL;      // (x / w here: the streaming pointwise path below does not go through conv2d_fwd_impl's checks)
    if (((in_scale == nullptr) != (in_shift == nullptr)) || (in_scale && !y_side) ||
        (in_scale && !fte_conv2d_bn_fwd_folds(n, h, wd, cin, cout, ksize, stride, s16))) return FTE_EINVAL;
    if (!ws || ws_bytes < fte_conv2d_bn_fwd_ws_bytes(n, h, wd, cin, cout, ksize, stride)) return FTE_EWORKSPACE;
    PlanBn bnplan;
    int rows = 0, e;
    Pw16Params pw;
    memset(&pw, 0, sizeof(pw));
    if (s16 && ksize == 1 && stride == 1 && pw16_plan((long)n * h * wd, cin, cout, PW_EPI_STATS, &pw)) {
        // the streaming pointwise kernel (pw16.hip): filter slice resident in LDS, statistics of the stored rows from its epilogue
        pw.A = (const unsigned short*)x; pw.W = (const unsigned short*)w; pw.OUT = (unsigned short*)z; pw.part = (float*)ws;
        pw.c0 = in_scale; pw.c1 = in_shift; pw.SIDE = (unsigned short*)y_side;
        const long M = (long)n * h * wd;
        const int sig[5] = {AL_MK, BL_NK, EPI_FWD, 8, 1};                         // tile id 8: the streaming pointwise kernel
        const int hr = igemm_prof_begin(sig, (int)M, cout, cin, 2.0 * M * cout * (double)cin, 2.0 * M * (cin + cout) + 2.0 * cin * cout, (hipStream_t)stream);
        hipError_t he = pw16_launch(pw, in_scale ? PW_PRO_FWD : PW_PRO_NONE, PW_EPI_STATS, (hipStream_t)stream);
        char sym[64];
        snprintf(sym, sizeof(sym), "pw16_kernel<%d,%d,%d,%d,%d>", cin, cout / pw.nct, pw16_waves(pw, in_scale ? PW_PRO_FWD : PW_PRO_NONE, PW_EPI_STATS), in_scale ? PW_PRO_FWD : PW_PRO_NONE, PW_EPI_STATS);
        igemm_prof_end(hr, sym, (hipStream_t)stream);
        if (he != hipSuccess) return (int)he;
        return rc(l_bn_finalize((const float*)ws, pw.nrb, gamma, beta, cout, eps, decay, mean, rstd, scale, shift, moving_mean, moving_var,
                                (hipStream_t)stream));
    }
    if (s16) {
        Plan16 guard;
        PlanS16 storage(true);
        e = conv2d_fwd_impl(x, w, true, nullptr, nullptr, nullptr, nullptr, nullptr, (uint16_t*)z, n, h, wd, cin, cout, ksize, stride,
                            nullptr, 0, stream, nullptr, nullptr, (float*)ws, &rows);
    } else {
        e = conv2d_fwd_impl(x, w, false, nullptr, nullptr, nullptr, nullptr, (float*)z, nullptr, n, h, wd, cin, cout, ksize, stride,
                            nullptr, 0, stream, nullptr, nullptr, (float*)ws, &rows);
    }
    if (e != FTE_OK) return e;
    return rc(l_bn_finalize((const float*)ws, rows, gamma, beta, cout, eps, decay, mean, rstd, scale, shift, moving_mean, moving_var,
                            (hipStream_t)stream));
}
size_t fte_conv2d_dgrad_bn_ws_bytes(int n, int h, int wd, int cin, int cout, int ksize, int stride) {
    PlanBn bnplan;
    return fte_conv2d_dgrad_ws_bytes(n, h, wd, cin, cout, ksize, stride);
}
int fte_conv2d_dgrad_bn(const void* dz, const void* w, const void* addin, const void* zbn, const void* ybn,
                        const float* gamma, const float* mean, const float* rstd, const float* bn_scale, const float* bn_shift,
                        void* g, float* dgamma, float* dbeta, float* coef,
                        int n, int h, int wd, int cin, int cout, int ksize, int stride, int s16, void* ws, size_t ws_bytes, void* stream) {
    if (!dz || !w || !zbn || !gamma || !mean || !rstd || !g || !dgamma || !dbeta || !coef || (s16 & ~1) || ((bn_scale == nullptr) != (bn_shift == nullptr)) ||
        (bn_scale && ybn))
        return FTE_EINVAL;
    const DgradBn bn = {zbn, ybn, gamma, mean, rstd, bn_scale, bn_shift, dgamma, dbeta, coef};
    PlanBn bnplan;
    Pw16Params pw;
    memset(&pw, 0, sizeof(pw));
    // (opt-in, FTE_PW16_DGRAD=1: measured SLOWER than the tile kernel's BN epilogue at 128 images -- 28x28 128<-256 69 vs 46 us, 14x14
    // 512<-256 87 vs 58: the K = 256 form runs four waves per CU, too few to hide the row pass's three input loads)
    static const bool pw_dgrad = getenv("FTE_PW16_DGRAD") && atoi(getenv("FTE_PW16_DGRAD")) == 1;
    if (pw_dgrad && s16 && ksize == 1 && stride == 1 && n > 0 && pw16_plan((long)n * h * wd, cout, cin, PW_EPI_BN, &pw)) {
        // the streaming pointwise kernel (pw16.hip): OUT[M, cin] = dz[M, cout] * W[cin, cout]^T (the HWIO pack's rows are its k-contiguous
        // filter rows), mask and sums in its row-coalesced epilogue, one partial row per block
        const size_t need = 2 * (size_t)pw.nrb * cin * sizeof(float);
        if (!ws || ws_bytes < need) return FTE_EWORKSPACE;
        pw.A = (const unsigned short*)dz; pw.W = (const unsigned short*)w; pw.OUT = (unsigned short*)g;
        pw.ADD = (const unsigned short*)addin;
        pw.Zm = (const unsigned short*)((!bn_scale && ybn) ? ybn : zbn);
        pw.Zx = (const unsigned short*)((!bn_scale && ybn) ? zbn : nullptr);
        pw.mu = mean; pw.rs = rstd; pw.sc = bn_scale; pw.sh = bn_shift;
        pw.part = (float*)ws; pw.pgx = (float*)ws + (size_t)pw.nrb * cin;
        hipError_t he = pw16_launch(pw, PW_PRO_NONE, PW_EPI_BN, (hipStream_t)stream);
        if (he != hipSuccess) return (int)he;
        return rc(l_bn_bwd_finalize(pw.part, pw.pgx, cin, pw.nrb, (long)n * h * wd, cin, gamma, mean, rstd, dgamma, dbeta, coef, (hipStream_t)stream));
    }
    if (s16) {
        Plan16 guard;
        PlanS16 storage(true);
        return conv2d_dgrad_impl(dz, w, true, nullptr, nullptr, nullptr, nullptr, nullptr, (uint16_t*)g, nullptr, nullptr,
                                 n, h, wd, cin, cout, ksize, stride, ws, ws_bytes, stream, (const uint16_t*)addin, nullptr, nullptr, &bn);
    }
    return conv2d_dgrad_impl(dz, w, false, (const float*)addin, nullptr, nullptr, nullptr, (float*)g, nullptr, nullptr, nullptr,
                             n, h, wd, cin, cout, ksize, stride, ws, ws_bytes, stream, nullptr, nullptr, nullptr, &bn);
}
int fte_bn_apply(const void* z, const float* scale, const float* shift, const void* res, void* y, long rows, int c, int relu, int flags, void* stream) {
    if (!z || !scale || !shift || !y || rows <= 0 || c % 4 || (flags & ~3) || (flags && c < 32)) return FTE_EINVAL;
    return rc(l_bn_apply(f32p(z), scale, shift, f32p(res), f32p(y), rows, c, relu, (hipStream_t)stream, flags));
}
int fte_bn_bwd_apply(const void* g, const void* z, const float* coef, void* dz, long rows, int c, int flags, void* stream) {
    if (!g || !z || !coef || !dz || rows <= 0 || c % 4 || (flags & ~3) || (flags && c < 32)) return FTE_EINVAL;
    return rc(l_bn_bwd_apply(f32p(g), f32p(z), coef, f32p(dz), rows, c, (hipStream_t)stream, flags));
}
size_t fte_gconv3x3_bn_ws_bytes(int n, int h, int wd, int c, int stride) {
    if (n <= 0 || h <= 0 || wd <= 0 || c <= 0 || c % 32 || (stride != 1 && stride != 2)) return 0;
    return align_up((size_t)l_gconv_bn_rows(n, h, wd, c) * 3 * c * sizeof(float));      // (h, wd: the input's side -- the larger grid)
}
int fte_gconv3x3_bn_fwd_bf16_s16(const uint16_t* x16, const uint16_t* wpk, uint16_t* z16, const float* gamma, const float* beta,
                                 float* mean, float* rstd, float* scale, float* shift, float* moving_mean, float* moving_var,
                                 float eps, float decay, const float* in_scale, const float* in_shift, uint16_t* y_side,
                                 int n, int h, int wd, int c, int stride, void* ws, size_t ws_bytes, void* stream) {
    if (!x16 || !wpk || !z16 || !gamma || !beta || !mean || !rstd || !scale || !shift || n <= 0 || h <= 0 || wd <= 0 || c <= 0 || c % 32 ||
        (stride != 1 && stride != 2) || (long)n * h * wd >= ((long)1 << 31) || ((moving_mean == nullptr) != (moving_var == nullptr)))
        return FTE_EINVAL;
    if (((in_scale == nullptr) != (in_shift == nullptr)) || (in_scale && (!y_side || stride != 1))) return FTE_EINVAL;
    if (!ws || ws_bytes < fte_gconv3x3_bn_ws_bytes(n, h, wd, c, stride)) return FTE_EWORKSPACE;
    const Pads ph = same_pads(h, 3, stride), pw = same_pads(wd, 3, stride);
    hipError_t e;
    int rows;
    if (stride == 1) {
        rows = l_gconv_bn_rows(n, h, wd, c);
        e = l_gconv_mfma16_bn(f32p(x16), wpk, f32p(z16), n, h, wd, c, h, wd, 0, 1, 1, 1, (float*)ws, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, (hipStream_t)stream,
                              in_scale, in_shift, y_side);
    } else {
        rows = l_gconv_bn_rows(n, ph.out, pw.out, c);
        e = l_gconv_mfma16_bn(f32p(x16), wpk, f32p(z16), n, ph.out, pw.out, c, h, wd, 1, ph.before, pw.before, 1, (float*)ws, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                              (hipStream_t)stream);
    }
    if (e != hipSuccess) return (int)e;
    return rc(l_bn_finalize((const float*)ws, rows, gamma, beta, c, eps, decay, mean, rstd, scale, shift, moving_mean, moving_var, (hipStream_t)stream));
}
int fte_gconv3x3_dgrad_bn_bf16_s16(const uint16_t* dz16, const uint16_t* wpk_dgrad, const uint16_t* zbn16, const float* gamma, const float* mean,
                                   const float* rstd, const float* bn_scale, const float* bn_shift, uint16_t* g16, float* dgamma, float* dbeta,
                                   float* coef, int n, int h, int wd, int c, int stride, void* ws, size_t ws_bytes, void* stream) {
    if (!dz16 || !wpk_dgrad || !zbn16 || !gamma || !mean || !rstd || !g16 || !dgamma || !dbeta || !coef || n <= 0 || h <= 0 || wd <= 0 || c <= 0 ||
        c % 32 || (stride != 1 && stride != 2) || (long)n * h * wd >= ((long)1 << 31) || ((bn_scale == nullptr) != (bn_shift == nullptr)))
        return FTE_EINVAL;
    if (!ws || ws_bytes < fte_gconv3x3_bn_ws_bytes(n, h, wd, c, stride)) return FTE_EWORKSPACE;
    const Pads ph = same_pads(h, 3, stride), pw = same_pads(wd, 3, stride);
    const int rows = l_gconv_bn_rows(n, h, wd, c);
    float* pg = (float*)ws;
    float* pgx = pg + (size_t)rows * c;
    hipError_t e;
    if (stride == 1)
        e = l_gconv_mfma16_bn(f32p(dz16), wpk_dgrad, f32p(g16), n, h, wd, c, h, wd, 0, 1, 1, 2, pg, pgx, zbn16, mean, rstd, bn_scale, bn_shift, (hipStream_t)stream);
    else
        e = l_gconv_mfma16_bn(f32p(dz16), wpk_dgrad, f32p(g16), n, h, wd, c, ph.out, pw.out, 2, ph.before, pw.before, 2, pg, pgx, zbn16, mean, rstd, bn_scale, bn_shift,
                              (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    return rc(l_bn_bwd_finalize(pg, pgx, c, rows, (long)n * h * wd, c, gamma, mean, rstd, dgamma, dbeta, coef, (hipStream_t)stream));
}

int fte_relu_bwd_s16(const uint16_t* dy16, const uint16_t* y16, uint16_t* g16, long n, void* stream) {
    if (!dy16 || !y16 || !g16 || n <= 0 || n % 4) return FTE_EINVAL;
    return rc(l_relu_bwd(f32p(dy16), f32p(y16), f32p(g16), n, (hipStream_t)stream, 2));
}
int fte_maxpool3x3s2_fwd_s16(const uint16_t* x16, uint16_t* y16, uint8_t* idx, int n, int h, int wd, int c, void* stream) {
    if (!x16 || !y16 || !idx || n <= 0 || c % 4) return FTE_EINVAL;
    const Pads ph = same_pads(h, 3, 2), pw = same_pads(wd, 3, 2);
    return rc(l_maxpool_fwd(f32p(x16), f32p(y16), idx, n, h, wd, c, ph.out, pw.out, ph.before, pw.before, (hipStream_t)stream, 2));
}
int fte_maxpool3x3s2_bwd_s16(const uint16_t* dy16, const uint8_t* idx, uint16_t* dx16, int n, int h, int wd, int c, void* stream) {
    if (!dy16 || !idx || !dx16 || n <= 0 || c % 4) return FTE_EINVAL;
    const Pads ph = same_pads(h, 3, 2), pw = same_pads(wd, 3, 2);
    return rc(l_maxpool_bwd(f32p(dy16), idx, f32p(dx16), n, h, wd, c, ph.out, pw.out, ph.before, pw.before, (hipStream_t)stream, 2));
}
int fte_gap_fwd_s16(const uint16_t* x16, float* y, int n, int hw, int c, void* stream) {
    if (!x16 || !y || n <= 0 || hw <= 0 || c <= 0 || c % 4) return FTE_EINVAL;
    return rc(l_gap_fwd(f32p(x16), y, n, hw, c, (hipStream_t)stream, 2));
}
int fte_gap_bwd_s16(const float* dy, uint16_t* dx16, int n, int hw, int c, void* stream) {
    if (!dy || !dx16 || n <= 0 || hw <= 0) return FTE_EINVAL;
    return rc(l_gap_bwd(dy, f32p(dx16), n, hw, c, (hipStream_t)stream, 2));
}
int fte_dropout_fwd(const float* x, float* mask, float* y, long n, float keep_prob, uint64_t seed, void* stream) {
    if (!x || !mask || !y || n <= 0 || !(keep_prob > 0.f)) return FTE_EINVAL;
    return rc(l_dropout_fwd(x, mask, y, n, keep_prob, seed, (hipStream_t)stream));
}
int fte_dropout_bwd(const float* dy, const float* mask, float* dx, long n, float keep_prob, void* stream) {
    if (!dy || !mask || !dx || n <= 0 || !(keep_prob > 0.f)) return FTE_EINVAL;
    return rc(l_scale_mask(dy, mask, dx, n, 1.f / keep_prob, (hipStream_t)stream));
}
int fte_im2col_first(const float* x, float* cols, int n, int h, int wd, int cin, int ksize, int stride, int kpad, void* stream) {
    if (!x || !cols || n <= 0 || kpad % 32 || kpad < ksize * ksize * cin) return FTE_EINVAL;
    const Pads ph = same_pads(h, ksize, stride), pw = same_pads(wd, ksize, stride);
    return rc(l_im2col_first(x, cols, n, h, wd, cin, ksize, stride, ph.out, pw.out, ph.before, pw.before, kpad, (hipStream_t)stream));
}

int fte_im2col_first_s16(const float* x, uint16_t* cols16, int n, int h, int wd, int cin, int ksize, int stride, int kpad, void* stream) {
    if (!x || !cols16 || n <= 0 || kpad % 32 || kpad < ksize * ksize * cin) return FTE_EINVAL;
    const Pads ph = same_pads(h, ksize, stride), pw = same_pads(wd, ksize, stride);
    return rc(l_im2col_first(x, f32p(cols16), n, h, wd, cin, ksize, stride, ph.out, pw.out, ph.before, pw.before, kpad, (hipStream_t)stream, 1));
}

int fte_preprocess_u8(const uint8_t* slots, float* out, int n, long slot_stride, int channels, int in_h, int in_w, int crop_h, int crop_w,
                      void* stream) {
    if (!slots || !out || n <= 0 || (channels != 1 && channels != 3) || in_h <= 0 || in_w <= 0 || crop_h <= 0 || crop_w <= 0 ||
        crop_h > in_h || crop_w > in_w || slot_stride < 64 || slot_stride % 64 || n > 65535) return FTE_EINVAL;
    return rc(k_preprocess_u8(slots, out, n, slot_stride, channels, in_h, in_w, crop_h, crop_w, (hipStream_t)stream));
}

// ------------------------------------------------------------------------------------------------
// grouped 3x3 conv, SE-gate pieces
int fte_gconv3x3_fwd(const float* x, const float* w, float* y, int n, int h, int wd, int c, int groups, int stride, void* stream) {
    if (!x || !w || !y || n <= 0 || groups <= 0 || c % groups || (stride != 1 && stride != 2)) return FTE_EINVAL;
    const Pads ph = same_pads(h, 3, stride), pw = same_pads(wd, 3, stride);
    return rc(l_gconv_fwd(x, w, y, n, h, wd, c, groups, ph.out, pw.out, stride, ph.before, pw.before, (hipStream_t)stream));
}
int fte_gconv3x3_pack_bf16(const float* w, uint16_t* wpk_fwd, uint16_t* wpk_dgrad, int c, int groups, void* stream) {
    if (!w || !wpk_fwd || !wpk_dgrad || groups <= 0 || c % groups || c % 32) return FTE_EINVAL;
    const int gw = c / groups;
    if (gw != 4 && gw != 8 && gw != 16 && gw != 32) return FTE_EINVAL;
    return rc(l_gconv_pack16(w, wpk_fwd, wpk_dgrad, c, groups, (hipStream_t)stream));
}
int fte_gconv3x3_bf16(const float* x, const uint16_t* wpk, float* y, int n, int h, int wd, int c, int stride, int dgrad, void* stream) {
    if (!x || !wpk || !y || n <= 0 || h <= 0 || wd <= 0 || c <= 0 || c % 32 || (stride != 1 && stride != 2) ||
        (long)n * h * wd >= ((long)1 << 31)) return FTE_EINVAL;
    const Pads ph = same_pads(h, 3, stride), pw = same_pads(wd, 3, stride);
    if (stride == 1) return rc(l_gconv_mfma16(x, wpk, y, n, h, wd, c, h, wd, 0, 1, 1, (hipStream_t)stream));
    if (!dgrad) return rc(l_gconv_mfma16(x, wpk, y, n, ph.out, pw.out, c, h, wd, 1, ph.before, pw.before, (hipStream_t)stream));
    return rc(l_gconv_mfma16(x, wpk, y, n, h, wd, c, ph.out, pw.out, 2, ph.before, pw.before, (hipStream_t)stream));
}
int fte_gconv3x3_bf16_s16(const uint16_t* x16, const uint16_t* wpk, uint16_t* y16, int n, int h, int wd, int c, int stride, int dgrad, void* stream) {
    if (!x16 || !wpk || !y16 || n <= 0 || h <= 0 || wd <= 0 || c <= 0 || c % 32 || (stride != 1 && stride != 2) ||
        (long)n * h * wd >= ((long)1 << 31)) return FTE_EINVAL;
    const Pads ph = same_pads(h, 3, stride), pw = same_pads(wd, 3, stride);
    const float* x = f32p(x16);
    float* y = f32p(y16);
    if (stride == 1) return rc(l_gconv_mfma16(x, wpk, y, n, h, wd, c, h, wd, 0, 1, 1, (hipStream_t)stream, 1));
    if (!dgrad) return rc(l_gconv_mfma16(x, wpk, y, n, ph.out, pw.out, c, h, wd, 1, ph.before, pw.before, (hipStream_t)stream, 1));
    return rc(l_gconv_mfma16(x, wpk, y, n, h, wd, c, ph.out, pw.out, 2, ph.before, pw.before, (hipStream_t)stream, 1));
}
size_t fte_gconv3x3_wgrad_bf16_ws_bytes(int n, int h, int wd, int c, int groups, int stride) {
    if (n <= 0 || h <= 0 || wd <= 0 || c <= 0 || groups <= 0 || c % groups || c % 32 || (stride != 1 && stride != 2)) return 0;
    const Pads ph = same_pads(h, 3, stride), pw = same_pads(wd, 3, stride);
    return align_up((size_t)l_gconv_wgrad16_chunks((long)n * ph.out * pw.out, c) * c * 9 * (c / groups) * sizeof(float));
}
int fte_gconv3x3_wgrad_bf16(const float* x, const float* dz, float* dw, int n, int h, int wd, int c, int groups, int stride,
                            void* ws, size_t ws_bytes, void* stream) {
    if (!x || !dz || !dw || n <= 0 || h <= 0 || wd <= 0 || groups <= 0 || c % groups || c % 32 || (stride != 1 && stride != 2) ||
        (long)n * h * wd >= ((long)1 << 31)) return FTE_EINVAL;
    const int gw = c / groups;
    if (gw != 4 && gw != 8 && gw != 16 && gw != 32) return FTE_EINVAL;
    if (!ws || ws_bytes < fte_gconv3x3_wgrad_bf16_ws_bytes(n, h, wd, c, groups, stride)) return FTE_EWORKSPACE;
    const Pads ph = same_pads(h, 3, stride), pw = same_pads(wd, 3, stride);
    return rc(l_gconv_wgrad16(x, dz, (float*)ws, dw, n, h, wd, c, groups, ph.out, pw.out, stride, ph.before, pw.before,
                              l_gconv_wgrad16_chunks((long)n * ph.out * pw.out, c), (hipStream_t)stream));
}
int fte_gconv3x3_wgrad_bf16_s16(const uint16_t* x16, const uint16_t* dz16, float* dw, int n, int h, int wd, int c, int groups, int stride,
                                void* ws, size_t ws_bytes, void* stream) {
    if (!x16 || !dz16 || !dw || n <= 0 || h <= 0 || wd <= 0 || groups <= 0 || c % groups || c % 32 || (stride != 1 && stride != 2) ||
        (long)n * h * wd >= ((long)1 << 31)) return FTE_EINVAL;
    const int gw = c / groups;
    if (gw != 4 && gw != 8 && gw != 16 && gw != 32) return FTE_EINVAL;
    if (!ws || ws_bytes < fte_gconv3x3_wgrad_bf16_ws_bytes(n, h, wd, c, groups, stride)) return FTE_EWORKSPACE;
    const Pads ph = same_pads(h, 3, stride), pw = same_pads(wd, 3, stride);
    return rc(l_gconv_wgrad16(f32p(x16), f32p(dz16), (float*)ws, dw, n, h, wd, c, groups, ph.out, pw.out, stride, ph.before, pw.before,
                              l_gconv_wgrad16_chunks((long)n * ph.out * pw.out, c), (hipStream_t)stream, 1));
}
int fte_gconv3x3_dgrad(const float* dz, const float* w, float* dx, int n, int h, int wd, int c, int groups, int stride, void* stream) {
    if (!dz || !w || !dx || n <= 0 || groups <= 0 || c % groups || (stride != 1 && stride != 2)) return FTE_EINVAL;
    const Pads ph = same_pads(h, 3, stride), pw = same_pads(wd, 3, stride);
    return rc(l_gconv_dgrad(dz, w, dx, n, h, wd, c, groups, ph.out, pw.out, stride, ph.before, pw.before, (hipStream_t)stream));
}
size_t fte_gconv3x3_wgrad_ws_bytes(int n, int h, int wd, int c, int groups, int stride) {
    const Pads ph = same_pads(h, 3, stride), pw = same_pads(wd, 3, stride);
    const int gw = c / groups;
    return align_up((size_t)l_gconv_wgrad_chunks((long)n * ph.out * pw.out, c, gw) * 9 * c * gw * sizeof(float)) + SCRATCH_BYTES;
}
int fte_gconv3x3_wgrad(const float* x, const float* dz, float* dw, int n, int h, int wd, int c, int groups, int stride,
                       void* ws, size_t ws_bytes, void* stream) {
    if (!x || !dz || !dw || n <= 0 || groups <= 0 || c % groups || (stride != 1 && stride != 2)) return FTE_EINVAL;
    const Pads ph = same_pads(h, 3, stride), pw = same_pads(wd, 3, stride);
    const int gw = c / groups;
    const int chunks = l_gconv_wgrad_chunks((long)n * ph.out * pw.out, c, gw);
    const size_t need = align_up((size_t)chunks * 9 * c * gw * sizeof(float));
    if (!ws || ws_bytes < need + SCRATCH_BYTES) return FTE_EWORKSPACE;
    hipError_t e = l_gconv_wgrad(x, dz, (float*)ws, n, h, wd, c, groups, ph.out, pw.out, stride, ph.before, pw.before, chunks, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    return rc(k_reduce_rows((const float*)ws, dw, nullptr, 1, chunks, 9L * c * gw, 1, 1.f, (float*)((char*)ws + need), (hipStream_t)stream));
}
int fte_bcast_add(float* dx, const float* v, int n, int hw, int c, float scale, void* stream) {
    if (!dx || !v || n <= 0 || c % 4) return FTE_EINVAL;
    return rc(l_bcast_add(dx, v, n, hw, c, scale, (hipStream_t)stream));
}
int fte_act_fwd(const float* x, float* y, long n, int kind, void* stream) {
    if (!x || !y || n <= 0 || (kind != 0 && kind != 1)) return FTE_EINVAL;
    return rc(l_act_fwd(x, y, n, kind, (hipStream_t)stream));
}
int fte_act_bwd(const float* dy, const float* y, float* dx, long n, int kind, void* stream) {
    if (!dy || !y || !dx || n <= 0 || (kind != 0 && kind != 1)) return FTE_EINVAL;
    return rc(l_act_bwd(dy, y, dx, n, kind, (hipStream_t)stream));
}
int fte_channel_scale_fwd(const float* x, const float* gate, float* y, int n, int hw, int c, void* stream) {
    if (!x || !gate || !y || n <= 0 || c % 4) return FTE_EINVAL;
    return rc(l_chscale_fwd(x, gate, y, n, hw, c, (hipStream_t)stream));
}
int fte_channel_scale_bwd(const float* dy, const float* x, const float* gate, float* dx, float* dgate, int n, int hw, int c,
                          int pre_sigmoid, void* stream) {
    if (!dy || !x || !gate || !dx || !dgate || n <= 0 || c <= 0 || c % 4) return FTE_EINVAL;
    return rc(l_chscale_bwd(dy, x, gate, dx, dgate, n, hw, c, pre_sigmoid ? 1 : 0, (hipStream_t)stream));
}

// SE gate on bf16 tensors (fte.h, bf16 STORAGE): scale, the gate-gradient reduction (no dx), and the one-pass input gradient
int fte_channel_scale_fwd_s16(const uint16_t* x16, const float* gate, uint16_t* y16, int n, int hw, int c, void* stream) {
    if (!x16 || !gate || !y16 || n <= 0 || c % 4) return FTE_EINVAL;
    return rc(l_chscale_fwd(f32p(x16), gate, f32p(y16), n, hw, c, (hipStream_t)stream, 1));
}
int fte_channel_scale_bwd_s16(const uint16_t* dy16, const uint16_t* x16, const float* gate, float* dgate, int n, int hw, int c,
                              int pre_sigmoid, void* stream) {
    if (!dy16 || !x16 || !gate || !dgate || n <= 0 || c <= 0 || c % 4) return FTE_EINVAL;
    return rc(l_chscale_bwd(f32p(dy16), f32p(x16), gate, nullptr, dgate, n, hw, c, pre_sigmoid ? 1 : 0, (hipStream_t)stream, 1));
}
int fte_channel_scale_bwd_apply_s16(const uint16_t* dy16, const float* gate, const float* dsq, uint16_t* dx16, int n, int hw, int c,
                                    float scale, void* stream) {
    if (!dy16 || !gate || !dsq || !dx16 || n <= 0 || c <= 0 || c % 4) return FTE_EINVAL;
    return rc(l_chscale_bwd_apply(f32p(dy16), gate, dsq, f32p(dx16), n, hw, c, scale, (hipStream_t)stream, 1));
}

// the SE residual block, fused (layers.hip "SE residual block")
static bool se_args_ok(int n, int hw, int c, int flags) { return n > 0 && hw > 0 && c > 0 && c % 4 == 0 && !(flags & ~3) && (!flags || c >= 32); }
int fte_se_squeeze(const void* z, const float* scale, const float* shift, const float* mean, const float* rstd, float* sq, float* xm,
                   int n, int hw, int c, int flags, void* stream) {
    if (!z || !scale || !shift || !sq || (xm && (!mean || !rstd)) || !se_args_ok(n, hw, c, flags)) return FTE_EINVAL;
    return rc(l_se_squeeze(f32p(z), scale, shift, mean, rstd, sq, xm, n, hw, c, (hipStream_t)stream, flags));
}
int fte_se_apply_fwd(const void* z, const float* scale, const float* shift, const float* gate, const void* shortcut, void* out,
                     int n, int hw, int c, int flags, void* stream) {
    if (!z || !scale || !shift || !gate || !shortcut || !out || !se_args_ok(n, hw, c, flags)) return FTE_EINVAL;
    return rc(l_se_apply(f32p(z), scale, shift, gate, f32p(shortcut), f32p(out), n, hw, c, (hipStream_t)stream, flags));
}
int fte_se_bwd_gate(const void* dy, const void* out, const void* z, const float* gamma, const float* beta, const float* mean,
                    const float* rstd, const float* gate, void* g, float* s1, float* s2, float* dgate, int n, int hw, int c, int flags, void* stream) {
    if (!dy || !out || !z || !gamma || !beta || !mean || !rstd || !gate || !g || !s1 || !s2 || !dgate || !se_args_ok(n, hw, c, flags)) return FTE_EINVAL;
    return rc(l_se_bwd_gate(f32p(dy), f32p(out), f32p(z), gamma, beta, mean, rstd, gate, f32p(g), s1, s2, dgate, n, hw, c, (hipStream_t)stream, flags));
}
int fte_se_bn_bwd_coef(const float* s1, const float* s2, const float* gate, const float* dsq, const float* xm, const float* gamma,
                       const float* mean, const float* rstd, float* dgamma, float* dbeta, float* coef, int n, int hw, int c, void* stream) {
    if (!s1 || !s2 || !gate || !dsq || !xm || !gamma || !mean || !rstd || !dgamma || !dbeta || !coef || !se_args_ok(n, hw, c, 0)) return FTE_EINVAL;
    return rc(l_se_bn_coef(s1, s2, gate, dsq, xm, gamma, mean, rstd, dgamma, dbeta, coef, n, hw, c, (hipStream_t)stream));
}
int fte_se_bn_bwd_apply(const void* g, const void* z, const float* coef, const float* gate, const float* dsq, void* dz,
                        int n, int hw, int c, int flags, void* stream) {
    if (!g || !z || !coef || !gate || !dsq || !dz || !se_args_ok(n, hw, c, flags)) return FTE_EINVAL;
    return rc(l_se_bn_apply(f32p(g), f32p(z), coef, gate, dsq, f32p(dz), n, hw, c, (hipStream_t)stream, flags));
}

// ------------------------------------------------------------------------------------------------
// ShuffleNet-v2: depthwise 3x3, channel gather
int fte_dwconv3x3_fwd(const float* x, const float* w, float* y, int n, int h, int wd, int c, int stride, void* stream) {
    if (!x || !w || !y || n <= 0 || c % 4 || (stride != 1 && stride != 2)) return FTE_EINVAL;
    const Pads ph = same_pads(h, 3, stride), pw = same_pads(wd, 3, stride);
    return rc(l_dwconv_fwd(x, w, y, n, h, wd, c, ph.out, pw.out, stride, ph.before, pw.before, (hipStream_t)stream));
}
int fte_dwconv3x3_dgrad(const float* dy, const float* w, float* dx, int n, int h, int wd, int c, int stride, void* stream) {
    if (!dy || !w || !dx || n <= 0 || c % 4 || (stride != 1 && stride != 2)) return FTE_EINVAL;
    const Pads ph = same_pads(h, 3, stride), pw = same_pads(wd, 3, stride);
    return rc(l_dwconv_dgrad(dy, w, dx, n, h, wd, c, ph.out, pw.out, stride, ph.before, pw.before, (hipStream_t)stream));
}
size_t fte_dwconv3x3_wgrad_ws_bytes(int n, int h, int wd, int c, int stride) {
    const Pads ph = same_pads(h, 3, stride), pw = same_pads(wd, 3, stride);
    return align_up((size_t)l_dwconv_wgrad_splits((long)n * ph.out * pw.out, c) * 9 * c * sizeof(float)) + SCRATCH_BYTES;
}
int fte_dwconv3x3_wgrad(const float* x, const float* dy, float* dw, int n, int h, int wd, int c, int stride,
                        void* ws, size_t ws_bytes, void* stream) {
    if (!x || !dy || !dw || n <= 0 || c % 4 || (stride != 1 && stride != 2)) return FTE_EINVAL;
    const Pads ph = same_pads(h, 3, stride), pw = same_pads(wd, 3, stride);
    const int splits = l_dwconv_wgrad_splits((long)n * ph.out * pw.out, c);
    const size_t need = align_up((size_t)splits * 9 * c * sizeof(float));
    if (!ws || ws_bytes < need + SCRATCH_BYTES) return FTE_EWORKSPACE;
    hipError_t e = l_dwconv_wgrad(x, dy, (float*)ws, n, h, wd, c, ph.out, pw.out, stride, ph.before, pw.before, splits, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    return rc(k_reduce_rows((const float*)ws, dw, nullptr, 1, splits, 9L * c, 1, 1.f, (float*)((char*)ws + need), (hipStream_t)stream));
}
int fte_channel_gather(const float* a, const float* b, float* out, const int32_t* table, long rows, int ca, int cb, int co, void* stream) {
    if (!a || !out || !table || rows <= 0 || co <= 0 || co % 4) return FTE_EINVAL;      // the table is read 4 entries at a time
    return rc(l_channel_gather(a, b ? b : a, out, table, rows, ca, cb, co, (hipStream_t)stream));
}
int fte_channel_gather_affine(const float* a, const float* b, float* out, const int32_t* table, int co,
                              float* out1, const int32_t* table1, int co1, long rows, int ca, int cb,
                              const float* scale_a, const float* shift_a, int relu_a,
                              const float* scale_b, const float* shift_b, int relu_b, void* stream) {
    if (!a || !out || !table || rows <= 0 || co <= 0 || co % 4 || (scale_a && !shift_a) || (scale_b && (!shift_b || !b))) return FTE_EINVAL;
    if (out1 ? (!table1 || co1 <= 0 || co1 % 4) : co1 != 0) return FTE_EINVAL;
    return rc(l_channel_gather_affine(a, b ? b : a, out, table, co, out1, table1, co1, rows, ca, cb, scale_a, shift_a, relu_a,
                                      scale_b, shift_b, relu_b, (hipStream_t)stream));
}

// ---- bf16 STORAGE twins of ShuffleNet-v2's layers: every activation-side tensor bf16, filters / tables / per-channel vectors unchanged
int fte_dwconv3x3_fwd_s16(const uint16_t* x16, const float* w, uint16_t* y16, int n, int h, int wd, int c, int stride, void* stream) {
    if (!x16 || !w || !y16 || n <= 0 || c % 4 || (stride != 1 && stride != 2)) return FTE_EINVAL;
    const Pads ph = same_pads(h, 3, stride), pw = same_pads(wd, 3, stride);
    return rc(l_dwconv_fwd(f32p(x16), w, f32p(y16), n, h, wd, c, ph.out, pw.out, stride, ph.before, pw.before, (hipStream_t)stream, 1));
}
int fte_dwconv3x3_dgrad_s16(const uint16_t* dy16, const float* w, uint16_t* dx16, int n, int h, int wd, int c, int stride, void* stream) {
    if (!dy16 || !w || !dx16 || n <= 0 || c % 4 || (stride != 1 && stride != 2)) return FTE_EINVAL;
    const Pads ph = same_pads(h, 3, stride), pw = same_pads(wd, 3, stride);
    return rc(l_dwconv_dgrad(f32p(dy16), w, f32p(dx16), n, h, wd, c, ph.out, pw.out, stride, ph.before, pw.before, (hipStream_t)stream, 1));
}
int fte_dwconv3x3_wgrad_s16(const uint16_t* x16, const uint16_t* dy16, float* dw, int n, int h, int wd, int c, int stride,
                            void* ws, size_t ws_bytes, void* stream) {
    if (!x16 || !dy16 || !dw || n <= 0 || c % 4 || (stride != 1 && stride != 2)) return FTE_EINVAL;
    const Pads ph = same_pads(h, 3, stride), pw = same_pads(wd, 3, stride);
    const int splits = l_dwconv_wgrad_splits((long)n * ph.out * pw.out, c);
    const size_t need = align_up((size_t)splits * 9 * c * sizeof(float));
    if (!ws || ws_bytes < need + SCRATCH_BYTES) return FTE_EWORKSPACE;
    hipError_t e = l_dwconv_wgrad(f32p(x16), f32p(dy16), (float*)ws, n, h, wd, c, ph.out, pw.out, stride, ph.before, pw.before, splits, (hipStream_t)stream, 1);
    if (e != hipSuccess) return (int)e;
    return rc(k_reduce_rows((const float*)ws, dw, nullptr, 1, splits, 9L * c, 1, 1.f, (float*)((char*)ws + need), (hipStream_t)stream));
}
int fte_channel_gather_s16(const uint16_t* a, const uint16_t* b, uint16_t* out, const int32_t* table, long rows, int ca, int cb, int co, void* stream) {
    if (!a || !out || !table || rows <= 0 || co <= 0 || co % 4) return FTE_EINVAL;
    return rc(l_channel_gather(f32p(a), f32p(b ? b : a), f32p(out), table, rows, ca, cb, co, (hipStream_t)stream, 1));
}
int fte_channel_gather_affine_s16(const uint16_t* a, const uint16_t* b, uint16_t* out, const int32_t* table, int co,
                                  uint16_t* out1, const int32_t* table1, int co1, long rows, int ca, int cb,
                                  const float* scale_a, const float* shift_a, int relu_a,
                                  const float* scale_b, const float* shift_b, int relu_b, void* stream) {
    if (!a || !out || !table || rows <= 0 || co <= 0 || co % 4 || (scale_a && !shift_a) || (scale_b && (!shift_b || !b))) return FTE_EINVAL;
    if (out1 ? (!table1 || co1 <= 0 || co1 % 4) : co1 != 0) return FTE_EINVAL;
    return rc(l_channel_gather_affine(f32p(a), f32p(b ? b : a), f32p(out), table, co, f32p(out1), table1, co1, rows, ca, cb, scale_a, shift_a, relu_a,
                                      scale_b, shift_b, relu_b, (hipStream_t)stream, 1));
}
int fte_bn_train_stats_s16(const void* z, const float* gamma, const float* beta, float* mean, float* rstd, float* scale, float* shift,
                           float* moving_mean, float* moving_var, long rows, int c, float eps, float decay, int flags,
                           void* ws, size_t ws_bytes, void* stream) {
    if (!z || !gamma || !beta || !mean || !rstd || !scale || !shift || rows <= 0 || c % 4 || c < 32 || (flags & ~3)) return FTE_EINVAL;
    if (!ws || ws_bytes < fte_bn_ws_bytes(c)) return FTE_EWORKSPACE;
    return rc(l_bn_train_stats(f32p(z), gamma, beta, rows, c, eps, decay, mean, rstd, scale, shift, moving_mean, moving_var,
                               (float*)ws, (hipStream_t)stream, flags));
}

}  // extern "C"
