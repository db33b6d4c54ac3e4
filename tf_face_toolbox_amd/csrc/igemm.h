// igemm.h -- parameter block and launcher of the gathered-GEMM kernel family (igemm.hip).
#pragma once
#include <hip/hip_runtime.h>

enum { AL_MK = 0, AL_KM = 1 };           // A operand: k-contiguous rows (im2col) | m-contiguous (k = pixel)
enum { BL_KN = 0, BL_NK = 1 };           // B operand: [k][n] n-contiguous | [n][k] k-contiguous (per tap)
enum { EPI_FWD = 0, EPI_DGRAD = 1 };
enum { TILE_128x128 = 0, TILE_256x64 = 1, TILE_128x64 = 2, TILE_64x64 = 3, TILE_192x64 = 4,      // 192x64: M = 576 = 9 taps x 64 channels in three whole tiles
       TILE_64x64_W1 = 5, TILE_64x64_W2 = 6 };  // 64x64 tile computed by ONE wave (64 threads, no barrier) / by two waves (128 threads)

struct IgemmParams {
    int M, N, K;          // K = NT*KC (tap modes) or number of pixels (AL_KM)
    int kchunk;           // K range per blockIdx.y (multiple of 32)
    int split_major;      // > 0: 1-D grid of tiles*splits with split = id % split_major (all tiles of one K range
                          // -- one pixel range for wgrad -- then share an XCD and its L2)
    int m_base;           // first GEMM row of this launch (rows [m_base, M) are tiled; lets one op be
                          // split into a big-tile main launch and a small-tile tail launch)
    // ---- A: gathered from an NHWC image -------------------------------------
    const float* A;
    unsigned a_bytes;     // size of the A tensor in bytes (< 2 GiB): buffer-load range check
    int a_OH, a_OW;       // grid over which the GEMM row (AL_MK) / reduction index (AL_KM) decomposes
    int a_IH, a_IW;       // source image
    int a_stride;         // source pixels per grid step
    int a_ld;             // floats per source pixel
    int a_KC;             // channels per tap
    int a_NT;             // taps
    int a_dh[9], a_dw[9]; // source pixel = grid*stride + (dh, dw); out of range -> zero
    // ---- B -------------------------------------------------------------------
    const float* B;
    unsigned b_bytes;
    int b_ld;
    int b_tapoff[9];      // BL_NK: element offset of tap t
    // ---- C / epilogue ----------------------------------------------------------
    int c_OH, c_OW;       // 0,0: row m stored at m*c_ld; else m -> (n, oh, ow) on this grid ...
    int c_FH, c_FW;       // ... placed at pixel (oh*c_step + c_ph, ow*c_step + c_pw) of a c_FH x c_FW image
    int c_step, c_ph, c_pw;
    int c_ld;
    long slab;            // EPI_FWD: Y offset per blockIdx.y (split-K partial slabs)
    float* PW;            // != NULL: store the raw accumulator tile to PW[(split*tiles + tile)*BM*BN + r*BN + c] and stop;
                          // igemm_fixup then sums the splits and applies the epilogue (split-K for fwd / dgrad tiles)
    // EPI_FWD:  v = acc + bias[n]; Z = v; Y = prelu(v, alpha[n]) + R
    float* Y; float* Z; const float* R; const float* bias; const float* alpha;
    // EPI_DGRAD: v = acc + ADD; RAW = v; DZ = v * prelu'(Zin, alpha[n % amod]);
    //            PA[prow0 + mtile][n] = sum_rows v*min(Zin,0); PB[...] = sum_rows DZ
    const float* ADD; float* RAW; const float* Zin; float* DZ; float* PA; float* PB;
    int amod, prow0;
    // EPI_DGRAD, stride 2: ncls > 1 merges the output-parity classes into one launch (see igemm_kernel)
    int ncls, cls_tiles, cls_mtiles, cls_rot;      // cls_rot: set by the launcher (class rotation, see the kernel)
    int cls_tap0[5], cls_ph[4], cls_pw[4];
    // bf16 operand copies: src16 != 0 -> A and B point at bf16 data (a_bytes / b_bytes count 2-byte elements' bytes) and the
    // BF = 2 kernels run; Y16 / DZ16 != NULL -> the epilogue also writes a bf16 copy of Y / DZ for the next consumer
    int src16;
    unsigned short* Y16; unsigned short* DZ16;
    // bf16 STORAGE (src16 launches only; fte_conv2d_*_s16): the epilogue's tensors as bf16 in HBM -- inputs R16 (shortcut), ADD16
    // (skip gradient), Zin16 (z of the previous layer) replace R / ADD / Zin; outputs Z16, RAW16 next to Y16 / DZ16; the fp32
    // outputs Y / DZ may then be NULL
    const unsigned short* R16; const unsigned short* ADD16; const unsigned short* Zin16;
    unsigned short* Z16; unsigned short* RAW16;
    // ---- BN fusion (the graph nets' conv -> BN pairs; only the BNM = 1 instantiations of the kernels read these) ----------------
    // EPI_FWD, SP != NULL: per-tile column statistics of the STORED output (no activation, no shortcut): n, mean, M2 of the tile's
    //   valid rows -> SP[((prow0 + mt) * 3 + {0, 1, 2}) * N + col] = bn_finalize_kernel's partial layout, one partial row per tile row
    float* SP;
    // EPI_DGRAD, bn_mu != NULL: the data gradient lands on the output of a BN (+ ReLU) layer; its mask and reduction pass run here:
    //   v = stored(acc + ADD);  g = mask ? v : 0;  DZ = g;  PA[row][n] = sum_rows g * (zb - mu[n]) * rs[n];  PB[row][n] = sum_rows g
    //   mask / zb:  bn_sc != NULL: fma(Zin, sc, sh) > 0, zb = Zin (BN + ReLU, mask recomputed from z -- bn_affine's expression)
    //               Zx / Zx16 != NULL: Zin > 0 with Zin = the stored block output, zb = Zx (BN + add + ReLU)
    //               neither: no mask, zb = Zin (BN without activation)
    const float* bn_mu; const float* bn_rs; const float* bn_sc; const float* bn_sh;
    const float* Zx; const unsigned short* Zx16;
    // ---- stream-K (igemm_sk_kernel, fp32 forward / data gradient; see igemm.hip "stream-K") -------------------------------------
    // sk_workers > 0: the launch is ONE grid of sk_workers resident blocks; worker w takes iterations [w*sk_base + min(w, sk_rem), ...)
    // of the linear (tile, K-step) space, sk_base (+1 for the first sk_rem workers) each.  A worker that starts inside a tile leaves its
    // raw accumulators in slab w of SKW (BM*BN floats each) and raises SKF[w]; the worker that holds K-step 0 of the tile adds the
    // slabs of its successors in worker order and runs the fused epilogue.  SKF: the caller passes sk_workers words behind the slabs; the
    // launcher normally replaces them by the library's per-stream words (igemm.hip sk_flags) and zeroes the caller's only as a fallback.
    int sk_workers, sk_base, sk_rem, sk_epoch, sk_acq;      // sk_epoch: the value a raised flag word holds (set by the launcher)
    float* SKW; unsigned* SKF;
    int ptiles;           // igemm16p_kernel (persistent blocks): tiles of the launch; set by its launcher
    int ptiles_dbg;       // diagnostic switches of that kernel (FTE_IGEMM16_DBG; 0 in production)
};

hipError_t igemm_launch(const IgemmParams& p, int al, int bl, int epi, int tile, int splits, hipStream_t st);
// the fix-up splits every tile into this many row chunks (one block each); a dgrad fix-up therefore writes
// FIXUP_CHUNKS partial rows (PA/PB) per tile row, numbered prow0 + mt*FIXUP_CHUNKS + chunk
constexpr int FIXUP_CHUNKS = 4;
// sums `splits` partial tiles written through PW and applies the EPI_FWD / EPI_DGRAD epilogue of the same params
hipError_t igemm_fixup(const IgemmParams& p, int epi, int tile, int splits, hipStream_t st);
void igemm_tile_dims(int tile, int* bm, int* bn);
// stream-K: resident blocks per CU the planner assumes for a tile shape (0: no stream-K instantiation of that shape), and the
// workspace (slabs + flag words) a launch of `workers` blocks needs
int igemm_sk_blocks_per_cu(int tile, int epi);      // min(compiled figure, the runtime's occupancy of that symbol on this device)
int igemm_num_cus();                                   // compute units of the current device (256 without one)
size_t igemm_sk_ws_bytes(int tile, int workers);

// igemm16.hip: the bf16 LDS-DMA kernel (k-contiguous bf16 A and B: conv forward / dgrad on the bf16 operand copies).  igemm_launch
// routes a launch there when igemm16_handles() says so; everything else runs on igemm.hip's kernels.
bool igemm16_handles(const IgemmParams& p, int al, int bl, int tile);
hipError_t igemm16_launch(const IgemmParams& p, int epi, int tile, int splits, hipStream_t st);

// operand precision of the MFMA products: false = fp32 (v_mfma_f32_32x32x2_f32, exact), true = operands rounded to bf16
// inside the kernel (v_mfma_f32_32x32x16_bf16, fp32 accumulate); storage in HBM is fp32 either way
void igemm_set_bf16(bool on);
bool igemm_get_bf16();

// launch records for the roofline measurement (see igemm.hip)
void igemm_prof_enable(bool on, bool clear);
bool igemm_prof_on();
// called by every launcher of the family while records are on: the kernel symbol of this dispatch with its template
// arguments, spelled as rocprofv3 --kernel-trace prints them (minus blanks), e.g. "igemm_kernel<128,128,2,2,1,0,0,0>"
void igemm_note_symbol(const char* family, const int* targs, int ntargs);
hipError_t igemm_prof_get_name(int i, char* buf, int buflen);
// records for kernels launched outside igemm_launch (wgrad16.hip): sig5 = {al, bl, epi, tile id, splits}
int igemm_prof_begin(const int* sig5, int rows, int n, int k, double flops, double bytes, hipStream_t st);
void igemm_prof_end(int handle, const char* sym, hipStream_t st);
int igemm_prof_count();
hipError_t igemm_prof_get(int i, int* sig, double* flops, float* ms);
hipError_t igemm_prof_get_shape(int i, int* mnk, double* bytes);      // GEMM shape {rows, N, K} and algorithmic bytes of record i
