// wino.h -- Winograd F(2x2, 3x3) path of the stride-1 3x3 convolutions (wino.hip): host launchers.
// The reference turns this algorithm on for every run (train.py:260, TF_ENABLE_WINOGRAD_NONFUSED=1); the layers are the 16
// resBlock convs of nets/sphere.py:38-45,58-70.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

// Geometry of one stride-1 SAME 3x3 layer over [n, h, w, c]: 2x2 output tiles, TH x TW of them per image (odd sizes: the last
// tile row / column is half outside the image), M tiles in all, MB = row blocks of 64 tiles.
struct WinoGeom {
    int n, h, w, th, tw;
    long M;
    int MB;
};
WinoGeom wino_geom(int n, int h, int w);

// "pack" layout of every transformed operand, C channels (C % 8 == 0), R rows (tiles, or output channels for a filter):
//   float index = ((((R / 64) * (C / 8) + c / 8) * 16 + t) * 64 + R % 64) * 8 + (((c % 8) / 4) ^ ((R % 64 / 16) & 1)) * 4 + c % 4
// i.e. per (row block, 8-channel step) one contiguous 32 KiB slab [16 t][64 rows][8 channels] -- exactly the LDS image of one
// K-step of wino_mm_kernel (the 16-byte halves of a row swapped for rows 16-31 and 48-63: conflict-free ds_read_b128 fragments).
inline size_t wino_pack_floats(long rows, int c) { return (size_t)((rows + 63) / 64) * 64 * (size_t)c * 16; }

// mode 0: V = B^T d B of the 4x4 input patch of every tile (forward: d = x; data gradient: d = dz)
// mode 1: U' = G' d G'^T of the 2x2 tile itself (filter gradient: d = dz), G' = [[1,0],[.5,.5],[.5,-.5],[0,1]]
// small: 256-thread blocks of half a row block each (mode 0; fits on a CU beside a resident block of the forward product)
hipError_t wino_transform_tiles(const float* x, float* pack, int n, int h, int w, int c, int mode, hipStream_t st, bool small = false);
// U = G g G^T of every 3x3 filter, packed with rows = the product's output channels.  w is HWIO [3][3][cin][cout];
// dgrad = 0: rows = cout, k = cin (forward);  dgrad = 1: rows = cin, k = cout, taps rotated by 180 degrees (data gradient)
hipError_t wino_transform_filter(const float* w, float* pack, int cin, int cout, int dgrad, hipStream_t st);

struct WinoMMParams {
    const float* V;       // pack of the transformed input tiles: rows = tiles, channels = K
    const float* U;       // pack of the transformed filters: rows = N output channels, channels = K
    int K, N;             // reduction channels, output channels (both % 64 == 0)
    WinoGeom g;
    // forward epilogue (igemm.h EPI_FWD): v = out + bias[n]; Z = v; Y = prelu(v, alpha[n]) + R
    float* Y; float* Z; const float* R; const float* bias; const float* alpha;
    // data-gradient epilogue (EPI_DGRAD): v = out + ADD; RAW = v; DZ = v * prelu'(Zin, alpha[n % amod]);
    //   PA[mb][n] = sum over the block's rows of v * min(Zin, 0);  PB[mb][n] = sum of DZ      (one partial row per row block)
    const float* ADD; float* RAW; const float* Zin; float* DZ; float* PA; float* PB;
    int amod;
};
hipError_t wino_mm(const WinoMMParams& p, int epi, hipStream_t st);

// filter gradient: slabs[s][t][cin][cout] = sum over the tiles of split s of V_t[tile][cin] * U'_t[tile][cout] with U' = G' e G'^T of
// the 2x2 tiles of dz computed inside the kernel (dz: [n, h, w, cout] NHWC); then
// dw[3][3][cin][cout] = A'^T (sum_s slabs) A' with A'^T = [[1,1,1,0],[0,1,-1,0],[0,1,1,-1]]
int wino_wgrad_splits(int cin, int cout);      // 0: shape not supported
hipError_t wino_wgrad(const float* V, const float* dz, float* slabs, float* dw, const WinoGeom& g, int cin, int cout, hipStream_t st);

// conv algorithm switch (api.hip): 0 = direct implicit GEMM, 1 = Winograd wherever it applies, 2 = auto (the planner's rule)
void wino_set_algo(int a);
int wino_get_algo();
