// wgrad16.h -- the resident filter-gradient kernel of the bf16 storage mode (wgrad16.hip).
#pragma once
#include <hip/hip_runtime.h>

struct Wgrad16Params {
    const unsigned short* x;     // [n, H, W, cin] bf16
    const unsigned short* dz;    // [n, H, W, cout] bf16
    float* out;                  // S partial slabs of 9 * cin * cout floats ([tap][cin][cout]); summed by the caller
    long slab;
    int n, H, W, cin, cout;
    int S, kper;                 // ranges of the padded-slot reduction, slots per range (a multiple of 64)
    unsigned x_bytes, dz_bytes;
    int dh[9], dw[9];
    int dbg;                     // diagnostic switches (FTE_WGRAD16_DBG; 0 in production)
    unsigned long long* stamps;  // diagnostic build only
    unsigned long long magic_is, magic_pw1;      // 2^40 / ((H+1)(W+1)) + 1, 2^40 / (W+1) + 1: scalar divisions of slot indices
};

// fills S / kper / shapes when the layer is one the kernel takes (3x3, stride 1, W <= 30, channel multiples, enough K-pieces per
// block); cfg selects the instantiation
bool wgrad16_plan(int n, int h, int w, int cin, int cout, Wgrad16Params* p, int* cfg);
hipError_t wgrad16_launch(const Wgrad16Params& p, int cfg, hipStream_t st);
// the same for 1x1 / stride-1 convs (wgrad16p_kernel): slabs of cin * cout floats; cfg = channel tile (128x256, 256x128, 128x128, 64x256, 256x64, 64x128, 128x64)
bool wgrad16p_plan(int n, int h, int w, int cin, int cout, Wgrad16Params* p, int* cfg);
hipError_t wgrad16p_launch(const Wgrad16Params& p, int cfg, hipStream_t st);
