// pgemm.h -- persistent fp32 MFMA GEMM for plain row-major products (pgemm.hip): C[M][N] = A[M][K] * B (+ ADD).
#pragma once
#include <hip/hip_runtime.h>

#include "igemm.h"

struct PgemmParams {
    const float* A; const float* B; float* C; const float* ADD;      // ADD (optional): same layout as C
    int M, N, K;             // N % 64 == 0, K % 32 == 0
    int lda, ldb, ldc;       // floats; B is [K][ldb] (BL_KN) or [N][ldb] (BL_NK)
    unsigned a_bytes, b_bytes, c_bytes;
    int tiles;               // filled by pgemm_launch
    int stagger;             // start delay per SIMD wave slot, in units of 64 cycles (de-phases co-resident blocks)
    int abl;                 // ablation bits (FTE_PGEMM_ABL; 0 in production)
};
bool pgemm_handles(const PgemmParams& p);
hipError_t pgemm_launch(PgemmParams p, int bl, hipStream_t st);
