// pw16.h -- streaming pointwise (1x1, stride 1) convolution of the bf16 storage mode with the batch-norm work of the graph nets
// folded into its loader and its epilogue (pw16.hip).
#pragma once
#include <hip/hip_runtime.h>

struct Pw16Params {
    const unsigned short* A;      // [M][K] bf16: the conv input (forward) / dz or the masked gradient g (data gradient)
    const unsigned short* A2;     // PRO_BWD: z of the batch norm that follows the conv, [M][K]
    const unsigned short* W;      // [N][K] bf16, k contiguous: forward = the [cout][cin] pack, data gradient = the HWIO pack [cin][cout]
    unsigned short* OUT;          // [M][N] bf16
    unsigned short* SIDE;         // [M][K]: the TRANSFORMED operand written back (the normalised activation y / the gradient dz), or NULL
    int M, K, N;
    int nrb, nct;                 // row blocks x column tiles = the grid
    const float* c0; const float* c1; const float* c2;      // loader coefficients per k: PRO_FWD scale, shift; PRO_BWD A, B, C0
    const unsigned short* ADD;    // EPI_BN / EPI_PLAIN: accumulated into the result, [M][N], or NULL
    const unsigned short* Zm;     // EPI_BN: the tensor the mask is taken from (z of the BN below, or its stored output), or NULL
    const unsigned short* Zx;     // EPI_BN: z of the BN below when the mask comes from the output (else NULL: Zm is z)
    const float* mu; const float* rs; const float* sc; const float* sh;      // EPI_BN: per output column
    float* part;                  // EPI_STATS: [nrb][3][N] (n, mean, M2);  EPI_BN: sum g, [nrb][N]
    float* pgx;                   // EPI_BN: sum g * xhat, [nrb][N]
};
enum { PW_PRO_NONE = 0, PW_PRO_FWD = 1, PW_PRO_BWD = 2 };
enum { PW_EPI_PLAIN = 0, PW_EPI_STATS = 1, PW_EPI_BN = 2 };

// true (and nrb / nct filled) when the kernel takes the shape: K in {64, 128, 256}, N a multiple of the column tile
bool pw16_plan(long M, int K, int N, int epi, Pw16Params* p);
hipError_t pw16_launch(const Pw16Params& p, int pro, int epi, hipStream_t st);
int pw16_waves(const Pw16Params& p, int pro, int epi);
