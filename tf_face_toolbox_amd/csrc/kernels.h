// kernels.h -- host launchers of the HBM-bound kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

hipError_t k_pack_weights_table(const float* params, unsigned short* w16t, const int* table, int nconv, long total, int transposed, hipStream_t st);
hipError_t k_conv_first_fwd(const float* x, const float* w, const float* bias, const float* alpha, float* z, float* y,
                            unsigned short* z16, unsigned short* y16, int n, int h, int wd, int cin, int cout, int ho, int wo, int stride, int pt, int pl, hipStream_t st);
int k_conv_first_wgrad_blocks(long npix);
hipError_t k_conv_first_wgrad(const float* x, const float* dz, const unsigned short* dz16, float* part, int n, int h, int wd, int cin, int cout, int ho, int wo,
                              int stride, int pt, int pl, int blocks, hipStream_t st);
constexpr long REDUCE_SCRATCH_FLOATS = 1 << 18;      // 1 MiB of scratch for the tall-and-narrow case of k_reduce_rows
hipError_t k_reduce_rows(const float* in, float* out, const float* bias, int bmod, long rows, long cols, int fold, float scale,
                         float* scratch, hipStream_t st, int act = 0);
// the same for two matrices of one shape in one launch (out2 may not alias out)
hipError_t k_reduce_rows2(const float* in, float* out, const float* in2, float* out2, const float* bias, int bmod, long rows, long cols,
                          int fold, float scale, float* scratch, hipStream_t st, int act = 0);
hipError_t k_sum(const float* a, long n, float scale, float* out, float* ws, bool sq, hipStream_t st);
hipError_t k_softmax_ce(const float* logits, const int32_t* labels, float* loss_rows, float* dlogits, int n, int c, int ld, float gs, hipStream_t st);
hipError_t k_to_bf16(const float* x, unsigned short* y, long n, hipStream_t st);
hipError_t k_pack_weights_bf16(const float* w, unsigned short* w16, unsigned short* w16t, int taps, int cin, int cout, hipStream_t st);
hipError_t k_focal_loss(const float* logits, const int32_t* labels, float* loss_rows, float* dlogits, int n, int c, int ld,
                        float gamma, float alpha, float gs, hipStream_t st);
hipError_t k_asoftmax(const float* s, const float* xn, const float* wn, const int32_t* labels, float lam, float* f, float* loss_rows,
                      float* G, float* rowcoef, int n, int c, int ld, float gs, hipStream_t st);
hipError_t k_asoftmax_colcoef(const float* G, const float* s, const float* wn, float* cc, int n, int c, int ld, hipStream_t st);
hipError_t k_row_norms(const float* a, float* out, int rows, int cols, int ld, hipStream_t st);
hipError_t k_col_norms(const float* a, float* out, int rows, int cols, int ld, hipStream_t st);
hipError_t k_add_scaled(float* a, const float* b, const float* rc, const float* cc, int rows, int cols, int ld, hipStream_t st);
hipError_t k_flip_w(const float* x, float* y, long rows, int w, int c, hipStream_t st);                    // rows = n * h
hipError_t k_axpby(float a, const float* x, float b, const float* y, float* out, long n, hipStream_t st);
hipError_t k_center_loss(const float* feat, const int32_t* labels, float* centers, float* loss_rows, float* dfeat,
                         int n, int d, int num_classes, float alpha, float gs, float* ws, hipStream_t st);
hipError_t k_center_update(const float* diff, const int32_t* labels, float* centers, int n, int d, int num_classes, float alpha, hipStream_t st);
hipError_t k_triplet(const float* feat, const int32_t* labels, float margin, bool soft, float lw, float* loss_rows, float* dfeat,
                     int n, int d, float* ws, hipStream_t st);
// out[m, n] = act(a[m, k] * (trans_w ? w[n][k]^T : w[k][n]) + bias) [* (mask > 0)] in one launch (kernels.hip "Small dense products")
hipError_t k_dense_small(const float* a, const float* w, const float* bias, const float* mask, float* out, int m, int n, int k,
                         bool trans_w, int act, bool bf16, hipStream_t st);
hipError_t k_momentum(float* w, float* acc, const float* g, long n, float lr, float mom, float wd, float gs, hipStream_t st);
hipError_t k_adam(float* w, float* m, float* v, const float* g, long n, float lr_t, float b1, float b2, float eps, float wd, float gs, hipStream_t st);
hipError_t k_preprocess_u8(const unsigned char* slots, float* out, int n, long slot_stride, int channels, int in_h, int in_w,
                           int crop_h, int crop_w, hipStream_t st);
