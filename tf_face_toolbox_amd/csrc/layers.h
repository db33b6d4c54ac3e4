// layers.h -- host launchers of the layer kernels (layers.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int BN_MAX_SPLITS = 512;     // row splits of the two-level per-channel reductions

hipError_t l_bn_train_stats(const float* z, const float* gamma, const float* beta, long rows, int C, float eps, float decay,
                            float* mean, float* rstd, float* scale, float* shift, float* mov_mean, float* mov_var,
                            float* part, hipStream_t st, int flags = 0);
hipError_t l_bn_infer_coef(const float* gamma, const float* beta, const float* mm, const float* mv, float eps, int C,
                           float* scale, float* shift, hipStream_t st);
hipError_t l_bn_apply(const float* z, const float* scale, const float* shift, const float* res, float* y, long rows, int C,
                      int relu, hipStream_t st, int flags = 0);
hipError_t l_relu_bwd(const float* dy, const float* y, float* g, long n, hipStream_t st, int flags = 0);
hipError_t l_bn_bwd(const float* dy, const float* ymask, const float* z, const float* gamma, const float* mean,
                    const float* rstd, const float* zsc, const float* zsf, float* gout, float* dz, float* dgamma, float* dbeta,
                    long rows, int C, float* part, hipStream_t st, int flags = 0);
hipError_t l_bn_finalize(const float* part, int splits, const float* gamma, const float* beta, int C, float eps, float decay,
                         float* mean, float* rstd, float* scale, float* shift, float* mov_mean, float* mov_var, hipStream_t st);
hipError_t l_bn_bwd_finalize(const float* pg, const float* pgx, long ld, int splits, long rows, int C, const float* gamma, const float* mean,
                             const float* rstd, float* dgamma, float* dbeta, float* coef, hipStream_t st);
hipError_t l_bn_bwd_apply(const float* g, const float* z, const float* coef, float* dz, long rows, int C, hipStream_t st, int flags = 0);
hipError_t l_maxpool_fwd(const float* x, float* y, uint8_t* idx, int n, int h, int w, int c, int ho, int wo, int pt, int pl, hipStream_t st, int flags = 0);
hipError_t l_maxpool_bwd(const float* dy, const uint8_t* idx, float* dx, int n, int h, int w, int c, int ho, int wo, int pt, int pl, hipStream_t st, int flags = 0);
hipError_t l_gap_fwd(const float* x, float* y, int n, int hw, int c, hipStream_t st, int flags = 0);
hipError_t l_gap_bwd(const float* dy, float* dx, int n, int hw, int c, hipStream_t st, int flags = 0);
hipError_t l_dropout_fwd(const float* x, float* mask, float* y, long n, float keep, uint64_t seed, hipStream_t st);
hipError_t l_scale_mask(const float* dy, const float* mask, float* dx, long n, float inv_keep, hipStream_t st);
hipError_t l_im2col_first(const float* x, float* cols, int n, int h, int w, int cin, int ks, int stride, int ho, int wo,
                          int pt, int pl, int kpad, hipStream_t st, int h16 = 0);
hipError_t l_gconv_fwd(const float* x, const float* w, float* y, int n, int h, int wd, int c, int groups, int ho, int wo,
                       int stride, int pt, int pl, hipStream_t st);
hipError_t l_gconv_dgrad(const float* dz, const float* w, float* dx, int n, int h, int wd, int c, int groups, int ho, int wo,
                         int stride, int pt, int pl, hipStream_t st);
hipError_t l_gconv_pack16(const float* w, unsigned short* wf, unsigned short* wd, int c, int groups, hipStream_t st);
hipError_t l_gconv_mfma16(const float* x, const unsigned short* wpk, float* y, int n, int h, int wd, int c, int hs, int ws,
                          int mode, int pt, int pl, hipStream_t st, int h16 = 0);
int l_gconv_bn_rows(int n, int h, int wd, int c);
hipError_t l_gconv_mfma16_bn(const float* x, const unsigned short* wpk, float* y, int n, int h, int wd, int c, int hs, int ws,
                             int mode, int pt, int pl, int bnf, float* part, float* pgx, const unsigned short* zbn, const float* mu,
                             const float* rs, const float* sc, const float* sh, hipStream_t st, const float* isc = nullptr,
                             const float* ish = nullptr, unsigned short* yside = nullptr);
int l_gconv_wgrad16_chunks(long npix, int c);
hipError_t l_gconv_wgrad16(const float* x, const float* dz, float* part, float* dw, int n, int h, int wd, int c, int groups,
                           int ho, int wo, int stride, int pt, int pl, int chunks, hipStream_t st, int h16 = 0);
int l_gconv_wgrad_chunks(long npix, int c, int gw);
hipError_t l_gconv_wgrad(const float* x, const float* dz, float* part, int n, int h, int wd, int c, int groups, int ho, int wo,
                         int stride, int pt, int pl, int chunks, hipStream_t st);
hipError_t l_act_fwd(const float* x, float* y, long n, int kind, hipStream_t st);
hipError_t l_act_bwd(const float* dy, const float* y, float* dx, long n, int kind, hipStream_t st);
// the SE residual block in one forward and two backward passes (layers.hip "SE residual block"); flags: bit 0 z / dz bf16, bit 1 activations bf16
hipError_t l_se_squeeze(const float* z, const float* scale, const float* shift, const float* mean, const float* rstd, float* sq, float* xm,
                        int n, int hw, int c, hipStream_t st, int flags);
hipError_t l_se_apply(const float* z, const float* scale, const float* shift, const float* gate, const float* res, float* out,
                      int n, int hw, int c, hipStream_t st, int flags);
hipError_t l_se_bwd_gate(const float* dy, const float* out, const float* z, const float* gamma, const float* beta, const float* mean,
                         const float* rstd, const float* gate, float* g, float* s1, float* s2, float* dgate, int n, int hw, int c,
                         hipStream_t st, int flags);
hipError_t l_se_bn_coef(const float* s1, const float* s2, const float* gate, const float* dsq, const float* xm, const float* gamma,
                        const float* mean, const float* rstd, float* dgamma, float* dbeta, float* coef, int n, int hw, int c, hipStream_t st);
hipError_t l_se_bn_apply(const float* g, const float* z, const float* coef, const float* gate, const float* dsq, float* dz,
                         int n, int hw, int c, hipStream_t st, int flags);
hipError_t l_chscale_fwd(const float* x, const float* gate, float* y, int n, int hw, int c, hipStream_t st, int h16 = 0);
hipError_t l_chscale_bwd_apply(const float* dy, const float* gate, const float* dsq, float* dx, int n, int hw, int c, float scale, hipStream_t st, int h16 = 0);
hipError_t l_chscale_bwd(const float* dy, const float* x, const float* gate, float* dx, float* dgate, int n, int hw, int c,
                         int pre_sigmoid, hipStream_t st, int h16 = 0);
hipError_t l_bcast_add(float* dx, const float* v, int n, int hw, int c, float scale, hipStream_t st);
hipError_t l_dwconv_fwd(const float* x, const float* w, float* y, int n, int h, int wd, int c, int ho, int wo, int stride, int pt, int pl, hipStream_t st, int h16 = 0);
hipError_t l_dwconv_dgrad(const float* dy, const float* w, float* dx, int n, int h, int wd, int c, int ho, int wo, int stride, int pt, int pl, hipStream_t st, int h16 = 0);
int l_dwconv_wgrad_splits(long npix, int c);
hipError_t l_dwconv_wgrad(const float* x, const float* dy, float* part, int n, int h, int wd, int c, int ho, int wo, int stride,
                          int pt, int pl, int splits, hipStream_t st, int h16 = 0);
hipError_t l_channel_gather(const float* a, const float* b, float* out, const int* table, long rows, int ca, int cb, int co, hipStream_t st, int h16 = 0);
hipError_t l_channel_gather_affine(const float* a, const float* b, float* out0, const int* table0, int co0,
                                   float* out1, const int* table1, int co1, long rows, int ca, int cb,
                                   const float* sca, const float* sfa, int relu_a, const float* scb, const float* sfb, int relu_b, hipStream_t st, int h16 = 0);
