/*
 * fte.h -- C ABI of libfte.so, the MI355X (gfx950) kernel library behind
 * tf_face_toolbox_amd's data-parallel training step.
 *
 * The reference (medivhna/TF_Face_Toolbox) has NO native boundary of its own:
 * its hot path is a TensorFlow-1.x graph whose arithmetic lives in stock TF
 * ops.  Each entry point below therefore replaces one TF op
 * (or one fused group of them) at the reference call site cited beside it.
 * A maintainer binds them with ctypes; nothing here takes or
 * returns a torch type.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; nothing is
 *     allocated, freed or retained by the library; scratch comes in through
 *     (ws, ws_bytes) and fte_*_ws_bytes() says how much a call needs;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it
 *     and the call returns without synchronising;
 *   - return value: 0 = ok, otherwise a negative FTE_E* code or a positive
 *     hipError_t; no C++ exception crosses the boundary;
 *   - every tensor pointer is 16-byte aligned and every tensor smaller than 2 GiB (the kernels move 16 bytes per lane
 *     through range-checked buffer loads); a violation is an error code, never an out-of-bounds access;
 *   - activations are NHWC fp32 (the layout data.py:275-279 hands over; the
 *     reference's NHWC->NCHW transpose, nets/sphere.py:53-54, is folded away),
 *     conv weights are TF HWIO [3,3,Cin,Cout], dense weights are [in,out].
 *   - TF 'SAME' padding (pad_before = pad_total/2) -- asymmetric (0,1) for
 *     stride 2 on even sizes.
 *   - all arithmetic is fp32 with fp32 accumulation (v_mfma_f32_32x32x2_f32
 *     for the GEMM-shaped work), the reference's dtype (nets/sphere.py:35).
 */
#ifndef FTE_H_
#define FTE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FTE_OK 0
#define FTE_EINVAL (-1)      /* unsupported shape / null pointer */
#define FTE_EWORKSPACE (-2)  /* ws_bytes too small */

/* Library / build identification: "fte <version> gfx950". */
const char* fte_version(void);

/* ---------------------------------------------------------------------------
 * Operand precision of every MFMA product of the library (convolutions and dense GEMMs; process-wide, not per stream).
 *   FTE_MFMA_F32  (default): v_mfma_f32_32x32x2_f32, exact fp32 -- the reference's arithmetic (dtype=tf.float32,
 *                 nets/sphere.py:35).
 *   FTE_MFMA_BF16: operand tiles are rounded to bf16 (round-to-nearest-even) on their way into LDS and multiplied by
 *                 v_mfma_f32_32x32x16_bf16 with fp32 accumulation (16x the fp32 MFMA rate).  Tensors in HBM, epilogues,
 *                 reductions, losses and the optimizer stay fp32 -- mixed precision as in BASELINE.json config 3.
 * Returns FTE_EINVAL for any other value.
 * ------------------------------------------------------------------------- */
#define FTE_MFMA_F32 0
#define FTE_MFMA_BF16 1
int fte_set_mfma_dtype(int dtype);
int fte_get_mfma_dtype(void);

/* ---------------------------------------------------------------------------
 * Convolution algorithm of the stride-1 3x3 layers in the fp32, BN-free entry points (fte_conv3x3_* / fte_conv2d_*): the reference
 * runs them on cuDNN's Winograd algorithm (train.py:260 sets TF_ENABLE_WINOGRAD_NONFUSED=1 for every run; the layers are
 * nets/sphere.py:41-42).  Process-wide; initialised from the environment variable FTE_CONV_ALGO = direct | winograd | auto.
 *   FTE_CONV_DIRECT   implicit GEMM over the nine taps (csrc/igemm.hip)
 *   FTE_CONV_WINOGRAD F(2x2,3x3) -- and F(3x3,2x2) for the filter gradient -- wherever the kernels exist (channels % 64 == 0)
 *   FTE_CONV_AUTO     (default) Winograd for the layers where it measured faster (>= 256 channels), direct elsewhere
 * Winograd needs the workspace fte_*_ws_bytes reports UNDER THE CURRENT SETTING (transformed tiles: 64 bytes per tile and channel);
 * with less the call runs the direct algorithm.  Same results to fp32 rounding (tests/test_gpu_wino.py: <= 2e-5 of max|ref|).
 * ------------------------------------------------------------------------- */
#define FTE_CONV_DIRECT 0
#define FTE_CONV_WINOGRAD 1
#define FTE_CONV_AUTO 2
int fte_set_conv_algo(int algo);
int fte_get_conv_algo(void);
/* The algorithm the CURRENT switch selects for a 3x3 layer in the fp32 mode, op = 0 forward, 1 data gradient, 2 filter gradient:
 * FTE_CONV_DIRECT or FTE_CONV_WINOGRAD (given the workspace of the matching *_ws_bytes query). */
int fte_conv3x3_algo(int n, int h, int wd, int cin, int cout, int stride, int op);
/* Bytes of the transformed-tile pack V = B^T d B of an [n, h, wd, c] tensor (16 floats per 2x2 tile and channel, row blocks of 64
 * tiles); 0 for shapes the Winograd kernels do not take. */
size_t fte_wino_pack_bytes(int n, int h, int wd, int c);

/* Measurement hook (bench.py's roofline leg; no reference counterpart).  While enabled, every
 * launch of the MFMA kernel family is bracketed by a HIP event pair ON THE LAUNCH STREAM and its
 * algorithmic FLOPs (2*rows*N*K of that launch) are recorded.  fte_prof_enable(1) clears and starts,
 * fte_prof_enable(0) pauses, fte_prof_enable(2) resumes without clearing (sampled recording: the event pair costs a
 * queue barrier per launch, ~5 % of a step at 64 images per GPU); after a device synchronise, fte_prof_get returns record i:
 * sig = {A layout, B layout, epilogue, tile id, split count} (identifies the kernel symbol:
 * igemm_kernel<BM,BN,WM,WN,sig[0],sig[1],sig[2]>; tile 0=128x128 1=256x64 2=128x64 3=64x64),
 * flops, and the launch's duration in milliseconds. */
int fte_prof_enable(int on);
int fte_prof_count(void);
int fte_prof_get(int i, int* sig, double* flops, float* ms);
/* the GEMM shape of record i, mnk = {rows, N, K} (conv forward: rows = n*ho*wo, N = cout, K = 9*cin), and its ALGORITHMIC
 * bytes: every operand and result tensor of that launch once -- what bench.py's per-shape roofline divides by. */
int fte_prof_get_shape(int i, int* mnk, double* bytes);
/* the kernel symbol record i was dispatched to, with its template arguments as `rocprofv3 --kernel-trace` prints them minus
 * blanks (e.g. "igemm_kernel<128,128,2,2,1,0,0,0>", "igemm16_kernel<128,128,4,2,0,2,4,0>"): bench.py's per-symbol table and
 * scripts/pmc_summary.py join the launch records with the profiler's rows on this string -- no name is rebuilt by hand. */
int fte_prof_get_name(int i, char* buf, int buflen);

/* ---------------------------------------------------------------------------
 * 3x3 convolution, TF-SAME, stride 1 or 2, Cin % 32 == 0, Cout % 64 == 0
 * (every conv of nets/sphere.py:41-42,61,65,69 except the first).
 * Implicit GEMM on fp32 MFMA: M = n*ho*wo, K = 9*cin, N = cout.
 * ------------------------------------------------------------------------- */

/* Replaces Conv2D + BiasAdd + the 6-op PReLU (nets/sphere.py:29-36) + residual
 * Add (nets/sphere.py:43):   z = conv(x,w) + bias ;  y = prelu(z, alpha) + res.
 * bias, alpha, res, z may be NULL (no bias / identity / no residual / do not
 * keep the pre-activation).  z is what backward needs (sign of z, min(z,0)). */
int fte_conv3x3_fwd(const float* x, const float* w, const float* bias, const float* alpha,
                    const float* res, float* z, float* y,
                    int n, int h, int wd, int cin, int cout, int stride,
                    void* ws, size_t ws_bytes, void* stream);
/* ws is optional (NULL/0 allowed): with it, leftover or too-few output tiles run as split-K big tiles
 * plus a fused fix-up instead of small tiles (faster tails and small per-GPU shards). */
size_t fte_conv3x3_fwd_ws_bytes(int n, int h, int wd, int cin, int cout, int stride);
/* fte_conv3x3_fwd that LEAVES the transformed input tiles in `vpack` (caller-owned, fte_wino_pack_bytes(n, h, wd, cin) bytes) for the
 * filter gradient of the same layer: the forward pass and Conv2DBackpropFilter read the same B^T d B of x (one transform instead of
 * two).  Only for layers fte_conv3x3_algo reports FTE_CONV_WINOGRAD for (op 0 and op 2); with vpack != NULL the call either runs the
 * Winograd algorithm or fails (FTE_EWORKSPACE) -- it never falls back silently.  vpack = NULL: exactly fte_conv3x3_fwd. */
int fte_conv3x3_fwd_keep(const float* x, const float* w, const float* bias, const float* alpha,
                         const float* res, float* z, float* y,
                         int n, int h, int wd, int cin, int cout, int stride, float* vpack,
                         void* ws, size_t ws_bytes, void* stream);

/* Replaces Conv2DBackpropInput fused with the PReLU gradient of the PRODUCING
 * layer (tf.gradients, data_parallel.py:33):
 *     g      = conv_transpose(dz, w) + addin          (gradient wrt the input x of this conv)
 *     raw    = g                                      (optional: x is also a residual shortcut)
 *     dzprev = g * prelu'(zprev, alpha_prev)          (zprev NULL: dzprev = g)
 *     dalpha_prev[c] = sum g*min(zprev,0) ; dbias_prev[c] = sum dzprev
 * x has shape [n,h,wd,cin]; dz has the conv's output shape.  addin, raw,
 * zprev, dalpha_prev, dbias_prev may be NULL. */
int fte_conv3x3_dgrad(const float* dz, const float* w, const float* addin,
                      const float* zprev, const float* alpha_prev,
                      float* raw, float* dzprev, float* dalpha_prev, float* dbias_prev,
                      int n, int h, int wd, int cin, int cout, int stride,
                      void* ws, size_t ws_bytes, void* stream);
size_t fte_conv3x3_dgrad_ws_bytes(int n, int h, int wd, int cin, int cout, int stride);

/* Replaces Conv2DBackpropFilter:  dw[3,3,cin,cout] = sum_pixels x (*) dz
 * (deterministic split-K: partial slabs in ws, then an ordered reduction). */
int fte_conv3x3_wgrad(const float* x, const float* dz, float* dw,
                      int n, int h, int wd, int cin, int cout, int stride,
                      void* ws, size_t ws_bytes, void* stream);
size_t fte_conv3x3_wgrad_ws_bytes(int n, int h, int wd, int cin, int cout, int stride);
/* fte_conv3x3_wgrad reading the V pack fte_conv3x3_fwd_keep left for this layer instead of transforming x again (x is still the
 * layer's input and must be valid).  With vpack != NULL the call runs the Winograd algorithm or fails; vpack = NULL: exactly
 * fte_conv3x3_wgrad. */
int fte_conv3x3_wgrad_kept(const float* x, const float* dz, float* dw,
                           int n, int h, int wd, int cin, int cout, int stride, const float* vpack,
                           void* ws, size_t ws_bytes, void* stream);


/* ---------------------------------------------------------------------------
 * Generic SAME convolution (kernel 1x1 or 3x3, stride 1 or 2, no dilation) on the same MFMA kernel
 * family: the convs of nets/resnet.py:47-61 (`conv_bn_relu`), nets/resnext.py:56-62 and the pointwise
 * convs of nets/shufflenet_v2.py:100-105.  Arguments as fte_conv3x3_*, plus `ksize`.
 * ------------------------------------------------------------------------- */
int fte_conv2d_fwd(const float* x, const float* w, const float* bias, const float* alpha, const float* res,
                   float* z, float* y, int n, int h, int wd, int cin, int cout, int ksize, int stride,
                   void* ws, size_t ws_bytes, void* stream);
size_t fte_conv2d_fwd_ws_bytes(int n, int h, int wd, int cin, int cout, int ksize, int stride);
int fte_conv2d_dgrad(const float* dz, const float* w, const float* addin, const float* zprev,
                     const float* alpha_prev, float* raw, float* dzprev, float* dalpha_prev, float* dbias_prev,
                     int n, int h, int wd, int cin, int cout, int ksize, int stride,
                     void* ws, size_t ws_bytes, void* stream);
size_t fte_conv2d_dgrad_ws_bytes(int n, int h, int wd, int cin, int cout, int ksize, int stride);
int fte_conv2d_wgrad(const float* x, const float* dz, float* dw, int n, int h, int wd, int cin, int cout,
                     int ksize, int stride, void* ws, size_t ws_bytes, void* stream);
size_t fte_conv2d_wgrad_ws_bytes(int n, int h, int wd, int cin, int cout, int ksize, int stride);

/* First layer of the BN nets (7x7 stride 2, Cin = 3; nets/resnet.py:109): cols[n*ho*wo, kpad] with k ordered
 * (r, s, c) like the HWIO weight rows and zero columns from ksize*ksize*cin to kpad (kpad % 32 == 0); the
 * stem is then fte_gemm_nn / fte_gemm_tn on cols. */
int fte_im2col_first(const float* x, float* cols, int n, int h, int wd, int cin, int ksize, int stride,
                     int kpad, void* stream);
/* the same with bf16 columns (bf16 STORAGE: the stem then runs as a 1x1 conv of `kpad` input channels on the bf16-source kernels --
 * fte_conv2d_bn_fwd / fte_conv2d_fwd_s16 / fte_conv2d_wgrad16 with cin = kpad, ksize = 1 -- and its output is stored as bf16) */
int fte_im2col_first_s16(const float* x, uint16_t* cols16, int n, int h, int wd, int cin, int ksize, int stride, int kpad, void* stream);

/* The loader's image transform on DECODED uint8 images (data.py:206-223 of the reference: convert_image_dtype, resize_images
 * to in_h x in_w -- TF-1.x bilinear, align_corners = False --, random_crop to crop_h x crop_w, random_flip_left_right,
 * (x - 0.5) / 0.5; the evaluation transform of data.py:153-191 is the same with the full window and no flip).  `slots` holds
 * n slots of slot_stride bytes (a multiple of 64): a 64-byte header of int32 {mode, h0, w0, y0, x0, flip, 0...} followed by
 * the h0 x w0 x channels bytes of the image (mode 0; y0 / x0 = the crop's corner in the RESIZED image) or by the finished
 * float32 crop (mode 1).  out is [n, crop_h, crop_w, channels] float32; every value is bit-equal to the host transform
 * (tf_face_toolbox_amd/_decode_worker.py).  The random draws stay on the host (they are the workers' seeded draws). */
int fte_preprocess_u8(const uint8_t* slots, float* out, int n, long slot_stride, int channels, int in_h, int in_w, int crop_h, int crop_w,
                      void* stream);

/* ---------------------------------------------------------------------------
 * layers.batch_norm(scale=True, center=True, fused=True, decay=0.999, epsilon=1e-3) in TRAINING mode
 * (nets/resnet.py:97-99; FusedBatchNorm / FusedBatchNormGrad).  z is [rows, c] (NHWC flattened).
 *   fwd: mean/var over the rows of THIS shard (biased var normalises; the moving variance gets the unbiased
 *        one), y = [relu](gamma*(z-mean)*rstd + beta [+ res]);  mean, rstd, scale, shift are kept for backward;
 *        moving_mean / moving_var may be NULL (replicas other than tower 0, data_parallel.py:242-243).
 *   bwd: g = dy * (ymask > 0) when ymask != NULL (the ReLU that followed), dgamma = sum g*xhat, dbeta = sum g,
 *        dz = gamma*rstd*(g - dbeta/rows - xhat*dgamma/rows).
 * ws >= fte_bn_ws_bytes(c).
 * ------------------------------------------------------------------------- */
size_t fte_bn_ws_bytes(int c);
int fte_bn_train_fwd(const float* z, const float* gamma, const float* beta, const float* res, float* y,
                     float* mean, float* rstd, float* scale, float* shift, float* moving_mean, float* moving_var,
                     long rows, int c, float eps, float decay, int relu, void* ws, size_t ws_bytes, void* stream);
int fte_bn_infer_fwd(const float* z, const float* gamma, const float* beta, const float* moving_mean,
                     const float* moving_var, const float* res, float* y, float* scale, float* shift,
                     long rows, int c, float eps, int relu, void* stream);
int fte_bn_train_bwd(const float* dy, const float* ymask, const float* z, const float* gamma, const float* mean,
                     const float* rstd, float* dz, float* dgamma, float* dbeta, long rows, int c,
                     void* ws, size_t ws_bytes, void* stream);
/* The same backward pass with the ReLU mask recomputed from z: g = dy * (fma(z, scale, shift) > 0) with the scale / shift
 * the forward pass kept -- the expression the forward kernels evaluate, so the mask is the forward's bit for bit and
 * the normalised activation is not read (nor need it exist: fte_channel_gather_affine). */
int fte_bn_train_bwd_zmask(const float* dy, const float* z, const float* gamma, const float* mean, const float* rstd,
                           const float* scale, const float* shift, float* dz, float* dgamma, float* dbeta, long rows, int c,
                           void* ws, size_t ws_bytes, void* stream);
/* Residual blocks (BN -> add shortcut -> ReLU, nets/resnet.py:97-112): g = dy * (y > 0) is needed twice, by the shortcut and by
 * this BN's backward.  The reduce pass writes it to g_out as a by-product and the apply pass reads it back -- no separate
 * fte_relu_bwd launch (3 tensor passes).  Bit-identical to fte_relu_bwd followed by fte_bn_train_bwd without a mask. */
int fte_bn_train_bwd_res(const float* dy, const float* y, const float* z, const float* gamma, const float* mean, const float* rstd,
                         float* g_out, float* dz, float* dgamma, float* dbeta, long rows, int c,
                         void* ws, size_t ws_bytes, void* stream);
/* The two halves of fte_bn_train_fwd / fte_bn_infer_fwd without the apply pass: batch statistics -> mean, rstd,
 * scale = gamma*rstd, shift = beta - mean*scale (+ the moving statistics), and the inference coefficients from the
 * moving statistics.  For consumers that apply scale / shift themselves (fte_channel_gather_affine). */
int fte_bn_train_stats(const float* z, const float* gamma, const float* beta, float* mean, float* rstd, float* scale, float* shift,
                       float* moving_mean, float* moving_var, long rows, int c, float eps, float decay,
                       void* ws, size_t ws_bytes, void* stream);
int fte_bn_infer_coef(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var,
                      float* scale, float* shift, int c, float eps, void* stream);
/* g = dy * (y > 0)  (tf.nn.relu gradient, materialised where a residual shortcut needs it) */
int fte_relu_bwd(const float* dy, const float* y, float* g, long n, void* stream);

/* layers.max_pool2d(kernel 3, stride 2, 'SAME') (nets/resnet.py:115); idx keeps the window position of
 * the first maximum (uint8 per element) for the gradient. */
int fte_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int n, int h, int wd, int c, void* stream);
int fte_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, float* dx, int n, int h, int wd, int c, void* stream);
/* tf.reduce_mean over the spatial axes (nets/resnet.py:142) */
int fte_gap_fwd(const float* x, float* y, int n, int hw, int c, void* stream);
int fte_gap_bwd(const float* dy, float* dx, int n, int hw, int c, void* stream);
/* layers.dropout(keep_prob) (nets/resnet.py:152): mask = U(seed, i) < keep, y = x*mask/keep */
int fte_dropout_fwd(const float* x, float* mask, float* y, long n, float keep_prob, uint64_t seed, void* stream);
int fte_dropout_bwd(const float* dy, const float* mask, float* dx, long n, float keep_prob, void* stream);

/* ---------------------------------------------------------------------------
 * Grouped 3x3 convolution: ONE kernel for the tf.split / 32 x layers.conv2d / tf.concat of
 * nets/resnext.py:41-51.  x [n,h,wd,c], w [groups][3][3][c/groups][c/groups] (the reference's 32 HWIO
 * variables `conv2_3x3_group_<i>/weights` stacked), c/groups in {4,8,16,32}, TF-SAME, stride 1 or 2.
 * ------------------------------------------------------------------------- */
int fte_gconv3x3_fwd(const float* x, const float* w, float* y, int n, int h, int wd, int c, int groups,
                     int stride, void* stream);
/* bf16 MFMA mode: the grouped 3x3 on the matrix cores.  A 32-channel slice (32 / gw whole groups) is a dense 3x3 conv 32 -> 32
 * with a block-diagonal filter; fte_gconv3x3_pack_bf16 builds that filter (bf16, k-contiguous) for the forward pass and --
 * mirrored and transposed -- for the data gradient.  fte_gconv3x3_bf16(x, wpk_fwd, y, ..., dgrad = 0) = forward,
 * fte_gconv3x3_bf16(dz, wpk_dgrad, dx, ..., dgrad = 1) = data gradient; n, h, wd are the FORWARD layer's input size in both,
 * TF-SAME, stride 1 or 2.  Operands rounded to bf16 (RNE), fp32 accumulate, fp32 tensors.
 * wpk_*: (c / 32) * 9 * 1024 uint16.  gw = c / groups in {4, 8, 16, 32}, c % 32 == 0. */
int fte_gconv3x3_pack_bf16(const float* w, uint16_t* wpk_fwd, uint16_t* wpk_dgrad, int c, int groups, void* stream);
int fte_gconv3x3_bf16(const float* x, const uint16_t* wpk, float* y, int n, int h, int wd, int c, int stride, int dgrad, void* stream);
/* ... and the filter gradient of those layers: per slice and tap a [32 ic] x [32 oc] product over the pixels (operands rounded to
 * bf16, fragments transposed out of LDS by ds_read_b64_tr_b16), ordered partials in ws, the groups' diagonal blocks summed into
 * dw [groups][3][3][gw][gw]. */
size_t fte_gconv3x3_wgrad_bf16_ws_bytes(int n, int h, int wd, int c, int groups, int stride);
int fte_gconv3x3_wgrad_bf16(const float* x, const float* dz, float* dw, int n, int h, int wd, int c, int groups, int stride,
                            void* ws, size_t ws_bytes, void* stream);
int fte_gconv3x3_dgrad(const float* dz, const float* w, float* dx, int n, int h, int wd, int c, int groups,
                       int stride, void* stream);
int fte_gconv3x3_wgrad(const float* x, const float* dz, float* dw, int n, int h, int wd, int c, int groups,
                       int stride, void* ws, size_t ws_bytes, void* stream);
size_t fte_gconv3x3_wgrad_ws_bytes(int n, int h, int wd, int c, int groups, int stride);

/* Squeeze-excitation gate pieces (nets/shufflenet_v2.py:79-85): activations kind 0 = ReLU, 1 = sigmoid
 * (bwd takes the OUTPUT y), and the per-(image, channel) scale y = x * gate[n,c] with its gradients
 * dx = dy*gate, dgate[n,c] = sum_hw dy*x.  The two 1x1 convs on the pooled vector are fte_gemm_*. */
/* dx[n,hw,c] += v[n,c]*scale: the squeeze (spatial mean) gradient broadcast back over the map */
int fte_bcast_add(float* dx, const float* v, int n, int hw, int c, float scale, void* stream);
/* The gate's dense layers in ONE launch each (fte_gemm_* make two: split-K + reduction, 12-15 us for these sizes):
 * out[m, n] = act(a[m, k] * op(w) + bias) [* (mask > 0)], op(w) = w[k][n] (trans_w = 0: layers.fully_connected forward) or w[n][k]^T
 * (trans_w = 1: the gradient w.r.t. the layer's input); act 0 none, 1 ReLU, 2 sigmoid; bias [n] and mask [m, n] optional (mask: the ReLU
 * gradient of the layer below, tf.nn.relu's grad).  n % 32 == 0, k % 128 == 0.  Operand precision follows fte_set_mfma_dtype. */
int fte_dense_small(const float* a, const float* w, const float* bias, const float* mask, float* out, int m, int n, int k,
                    int trans_w, int act, void* stream);
int fte_act_fwd(const float* x, float* y, long n, int kind, void* stream);
int fte_act_bwd(const float* dy, const float* y, float* dx, long n, int kind, void* stream);
int fte_channel_scale_fwd(const float* x, const float* gate, float* y, int n, int hw, int c, void* stream);
/* pre_sigmoid != 0: dgate is multiplied by gate * (1 - gate), i.e. it is the gradient w.r.t. the pre-sigmoid value */
int fte_channel_scale_bwd(const float* dy, const float* x, const float* gate, float* dx, float* dgate,
                          int n, int hw, int c, int pre_sigmoid, void* stream);

/* ---------------------------------------------------------------------------
 * bf16 OPERAND COPIES (mixed precision with bf16 storage of what the MFMAs read; BASELINE.json config 3).
 * The *16 convolutions read bf16 copies of their two operands -- half the bytes per K-step through the load path, no
 * conversion in the loop -- and can write a bf16 copy of their result for the next consumer; everything else (fp32
 * accumulation, bias / PReLU / residual / PReLU-gradient epilogues, the fp32 result tensors, reductions) is unchanged.
 * Results are bit-identical to the FTE_MFMA_BF16 mode of the fp32-source entry points (same rounding, same order).
 *   fte_to_bf16            y16[i] = bf16(x[i]) (round to nearest even), n % 4 == 0
 *   fte_pack_weights_bf16  w [k*k][cin][cout] (HWIO) -> w16 (same layout: dgrad's operand) and / or w16t [k*k][cout][cin]
 *                          (forward's operand); once per optimizer step
 *   fte_conv2d_fwd16       = fte_conv2d_fwd with x16, w16t; y16 (optional) receives bf16(y)
 *   fte_conv2d_dgrad16     = fte_conv2d_dgrad with dz16, w16; dzprev16 (optional) receives bf16(dzprev)
 *   fte_conv2d_wgrad16     = fte_conv2d_wgrad with x16, dz16 (cin % 8 == 0)
 * Workspace sizes are those of the fp32-source functions.
 * ------------------------------------------------------------------------- */
int fte_to_bf16(const float* x, uint16_t* y16, long n, void* stream);
int fte_pack_weights_bf16(const float* w, uint16_t* w16, uint16_t* w16t, int ksize, int cin, int cout, void* stream);
/* every filter of a net in ONE launch per layout (a step of a 50-layer net made 36-53 pack launches of ~7 us otherwise): `table` is a
 * DEVICE array of nconv <= 64 rows {source offset in `params` (floats), destination offset in `dst` (bf16 elements, a multiple of 8),
 * taps, cin, cout, first index of this conv in the flattened walk / 4}, cin % 4 == 0 and cout % 4 == 0, total = sum of
 * taps*cin*cout.  transposed = 1 writes the [tap][cout][cin] packs (forward), 0 the HWIO packs (data gradient). */
int fte_pack_weights_bf16_table(const float* params, uint16_t* dst, const int32_t* table, int nconv, long total, int transposed, void* stream);
int fte_conv2d_fwd16(const uint16_t* x16, const uint16_t* w16t, const float* bias, const float* alpha, const float* res,
                     float* z, float* y, uint16_t* y16, int n, int h, int wd, int cin, int cout, int ksize, int stride,
                     void* ws, size_t ws_bytes, void* stream);
int fte_conv2d_dgrad16(const uint16_t* dz16, const uint16_t* w16, const float* addin, const float* zprev,
                       const float* alpha_prev, float* raw, float* dzprev, uint16_t* dzprev16, float* dalpha_prev,
                       float* dbias_prev, int n, int h, int wd, int cin, int cout, int ksize, int stride,
                       void* ws, size_t ws_bytes, void* stream);
int fte_conv2d_wgrad16(const uint16_t* x16, const uint16_t* dz16, float* dw, int n, int h, int wd, int cin, int cout,
                       int ksize, int stride, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * bf16 STORAGE (the third precision contract, next to fp32 and "bf16 operands, fp32 tensors"; BASELINE.json config 3's
 * precision, SURVEY.md section 7 step 8 "bf16 storage path, fp32 master weights").  The activations the backward pass keeps -- z, y -- and the
 * gradients that travel between layers -- dz, and the skip-path gradient `raw` -- live in HBM as bf16 ONLY: the epilogues
 * move 6-8 bytes per output element instead of 10-18.  Arithmetic is unchanged: fp32 accumulation, fp32 bias / PReLU /
 * residual / PReLU-gradient math, fp32 dalpha / dbias / filter gradients, fp32 master weights and optimizer.  Every stored
 * value is rounded exactly once, to nearest even, where it is written; every consumer sees the rounded value:
 *   y16  = bf16( prelu(acc + bias) + float(res16) )       z16 = bf16( acc + bias )
 *   g    = acc + float(addin16)        raw16 = bf16(g)     dz = g * prelu'(float(zprev16))      dzprev16 = bf16(dz)
 *   dalpha = sum g * min(float(zprev16), 0)                dbias = sum dz                       (fp32 sums of unrounded terms)
 * oracle/ops.py storage_rounding('bf16') rounds at the same points (tests/test_gpu_bf16_storage.py).
 *   fte_conv2d_fwd_s16          as fte_conv2d_fwd16 with a bf16 shortcut, bf16 z / y outputs; z32 / y32 (optional) also
 *                               receive the unrounded fp32 values (the last conv layer, whose consumer is the dense layer)
 *   fte_conv2d_dgrad_s16        as fte_conv2d_dgrad16 with bf16 skip gradient / z inputs and bf16 raw / dz outputs
 *   fte_conv3x3_first_fwd_s16   the first layer (fp32 images in, bf16 z / y out)
 *   fte_conv3x3_first_wgrad_s16 its filter gradient from a bf16 dz
 * The filter gradient of every other layer is fte_conv2d_wgrad16 (bf16 x and dz in, fp32 dw out).
 * ------------------------------------------------------------------------- */
int fte_conv2d_fwd_s16(const uint16_t* x16, const uint16_t* w16t, const float* bias, const float* alpha, const uint16_t* res16,
                       uint16_t* z16, uint16_t* y16, float* z32, float* y32, int n, int h, int wd, int cin, int cout,
                       int ksize, int stride, void* ws, size_t ws_bytes, void* stream);
int fte_conv2d_dgrad_s16(const uint16_t* dz16, const uint16_t* w16, const uint16_t* addin16, const uint16_t* zprev16,
                         const float* alpha_prev, uint16_t* raw16, uint16_t* dzprev16, float* dalpha_prev, float* dbias_prev,
                         int n, int h, int wd, int cin, int cout, int ksize, int stride, void* ws, size_t ws_bytes, void* stream);
int fte_conv3x3_first_fwd_s16(const float* x, const float* w, const float* bias, const float* alpha, uint16_t* z16, uint16_t* y16,
                              int n, int h, int wd, int cin, int cout, int stride, void* stream);
int fte_conv3x3_first_wgrad_s16(const float* x, const uint16_t* dz16, float* dw, int n, int h, int wd, int cin, int cout, int stride,
                                void* ws, size_t ws_bytes, void* stream);

/* bf16 STORAGE for the BN nets (nets/resnet.py, nets/resnext.py: BASELINE.json configs[2] "bf16").  Same contract as above: the
 * tensors between the layers are bf16 in HBM, every kernel computes in fp32 and rounds once where it stores.  `flags` says which
 * tensors of a call are bf16: FTE_S16_Z = the pre-activation side (z, and dz in backward), FTE_S16_A = the activation side (y, the
 * shortcut `res`, dy, the masked gradient g_out); the other side is fp32 (the stem, whose conv output comes from an fp32 GEMM).
 * Pointers are void*: bf16 (uint16_t) or float elements per the flags.  c % 4 == 0 and c >= 32.
 *   fte_bn_train_fwd_s16 / fte_bn_infer_fwd_s16   = fte_bn_train_fwd / fte_bn_infer_fwd
 *   fte_bn_train_bwd_s16   the three backward forms in one: g_out + y given = the residual form (fte_bn_train_bwd_res), scale + shift
 *                          given = the ReLU mask recomputed from z (fte_bn_train_bwd_zmask), neither = fte_bn_train_bwd (y = optional mask)
 *   fte_relu_bwd_s16, fte_maxpool3x3s2_{fwd,bwd}_s16, fte_gap_{fwd,bwd}_s16 (features / their gradient stay fp32)
 *   fte_gconv3x3_bf16_s16, fte_gconv3x3_wgrad_bf16_s16   the grouped 3x3 on the bf16 MFMA with bf16 x / y / dz in HBM (dw fp32) */
#define FTE_S16_Z 1
#define FTE_S16_A 2
int fte_bn_train_fwd_s16(const void* z, const float* gamma, const float* beta, const void* res, void* y,
                         float* mean, float* rstd, float* scale, float* shift, float* moving_mean, float* moving_var,
                         long rows, int c, float eps, float decay, int relu, int flags, void* ws, size_t ws_bytes, void* stream);
int fte_bn_infer_fwd_s16(const void* z, const float* gamma, const float* beta, const float* moving_mean, const float* moving_var,
                         const void* res, void* y, float* scale, float* shift, long rows, int c, float eps, int relu, int flags, void* stream);
int fte_bn_train_bwd_s16(const void* dy, const void* y, const void* z, const float* gamma, const float* mean, const float* rstd,
                         const float* scale, const float* shift, void* g_out, void* dz, float* dgamma, float* dbeta,
                         long rows, int c, int flags, void* ws, size_t ws_bytes, void* stream);
int fte_relu_bwd_s16(const uint16_t* dy16, const uint16_t* y16, uint16_t* g16, long n, void* stream);
int fte_maxpool3x3s2_fwd_s16(const uint16_t* x16, uint16_t* y16, uint8_t* idx, int n, int h, int wd, int c, void* stream);
int fte_maxpool3x3s2_bwd_s16(const uint16_t* dy16, const uint8_t* idx, uint16_t* dx16, int n, int h, int wd, int c, void* stream);
int fte_gap_fwd_s16(const uint16_t* x16, float* y, int n, int hw, int c, void* stream);
int fte_gap_bwd_s16(const float* dy, uint16_t* dx16, int n, int hw, int c, void* stream);
int fte_gconv3x3_bf16_s16(const uint16_t* x16, const uint16_t* wpk, uint16_t* y16, int n, int h, int wd, int c, int stride, int dgrad, void* stream);
int fte_gconv3x3_wgrad_bf16_s16(const uint16_t* x16, const uint16_t* dz16, float* dw, int n, int h, int wd, int c, int groups, int stride,
                                void* ws, size_t ws_bytes, void* stream);
/* ---------------------------------------------------------------------------
 * BN FUSION: the conv -> batch_norm (-> ReLU) pairs of nets/resnet.py:47-61 (`conv_bn_relu`), nets/resnext.py:34-67,
 * nets/shufflenet_v2.py:120-135.  TF runs Conv2D, FusedBatchNorm (statistics pass + normalise pass) and Relu as separate
 * kernels, and FusedBatchNormGrad re-reads dy and x for its two sums; here the PRODUCING kernel leaves those per-channel
 * sums behind, so each BN tensor makes one HBM round trip less in each direction:
 *   forward   the conv epilogue keeps (n, mean, M2) of the values it stores -- per tile row and channel, Chan-merged, no
 *             E[x^2] - E[x]^2 -- in `ws`; one finalize launch merges the tile rows in a fixed order -> mean, rstd (biased batch
 *             variance, eps inside the root), scale = gamma * rstd, shift = beta - mean * scale, moving statistics (decay, unbiased
 *             variance) exactly as fte_bn_train_stats.  The caller normalises with fte_bn_apply (or the consumer folds it in).
 *   backward  the data gradient that lands on the BN layer's OUTPUT applies the ReLU mask in its epilogue, stores the masked gradient
 *             g and leaves sum g, sum g * xhat per tile row; one finalize launch -> dgamma, dbeta and coef[3 c] = (A, B, C0) of
 *             dz = A g + B z + C0, which fte_bn_bwd_apply evaluates (one read of g and z, one write of dz).
 *             mask:  bn_scale / bn_shift given -> fma(zbn, scale, shift) > 0, the expression the forward pass evaluated (BN + ReLU);
 *                    ybn given -> ybn > 0, ybn = the stored output of BN + add + ReLU (g is then also the shortcut's gradient);
 *                    neither -> no mask (BN without activation).
 * s16 = 0: fp32 tensors (w: HWIO fp32);  s16 = 1: bf16 storage -- x / dz / addin / zbn / ybn / z / g are bf16, w is the bf16 pack
 * (forward: [tap][cout][cin], data gradient: HWIO; fte_pack_weights_bf16), and statistics / sums are taken of the ROUNDED values that
 * are stored.  Sums are fp32, deterministic (fixed merge order, no atomics).  Shapes as fte_conv2d_fwd / fte_conv2d_dgrad.
 * The grouped 3x3 twins run on the bf16 MFMA with bf16 tensors (4 / 8 / 16 / 32 channels per group, c % 32 == 0).
 * ------------------------------------------------------------------------- */
size_t fte_conv2d_bn_fwd_ws_bytes(int n, int h, int wd, int cin, int cout, int ksize, int stride);
/* in_scale / in_shift / y_side (all three or none): `x` is then the PRE-normalisation tensor of the batch norm IN FRONT of this conv
 * (conv -> BN -> ReLU -> conv chains); the operand loader applies y = relu(in_scale[k] * x + in_shift[k]) -- the expression and the
 * bf16 rounding of fte_bn_apply -- on the way to the matrix cores and writes the normalised rows to y_side, which only the filter
 * gradient reads: the fte_bn_apply launch between the two convs and its pass over the tensor disappear.  Taken by the streaming
 * pointwise kernel only: fte_conv2d_bn_fwd_folds() says whether a shape qualifies (bf16 storage, 1x1, stride 1, cin 64 / 128 / 256). */
int fte_conv2d_bn_fwd_folds(int n, int h, int wd, int cin, int cout, int ksize, int stride, int s16);
int fte_conv2d_bn_fwd(const void* x, const void* w, void* z, const float* gamma, const float* beta, float* mean, float* rstd,
                      float* scale, float* shift, float* moving_mean, float* moving_var, float eps, float decay,
                      const float* in_scale, const float* in_shift, void* y_side,
                      int n, int h, int wd, int cin, int cout, int ksize, int stride, int s16, void* ws, size_t ws_bytes, void* stream);
size_t fte_conv2d_dgrad_bn_ws_bytes(int n, int h, int wd, int cin, int cout, int ksize, int stride);
int fte_conv2d_dgrad_bn(const void* dz, const void* w, const void* addin, const void* zbn, const void* ybn,
                        const float* gamma, const float* mean, const float* rstd, const float* bn_scale, const float* bn_shift,
                        void* g, float* dgamma, float* dbeta, float* coef,
                        int n, int h, int wd, int cin, int cout, int ksize, int stride, int s16, void* ws, size_t ws_bytes, void* stream);
/* y = [relu](scale[c] * z + shift[c] [+ res]) with given coefficients; flags: FTE_S16_Z (z bf16), FTE_S16_A (res, y bf16) */
int fte_bn_apply(const void* z, const float* scale, const float* shift, const void* res, void* y, long rows, int c, int relu, int flags, void* stream);
/* dz = coef[c] * g + coef[C + c] * z + coef[2C + c], g already masked; flags: FTE_S16_Z (z, dz bf16), FTE_S16_A (g bf16) */
int fte_bn_bwd_apply(const void* g, const void* z, const float* coef, void* dz, long rows, int c, int flags, void* stream);
size_t fte_gconv3x3_bn_ws_bytes(int n, int h, int wd, int c, int stride);
/* (in_scale / in_shift / y_side as above; stride 1 only) */
int fte_gconv3x3_bn_fwd_bf16_s16(const uint16_t* x16, const uint16_t* wpk, uint16_t* z16, const float* gamma, const float* beta,
                                 float* mean, float* rstd, float* scale, float* shift, float* moving_mean, float* moving_var,
                                 float eps, float decay, const float* in_scale, const float* in_shift, uint16_t* y_side,
                                 int n, int h, int wd, int c, int stride, void* ws, size_t ws_bytes, void* stream);
int fte_gconv3x3_dgrad_bn_bf16_s16(const uint16_t* dz16, const uint16_t* wpk_dgrad, const uint16_t* zbn16, const float* gamma, const float* mean,
                                   const float* rstd, const float* bn_scale, const float* bn_shift, uint16_t* g16, float* dgamma, float* dbeta,
                                   float* coef, int n, int h, int wd, int c, int stride, void* ws, size_t ws_bytes, void* stream);

/* ShuffleNet-v2's layers on bf16 tensors (nets/shufflenet_v2.py): depthwise 3x3 forward / data gradient / filter gradient (fp32 filter
 * and dw), the channel gather and the gather with batch norm folded in (sources and results bf16, tables / scale / shift unchanged),
 * and the statistics-only pass of a folded batch norm (`flags` as above: FTE_S16_Z = z is bf16). */
/* the SE gate on bf16 tensors: y = x * gate; dgate = sum_hw dy * x (reduction only); dx = dy * gate + dsq * scale in ONE pass -- written
 * once, rounded once (gate, dgate, dsq are [n, c] fp32) */
int fte_channel_scale_fwd_s16(const uint16_t* x16, const float* gate, uint16_t* y16, int n, int hw, int c, void* stream);
int fte_channel_scale_bwd_s16(const uint16_t* dy16, const uint16_t* x16, const float* gate, float* dgate, int n, int hw, int c,
                              int pre_sigmoid, void* stream);
int fte_channel_scale_bwd_apply_s16(const uint16_t* dy16, const float* gate, const float* dsq, uint16_t* dx16, int n, int hw, int c,
                                    float scale, void* stream);
/* The SE residual block  z -> BN (no activation) -> y * gate(mean_hw y) -> + shortcut -> ReLU  (nets/resnet.py:63-92 with the gate of
 * nets/shufflenet_v2.py:79-85) in ONE pass over the tensor forward and TWO backward; the BN output and the gated tensor never exist
 * in HBM.  flags: bit 0 (FTE_S16_Z) z / dz are bf16, bit 1 (FTE_S16_A) shortcut / out / dy / g are bf16; gate, sq, xm, s1, s2, dgate, dsq
 * are [n, c] fp32; scale / shift / mean / rstd are the BN layer's (fte_conv2d_bn_fwd, fte_bn_train_stats, fte_bn_infer_coef).
 *   fte_se_squeeze       sq = scale * mean_hw(z) + shift (= mean_hw of the BN output); xm (optional) = (mean_hw(z) - mean) * rstd
 *   fte_se_apply_fwd     out = relu(fma(z, scale, shift) * gate + shortcut)
 *   fte_se_bwd_gate      g = dy * (out > 0) (the shortcut's gradient; stored, and the stored value is what is summed);
 *                        s1 = sum_hw g, s2 = sum_hw g * xhat, dgate = (gamma * s2 + beta * s1) * gate * (1 - gate) (w.r.t. the pre-sigmoid)
 *   fte_se_bn_bwd_coef   dsq = gradient of the squeeze (from the gate's dense layers): the batch-norm backward of
 *                        dy_bn = g * gate + dsq / hw from the per-image sums -- dbeta = sum_n (gate * s1 + dsq), dgamma = sum_n (gate * s2 +
 *                        dsq * xm), coef[3][c] of dz = A dy_bn + B z + C0 (FusedBatchNormGrad, nets/resnet.py:97-99)
 *   fte_se_bn_bwd_apply  dz = A * (g * gate + dsq / hw) + B * z + C0 */
int fte_se_squeeze(const void* z, const float* scale, const float* shift, const float* mean, const float* rstd, float* sq, float* xm,
                   int n, int hw, int c, int flags, void* stream);
int fte_se_apply_fwd(const void* z, const float* scale, const float* shift, const float* gate, const void* shortcut, void* out,
                     int n, int hw, int c, int flags, void* stream);
int fte_se_bwd_gate(const void* dy, const void* out, const void* z, const float* gamma, const float* beta, const float* mean,
                    const float* rstd, const float* gate, void* g, float* s1, float* s2, float* dgate, int n, int hw, int c, int flags, void* stream);
int fte_se_bn_bwd_coef(const float* s1, const float* s2, const float* gate, const float* dsq, const float* xm, const float* gamma,
                       const float* mean, const float* rstd, float* dgamma, float* dbeta, float* coef, int n, int hw, int c, void* stream);
int fte_se_bn_bwd_apply(const void* g, const void* z, const float* coef, const float* gate, const float* dsq, void* dz,
                        int n, int hw, int c, int flags, void* stream);
int fte_dwconv3x3_fwd_s16(const uint16_t* x16, const float* w, uint16_t* y16, int n, int h, int wd, int c, int stride, void* stream);
int fte_dwconv3x3_dgrad_s16(const uint16_t* dy16, const float* w, uint16_t* dx16, int n, int h, int wd, int c, int stride, void* stream);
int fte_dwconv3x3_wgrad_s16(const uint16_t* x16, const uint16_t* dy16, float* dw, int n, int h, int wd, int c, int stride,
                            void* ws, size_t ws_bytes, void* stream);
int fte_channel_gather_s16(const uint16_t* a, const uint16_t* b, uint16_t* out, const int32_t* table, long rows, int ca, int cb, int co, void* stream);
int fte_channel_gather_affine_s16(const uint16_t* a, const uint16_t* b, uint16_t* out, const int32_t* table, int co,
                                  uint16_t* out1, const int32_t* table1, int co1, long rows, int ca, int cb,
                                  const float* scale_a, const float* shift_a, int relu_a,
                                  const float* scale_b, const float* shift_b, int relu_b, void* stream);
int fte_bn_train_stats_s16(const void* z, const float* gamma, const float* beta, float* mean, float* rstd, float* scale, float* shift,
                           float* moving_mean, float* moving_var, long rows, int c, float eps, float decay, int flags,
                           void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * ShuffleNet-v2 (nets/shufflenet_v2.py).  Depthwise 3x3, TF-SAME, stride 1 or 2: the DepthwiseConv2dNative half of
 * layers.separable_conv2d (:98,104; the pointwise half is fte_conv2d_* with ksize 1).  x [n,h,wd,c], w [3,3,c].
 * HBM-bound (9 MAC per element).
 * fte_channel_gather: out[row,k] = table[k] < 0 ? 0 : (table[k]>>16 ? b : a)[row, table[k] & 0xffff] -- one kernel
 * for _channel_split (:60-64), tf.concat + _channel_shuffle (:66-77,112-113) and their gradients; the
 * concatenated tensor is never materialised.  table is a device int32[co].
 * ------------------------------------------------------------------------- */
int fte_dwconv3x3_fwd(const float* x, const float* w, float* y, int n, int h, int wd, int c, int stride, void* stream);
int fte_dwconv3x3_dgrad(const float* dy, const float* w, float* dx, int n, int h, int wd, int c, int stride, void* stream);
int fte_dwconv3x3_wgrad(const float* x, const float* dy, float* dw, int n, int h, int wd, int c, int stride,
                        void* ws, size_t ws_bytes, void* stream);
size_t fte_dwconv3x3_wgrad_ws_bytes(int n, int h, int wd, int c, int stride);
int fte_channel_gather(const float* a, const float* b, float* out, const int32_t* table, long rows,
                       int ca, int cb, int co, void* stream);
/* The gather with batch norm applied to a source on the way: a source whose scale is not NULL contributes
 * [relu](fma(src[row,ch], scale[ch], shift[ch])).  conv3_1x1's BN + ReLU output (:110) and the stride-2 shortcut's (:96-101)
 * feed only the concat / shuffle / split: they are normalised inside the gather and never written to HBM.
 * out1 / table1 / co1 (optional, NULL / NULL / 0): a second output of the same two sources in the same launch -- the two
 * halves a block hands to the next one (forward), the gradients of the two sources (backward). */
int fte_channel_gather_affine(const float* a, const float* b, float* out, const int32_t* table, int co,
                              float* out1, const int32_t* table1, int co1, long rows, int ca, int cb,
                              const float* scale_a, const float* shift_a, int relu_a,
                              const float* scale_b, const float* shift_b, int relu_b, void* stream);

/* ---------------------------------------------------------------------------
 * First conv of the net (Cin = 1 or 3; nets/sphere.py:57, nets/shufflenet_v2.py:148-149): K = 9*Cin
 * is too short for a GEMM -- HBM-bound direct convolution, fused bias+PReLU (both optional).
 * cout = 64, or 32 for the 24-wide ShuffleNet-v2 stem stored 32 channels wide (w is [3,3,cin,cout]).
 * ------------------------------------------------------------------------- */
int fte_conv3x3_first_fwd(const float* x, const float* w, const float* bias, const float* alpha,
                          float* z, float* y, int n, int h, int wd, int cin, int cout,
                          int stride, void* stream);
int fte_conv3x3_first_wgrad(const float* x, const float* dz, float* dw,
                            int n, int h, int wd, int cin, int cout, int stride,
                            void* ws, size_t ws_bytes, void* stream);
size_t fte_conv3x3_first_wgrad_ws_bytes(int n, int h, int wd, int cin, int cout, int stride);

/* ---------------------------------------------------------------------------
 * Dense layers (MatMul + BiasAdd and their gradients: nets/sphere.py:73-74,
 * 86-90).  Row-major fp32; k % 32 == 0 for nn/nt, n % 64 == 0 everywhere.
 * ------------------------------------------------------------------------- */

/* y[m,n] = x[m,k] @ w[k,n] (+ bias[n]) */
int fte_gemm_nn(const float* x, const float* w, const float* bias, float* y,
                int m, int n, int k, void* ws, size_t ws_bytes, void* stream);
/* the same followed by an activation in the same pass over y: act 0 none, 1 ReLU, 2 sigmoid (the squeeze-excitation gate's two dense
 * layers, nets/shufflenet_v2.py:79-85; y = act(x @ w + bias)) */
int fte_gemm_nn_act(const float* x, const float* w, const float* bias, float* y,
                    int m, int n, int k, int act, void* ws, size_t ws_bytes, void* stream);
/* dx[m,k] = dy[m,n] @ w[k,n]^T, with the same fused PReLU-gradient epilogue as
 * fte_conv3x3_dgrad (zprev has dx's shape; alpha_prev has `amod` entries and
 * column j uses alpha_prev[j % amod] -- the flattened H*W*C feature map that
 * feeds nets/sphere.py:72-74). */
int fte_gemm_nt(const float* dy, const float* w, const float* zprev, const float* alpha_prev,
                int amod, float* raw, float* dx, float* dalpha_prev,
                int m, int n, int k, void* ws, size_t ws_bytes, void* stream);
/* dw[k,n] = x[m,k]^T @ dy[m,n] */
int fte_gemm_tn(const float* x, const float* dy, float* dw,
                int m, int n, int k, void* ws, size_t ws_bytes, void* stream);
size_t fte_gemm_ws_bytes(int m, int n, int k);

/* ---------------------------------------------------------------------------
 * Loss heads
 * ------------------------------------------------------------------------- */

/* Replaces SparseSoftmaxCrossEntropyWithLogits + mean + its gradient
 * (tf.losses.sparse_softmax_cross_entropy, nets/sphere.py:109):
 *   loss_rows[i] = -log softmax(logits[i,:c])[labels[i]]
 *   dlogits[i,j] = (softmax - onehot) * grad_scale   (columns c..ld-1 get 0)
 * logits/dlogits are [n, ld] with ld >= c (padded classifier width). */
int fte_softmax_ce_fwd_bwd(const float* logits, const int32_t* labels, float* loss_rows,
                           float* dlogits, int n, int c, int ld, float grad_scale, void* stream);

/* focal_loss (loss.py:18-27; note the reference's swapped-looking defaults gamma = 1.0, alpha = 2.0 are kept as named):
 * loss_rows[i] = gamma * (1 - p_y)^alpha * CE_i, dlogits = grad_scale * d(loss_rows[i])/dlogits (through BOTH the
 * cross-entropy and the softmax-score factor, as tf.gradients does).  alpha >= 1.  Same layout rules as above. */
int fte_focal_loss_fwd_bwd(const float* logits, const int32_t* labels, float* loss_rows, float* dlogits,
                           int n, int c, int ld, float gamma, float alpha, float grad_scale, void* stream);

/* A-softmax (SphereFace, m = 4; README.md:14,19 claims it, the code is not in
 * the reference tree -- SURVEY.md Appendix A.9).  s = x @ w is the raw dot
 * product [n, ld]; xn = |x_i| (n), wn = |W_j| (c).  Produces the margin logits
 * f (optional), the per-row loss, G = dLoss/ds (the matrix that feeds the two
 * gradient GEMMs), rowcoef (dx += rowcoef_i * x_i) and, via
 * fte_asoftmax_colcoef, colcoef (dw[:,j] += colcoef_j * w[:,j]). */
int fte_asoftmax_fwd_bwd(const float* s, const float* xn, const float* wn, const int32_t* labels,
                         float lambda, float* f, float* loss_rows, float* G, float* rowcoef,
                         int n, int c, int ld, float grad_scale, void* stream);
int fte_asoftmax_colcoef(const float* G, const float* s, const float* wn, float* colcoef,
                         int n, int c, int ld, void* stream);
/* out[i] = sqrt(sum_j a[i,j]^2) over rows of [rows, ld] (cols used) */
int fte_row_norms(const float* a, float* out, int rows, int cols, int ld, void* stream);
/* out[j] = sqrt(sum_i a[i,j]^2) over columns */
int fte_col_norms(const float* a, float* out, int rows, int cols, int ld, void* stream);
/* The flip-averaged inference path (nets/sphere.py:97-101, evaluate.py:62-63): y[n,h,w',c] = x[n,h,wd-1-w',c] replaces
 * tf.reverse(images, axis=[2]) (x != y; 16-byte aligned when c % 4 == 0) and out = a*x + b*y the mean of the two embeddings
 * (a = b = 0.5; out may alias x or y). */
int fte_flip_width(const float* x, float* y, int n, int h, int wd, int c, void* stream);
int fte_axpby(float a, const float* x, float b, const float* y, float* out, long n, void* stream);
/* a[i,j] += rc[i] * b[i,j]   (rc NULL -> skip) ;  a[i,j] += cc[j] * b[i,j]  (cc NULL -> skip) */
int fte_add_scaled_rows_cols(float* a, const float* b, const float* rc, const float* cc,
                             int rows, int cols, int ld, void* stream);

/* center loss (loss.py:29-45): loss_rows[i] = sum_j (f_ij - c_{y_i} j)^2 (caller takes the mean over n*d);
 * dfeat = 2(f - c_y)*grad_scale; then centers[y] -= (1-alpha)(c_y - f), duplicates accumulating
 * (scatter_sub).  Every gather is served from the centers as they were BEFORE the update
 * (loss.py:37 before :39).  In place on `centers` [num_classes, d]; ws >= n*d floats, and after the call
 * ws[0 : n*d] holds diff = f - c_y (what fte_center_scatter_update consumes).  alpha == 1 evaluates loss and gradient
 * only (no update launch).  A label outside [0, num_classes) gives a NaN loss / gradient row and no update -- never an
 * out-of-bounds access. */
int fte_center_loss_fwd_bwd_update(const float* feat, const int32_t* labels, float* centers,
                                   float* loss_rows, float* dfeat, int n, int d, int num_classes, float alpha,
                                   float grad_scale, void* ws, size_t ws_bytes, void* stream);
/* the scatter_sub half of loss.py:38-39 on its own: centers[labels[i]] += (1 - alpha) * diff[i] for i < n, duplicates
 * accumulating.  Used by the opt-in replica reconciliation (DataParallel(sync_centers=True)): every rank evaluates the loss
 * with alpha = 1, the ranks all-gather their (labels, diff) rows and each applies ALL of them, so the replicas keep ONE
 * table equal to the single-tower update of the global batch; the default reproduces the reference's per-tower tables. */
int fte_center_scatter_update(const float* diff, const int32_t* labels, float* centers, int n, int d, int num_classes,
                              float alpha, void* stream);

/* batch-hard triplet (loss.py:47-78): per-sample loss [n] and d(sum w_i*loss_i)/dfeat.
 * soft_margin != 0 selects softplus(pos - neg) (margin=None in the reference, loss.py:74-75) and `margin` is ignored; otherwise
 * max(0, pos - neg + margin) for ANY margin, negative ones included (loss.py:76-77).  ws >= 3*n*n floats. */
int fte_batch_hard_triplet_fwd_bwd(const float* feat, const int32_t* labels, float margin, int soft_margin,
                                   float loss_weight, float* loss_rows, float* dfeat,
                                   int n, int d, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * Flat-arena reductions and optimizers (ApplyMomentum / ApplyAdam per variable,
 * data_parallel.py:65-69,191-196; L2Loss + AddN, nets/net_base.py:105).
 * ------------------------------------------------------------------------- */

/* out[j] = scale * sum_{r<rows} in[r*cols + j]  (+ bias[j % bmod] when bias != NULL).
 * If fold > 1, column j of `out` (cols/fold of them) sums in[r, j + t*(cols/fold)] over t too. */
int fte_reduce_rows(const float* in, float* out, const float* bias, int bmod,
                    long rows, long cols, int fold, float scale, void* stream);
/* out[0] = scale * sum_i a[i]^2   (ws >= 1024 floats) */
int fte_sumsq(const float* a, long n, float scale, float* out, void* ws, size_t ws_bytes, void* stream);
/* out[0] = scale * sum_i a[i]     (ws >= 1024 floats) */
int fte_sum(const float* a, long n, float scale, float* out, void* ws, size_t ws_bytes, void* stream);

/* acc = mom*acc + (gscale*g + wd*w) ; w -= lr*acc      (MomentumOptimizer, Appendix A.7) */
int fte_momentum_update(float* w, float* acc, const float* g, long n,
                        float lr, float mom, float wd, float gscale, void* stream);
/* TF AdamOptimizer (epsilon outside the bias correction); t = 1-based step */
int fte_adam_update(float* w, float* m, float* v, const float* g, long n,
                    float lr, float b1, float b2, float eps, float wd, float gscale, int t, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FTE_H_ */
