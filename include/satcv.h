/*
 * satcv.h -- C ABI of the MI355X-native U-Net / ASPP tile pipeline (libsatcv.so).
 *
 * The reference (mjevans26/Satellite_ComputerVision) has no FFI for this path:
 * every tensor op of utils/model_tools.py is delegated to tf.keras layers.  Each
 * entry point below therefore replaces one Keras layer call site (cited) and is
 * what the reference-side binding would call instead of TensorFlow (see
 * INTEGRATION.md for the ctypes stub).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++/torch types.
 *   - every function returns int: 0 = SATCV_OK, <0 = error; satcv_last_error()
 *     returns a thread-local message.
 *   - all tensor pointers are DEVICE pointers owned by the caller; the library
 *     never allocates or frees tensor memory and never synchronises the stream.
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous.
 *   - activations are NHWC (channels contiguous), storage type `dtype`
 *     (SATCV_F32 or SATCV_BF16), channel counts padded to a multiple of 16.
 *     Accumulation, BN statistics, losses, gradients of weights and the
 *     optimizer are always fp32.
 */
#ifndef SATCV_H
#define SATCV_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Per-channel statistics rows (BatchNorm forward sum / sum-of-squares, backward sum g / sum g*xhat) are accumulated in DOUBLE:
 * the replica rows are filled with atomics whose order varies from run to run, and fp32 rounding differences in the resulting
 * scale/shift flip ReLU masks (a batch-64 U-Net's gradient moved by 2e-3 between two runs on the same batch).  In double the
 * order only matters at 1e-16, so the fp32 quantities derived from the rows are reproducible. */
typedef double satcv_stat_t;

enum { SATCV_OK = 0, SATCV_ERR_INVALID = -1, SATCV_ERR_HIP = -2, SATCV_ERR_UNSUPPORTED = -3 };
enum { SATCV_F32 = 0, SATCV_BF16 = 1,
       SATCV_FP8 = 2 /* OCP e4m3fn storage, inference only: ingest, pack_weights, conv2d_igemm (pipelined kernel), maxpool,
                        affine_requant, head_fwd */,
       SATCV_FP8X = 3 /* same NHWC e4m3 activations, but weights packed in 16-channel granules ([tap][K/16][Npad][16]) and
                         the convolution on the block-scaled K=64 MFMA (2x the bf16 rate); needs channels % 64 == 0.
                         Only pack_weights and conv2d_igemm take it. */,
       SATCV_F64 = 4 /* the statistics rows (satcv_stat_t); only satcv_allreduce takes it */ };
/* rows of replicated per-channel accumulators (sum rows in order to consume) */
#define SATCV_STAT_ROWS 32

const char* satcv_version(void);
const char* satcv_last_error(void);
/* out[0]=CU count, out[1]=LDS bytes/CU, out[2]=wave size, out[3]=gfx arch number (950) */
int satcv_device_info(int32_t* out4);

/* ---------------------------------------------------------------- data movement
 * f32 NHWC (n,h,w,c) -> dtype NHWC (n,h,w,cpad), zero-filling channels [c,cpad).
 * Replaces the implicit float32 feed of Model.predict/fit (utils/prediction_tools.py:152). */
int satcv_ingest_nhwc(const float* src, void* dst, int64_t npix, int32_t c, int32_t cpad,
                      int32_t dtype, void* stream);
/* same with a multiplier (the fp8 path feeds x/q_in so that the reflectances use the upper e4m3 range) */
int satcv_ingest_nhwc_scaled(const float* src, void* dst, int64_t npix, int32_t c, int32_t cpad, float mul, int32_t dtype, void* stream);
/* planar CHW (c,h,w) per tile, any of u8/u16/f32 (src_kind 0/1/2), scaled by `scale`
 * -> dtype NHWC (n,h,w,cpad).  Mirrors the CHW->HWC + rescale of utils/processing.py:544-613. */
int satcv_ingest_chw(const void* src, int32_t src_kind, float scale, void* dst, int32_t n,
                     int32_t c, int32_t h, int32_t w, int32_t cpad, int32_t dtype, void* stream);

/* Keras HWIO fp32 master kernel (kh,kw,cin,cout) -> packed operand images.
 *   fwd   : [kh*kw][cin_pad/8][cout_pad][8]   (B operand of the forward implicit GEMM)
 *   dgrad : [kh*kw][cout_pad8/8][cin_pad32][8] spatially flipped, in/out swapped
 * transposed=1: source is a Conv2DTranspose kernel (kh,kw,cout,cin) (utils/model_tools.py:306);
 *   fwd   : [1][cin_pad/8][kh*kw*cout][8]  (N index = (i*kw+j)*cout + o)
 *   dgrad : [1][kh*kw*cout/8][cin_pad32][8] (K index = (i*kw+j)*cout + o)
 * Either destination may be NULL. */
int satcv_pack_weights(const float* src, void* dst_fwd, void* dst_dgrad, int32_t kh, int32_t kw,
                       int32_t cin, int32_t cout, int32_t cin_pad, int32_t transposed,
                       int32_t dtype, void* stream);
/* All layers in one launch: a DEVICE array of jobs (mode 0 conv fwd, 1 conv dgrad, 2 transposed-conv fwd, 3 transposed-conv dgrad;
 * kpad / npad = K and N paddings of the image exactly as satcv_pack_weights derives them) and the exclusive prefix sum of
 * satcv_pack_job_items() over the jobs.  satcv_pack_job_items() is the job's number of 16-byte output items ROUNDED UP TO 256: a block of
 * 256 consecutive items then belongs to one job and the kernel looks its job up once per block; total_items must be the sum of those values
 * (SATCV_ERR_INVALID otherwise). */
typedef struct { const float* src; void* dst; int32_t mode, taps, cin, cout, kpad, npad; } satcv_pack_job;
int64_t satcv_pack_job_items(const satcv_pack_job* job);
int satcv_pack_weights_batched(const satcv_pack_job* jobs_dev, const int64_t* prefix_dev, int32_t njobs, int64_t total_items, int32_t dtype,
                               void* stream);

/* ------------------------------------------------------------ implicit-GEMM conv
 * One descriptor drives Conv2D forward, its data gradient, Conv2DTranspose(k==s)
 * forward (depth-to-space store) and its data gradient (space-to-depth load).
 * Replaces layers.Conv2D (utils/model_tools.py:178,312,315,405,544-549) and
 * layers.Conv2DTranspose (:306), plus the input-side BatchNormalization+ReLU of the
 * PREVIOUS layer (:179-180, :308-309) which is applied while staging the tile. */
typedef struct satcv_conv_desc {
  const void* x0;          /* source 0, NHWC (n, hin, win, c0)                       */
  const void* x1;          /* optional source 1 (concat along channels), or NULL      */
  int32_t c0, c1;          /* channels of each source (multiples of 16); K = c0+c1     */
  const float* in_scale;   /* optional per-input-channel affine (+ReLU) applied on load */
  const float* in_shift;
  int32_t in_relu;
  const void* w;           /* packed weights (satcv_pack_weights)                     */
  const float* bias;       /* [cstat] or NULL                                         */
  void* y;                 /* output NHWC, channel stride ldy                         */
  int32_t ldy;
  satcv_stat_t* stats;     /* optional [SATCV_STAT_ROWS][2][stats_ld] sum / sum-of-squares
                              of the stored outputs (atomically accumulated)          */
  int32_t stats_ld;
  int32_t n, h, w_;        /* GEMM pixel grid (= output grid, or input grid for mode_out=1) */
  int32_t cout;            /* GEMM N extent (valid columns)                           */
  int32_t cout_pad;        /* N stride of the packed weights                          */
  int32_t kh, kw, dil;     /* taps and dilation ('same' zero padding)                 */
  int32_t mode_in;         /* 0: plain; 1: space-to-depth gather by factor f          */
  int32_t mode_out;        /* 0: plain; 1: depth-to-space scatter by factor f         */
  int32_t f;
  int32_t cstat;           /* modulus mapping N index -> bias/stats channel            */
  int32_t out_relu;        /* clamp outputs at 0 before storing                        */
  int32_t dtype;
  int32_t accumulate;      /* 1: y += result instead of y = result (fan-out of a tensor into several convs);
                            * 2: y = ReLU(y + result), the result rounded to the storage type first (residual join written in place
                            *    over the shortcut; inference) */
  /* strided convolution (ResNet-style, symmetric zero padding dil*(k-1)/2): output grid (n,h,w_), input grid
   * (n,hin,win) with h = (hin-1)/stride+1.  stride 0/1 = dense.  Generic kernel only. */
  int32_t stride, hin, win;
  /* optional per-channel multiplier of the accumulator (indexed like bias): y = act(acc*out_scale + bias).  The folded
   * inference path puts BatchNorm and the fp8 quantisation scales here (q_in*w_scale*bn_scale/q_out). */
  const float* out_scale;
  /* optional fused max-pool (window == stride == pool_f, 'valid') of the STORED outputs, written to pool_y with channel stride
   * pool_ld: layers.MaxPooling2D (utils/model_tools.py:281) of an encoder block in the folded inference graph.  Pipelined kernel
   * only (satcv_conv2d_igemm_pipelined), mode_out 0, h and w_ divisible by pool_f. */
  void* pool_y; int32_t pool_ld, pool_f;
  /* optional: the REDUCE pass of the BatchNormalization backward (satcv_bn_bwd_reduce, utils/model_tools.py:179-180 differentiated) of
   * the layer whose activation gradient this launch writes (the launch is a data gradient: y = dL/d act), done in the epilogue while
   * the tile is on chip.  With bst_y set, `stats` receives  sum g  and  sum g * xhat  in the layout satcv_bn_bwd_finalize reads
   * (instead of sum / sum of squares), where g = y as stored, zeroed where bst_scale*v + bst_shift <= 0 (unless bst_relu == 0),
   * xhat = (v - bst_mean) * bst_rstd and v = that layer's raw convolution output at the same pixel and channel: bst_y (channel
   * stride bst_ld) for channels < bst_split, bst_y1 (stride bst_ld1, channel - bst_split) above it -- bst_y1 NULL: one source.
   * Pipelined kernel, bf16, plain output mode, no accumulate / out_relu / pool_y, and only when the map is a whole number of the
   * kernel's tiles: satcv_conv2d_igemm_pipelined() answers 0 otherwise and the caller keeps the separate reduce launch. */
  const void* bst_y; const void* bst_y1; int32_t bst_ld, bst_ld1, bst_split;
  const float* bst_scale; const float* bst_shift; const float* bst_mean; const float* bst_rstd; int32_t bst_relu;
  /* which tile family may serve this launch: 0 = the library option igemm_m16 decides (default 1: the 16x16x32 tiles only for launches that
   * write statistics, so that inference results are bit-identical across batch splits); 2 = every eligible launch may run on the 16x16x32 /
   * persistent tiles, whose choice depends on the launch's tile count -- what a TRAINING plan asks for, per launch (round 6: this used to be
   * a process-global option toggled around the plan's steps).  Option igemm_m16 = 0 (SATCV_M16=0) still turns those tiles off. */
  int32_t tile_policy;
} satcv_conv_desc;
int satcv_conv2d_igemm(const satcv_conv_desc* d, void* stream);
/* 1 if this descriptor runs on the pipelined kernel (required by out_scale / pool_y / the fp8 dtypes), else 0; no launch. */
int satcv_conv2d_igemm_pipelined(const satcv_conv_desc* d);

/* Weight gradient of the same convolutions: dW[tap][ci][co] = sum_p X[p+tap][ci]*dY[p][co],
 * written as Keras HWIO fp32.  X is staged with the same optional affine+ReLU as the
 * forward.  Needs a caller-provided fp32 workspace (satcv_conv2d_wgrad_workspace). */
typedef struct satcv_wgrad_desc {
  const void* x0; const void* x1; int32_t c0, c1;
  const float* in_scale; const float* in_shift; int32_t in_relu;
  const void* dy; int32_t lddy;      /* (n, h*f, w*f, lddy) when mode_dy=1 else (n,h,w,lddy) */
  float* dw;                         /* (kh,kw,cin,cout) fp32, or (f,f,cout,cin) when transposed */
  int32_t cin, cout;                 /* real (unpadded) extents written to dw */
  int32_t n, h, w_;
  int32_t kh, kw, dil;
  int32_t mode_dy;                   /* 1: dy is gathered space-to-depth by f (Conv2DTranspose) */
  int32_t f;
  int32_t transposed;                /* dw layout (f,f,cout,cin) */
  float* workspace; int64_t workspace_bytes;
  int32_t dtype;
  int32_t accumulate;                /* dw += result (a layer applied to several inputs: shared weights) */
  int32_t whole_chip;                /* 1: nothing runs beside this launch (the last weight gradient of a backward pass): one workgroup
                                        per CU instead of the 160 that leave room for the other stream's kernels */
  int32_t defer_reduce;              /* 1: the launch only writes its fp32 partial slabs to `workspace` (which then belongs to this layer
                                        until the sum has run); the caller sums them later, many layers per launch:
                                        satcv_conv2d_wgrad_reduce_job + satcv_reduce_slabs_batched.  Plain 1x1 / 3x3 path only. */
} satcv_wgrad_desc;
int64_t satcv_conv2d_wgrad_workspace(const satcv_wgrad_desc* d);
int satcv_conv2d_wgrad(const satcv_wgrad_desc* d, void* stream);

/* Deferred, batched slab sum (round 5).  Every weight-gradient launch ends in an ordered sum of its workgroups' fp32 partial slabs into the
 * Keras-layout gradient -- a launch of 5-30 us per layer, 21 per training step.  With defer_reduce the producers skip it; the caller
 * collects one job per layer (the slab geometry the library chose), keeps the layers' workspaces apart, and runs ONE launch over a DEVICE
 * array of jobs + the exclusive prefix sum of satcv_reduce_job_items().  The summation order is fixed (a job's `lanes` partial sums over
 * slabs l, l + lanes, ..., combined in increasing lane order): results are bit-reproducible.  satcv_reduce_job_items() is the job's item
 * count ROUNDED UP TO 16, so that every prefix is a multiple of 16 and a lane group never straddles a wavefront; total_items must be the sum
 * of those values (SATCV_ERR_INVALID otherwise). */
typedef struct satcv_reduce_job {
  const float* ws; float* dw;
  int32_t nslab, taps, kpad, npad, cin, nvalid, transposed, accumulate, lanes, pad_;
} satcv_reduce_job;
int satcv_conv2d_wgrad_reduce_job(const satcv_wgrad_desc* d, satcv_reduce_job* job);      /* the sum satcv_conv2d_wgrad(d) deferred */
int64_t satcv_reduce_job_items(const satcv_reduce_job* job);
int satcv_reduce_slabs_batched(const satcv_reduce_job* jobs_dev, const int64_t* prefix_dev, int32_t njobs, int64_t total_items, void* stream);

/* Fused backward of a thin conv -> BatchNormalization -> ReLU block (conv_batch_act, utils/model_tools.py:174-186; Keras autodiff of
 * Conv2D :178 + BatchNormalization :179 + Activation :180 inside Model.fit): ONE launch replaces satcv_bn_bwd_apply + the data-gradient
 * satcv_conv2d_igemm + satcv_conv2d_wgrad of the layer.  dy = scale * (g * [scale*y+shift > 0] - c1 - xhat * c2) is formed in
 * registers from g (gradient w.r.t. the activated output) and yraw (the conv's stored output) with coef = [c1 | c2] from
 * satcv_bn_bwd_finalize, never written to memory, and used for both products:
 *     dx (n, h, w, lddx)  = conv3x3(dy, w_dgrad)           w_dgrad: the data-gradient operand image of satcv_pack_weights
 *     dw (3, 3, cin, cout) [+]= sum_pixels x (x) dy        x = x0 (| x1) with the optional in_scale / in_shift / in_relu of the forward
 * Limits (else satcv_conv2d_bwd_fused_workspace returns -1 and the caller keeps the three launches): bf16, 3x3, dilation 1, stored input
 * channels c0 + c1 = cin in {32, 64} with cout 32, or 64 with cout 64, maps of whole 8 x 32 tiles (h % 8 == 0, w_ % 32 == 0), g and yraw
 * with the same channel stride ldg.  workspace: fp32 partial sums, one slab per resident workgroup (size from the query). */
typedef struct satcv_bwdf_desc {
  const void* g; const void* yraw; int32_t ldg;
  const float* bn_scale; const float* bn_shift; const float* bn_mean; const float* bn_rstd;
  const float* bn_coef;                /* [2][cout] */
  int32_t linear;                      /* 1: BatchNormalization without ReLU (no mask) */
  const void* x0; const void* x1; int32_t c0, c1;
  const float* in_scale; const float* in_shift; int32_t in_relu;
  const void* w_dgrad;
  void* dx; int32_t lddx;
  float* dw; int32_t cin, cout;
  int32_t n, h, w_;
  int32_t kh, kw, dil;
  float* workspace; int64_t workspace_bytes;
  int32_t dtype;
  int32_t accumulate;                  /* dw += result (shared weights) */
  /* optional: the sums of the BatchNorm backward of the layer BELOW -- the BatchNormalization + ReLU whose scale / shift are this
   * launch's in_scale / in_shift (needs in_relu = 1) -- formed from the stored dx and the staged input: what satcv_bn_bwd_reduce would
   * compute in a pass of its own over dx and that layer's raw output.  [ROWS][2][bst_sums_ld] rows as for satcv_bn_bwd_finalize. */
  satcv_stat_t* bst_sums; int32_t bst_sums_ld;
  const float* bst_mean; const float* bst_rstd;
  int32_t bst_act_form;                /* 1: the input x is itself an activation (in_scale NULL: a max-pooled encoder output); the rows get
                                          sum dx [x > 0], sum dx x for satcv_bn_bwd_finalize2 */
  /* optional, encoder_block (utils/model_tools.py:262-286): the block's activated output is ALSO max-pooled 2 x 2 -- g is the gradient
   * of the skip, dpool (n, h/2, w/2, lddp) the gradient of MaxPooling2D's output and amax the arg-max bytes satcv_bn_relu_pool_amax
   * wrote: the launch uses g + (amax == position in the window ? dpool : 0).  Limits: 32 -> 64 channels; or 16 stored channels -> 32
   * with dx == NULL (the first block: fed by the model input, no data gradient). */
  const void* dpool; int32_t lddp; const void* amax;
  /* optional, the block under the 1 x 1 head (utils/model_tools.py:405): g is not read -- it is formed in the loader as
   * g[p][c] = bf16(dlogits[p][0] w[c][0] + dlogits[p][1] w[c][1]), what satcv_head_bwd would have stored as dx (pass dx = NULL there).
   * hg_dlogits (npix, 2) fp32, hg_w the head's Keras kernel (cout, 2) fp32, hg_ncls == 2; 32 -> 32 channels only. */
  const float* hg_dlogits; const float* hg_w; int32_t hg_ncls;
  int32_t defer_reduce;                /* 1: leave the weight-gradient slabs in `workspace` (satcv_conv2d_bwd_fused_reduce_job describes the
                                          deferred sum for satcv_reduce_slabs_batched) */
} satcv_bwdf_desc;
int64_t satcv_conv2d_bwd_fused_workspace(const satcv_bwdf_desc* d);
int satcv_conv2d_bwd_fused(const satcv_bwdf_desc* d, void* stream);
int satcv_conv2d_bwd_fused_reduce_job(const satcv_bwdf_desc* d, satcv_reduce_job* job);

/* Fused backward of decoder_block's up-sampling path (utils/model_tools.py:306-309: Conv2DTranspose(filters, up_size, strides = up_size) ->
 * concatenate([skip, up]) -> BatchNormalization -> Activation('relu'), differentiated by Keras inside Model.fit), round 5.  For the `up`
 * channels of the concatenation ONE launch replaces satcv_bn_bwd_apply (their half) + the space-to-depth data gradient + the weight gradient:
 *     dup = scale * (g * [scale*yup+shift > 0] - c1 - xhat * c2)   in registers, never stored (g = gradient of the activated concatenation,
 *           its `up` channels; yup = the transposed convolution's stored output; c1, c2 from satcv_bn_bwd_finalize of the concatenation)
 *     dx (n, h, w, lddx)        = sum_{ij, co} dup[(2y+i, 2x+j)][co] K[i][j][co][ci]      w_dgrad: the data-gradient image of satcv_pack_weights(transposed)
 *     dw (2, 2, cout, cin) fp32 = sum_pixels dup x^T                                       (Keras Conv2DTranspose kernel layout)
 * and, with bst_sums, the sums of the BatchNorm backward of the layer whose output x is (as satcv_conv2d_bwd_fused does).  The bias gradient
 * of the transposed convolution is identically zero under the BatchNormalization that follows it.  bf16, f = 2, cout 32 / 64, cin a
 * multiple of 64, rows of w a multiple of 64 / 32 pixels: satcv_convt_bwd_fused_workspace() answers -1 otherwise and the caller keeps the
 * three launches.  The `skip` channels keep satcv_bn_bwd_apply (c = their count). */
typedef struct satcv_ctbf_desc {
  const void* g; int32_t ldg;            /* (n, 2h, 2w, ldg), already offset to the first `up` channel */
  const void* yup; int32_t ldy;          /* (n, 2h, 2w, ldy) */
  const float* bn_scale; const float* bn_shift; const float* bn_mean; const float* bn_rstd;   /* [cout], of the `up` channels */
  const float* bn_c1; const float* bn_c2; int32_t linear;
  const void* x; int32_t ldx;            /* (n, h, w, ldx): the transposed convolution's input, with its pending BatchNorm + ReLU */
  const float* in_scale; const float* in_shift; int32_t in_relu;
  const void* w_dgrad; int32_t w_npad;   /* [4 cout / 8][w_npad][8] */
  void* dx; int32_t lddx;
  float* dw; int32_t cin, cout;
  int32_t n, h, w_, f;                   /* INPUT grid; f = 2 */
  float* workspace; int64_t workspace_bytes;
  int32_t dtype, accumulate, defer_reduce;
  satcv_stat_t* bst_sums; int32_t bst_sums_ld; const float* bst_mean; const float* bst_rstd;
} satcv_ctbf_desc;
int64_t satcv_convt_bwd_fused_workspace(const satcv_ctbf_desc* d);
int satcv_convt_bwd_fused(const satcv_ctbf_desc* d, void* stream);
int satcv_convt_bwd_fused_reduce_job(const satcv_ctbf_desc* d, satcv_reduce_job* job);

/* --------------------------------------------------------------- batch norm
 * layers.BatchNormalization (utils/model_tools.py:179,308,313,316): eps, momentum as given.
 * Training: consume the [ROWS][2][ld] sum/sumsq rows (and zero them), produce per-channel
 * scale=gamma*rstd, shift=beta-mean*scale, mean, rstd, and update the moving statistics
 * `updates` times (2 reproduces the double cba1 call of conv_block.call, :238-239). */
int satcv_bn_finalize_train(satcv_stat_t* stats, int32_t stats_ld, int32_t c, float count,
                            const float* gamma, const float* beta, float eps, float momentum,
                            int32_t updates, int32_t bessel, float* moving_mean, float* moving_var,
                            float* scale, float* shift, float* mean, float* rstd, void* stream);
/* Inference: scale/shift from the moving statistics. */
int satcv_bn_affine_infer(const float* gamma, const float* beta, const float* moving_mean,
                          const float* moving_var, float eps, int32_t c, float* scale, float* shift,
                          void* stream);
/* The same for every BatchNormalization of an inference plan in one launch: `jobs_device` is a DEVICE array of njobs entries. */
typedef struct satcv_bn_affine_job {
  const float* gamma; const float* beta; const float* moving_mean; const float* moving_var;
  float* scale; float* shift; int64_t c;
  /* optional (NULL: not written): bias_eff = scale * conv_bias + shift -- the additive term of a convolution whose epilogue applies
   * its own inference BatchNormalization (out_scale = scale, bias = bias_eff), as the residual joins of the ResNet backbone do */
  const float* conv_bias; float* bias_eff;
} satcv_bn_affine_job;
int satcv_bn_affine_infer_batched(const satcv_bn_affine_job* jobs_device, int32_t njobs, float eps, void* stream);

/* act = relu(scale*yraw+shift) (written if act != NULL), pooled = maxpool_f(act) ('valid'),
 * optional sum/sumsq rows of `act` (feeds the decoder's concat BatchNormalization).
 * Replaces Activation('relu') + MaxPooling2D (utils/model_tools.py:180,281). */
int satcv_bn_relu_pool(const void* yraw, const float* scale, const float* shift, void* act, int32_t act_ld,
                       void* pooled, satcv_stat_t* stats, int32_t stats_ld, int32_t n, int32_t h,
                       int32_t w_, int32_t c, int32_t f, int32_t dtype, void* stream);
/* the same, also writing amax (n, h/f, w/f, c) bytes: the position i * f + j of every window's first maximum (what the backward of
 * MaxPooling2D routes the gradient to) -- read by the pooled form of satcv_conv2d_bwd_fused instead of re-scanning the windows */
int satcv_bn_relu_pool_amax(const void* yraw, const float* scale, const float* shift, void* act, int32_t act_ld,
                            void* pooled, void* amax, satcv_stat_t* stats, int32_t stats_ld, int32_t n, int32_t h,
                            int32_t w_, int32_t c, int32_t f, int32_t dtype, void* stream);

/* Backward of  a = relu(scale*y+shift)  (training-mode BN):
 *   g  = (da [+ unpool(dpool)]) * (a > 0)
 *   reduce : sums[row][0][ch] += g ; sums[row][1][ch] += g * xhat
 *   finalize: dbeta=sum g, dgamma=sum g*xhat -> grads, coef (c1=dbeta/M, c2=dgamma/M), zero sums
 *   apply  : dy = scale*(g - c1 - xhat*c2)  (+ optional dbias[ch] += sum dy)            */
typedef struct satcv_bnbwd_desc {
  const void* da; int32_t ldda;        /* grad w.r.t. activation, may be NULL if only dpool */
  const void* dpool; int32_t lddp;     /* optional grad of maxpool_f(a), (n,h/f,w/f,lddp)   */
  int32_t f;
  const void* yraw; int32_t ldy;
  const float* scale; const float* shift; const float* mean; const float* rstd;
  satcv_stat_t* sums; int32_t sums_ld; /* [ROWS][2][sums_ld]                                */
  const float* coef;                   /* [2][c] from finalize (apply only)                 */
  void* dy; int32_t lddy_out;          /* apply output                                      */
  float* dbias;                        /* optional [c], atomically accumulated (apply)      */
  int32_t n, h, w_, c;
  int32_t dtype;
  int32_t linear;                      /* 1: BatchNormalization without the ReLU (residual branch): g = da, no mask   */
  /* optional second source (c_split > 0): the BatchNormalization of concat([skip, up]) (utils/model_tools.py:307-308) in ONE pass
   * over the gradient `da` of the concatenation -- channels [0, c_split) are read from yraw / written to dy as above, channels
   * [c_split, c) from yraw1 (stride ldy1) / to dy1 (stride lddy1).  Dense form only (no dpool, no dbias).  satcv_bn_bwd_apply with
   * dy1 == NULL applies the first c_split channels only (the others' gradient is formed by satcv_convt_bwd_fused in its loader).   */
  const void* yraw1; int32_t ldy1;
  void* dy1; int32_t lddy1;
  int32_t c_split;
  /* optional (apply pass of the two-source form): the first source is the ACTIVATED output a = relu(BN(y_enc)) of an encoder block
   * (the skip of decoder_block, utils/model_tools.py:307) and dy its gradient: also accumulate the sums of THAT BatchNorm's backward
   * in the activated form -- rows [0]: sum dy [a > 0], rows [1]: sum dy a over the c_split skip channels -- which
   * satcv_bn_bwd_finalize2 converts.  Saves the pass satcv_bn_bwd_reduce would make over dy and y_enc. */
  satcv_stat_t* sk_sums; int32_t sk_sums_ld;
} satcv_bnbwd_desc;
int satcv_bn_bwd_reduce(const satcv_bnbwd_desc* d, void* stream);
int satcv_bn_bwd_finalize(satcv_stat_t* sums, int32_t sums_ld, int32_t c, float count, float* dgamma,
                          float* dbeta, float* coef, int32_t accumulate, void* stream);
int satcv_bn_bwd_apply(const satcv_bnbwd_desc* d, void* stream);
/* finalize over two sets of rows: `sums` in the raw form satcv_bn_bwd_reduce writes (may be NULL), `act_sums` in the activated form
 * (sum g [a > 0], sum g a with a = relu(scale * y + shift)) written by producers that only see the activated tensor: the sk_sums of
 * satcv_bn_bwd_apply, a data-gradient launch whose bst_y is the max-pooled activation with unit bst_scale / bst_rstd and zero
 * bst_shift / bst_mean.  sum g xhat = (sum g a - (shift + scale mean) sum g) rstd / scale.  Zeroes both sets of rows. */
int satcv_bn_bwd_finalize2(satcv_stat_t* sums, int32_t sums_ld, satcv_stat_t* act_sums, int32_t act_sums_ld, int32_t c, float count,
                           const float* scale, const float* shift, const float* mean, const float* rstd, float* dgamma, float* dbeta,
                           float* coef, int32_t accumulate, void* stream);

/* Residual blocks of the atrous CNN family (utils/model_tools.py:922-979: ReLU(BN(conv) + shortcut), plain Conv2D layers):
 * satcv_relu_bwd : g[i] = act[i] > 0 ? g[i] : 0 in place (gradient through the ReLU of a materialised activation; the masked
 *                  gradient then serves BOTH addends of the sum).
 * satcv_bias_grad: dbias[ch] += sum over pixels of dy[pix][ch] (a Conv2D that is not followed by BatchNormalization).  With
 *                  `partials` (satcv_bias_grad_workspace bytes) the workgroups write rows that a second launch adds in fixed order
 *                  (bit-reproducible); NULL: float atomics. */
int satcv_relu_bwd(const void* act, void* g, int64_t count, int32_t dtype, void* stream);
int64_t satcv_bias_grad_workspace(int64_t npix, int32_t c);
int satcv_bias_grad(const void* dy, int32_t lddy, int64_t npix, int32_t c, int32_t dtype, float* dbias, float* partials, void* stream);

/* ------------------------------------------ ResNet / DeepLab-v3 inference helpers
 * The reference has no DeepLab-v3/ResNet-50 code (README.md:8 only names it); these ops serve the build-defined
 * configuration of SURVEY.md section 8a row A9.
 * satcv_maxpool: window k, stride s, symmetric padding (ResNet stem 3x3/2).
 * satcv_add_act: out = relu?(affine?(y) + affine?(res)) -- the residual join of a bottleneck.
 * satcv_upsample_head: bilinear x factor (half-pixel centres) of fp32 logits + softmax/argmax or sigmoid/threshold.
 * (satcv_head_fwd with activation 2 writes the raw logits.) */
int satcv_maxpool(const void* x, void* out, int32_t n, int32_t h, int32_t w_, int32_t c, int32_t k, int32_t s,
                  int32_t pad, int32_t dtype, void* stream);
int satcv_add_act(const void* y, const float* y_scale, const float* y_shift, const void* res,
                  const float* res_scale, const float* res_shift, int32_t relu, void* out, int64_t npix,
                  int32_t c, int32_t dtype, void* stream);
/* out = Q(relu?(scale[c]*x + shift[c])) channel-wise between two NHWC tensors with their own channel strides; dtype_out is
 * dtype or SATCV_FP8 (or SATCV_BF16 from SATCV_FP8: where the hybrid fp8 graph hands a tensor to its bf16 levels).  Uses in the folded fp8 inference path: the skip half of concat([skip, up]) -> BN -> ReLU with the
 * requantisation q_skip/q_cat folded into scale/shift (utils/model_tools.py:307-309), and BN+ReLU+quantisation of the
 * first conv block, which is computed in bf16 because the e4m3 rounding of the INPUT bands costs the most accuracy. */
int satcv_affine_requant(const void* x, int32_t ldx, const float* scale, const float* shift, int32_t relu, void* out, int32_t ldo,
                         int64_t npix, int32_t c, int32_t dtype, int32_t dtype_out, void* stream);

int satcv_upsample_head(const float* logits, int32_t n, int32_t h, int32_t w_, int32_t ncls, int32_t factor,
                        int32_t activation, float thresh, float* probs, int32_t* classes, void* stream);

/* ---------------------------------------------------------------- dropout
 * layers.SpatialDropout2D / layers.Dropout (utils/model_tools.py:311, 351, 363, 402), training only.
 * satcv_dropout_mask: counter-based Bernoulli mask, values 0 or 1/(1-rate).
 * satcv_dropout_apply: out = act(x) * mask with act(x) = relu?(scale*x+shift) if scale else x;
 *   mask_mode 0: mask is (n, ldm) [whole feature maps dropped], 1: mask is (n*hw, ldm) [per element];
 *   ldm >= c lets a channel slice of a wider mask be applied.  With scale == NULL this is also the
 *   backward (g * mask). */
int satcv_dropout_mask(uint64_t seed, uint64_t offset, float rate, int64_t count, float* mask, void* stream);
int satcv_dropout_apply(const void* x, int32_t ldx, const float* scale, const float* shift, int32_t relu,
                        const float* mask, int32_t ldm, int32_t mask_mode, void* out, int32_t ldo, int32_t n,
                        int32_t hw, int32_t c, int32_t dtype, void* stream);

/* ------------------------------------------------------------------- head
 * Conv2D(ncls,(1,1),activation) + argmax/threshold (utils/model_tools.py:405-406, :660-661).
 * activation 0: softmax, classes = argmax (ties -> lowest index);
 *            1: sigmoid, classes = probs > thresh.
 * x is the raw output of the last decoder conv; its BN+ReLU is applied on load. */
typedef struct satcv_head_desc {
  const void* x; int32_t ldx; int32_t cin;
  const float* in_scale; const float* in_shift;
  const float* w;            /* fp32 (cin, ncls) = Keras (1,1,cin,ncls) */
  const float* b;            /* [ncls] */
  int32_t ncls; int32_t activation; float thresh;
  float* probs;              /* (npix, ncls) fp32 */
  int32_t* classes;          /* (npix) int32 */
  const float* dlogits;      /* bwd: (npix, ncls) fp32 */
  void* dx; int32_t lddx;    /* bwd: grad w.r.t. the activated input, storage dtype */
  float* dw; float* db;      /* bwd: fp32, atomically accumulated */
  /* bwd, optional: fused first pass of the BatchNorm+ReLU backward of x: with g = dx*(a>0), xhat = (x-mean)*rstd the kernel
   * accumulates per channel sum(g), sum(g*xhat) into bnr_sums ([SATCV_STAT_ROWS][2][bnr_sums_ld]) -- replaces
   * satcv_bn_bwd_reduce for that tensor (register-resident head kernel only) */
  const float* bnr_mean; const float* bnr_rstd; satcv_stat_t* bnr_sums; int32_t bnr_sums_ld;
  int64_t npix;
  int32_t dtype;
  /* bwd, optional: [satcv_head_bwd_workspace(d) bytes] -- every workgroup writes its dW / db partial sums here instead of adding
   * them to dw / db with float atomics; satcv_head_bwd_finalize then adds the rows in fixed order (bit-reproducible gradients) */
  float* partials;
} satcv_head_desc;
int satcv_head_fwd(const satcv_head_desc* d, void* stream);
int satcv_head_bwd(const satcv_head_desc* d, void* stream);
/* bytes of `partials` for this descriptor (0: shape outside the register-resident kernel, which then uses atomics) */
int64_t satcv_head_bwd_workspace(const satcv_head_desc* d);
int satcv_head_bwd_finalize(const satcv_head_desc* d, void* stream);

/* CRC-32C (Castagnoli) of a HOST buffer, continuing from crc_in (0 to start): the checksum of the TFRecord framing that
 * tf.io.TFRecordWriter / tf.data.TFRecordDataset use (utils/prediction_tools.py:221, 404; utils/processing.py:416). */
uint32_t satcv_crc32c(const void* data, uint64_t nbytes, uint32_t crc_in);

/* ------------------------------------------------------------------ tile input pipeline
 * Device version of UNETDataGenerator.__getitem__ (utils/processing.py:456-755) for one source / the labels of one batch.
 * src: planes (n, c, hin, win) of kind 0 u8, 1 u16, 2 f32, 3 i16, 4 f64, 5 i32, 6 i64 (what np.load returned).
 *   x / rescale (float64; 0 = none)                                     -- :551-552, 601, 613 (255 / 10000 / 100 / 2000)
 *   nan_mask: append the mask channel; with replace (to_fit) values that are NaN or < -5000 -- in this channel or an
 *   EARLIER one of the image, as coded -- become N(0,1) draws (counter RNG on seed) and the mask is 1     -- :553-583
 *   nan_mask == 2 (SiameseDataGenerator._get_unet_data, :797-805): mask channel = 1 where NO band of the pixel is NaN or
 *   below -1 after the rescale; NaN elements (only) become U[0,1) draws; the colour augmentation that follows takes its mean
 *   over the replaced values, as the reference does
 *   centre trim to (h, w_)                                                                                 -- :586-590
 *   ch_mean != NULL: (x - mean)*contra_mul + mean*bright_mul, mean = nanmean over (h, w_) per image and channel from
 *   satcv_tile_channel_mean                                             -- utils/array_tools.py:159-186
 *   flip_v, flip_h, rot (np.rot90 k) of the stacked batch               -- utils/array_tools.py:188-213, processing.py:748
 * dst: fp32 NHWC (n, ho, wo, ldc) written at channel offset coff (ho, wo = w_, h when rot is odd). */
typedef struct {
  const void* src; int32_t src_kind;
  int32_t n, c, hin, win, h, w_;
  double rescale;
  int32_t nan_mask, replace;
  uint64_t seed;
  const double* ch_mean;
  double contra_mul, bright_mul;
  int32_t flip_v, flip_h, rot;
  float* dst; int32_t ldc, coff;
} satcv_tile_desc;
int satcv_tile_channel_mean(const satcv_tile_desc* d, double* mean_out /* (n, c) */, void* stream);
int satcv_tile_ingest(const satcv_tile_desc* d, void* stream);
/* labels (n, 1, hin, win) -> merge_classes (utils/array_tools.py:26-44) as 256-entry look-up tables (-1 = keep): lut on the
 * land-cover value, then lu_lut on the optional land-use array -> trim -> morph -> tf.one_hot(depth nclasses) fp32
 * (utils/processing.py:656-704). */
int satcv_label_onehot(const void* lc, int32_t lc_kind, const int32_t* lut, const void* lu, int32_t lu_kind, const int32_t* lu_lut,
                       int32_t n, int32_t hin, int32_t win, int32_t h, int32_t w_, int32_t nclasses, int32_t flip_v, int32_t flip_h,
                       int32_t rot, float* dst, int32_t ldc, int32_t coff, void* stream);

/* ------------------------------------------------------------------ losses
 * Each writes loss_out[0] += mean loss contribution (caller zeroes) and
 * dlogits = dL/dlogits (through the head activation).
 * kind 0: weighted_categorical_crossentropy (utils/model_tools.py:25-40), weights[ncls]
 * kind 1: weighted_bce on probabilities (:96-112), weights[0] = pos_weight                 */
int satcv_loss_fwd_bwd(int32_t kind, const float* probs, const float* y_true,
                       const float* weights, int32_t ncls, int32_t activation, int64_t npix,
                       float grad_scale, float* loss_out, float* dlogits, void* stream);

/* Ratio-type losses that need global sums before the gradient exists (two passes inside):
 * kind 2: gen_dice (utils/model_tools.py:42-94; class_weights = global_weights or NULL for the
 *         per-image 1/count^2 weights with `eps` for empty classes), 3: iou_loss (:131-140),
 *         4: mse_4d (:142-166, mean over finite elements).
 * workspace: nimg*3*ncls floats (zeroed here).  Writes loss_out[0] += loss and dL/dlogits. */
int satcv_loss_global_fwd_bwd(int32_t kind, const float* probs, const float* y_true,
                              const float* class_weights, int32_t ncls, int32_t activation,
                              int32_t nimg, int64_t pix_per_img, float eps, float grad_scale,
                              float* workspace, float* loss_out, float* dlogits, void* stream);

/* confusion[t*ncls + p] += 1 over pixels (MeanIoU / accuracy; notebooks nb:275) */
int satcv_confusion(const int32_t* classes, const float* y_true, int32_t ncls, int64_t npix,
                    int64_t* confusion, void* stream);

/* --------------------------------------------------------------- optimizer
 * Keras Adam (epsilon outside the bias correction).  `state` is a device
 * float[4]: {lr, step_count, grad_scale, unused}; the kernel increments
 * step_count itself so that the launch is graph-replayable. */
int satcv_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float beta1,
                    float beta2, float eps, float* state, const float* lr_mul, void* stream);
/* The same update on a SUB-RANGE of the flat buffers (pointers already offset, n elements), with the step counter incremented only when
 * bump != 0: a step may be applied in several launches -- the part of the parameters whose gradients are final early in the backward pass
 * on a second stream, beside the rest of the backward pass (engine.Plan, round 6) -- as long as exactly ONE of them, the last to run,
 * bumps: every part then sees the same step number. */
int satcv_adam_step_part(float* p, const float* g, float* m, float* v, int64_t n, float beta1,
                         float beta2, float eps, float* state, const float* lr_mul, int32_t bump, void* stream);

/* Clears the two buffers a training step accumulates into -- the flat gradient (bytes_a, a multiple of 16, 16-byte aligned) and a small
 * second one (the loss scalar; bytes_b a multiple of 4) -- in ONE launch of the library's own (the optimizer loop of Model.fit,
 * notebooks/UNET_G4G_2019_solar.ipynb:1267: Keras starts every step from zero gradients).  Either may be empty. */
int satcv_zero2(void* a, int64_t bytes_a, void* b, int64_t bytes_b, void* stream);

/* ---------------------------------------------------- data-parallel collectives
 * The reference has no multi-device code; this is the north-star's data-parallel `Model.fit`
 * (notebooks/UNET_G4G_2019_solar.ipynb:1267-1275 run on N replicas): one process per GPU, ONE exchange per step -- the
 * sum of the flat fp32 gradient over the replicas, RCCL over xGMI.  RCCL is bound at run time (dlopen of librccl.so.1, or the
 * path in SATCV_RCCL_LIB): SATCV_ERR_UNSUPPORTED from every entry point below when it cannot be loaded.
 *
 * satcv_comm_unique_id: rank 0 fills 128 bytes and hands them to every rank through any side channel.
 * satcv_comm_init:      collective over all ranks, on the calling thread's CURRENT device; the handle is owned by the
 *                       caller and released with satcv_comm_destroy (also collective).  One communicator per device/process.
 * satcv_allreduce_grads: sum of grads[lo, hi) over the ranks, in place, asynchronous on `stream`, sent in buckets of
 *                       `bucket_elems` floats cut from the END of the range (reverse-layer order: the layers nearest the loss are
 *                       final first) inside one RCCL group.  payload SATCV_F32, or SATCV_BF16: the range is rounded to bf16
 *                       into `scratch` (caller-owned, >= (hi - lo) * 2 bytes), summed in bf16 on the wire (half the bytes: 37 MB
 *                       instead of 74 MB for get_unet_model(2, 4)) and widened back.  The mean is the optimizer's business
 *                       (satcv_adam_step's grad_scale = 1 / world).
 * satcv_allreduce:      in-place sum (average != 0: mean) of `count` elements of SATCV_F32 / SATCV_BF16 / SATCV_F64 -- the
 *                       BatchNorm statistics rows under SyncBN, loss / confusion-matrix scalars at log points. */
#define SATCV_COMM_ID_BYTES 128
typedef struct satcv_comm satcv_comm;
int satcv_comm_unique_id(void* id128);
int satcv_comm_init(satcv_comm** comm_out, int32_t rank, int32_t world, const void* id128);
int satcv_comm_destroy(satcv_comm* comm);
int satcv_comm_info(const satcv_comm* comm, int32_t* rank, int32_t* world);
int satcv_allreduce_grads(satcv_comm* comm, float* grads, int64_t lo, int64_t hi, int64_t bucket_elems, int32_t payload,
                          void* scratch, void* stream);
int satcv_allreduce(satcv_comm* comm, void* buf, int64_t count, int32_t dtype, int32_t average, void* stream);

/* ------------------------------------------------------------ ConvLSTM2D family
 * The reference's LSTM builders (utils/model_tools.py:666-768 build_lstm_layers / build_lstm_layers2, :773-920 get_lstm_model /
 * get_lstm_autoencoder / get_hybrid_model, :1016-1060 get_hierarchical_model) call tf.keras.layers.ConvLSTM2D(filters, [3, 3],
 * padding 'same', activation None, optional dilation_rate on the INPUT convolution).  Per time step
 *     z = conv(x_t, kernel) + bias + conv(h_{t-1}, recurrent_kernel)          (gate order i, f, c, o along the 4 F channels)
 *     i, f, o = rec_act(z_i, z_f, z_o);  g = act(z_c);  c_t = f c_{t-1} + i g;  h_t = o act(c_t)
 * Both convolutions are satcv_conv2d_igemm launches (sequences are stored time-major, so the input convolution of ALL steps is
 * one launch over T * B images); these entry points are the cell arithmetic.
 *
 * satcv_ingest_seq: (B, T, H, W, C) float32 (the Keras input layout) -> (T, B, H, W, cpad) storage type. */
int satcv_ingest_seq(const float* src, void* dst, int32_t batch, int32_t steps, int32_t h, int32_t w_, int32_t c, int32_t cpad,
                     int32_t dtype, void* stream);
typedef struct satcv_lstm_gates_desc {
  /* forward inputs: the two convolution outputs of this step (storage type), (npix, 4 F) with leading dimensions ldx / ldh_g;
   * hg NULL at t = 0 (h_0 = 0); c_prev (npix, F) float32 or NULL (c_0 = 0) */
  const void* xg; int32_t ldx; const void* hg; int32_t ldh_g; const float* c_prev;
  float* c_out;                        /* c_t (npix, F) float32 (backward: c_t as an INPUT)                                       */
  void* h_out; int32_t ldh;            /* h_t (npix, ldh) storage type                                                            */
  void* gates_out;                     /* post-activation (i, f, g, o) (npix, 4 F) storage type: forward output (NULL at inference),
                                          backward input                                                                          */
  satcv_stat_t* stats; int32_t stats_ld;   /* optional BatchNorm statistics of the stored h_t, rows as in satcv_conv_desc           */
  /* backward: dL/dh_t from up to two producers (the layer above; the recurrent data gradient of step t + 1), dc_{t+1 -> t} or NULL */
  const void* dh_a; int32_t lddh_a; const void* dh_b; int32_t lddh_b; const float* dc_next;
  void* dz_out; int32_t lddz;          /* dL/dz_t (npix, lddz >= 4 F) storage type: the dy of both convolutions' gradients        */
  float* dc_prev_out;                  /* dc_{t -> t-1} (npix, F) float32                                                         */
  int64_t npix; int32_t filters;
  int32_t rec_act;                     /* 0 hard_sigmoid = clip(0.2 z + 0.5, 0, 1) (Keras 2.x ConvLSTM2D default), 1 sigmoid (Keras 3) */
  int32_t act;                         /* 0 linear (activation=None, as every reference call site), 1 tanh (the Keras default)    */
  int32_t dtype;
} satcv_lstm_gates_desc;
int satcv_convlstm_gates_fwd(const satcv_lstm_gates_desc* d, void* stream);
int satcv_convlstm_gates_bwd(const satcv_lstm_gates_desc* d, void* stream);

/* Conv2D(cout <= 16, 1x1) over the channel concatenation of one or two sources -- the `dense` layers and fusion heads of the LSTM
 * family (utils/model_tools.py:797, 847-857, 902-910, 1050-1056).  A source is an NHWC tensor (float32 or bf16) with an optional
 * pending BatchNorm + ReLU and, if hs / ws are set, on a coarser grid that is read through tf.image.resize(..., 'nearest')
 * (half-pixel centres: source index min(floor((i + 0.5) in / out), in - 1)).  activation: 0 softmax (+ argmax classes), 1 sigmoid,
 * 2 linear, 3 ReLU clipped at max_value (max_value <= 0: plain ReLU; layers.ReLU(max_value=2.0) is the reference's default head).
 * Backward: `dout` is dL/d out for activations 2 / 3 (for softmax / sigmoid heads pass the loss kernel's dL/dlogits with activation 2);
 * dw (rows of every source in order, (sum cin, cout)) and db are ACCUMULATED (zero them first); src[i].dx receives the gradient of the
 * source's ACTIVATED values on the source's own grid. */
typedef struct satcv_dense_src {
  const void* x; int32_t ld, cin, dtype;
  const float* in_scale; const float* in_shift; int32_t in_relu;
  int32_t hs, ws;                      /* 0, 0: the output grid                                                                   */
  void* dx; int32_t lddx, dx_dtype;    /* backward output or NULL                                                                 */
} satcv_dense_src;
typedef struct satcv_dense_desc {
  satcv_dense_src src[2]; int32_t nsrc;
  const float* w; const float* b; int32_t cout;
  int32_t activation; float max_value;
  float* out; int32_t* classes; float* z_out;      /* (npix, cout) float32; classes (npix) for softmax or NULL; z_out optional     */
  int64_t npix; int32_t h, w_;                     /* output grid (npix = images * h * w_)                                         */
  const float* dout; float* dz_out; float* dw; float* db;
} satcv_dense_desc;
int satcv_dense_small_fwd(const satcv_dense_desc* d, void* stream);
int satcv_dense_small_bwd(const satcv_dense_desc* d, void* stream);

/* ------------------------------------------------------- stream utilities */
int satcv_graph_begin(void* stream);
int satcv_graph_end(void* stream, void** graph_exec_out);
int satcv_graph_launch(void* graph_exec, void* stream);
int satcv_graph_destroy(void* graph_exec);
/* event pairs recorded around every igemm launch while enabled; total ms + count returned.  kind (bit of kind_mask): 0 = 3x3 conv
 * forward / data gradient, 1 = 1x1 / transposed-conv GEMMs, 2 = weight gradients, 3 = fused thin-layer backward (data + weight gradient) */
int satcv_prof_enable(int32_t kind_mask);
int satcv_prof_collect(int32_t kind, double* total_ms, int64_t* launches, double* flops);
/* kernel-selection knobs (tests force a tile configuration on small shapes; probes A/B variants in one process).  Keys:
 * "igemm_db"  0 = 128x128 single-buffered tile only, 1 = automatic (default), 2 = the double-buffered 256x128 tile wherever
 *             its shape limits allow;
 * "igemm_thin" 0 = general kernel, 1 (default) = the persistent weights-stationary kernel for thin 3x3 layers (Cin 16/32/64 -> Cout 32/64,
 *             maps of whole 8 x 32 tiles), whatever the batch size (2 = the same; kept for older callers);
 *             "igemm_thin_launches" (read-only) counts the launches it has served in this process (tests check the path taken).
 * "wgrad_db"  1 (default) = double-buffered weight-gradient kernel where its limits allow, 0 = the single-buffered one, 2 = the
 *             64 x 128 block for 1 x 1 / transposed-conv gradients instead of the 128 x 256 one.
 * Returns SATCV_ERR_INVALID for an unknown key.  Not thread-safe against concurrent launches. */
int satcv_set_option(const char* key, int32_t value);
int satcv_get_option(const char* key, int32_t* value);

#ifdef __cplusplus
}
#endif
#endif
