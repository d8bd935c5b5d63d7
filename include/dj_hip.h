/* dj_hip.h -- C ABI of the MI355X (gfx950) compute library behind the
 * ResNet50-DCT + SSD300 hot path.
 *
 * The reference (Shulk97/JPEG_detection_Resnet_SSD) has no native boundary of its own:
 * its "operator API" is Keras 2.2.4 layers executed by TensorFlow 1.8 ops.  Each entry
 * point below replaces the TF op(s) that one Keras layer / loss / optimizer launches on
 * the hot path; the citation names the reference call site it stands in for
 * (paths relative to the reference root; L/ = localisation_part/, C/ = classification_part/).
 *
 * Conventions (all entry points):
 *   - tensors are caller-owned DEVICE pointers to float32, activations NHWC, conv
 *     kernels HWIO (the Keras layout), Conv2DTranspose kernels (kh,kw,out,in);
 *   - `ld_*` = floats between consecutive pixels (>= channels: lets a tensor be a
 *     channel slice of a wider concat buffer);
 *   - `stream` is a hipStream_t passed as void*; nothing here allocates or synchronises, so
 *     calls are capturable in a hipGraph;
 *   - state: every launch is a function of its arguments and of THREE library settings, all
 *     safe to read and write from any thread and none of them touched by a compute call:
 *     the arithmetic mode (a process default, dj_set_compute_mode, that a thread overrides
 *     for its own launches with dj_set_thread_compute_mode -- two models of different modes,
 *     or two threads, do not disturb each other), the per-geometry launch overrides of the
 *     autotuner (dj_conv2d_tune_set: a mutex-guarded table keyed by geometry AND arithmetic
 *     mode; an entry only selects among kernels that compute the same result), and the test
 *     switch dj_set_fast_path.  The last error text (dj_last_error) is per thread;
 *   - return 0 on success, <0 on error (message: dj_last_error()); never throws.
 */
#ifndef DJ_HIP_H
#define DJ_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* dj_last_error(void);
/* Version of this interface: bumped whenever an entry point is added or the meaning of an argument changes
 * (2: the `beta` / `relu` bit fields, the workspace and BatchNormalization-backward-statistics entry points, the per-thread
 * arithmetic mode; 3: the dtype-carrying `_t` entry points for 16-bit tensors in HBM, the tuner direction 9).  A binding
 * written for one version must refuse a library that reports another (jpeg_detection_resnet_ssd_amd/_lib.py does). */
#define DJ_ABI_VERSION 3
int dj_abi_version(void);

/* Storage type of a tensor in HBM.  Everything is fp32 unless an entry point's name ends in `_t`: those take, next to
 * each ACTIVATION / GRADIENT tensor, one of these codes (BASELINE config 5, "fp16-input / fp32-accumulate": under
 * K.set_floatx('float16') the backbone's conv outputs and block sums are held as fp16, their gradients as bf16 -- the
 * exponent range gradients need --; weights, their gradients, the optimizer state, BatchNormalization statistics and all
 * arithmetic outside the MFMA operands stay fp32).  `ld_*` are in ELEMENTS of the tensor they describe. */
#define DJ_F32 0
#define DJ_F16 1
#define DJ_BF16 2

/* Geometry of one keras.layers.Conv2D (third-party; used at
 * L/models/keras_ssd300_dct_j2d_resnet.py:77-96,128-160,483-545,562-675).  TF 'same'
 * padding is resolved by the caller into pad_top/pad_left (asymmetric for even kernels). */
typedef struct dj_conv2d_desc {
  int batch;
  int in_h, in_w, in_c;
  int out_h, out_w, out_c;
  int kernel_h, kernel_w;
  int stride_h, stride_w;
  int dilation_h, dilation_w;
  int pad_top, pad_left;
  int ld_x; /* pixel stride of x / dx */
  int ld_y; /* pixel stride of y / dy */
} dj_conv2d_desc;

/* Leading dimension of the `stats` partial buffer ([rows][2][out_c] floats): one row per 64 output
 * pixels, independent of the tile configuration. */
int dj_conv2d_fwd_stats_rows(const dj_conv2d_desc* d);

/* y = act(conv(pro(x), w) + bias).  Replaces Conv2D.call (TF conv2d + bias_add [+ relu]).
 *   pro_scale/pro_shift ([in_c], optional): x is read as x*scale+shift (then ReLU if
 *     pro_relu) on in-bounds pixels only -- the BatchNormalization+Activation that Keras
 *     runs between two convs (L/models/...resnet.py:80-81,90-91) folded into the load.
 *   stats (optional): per-row-tile column sums / sums of squares of conv(x,w) WITHOUT bias,
 *     consumed by dj_bn_finalize (training-mode BatchNormalization statistics). */
/* `relu` argument of dj_conv2d_nhwc_fwd: DJ_CONV_RELU = fused ReLU epilogue; DJ_CONV_Y_ZEROED = the caller guarantees y is
 * all zeros on entry, so a split-K launch (atomic accumulation) does not clear it first. */
#define DJ_CONV_RELU 1
#define DJ_CONV_Y_ZEROED 2
#define DJ_CONV_STATS_MAY_SPLIT 4 /* dj_conv2d_nhwc_fwd_ws only, see there */
int dj_conv2d_nhwc_fwd(const dj_conv2d_desc* d, const float* x, const float* w, const float* bias,
                       float* y, const float* pro_scale, const float* pro_shift, int pro_relu,
                       int relu, float* stats, void* stream);

/* dj_conv2d_nhwc_fwd with a caller-owned workspace (the reference's TF convolution is bit-reproducible in the forward
 * pass, L/models/keras_ssd300_dct_j2d_resnet.py:481-675 are the small-map layers this matters for): when the tuned
 * launch splits its reduction over workgroups and `workspace_floats` >= dj_conv2d_fwd_workspace_floats(d, ...), the K
 * chunks store their partial tiles to the workspace and a second kernel adds them in a fixed order, adds the bias,
 * applies ReLU and -- if `stats` is given -- takes the BatchNormalization statistics of the sum; without (enough)
 * workspace the launch accumulates with fp32 atomics in arrival order like dj_conv2d_nhwc_fwd.
 * DJ_CONV_STATS_MAY_SPLIT in `relu`: a launch with `stats` is tuned like one without and may be split (its statistics
 * then come from the reduction); otherwise a launch with `stats` is never split. */
int dj_conv2d_nhwc_fwd_ws(const dj_conv2d_desc* d, const float* x, const float* w, const float* bias,
                          float* y, const float* pro_scale, const float* pro_shift, int pro_relu,
                          int relu, float* stats, float* workspace, long workspace_floats, void* stream);
/* floats of workspace that make the CURRENT tuning choice for this geometry run without atomics (0: not split; < 0: bad
 * descriptor) */
long dj_conv2d_fwd_workspace_floats(const dj_conv2d_desc* d, int may_split_stats);

/* dx (+)= conv_transpose(dy, w) [+ bias].  Gradient of Conv2D w.r.t. its input (TF
 * Conv2DBackpropInput); with `bias` it is also the forward of keras.layers.Conv2DTranspose
 * (L/models/...resnet.py:1709-1711), whose (kh,kw,out,in) kernel is the HWIO kernel of
 * the convolution it transposes.  beta: bit 0 = accumulate into dx; DJ_DGRAD_NO_SPLIT = never split the reduction over
 * workgroups (a split launch adds with fp32 atomics in arrival order; the forward use as Conv2DTranspose sets it so that
 * the forward pass stays bit-reproducible). */
#define DJ_DGRAD_NO_SPLIT 2
int dj_conv2d_nhwc_dgrad(const dj_conv2d_desc* d, const float* dy, const float* w, const float* bias,
                         float* dx, int beta, void* stream);
/* Input gradient + the BatchNormalization BACKWARD statistics of the tensor it produces, in one launch.  dx here is
 * g = dL/d relu(bn(z)) for the BatchNormalization layer whose ONLY consumer is this convolution
 * (L/models/keras_ssd300_dct_j2d_resnet.py:77-96: conv -> BatchNormalization -> Activation('relu') -> conv inside the
 * bottleneck blocks); z [batch*in_h*in_w][ld_z] is that layer's input, mean / invstd its saved batch statistics, scale /
 * shift its affine (both NULL: no ReLU behind it, nothing is masked).  Besides storing dx the epilogue writes
 *   partial[r][0][c] = sum g_m,   partial[r][1][c] = sum g_m * (z - mean[c]) * invstd[c],   g_m = g where z*scale+shift > 0
 * over rows 64 r .. 64 r + 63, partial = [ceil(batch*in_h*in_w / 64)][2][in_c]: what dj_bn_bwd_reduce(mask_mode 2 / 0)
 * computes in a pass of its own over g and z, in the layout dj_bn_bwd_finalize reads.  Never split over K, never
 * accumulating; not for strided 1x1 convolutions (their dx is scattered). */
int dj_conv2d_nhwc_dgrad_bnbwd(const dj_conv2d_desc* d, const float* dy, const float* w, float* dx, const float* z, int ld_z,
                               const float* mean, const float* invstd, const float* scale, const float* shift,
                               float* partial, void* stream);

/* Residual Add + ReLU evaluated inside the consumer (the first 1x1 conv of the next bottleneck block,
 * L/models/keras_ssd300_dct_j2d_resnet.py:96-99 then :66-68): the conv's input is
 *   a[m][c] = relu(x[m][c]*pro_scale[c] + pro_shift[c] + res[m][c]*res_scale[c] + res_shift[c])
 * (res_scale/res_shift NULL: res taken as is -- an identity shortcut), computed while the A tile is staged; when
 * sum_out != NULL the tensor `a` -- the output of Add()+Activation('relu') that later layers and the backward pass
 * read -- is stored there by the same launch.  Only for 1x1, stride-1, unpadded convs with in_c % 32 == 0
 * (dj_conv2d_fwd_addrelu_supported() != 0); any other geometry is an error, never a silent slow path. */
int dj_conv2d_fwd_addrelu_supported(const dj_conv2d_desc* d);
int dj_conv2d_nhwc_fwd_addrelu(const dj_conv2d_desc* d, const float* x, const float* w, const float* bias, float* y,
                               const float* pro_scale, const float* pro_shift, const float* res, int ld_res,
                               const float* res_scale, const float* res_shift, float* sum_out, int ld_sum, int relu,
                               float* stats, void* stream);
/* ... with the split-K workspace of dj_conv2d_nhwc_fwd_ws (same rules, same size query) */
int dj_conv2d_nhwc_fwd_addrelu_ws(const dj_conv2d_desc* d, const float* x, const float* w, const float* bias, float* y,
                                  const float* pro_scale, const float* pro_shift, const float* res, int ld_res,
                                  const float* res_scale, const float* res_shift, float* sum_out, int ld_sum, int relu,
                                  float* stats, float* workspace, long workspace_floats, void* stream);

/* Convolution + the training-mode BatchNormalization that follows it (keras BatchNormalization(axis=3) after Conv2D,
 * L/models/keras_ssd300_dct_j2d_resnet.py:66-99): the conv epilogue adds each tile's column sums / sums of squares to
 * `acc` with fp64 atomics, the workgroup that finishes last turns them into scale = gamma/sqrt(var+eps),
 * shift = beta - mean*scale, save_mean / save_invstd (for the backward pass) and the momentum update of the moving
 * statistics (Bessel-corrected variance), and leaves acc / ticket zero again.  Replaces conv `stats` +
 * dj_bn_train_finalize: no launch between the conv and its consumer.  `acc` = DJ_BN_ACC_REPLICAS*2*out_c doubles and
 * `ticket` = one unsigned, zero before the first launch.  res != NULL selects the residual-add prologue of
 * dj_conv2d_nhwc_fwd_addrelu. */
#define DJ_BN_ACC_REPLICAS 16
typedef struct dj_bn_train {
  double* acc;
  unsigned* ticket;
  const float* gamma;
  const float* beta;
  float* moving_mean; /* NULL: no moving-statistics update */
  float* moving_var;
  float* scale;
  float* shift;
  float* save_mean;
  float* save_invstd;
  float eps, momentum;
} dj_bn_train;
int dj_conv2d_nhwc_fwd_bn(const dj_conv2d_desc* d, const float* x, const float* w, const float* bias, float* y,
                          const float* pro_scale, const float* pro_shift, int pro_relu, const float* res, int ld_res,
                          const float* res_scale, const float* res_shift, float* sum_out, int ld_sum,
                          const dj_bn_train* bn, void* stream);

/* Debug/test switch: 0 forces the generic (branchy, any-shape) implicit-GEMM kernel, 1 (default) lets the
 * launcher pick the branch-free buffer-load kernel whenever its alignment preconditions hold. */
void dj_set_fast_path(int enable);

/* Arithmetic of the implicit-GEMM kernels: 0 = fp32 (default; what every parity claim at 1e-3 refers to): fp32 tensors,
 * fp32 results, from whichever of two kernel families the tuning table names for a geometry -- v_mfma_f32_32x32x2_f32, or
 * the split-bf16 kernels of mode 4 (measured against the fp64 oracle the two are equally far from it, 1.4e-7 .. 4.6e-7
 * rel-L2 per GEMM; configuration indices [N, 2N) of dj_conv2d_tune_set select the latter);
 * 5 = fp32 MFMA instructions only (indices [0, N); the behaviour of mode 0 before round 3's split kernels);
 * 4 ("float32x6") = every product as six bf16 MFMAs on operands split into three bf16 pieces when they go to LDS
 * (hi*hi + hi*mid + mid*hi + hi*lo + mid*mid + lo*hi; dropped: 2^-24 of a product), fp32 accumulation: fp32 results at
 * 6/16 of the fp32 MFMA's matrix-pipe time; 3 ("float32x3") = two pieces, three MFMAs: ~4e-6 rel-L2 per GEMM;
 * (BASELINE config 5, "fp16 MFMA":) 1 = forward GEMMs round both operands to fp16 and gradient GEMMs (dgrad, wgrad,
 * Conv2DTranspose forward) to bf16 as the fragments leave LDS, v_mfma_f32_32x32x8_{f16,bf16} with fp32 accumulation;
 * 2 = bf16 in every GEMM.  Tensors in HBM (activations, weights = the fp32 master copy, gradients, optimizer state)
 * stay fp32 in every mode.  Sets the process-wide default; returns the previous default. */
int dj_set_compute_mode(int mode);
/* Override for the calling thread's launches (-1: follow the process default); returns the previous override.
 * dj_get_compute_mode reports the mode the calling thread's next launch would use. */
int dj_set_thread_compute_mode(int mode);
int dj_get_compute_mode(void);

/* Launch-configuration overrides per conv geometry, filled by the plan-time autotuner: `dir` 0 fwd, 1 dgrad,
 * 2 wgrad (+4: forward that takes BN statistics; 9: the input gradient of dj_conv2d_nhwc_dgrad_bnbwd, which runs other
 * kernels than a plain input gradient and is never split -- without an entry of its own it takes the tile shape of the
 * dir-1 entry); cfg in [0, dj_conv2d_tune_configs()) selects the tile shape
 * (128x128, 128x64, 64x64, 128x32) and schedule variant -- in mode 0 the count is twice that of the other modes, the upper
 * half naming the split-bf16 (float32x6) variants --, `splits` the split-K factor; cfg < 0 removes the override. */
int dj_conv2d_tune_configs(void);
int dj_conv2d_tune_set(int dir, const dj_conv2d_desc* d, int cfg, int splits);
int dj_conv2d_default_config(int dir, const dj_conv2d_desc* d, int* cfg, int* splits);

/* dw = sum over pixels pro(x)^T dy  (TF Conv2DBackpropFilter).  dw is HWIO, overwritten.  The pixel reduction is
 * split over workgroups that accumulate with fp32 atomics; `dw_zeroed` = 1 promises dw already holds zeros (the engine
 * clears its whole flat gradient buffer once per step instead of one memset per layer). */
int dj_conv2d_nhwc_wgrad(const dj_conv2d_desc* d, const float* x, const float* dy, float* dw,
                         const float* pro_scale, const float* pro_shift, int pro_relu, int dw_zeroed, void* stream);

/* ---- the three convolution GEMMs over tensors that carry their storage type (arithmetic mode 1 only) ----
 * Same semantics as the float entry points they generalise; a 16-bit operand needs the branch-free kernel's
 * preconditions (channel counts that are multiples of 32, 16-byte aligned bases, stride-1 or 1x1 input gradients) and is an
 * error otherwise -- never a silent conversion pass.  A 16-bit RESULT is written by one K range: such a launch is never
 * split over workgroups, whatever the tuner registered.
 * dj_conv2d_nhwc_fwd_t = dj_conv2d_nhwc_fwd_ws (res == NULL) / dj_conv2d_nhwc_fwd_addrelu_ws (res != NULL: then pro_relu is
 * ignored, the residual operand has x's type): x, res fp32 | fp16; y fp32 | fp16 (rounded AFTER the fp32 bias / ReLU /
 * statistics epilogue); sum_out fp32 | fp16.  Keras call sites: L/models/keras_ssd300_dct_j2d_resnet.py:77-99,128-163. */
int dj_conv2d_nhwc_fwd_t(const dj_conv2d_desc* d, const void* x, int dt_x, const void* w, int dt_w, const float* bias, void* y,
                         int dt_y, const float* pro_scale, const float* pro_shift, int pro_relu, int relu, float* stats,
                         const void* res, int ld_res, const float* res_scale, const float* res_shift, void* sum_out,
                         int ld_sum, int dt_sum, float* workspace, long workspace_floats, void* stream);
/* dj_conv2d_nhwc_dgrad (z == NULL) / dj_conv2d_nhwc_dgrad_bnbwd (z != NULL: beta must be 0, bias NULL): dy fp32 | bf16,
 * dx any type (bf16 for the gradient of a fp16 activation; accumulates in that type when beta & 1), z fp32 | fp16. */
int dj_conv2d_nhwc_dgrad_t(const dj_conv2d_desc* d, const void* dy, int dt_dy, const void* w, int dt_w, const float* bias,
                           void* dx, int dt_dx, int beta, const void* z, int ld_z, int dt_z, const float* mean, const float* invstd,
                           const float* scale, const float* shift, float* partial, void* stream);
/* `w` of the two entry points above: the fp32 master weights, or their 16-bit shadow in the type the GEMM multiplies in
 * (fp16 forward, bf16 input gradient), refreshed once per step by dj_shadow_weights: w16[i] = fp16(w[i]) and / or
 * wbf[i] = bf16(w[i]) for i < n (n a multiple of 4; either output may be NULL). */
int dj_shadow_weights(const float* w, void* w16, void* wbf, long n, void* stream);
/* dj_conv2d_nhwc_wgrad: x fp32 | fp16, dy fp32 | bf16; dw is the fp32 gradient of the master weights. */
int dj_conv2d_nhwc_wgrad_t(const dj_conv2d_desc* d, const void* x, int dt_x, const void* dy, int dt_dy, float* dw,
                           const float* pro_scale, const float* pro_shift, int pro_relu, int dw_zeroed, void* stream);

/* ---- blocked column reductions (two-stage, deterministic) ------------------------- */
/* Rows of the `partial` buffers written by the *_partial / *_reduce entry points for a
 * [rows][C] operand: partial is [dj_reduce_rows(rows)][2][C] floats. */
int dj_reduce_rows(long rows);
/* partial[blk][0][c] = sum x, partial[blk][1][c] = sum x^2 over the block's rows.  Batch statistics of
 * a BatchNormalization whose input is not a conv output (input BNs, L/models/...resnet.py:446,458,1716). */
int dj_colstats_partial(const float* x, long rows, int C, int ld, float* partial, void* stream);
/* partial[blk][0][c] = sum dy (bias gradient of Conv2D / Dense: TF BiasAddGrad). */
int dj_colsum_partial(const float* dy, long rows, int C, int ld, float* partial, void* stream);
/* out[c] (+)= sum_blk partial[blk][which][c] (accumulated in double). */
int dj_colreduce_finalize(const float* partial, int nrows, int C, int which, float* out, int beta, void* stream);
/* out[c] (+)= sum over rows of x[r][c] in one launch -- bias gradients (BiasAddGrad) of the SSD head convs, whose
 * gradient tensors have at most a few thousand rows. */
int dj_colsum_direct(const float* x, long rows, int C, int ld, float* out, int beta, void* stream);
/* dj_colsum_direct for up to DJ_COLSUM_PARTS tensors in one launch (the bias gradients of the SSD head convolutions,
 * deferred to the end of the backward pass). */
#define DJ_COLSUM_PARTS 32
typedef struct dj_colsum_part {
  const float* x;
  float* out;
  long rows;
  int C, ld, beta;
} dj_colsum_part;
int dj_colsum_multi(const dj_colsum_part* parts, int n_parts, void* stream);

/* ---- keras.layers.BatchNormalization(axis=3) (Keras 2.2.4 defaults eps 1e-3, momentum 0.99) ----
 * Training forward = statistics (conv epilogue `stats` or dj_colstats_partial) -> dj_bn_train_finalize
 * -> (scale, shift); the normalisation itself is applied by the consumer (conv prologue or
 * dj_affine_act).  `conv_bias` (optional) is the bias the producing conv added AFTER its statistics
 * were taken.  moving_* (optional) are updated in place (variance Bessel-corrected, as
 * tf.nn.fused_batch_norm reports it). */
int dj_bn_train_finalize(const float* partial, int nrows, long count, const float* conv_bias, const float* gamma,
                         const float* beta, float eps, float momentum, float* moving_mean, float* moving_var,
                         float* scale, float* shift, float* save_mean, float* save_invstd, int C, void* stream);
/* Inference mode: scale = gamma*rsqrt(moving_var+eps), shift = beta - moving_mean*scale. */
int dj_bn_infer_coeffs(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var,
                       float eps, float* scale, float* shift, int C, void* stream);
/* y = act(x*scale+shift [+ res*res_scale+res_shift]) -- BatchNormalization apply, Activation('relu'),
 * Add()+Activation('relu') (L/models/...resnet.py:96-99,160-163) in one pass.  Null scale = identity. */
int dj_affine_act(const float* x, int ldx, const float* scale, const float* shift, const float* res, int ldres,
                  const float* res_scale, const float* res_shift, float* y, int ldy, long rows, int C, int relu,
                  void* stream);
/* Backward of BN(+ReLU): dy is masked by the ReLU (mask_mode 0 none, 1: y > 0 with y the materialised
 * ReLU output, 2: z*scale+shift > 0), partial[blk] = (sum dy_m, sum dy_m*xhat). */
int dj_bn_bwd_reduce(const float* dy, int ld_dy, const float* z, int ld_z, const float* y, int ld_y,
                     const float* mean, const float* invstd, const float* scale, const float* shift, int mask_mode,
                     long rows, int C, float* partial, void* stream);
/* dgamma, dbeta and k0,k1,k2 with dz = k0*dy_m + k1*z + k2. */
int dj_bn_bwd_finalize(const float* partial, int nrows, long count, const float* gamma, const float* mean,
                       const float* invstd, float* dgamma, float* dbeta, float* k0, float* k1, float* k2, int C,
                       void* stream);
int dj_bn_bwd_apply(const float* dy, int ld_dy, const float* z, int ld_z, const float* y, int ld_y,
                    const float* scale, const float* shift, int mask_mode, const float* k0, const float* k1,
                    const float* k2, float* dz, int ld_dz, long rows, int C, float* dmasked, int ld_dm, int dm_beta,
                    void* stream);
/* ... optional second output of dj_bn_bwd_apply: dmasked[r][c] (+)= dy_m, the ReLU-masked upstream gradient itself --
 * after Add()+Activation('relu') that is the identity shortcut's gradient, written in the same pass (no dj_relu_bwd). */
/* dx (+)= dy * [y > 0]  (Activation('relu') gradient; `activation="relu"` convs of the SSD head). */
int dj_relu_bwd(const float* dy, int ld_dy, const float* y, int ld_y, float* dx, int ld_dx, long rows, int C,
                int beta, void* stream);
/* dst[r][c] (+)= src[r][c]: Concatenate / Reshape+Concatenate(axis=1) and their gradients
 * (L/models/...resnet.py:461,836-879,1713-1714). */
int dj_copy2d(const float* src, long ld_src, float* dst, long ld_dst, long rows, long cols, int beta, void* stream);
/* The same for up to DJ_COPY_PARTS (src, dst) pairs in one launch: Concatenate(axis=1) over the six SSD sources and its
 * gradient (L/models/keras_ssd300_dct_j2d_resnet.py:825-879). */
#define DJ_COPY_PARTS 8
typedef struct dj_copy_part {
  const float* src;
  float* dst;
  long ld_src, ld_dst, rows, cols;
  int beta; /* 1: dst += src */
} dj_copy_part;
int dj_copy2d_multi(const dj_copy_part* parts, int n_parts, void* stream);
/* The elementwise passes above over tensors that carry their storage type (each tensor pointer is followed by its DJ_F32 /
 * DJ_F16 / DJ_BF16 code; any mix; arithmetic in fp32, one rounding where a 16-bit tensor is written).  Same Keras call
 * sites: BatchNormalization / Activation / Add at L/models/keras_ssd300_dct_j2d_resnet.py:80-99,135-163. */
int dj_affine_act_t(const void* x, int dt_x, int ldx, const float* scale, const float* shift, const void* res, int dt_res,
                    int ldres, const float* res_scale, const float* res_shift, void* y, int dt_y, int ldy, long rows, int C,
                    int relu, void* stream);
int dj_bn_bwd_reduce_t(const void* dy, int dt_dy, int ld_dy, const void* z, int dt_z, int ld_z, const void* y, int dt_y,
                       int ld_y, const float* mean, const float* invstd, const float* scale, const float* shift,
                       int mask_mode, long rows, int C, float* partial, void* stream);
int dj_bn_bwd_apply_t(const void* dy, int dt_dy, int ld_dy, const void* z, int dt_z, int ld_z, const void* y, int dt_y,
                      int ld_y, const float* scale, const float* shift, int mask_mode, const float* k0, const float* k1,
                      const float* k2, void* dz, int dt_dz, int ld_dz, long rows, int C, void* dmasked, int dt_dm, int ld_dm,
                      int dm_beta, void* stream);
int dj_relu_bwd_t(const void* dy, int dt_dy, int ld_dy, const void* y, int dt_y, int ld_y, void* dx, int dt_dx, int ld_dx,
                  long rows, int C, int beta, void* stream);
/* dst (+)= src with a change of storage type on the way: a 16-bit backbone tensor handed to a layer that works on fp32
 * (L2Normalization, pooling, Concatenate), and the gradient coming back. */
int dj_copy2d_t(const void* src, int dt_src, long ld_src, void* dst, int dt_dst, long ld_dst, long rows, long cols, int beta,
                void* stream);
/* UpSampling2D() nearest x2 (L/models/...resnet.py:1669) written into a channel slice. */
int dj_upsample2x(const float* x, int ldx, float* y, int ldy, int B, int H, int W, int C, void* stream);

/* ---- L2Normalization (L/keras_layers/keras_layer_L2Normalization.py:61-63) ---- */
int dj_l2norm_fwd(const float* x, int ldx, const float* gamma, float* y, int ldy, float* rnorm, long rows, int C,
                  void* stream);
/* dx (optional, (+)=) and dgamma_partial (optional, [dj_reduce_rows(rows)][2][C], slot 0). */
int dj_l2norm_bwd(const float* dy, int ld_dy, const float* x, int ldx, const float* gamma, const float* rnorm,
                  float* dx, int ld_dx, float* dgamma_partial, long rows, int C, int beta, void* stream);

/* ---- keras.layers.MaxPooling2D: `pool5_ssd` (3,3)/1/'same' (L/models/...resnet.py:481,1110) and the ResNet50RGB
 * stem ZeroPadding2D(1) + (3,3)/2 (C/vgg_jpeg_keras/networks/resnet_dct.py:165-314).  pt/pl = leading pads;
 * pad_zero = 1 when the padding comes from a ZeroPadding2D (zeros take part in the max), 0 for TF 'same'/'valid'.
 * The gradient goes to the first maximum of each window in row-major order.  `argmax` (optional, B*OH*OW*C bytes):
 * forward stores each window's winning tap there and backward reads it instead of rescanning the windows
 * (then `x` may be NULL in backward). ---- */
int dj_maxpool2d_fwd(const float* x, float* y, int B, int H, int W, int C, int OH, int OW, int kh, int kw, int sh,
                     int sw, int pt, int pl, int pad_zero, unsigned char* argmax, void* stream);
int dj_maxpool2d_bwd(const float* x, const float* dy, float* dx, int B, int H, int W, int C, int OH, int OW, int kh,
                     int kw, int sh, int sw, int pt, int pl, int pad_zero, int beta, const unsigned char* argmax,
                     void* stream);

/* ---- Activation('softmax') on the last axis (L/models/...resnet.py:873; Dense(..., softmax)) ---- */
int dj_softmax_fwd(const float* x, float* y, long rows, int C, void* stream);
int dj_softmax_bwd(const float* p, const float* dp, long ld_dp, float* dx, long rows, int C, int beta, void* stream);

/* ---- SSDLoss.compute_loss (L/keras_loss_function/keras_ssd_loss.py:98-211) ----
 * y_true / y_pred: [nbox][n_cls+12] with nbox = batch * boxes-per-image; mining is batch-wide.
 * out5 = {loss (already the Keras batch mean), n_positive, n_negative_kept, class part, loc part}. */
long dj_ssd_loss_workspace_floats(long nbox);
int dj_ssd_loss_fwd(const float* y_true, const float* y_pred, long nbox, int n_cls, int neg_pos_ratio, int n_neg_min,
                    float alpha, float* workspace, float* out5, void* stream);
int dj_ssd_loss_bwd(const float* y_true, const float* y_pred, long nbox, int n_cls, float alpha, float upstream,
                    const float* workspace, const float* out5, float* d_pred, void* stream);

/* ---- keras.losses.categorical_crossentropy on probabilities (C/config/resnet/config_file.py:64) ---- */
int dj_categorical_crossentropy(const float* y_true, const float* probs, long rows, int C, float upstream,
                                float* loss_rows, float* d_probs, float* out_mean, void* stream);

/* ---- keras.optimizers.SGD.get_updates (L/training_dct_pascal_j2d_resnet.py:152;
 * C/config/resnet/config_file.py:58-63) with the l2 kernel_regularizer gradient (2*l2*w) and the
 * data-parallel 1/world_size folded in.  lr_t = lr / (1 + decay * iterations) is computed by the caller. */
int dj_sgd_momentum_update(float* param, const float* grad, float* velocity, long n, float lr_t, float momentum,
                           int nesterov, float l2, float grad_scale, float* sumsq, void* stream);

/* ---- DecodeDetections (L/keras_layers/keras_layer_DecodeDetections.py:109-265; numpy twin
 * L/ssd_encoder_decoder/ssd_output_decoder.py:111-226): y_pred [batch][n_boxes][n_classes+12] (centroid offsets, anchors,
 * variances in the last 12 columns) -> out [batch][top_k][6] = (class, confidence, xmin, ymin, xmax, ymax) sorted by
 * confidence, zero padded.  Per class: confidence > thresh, greedy NMS (drop IoU > iou_threshold), at most
 * nms_max_output_size.  Limits: n_boxes <= 12288, (n_classes-1)*nms_max_output_size <= 8192. ---- */
long dj_decode_detections_workspace_floats(int batch, int n_boxes, int n_classes, int nms_max_output_size);
int dj_decode_detections(const float* y_pred, int batch, int n_boxes, int n_classes, float confidence_thresh,
                         float iou_threshold, int top_k, int nms_max_output_size, int normalize_coords, int img_height,
                         int img_width, float* workspace, float* out, void* stream);

/* ---- DecodeDetectionsFast (L/keras_layers/keras_layer_DecodeDetectionsFast.py:108-215, mode='inference_fast'): each box
 * keeps its arg-max class and that confidence, background boxes are dropped, threshold, ONE class-agnostic greedy NMS
 * (at most nms_max_output_size <= 8192), top-k.  Same tensors and output layout as dj_decode_detections. ---- */
long dj_decode_detections_fast_workspace_floats(int batch, int n_boxes, int nms_max_output_size);
int dj_decode_detections_fast(const float* y_pred, int batch, int n_boxes, int n_classes, float confidence_thresh,
                              float iou_threshold, int top_k, int nms_max_output_size, int normalize_coords,
                              int img_height, int img_width, float* workspace, float* out, void* stream);

/* ---- SSDInputEncoder.__call__ for coords='centroids' (L/ssd_encoder_decoder/ssd_input_encoder.py:277-418,
 * matching_utils.py:22-116, L/bounding_box_utils/bounding_box_utils.py:283-383): labels [batch][max_gt][5] =
 * (class id, xmin, ymin, xmax, ymax) in pixels (float64, as the reference computes), n_gt [batch] valid rows,
 * anchors [n_boxes][8] = template anchor (cx, cy, w, h) + 4 variances (float64) -> y_true [batch][n_boxes]
 * [n_classes + 12] fp32.  n_classes includes the background class.  border_pixels: 0 'half', 1 'include', -1
 * 'exclude'; multi = 1 for matching_type 'multi'.  Limits: max_gt <= 128, n_boxes <= 12288.  Degenerate boxes
 * (xmax <= xmin ...) must be rejected by the caller, as the reference raises DegenerateBoxError on the host. ---- */
int dj_ssd_encode_targets(const double* labels, const int* n_gt, int batch, int max_gt, const double* anchors,
                          int n_boxes, int n_classes, int img_height, int img_width, int normalize_coords,
                          int border_pixels, int multi, double pos_iou_threshold, double neg_iou_limit,
                          int background_id, float* y_true, void* stream);

/* ---- GlobalAveragePooling2D (C/vgg_jpeg_keras/networks/resnet_dct.py:415) ---- */
int dj_global_avg_pool_fwd(const float* x, float* y, int B, int HW, int C, void* stream);
int dj_global_avg_pool_bwd(const float* dy, float* dx, int B, int HW, int C, int beta, void* stream);

#ifdef __cplusplus
}
#endif
#endif
