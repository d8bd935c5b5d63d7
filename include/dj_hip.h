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
 *   - `stream` is a hipStream_t passed as void*; nothing here allocates, synchronises
 *     or touches global state, so calls are capturable in a hipGraph;
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
int dj_abi_version(void);

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

/* Number of row tiles the forward launcher uses for this geometry == leading dimension
 * of the `stats` partial buffer ([rows][2][out_c] floats). */
int dj_conv2d_fwd_stats_rows(const dj_conv2d_desc* d);

/* y = act(conv(pro(x), w) + bias).  Replaces Conv2D.call (TF conv2d + bias_add [+ relu]).
 *   pro_scale/pro_shift ([in_c], optional): x is read as x*scale+shift (then ReLU if
 *     pro_relu) on in-bounds pixels only -- the BatchNormalization+Activation that Keras
 *     runs between two convs (L/models/...resnet.py:80-81,90-91) folded into the load.
 *   stats (optional): per-row-tile column sums / sums of squares of conv(x,w) WITHOUT bias,
 *     consumed by dj_bn_finalize (training-mode BatchNormalization statistics). */
int dj_conv2d_nhwc_fwd(const dj_conv2d_desc* d, const float* x, const float* w, const float* bias,
                       float* y, const float* pro_scale, const float* pro_shift, int pro_relu,
                       int relu, float* stats, void* stream);

/* dx (+)= conv_transpose(dy, w) [+ bias].  Gradient of Conv2D w.r.t. its input (TF
 * Conv2DBackpropInput); with `bias` it is also the forward of keras.layers.Conv2DTranspose
 * (L/models/...resnet.py:1709-1711), whose (kh,kw,out,in) kernel is the HWIO kernel of
 * the convolution it transposes.  beta=1 accumulates into dx. */
int dj_conv2d_nhwc_dgrad(const dj_conv2d_desc* d, const float* dy, const float* w, const float* bias,
                         float* dx, int beta, void* stream);

/* dw = sum over pixels pro(x)^T dy  (TF Conv2DBackpropFilter).  dw is HWIO, overwritten. */
int dj_conv2d_nhwc_wgrad(const dj_conv2d_desc* d, const float* x, const float* dy, float* dw,
                         const float* pro_scale, const float* pro_shift, int pro_relu, void* stream);

#ifdef __cplusplus
}
#endif
#endif
