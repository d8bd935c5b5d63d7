// BatchNormalization (training / inference), ReLU, residual add, L2Normalization and the
// blocked column reductions they share.  All HBM-bound: 16-byte coalesced accesses,
// two-stage deterministic reductions (per-block partials, then a per-channel finalize in
// double precision).  Reference call sites: keras.layers.BatchNormalization(axis=3) at
// localisation_part/models/keras_ssd300_dct_j2d_resnet.py:80,90,96,135,145,151,160,446,458,1716;
// Add + Activation('relu') :98-99,162-163; L2Normalization
// localisation_part/keras_layers/keras_layer_L2Normalization.py:61-63.
#include "../../include/dj_hip.h"
#include "dj_common.h"

#define DJ_RB 64  // rows per reduction block

template <int VEC>
struct VecIO;
template <>
struct VecIO<4> {
  static __device__ __forceinline__ f32x4 ld(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ void st(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <>
struct VecIO<1> {
  static __device__ __forceinline__ f32x4 ld(const float* p) { return f32x4{p[0], 0.f, 0.f, 0.f}; }
  static __device__ __forceinline__ void st(float* p, f32x4 v) { p[0] = v.x; }
};

// ---------------------------------------------------------------------------------
// Blocked column reduction: out[blk][q][c] = sum over the block's rows of f_q(row, c), q in {0,1}
// ---------------------------------------------------------------------------------
// Every functor maps (row, first column of a group of VEC) -> two per-column contributions.
struct ColStatsF {  // (x, x^2)
  const float* x;
  int ld;
  template <int VEC>
  __device__ __forceinline__ void eval(long r, int c, f32x4& v0, f32x4& v1) const {
    f32x4 v = VecIO<VEC>::ld(x + r * ld + c);
    v0 = v;
    v1 = v * v;
  }
};

// dy masked by a ReLU, and dy*xhat.  mask_mode 0: none; 1: relu output tensor `y` > 0;
// 2: own affine z*scale+shift > 0.
struct BnBwdF {
  const float* dy;
  int ld_dy;
  const float* z;
  int ld_z;
  const float* y;
  int ld_y;
  const float* mean;
  const float* invstd;
  const float* scale;
  const float* shift;
  int mask_mode;
  template <int VEC>
  __device__ __forceinline__ void eval(long r, int c, f32x4& v0, f32x4& v1) const {
    using IO = VecIO<VEC>;
    f32x4 g = IO::ld(dy + r * ld_dy + c);
    f32x4 zz = IO::ld(z + r * ld_z + c);
    f32x4 m = {1.f, 1.f, 1.f, 1.f};
    if (mask_mode == 1)
      m = IO::ld(y + r * ld_y + c);
    else if (mask_mode == 2)
      m = zz * IO::ld(scale + c) + IO::ld(shift + c);
    g.x = (m.x > 0.f) ? g.x : 0.f;
    g.y = (m.y > 0.f) ? g.y : 0.f;
    g.z = (m.z > 0.f) ? g.z : 0.f;
    g.w = (m.w > 0.f) ? g.w : 0.f;
    v0 = g;
    v1 = g * (zz - IO::ld(mean + c)) * IO::ld(invstd + c);
  }
};

// dy * x * rnorm[row]  (dgamma of L2Normalization), second slot unused
struct L2DgF {
  const float* dy;
  const float* x;
  const float* rnorm;
  int ld_dy, ld_x;
  template <int VEC>
  __device__ __forceinline__ void eval(long r, int c, f32x4& v0, f32x4& v1) const {
    v0 = VecIO<VEC>::ld(dy + r * ld_dy + c) * VecIO<VEC>::ld(x + r * ld_x + c) * rnorm[r];
    v1 = f32x4{0.f, 0.f, 0.f, 0.f};
  }
};

// plain column sum of dy (bias gradients)
struct ColSumF {
  const float* dy;
  int ld;
  template <int VEC>
  __device__ __forceinline__ void eval(long r, int c, f32x4& v0, f32x4& v1) const {
    v0 = VecIO<VEC>::ld(dy + r * ld + c);
    v1 = f32x4{0.f, 0.f, 0.f, 0.f};
  }
};

// block = 256 threads as (TX column groups of VEC) x (256/TX rows); grid = (row blocks, column blocks)
template <typename F, int VEC>
__global__ __launch_bounds__(256) void dj_colreduce_kernel(F f, long rows, int C, int tx_log2, float* partial) {
  __shared__ f32x4 red[2][256];
  const int TX = 1 << tx_log2;
  const int TY = 256 >> tx_log2;
  const int tx = threadIdx.x & (TX - 1), ty = threadIdx.x >> tx_log2;
  const int c = (blockIdx.y * TX + tx) * VEC;
  const long r0 = (long)blockIdx.x * DJ_RB;
  const long r1 = (r0 + DJ_RB < rows) ? r0 + DJ_RB : rows;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    for (long r = r0 + ty; r < r1; r += TY) {
      f32x4 a, b;
      f.template eval<VEC>(r, c, a, b);
      s0 += a;
      s1 += b;
    }
  }
  red[0][threadIdx.x] = s0;
  red[1][threadIdx.x] = s1;
  __syncthreads();
  if (ty == 0 && c < C) {
    for (int j = 1; j < TY; ++j) {
      s0 += red[0][j * TX + tx];
      s1 += red[1][j * TX + tx];
    }
    float* p0 = partial + ((size_t)blockIdx.x * 2 + 0) * C + c;
    float* p1 = partial + ((size_t)blockIdx.x * 2 + 1) * C + c;
    VecIO<VEC>::st(p0, s0);
    VecIO<VEC>::st(p1, s1);
  }
}

template <typename F>
static int launch_colreduce(F f, long rows, int C, bool vec4, float* partial, hipStream_t s, const char* name) {
  const int groups = vec4 ? C / 4 : C;
  int tx_log2 = 6;
  while (tx_log2 > 2 && (1 << (tx_log2 - 1)) >= groups) --tx_log2;
  int TX = 1 << tx_log2;
  dim3 grid((unsigned)dj_cdiv(rows, DJ_RB), (unsigned)dj_cdiv(groups, TX));
  if (vec4)
    hipLaunchKernelGGL((dj_colreduce_kernel<F, 4>), grid, dim3(256), 0, s, f, rows, C, tx_log2, partial);
  else
    hipLaunchKernelGGL((dj_colreduce_kernel<F, 1>), grid, dim3(256), 0, s, f, rows, C, tx_log2, partial);
  DJ_CHECK_LAUNCH(name);
  return DJ_OK;
}

static inline bool al16p(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// Column sums of the two partial slots for DJ_FIN_CH channels per block: 1024 threads = 16 channels x 64 row lanes
// (a 38x38x256 layer has 722 partial rows but only 256 channels: few channels per block keeps 16 blocks busy and
// 12 rows per lane), 64-byte row segments, double accumulation.  Valid in threads with ty == 0 (c < C).
#define DJ_FIN_THREADS 1024
#define DJ_FIN_CH 16
#define DJ_FIN_LANES (DJ_FIN_THREADS / DJ_FIN_CH)
__device__ __forceinline__ void dj_partial_sums(const float* partial, int nrows, int C, int which_mask, double& s0,
                                                double& s1) {
  __shared__ double red0[DJ_FIN_LANES][DJ_FIN_CH + 1];
  __shared__ double red1[DJ_FIN_LANES][DJ_FIN_CH + 1];
  const int tx = threadIdx.x & (DJ_FIN_CH - 1), ty = threadIdx.x / DJ_FIN_CH;
  const int c = blockIdx.x * DJ_FIN_CH + tx;
  double a = 0.0, b = 0.0;
  if (c < C) {
    // four rows in flight per thread: the loop is latency-bound (dependent loads otherwise)
    double a1 = 0.0, b1 = 0.0, a2 = 0.0, b2 = 0.0, a3 = 0.0, b3 = 0.0;
    constexpr int L = DJ_FIN_LANES;
    int r = ty;
    for (; r + 3 * L < nrows; r += 4 * L) {
      const float* p = partial + (size_t)r * 2 * C + c;
      float x0 = 0.f, x1 = 0.f, x2 = 0.f, x3 = 0.f, y0 = 0.f, y1 = 0.f, y2 = 0.f, y3 = 0.f;
      if (which_mask & 1) {
        x0 = p[0];
        x1 = p[(size_t)(2 * L) * C];
        x2 = p[(size_t)(4 * L) * C];
        x3 = p[(size_t)(6 * L) * C];
      }
      if (which_mask & 2) {
        y0 = p[C];
        y1 = p[(size_t)(2 * L + 1) * C];
        y2 = p[(size_t)(4 * L + 1) * C];
        y3 = p[(size_t)(6 * L + 1) * C];
      }
      a += (double)x0;
      a1 += (double)x1;
      a2 += (double)x2;
      a3 += (double)x3;
      b += (double)y0;
      b1 += (double)y1;
      b2 += (double)y2;
      b3 += (double)y3;
    }
    for (; r < nrows; r += L) {
      if (which_mask & 1) a += (double)partial[((size_t)r * 2 + 0) * C + c];
      if (which_mask & 2) b += (double)partial[((size_t)r * 2 + 1) * C + c];
    }
    a += a1 + a2 + a3;
    b += b1 + b2 + b3;
  }
  red0[ty][tx] = a;
  red1[ty][tx] = b;
  __syncthreads();
  if (ty < 8) {   // two-level: 8 lanes fold 8 rows each, then lane 0 folds those
    for (int j = ty + 8; j < DJ_FIN_LANES; j += 8) {
      a += red0[j][tx];
      b += red1[j][tx];
    }
    red0[ty][tx] = a;
    red1[ty][tx] = b;
  }
  __syncthreads();
  if (ty == 0) {
    for (int j = 1; j < 8; ++j) {
      a += red0[j][tx];
      b += red1[j][tx];
    }
  }
  s0 = a;
  s1 = b;
}

extern "C" int dj_reduce_rows(long rows) { return dj_cdiv(rows, DJ_RB); }

extern "C" int dj_colstats_partial(const float* x, long rows, int C, int ld, float* partial, void* stream) {
  DJ_CHECK_ARG(x && partial && rows > 0 && C > 0 && ld >= C, "colstats: bad arguments");
  ColStatsF f{x, ld};
  bool v4 = C % 4 == 0 && ld % 4 == 0 && al16p(x) && al16p(partial);
  return launch_colreduce(f, rows, C, v4, partial, (hipStream_t)stream, "dj_colstats_partial");
}

extern "C" int dj_colsum_partial(const float* dy, long rows, int C, int ld, float* partial, void* stream) {
  DJ_CHECK_ARG(dy && partial && rows > 0 && C > 0 && ld >= C, "colsum: bad arguments");
  ColSumF f{dy, ld};
  bool v4 = C % 4 == 0 && ld % 4 == 0 && al16p(dy) && al16p(partial);
  return launch_colreduce(f, rows, C, v4, partial, (hipStream_t)stream, "dj_colsum_partial");
}

// out[c] (+)= sum_r partial[r][which][c]
__global__ __launch_bounds__(DJ_FIN_THREADS) void dj_colreduce_finalize_kernel(const float* partial, int nrows, int C,
                                                                               int which, float* out, int beta) {
  double s0, s1;
  dj_partial_sums(partial, nrows, C, which == 0 ? 1 : 2, s0, s1);
  int c = blockIdx.x * DJ_FIN_CH + (threadIdx.x & (DJ_FIN_CH - 1));
  if ((threadIdx.x / DJ_FIN_CH) != 0 || c >= C) return;
  float v = (float)(which == 0 ? s0 : s1);
  if (beta) v += out[c];
  out[c] = v;
}

extern "C" int dj_colreduce_finalize(const float* partial, int nrows, int C, int which, float* out, int beta,
                                     void* stream) {
  DJ_CHECK_ARG(partial && out && nrows > 0 && C > 0 && (which == 0 || which == 1), "colreduce_finalize: bad arguments");
  hipLaunchKernelGGL(dj_colreduce_finalize_kernel, dim3(dj_cdiv(C, DJ_FIN_CH)), dim3(DJ_FIN_THREADS), 0, (hipStream_t)stream,
                     partial, nrows, C, which, out, beta);
  DJ_CHECK_LAUNCH("dj_colreduce_finalize");
  return DJ_OK;
}

// out[c] (+)= sum_r x[r][c] in ONE launch (1024 threads = 16 columns x 64 row lanes, double accumulation): the bias
// gradients of the SSD head convolutions have at most a few thousand rows, where two launches cost more than the sum
__device__ __forceinline__ void dj_colsum_direct_body(const float* x, long rows, int C, int ld, float* out, int beta) {
  __shared__ double red[DJ_FIN_LANES][DJ_FIN_CH + 1];
  const int tx = threadIdx.x & (DJ_FIN_CH - 1), ty = threadIdx.x / DJ_FIN_CH;
  const int c = blockIdx.x * DJ_FIN_CH + tx;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (c < C) {
    long r = ty;
    for (; r + 3 * DJ_FIN_LANES < rows; r += 4 * DJ_FIN_LANES) {
      const float* p = x + r * ld + c;
      float v0 = p[0], v1 = p[(long)DJ_FIN_LANES * ld], v2 = p[(long)2 * DJ_FIN_LANES * ld], v3 = p[(long)3 * DJ_FIN_LANES * ld];
      a0 += (double)v0;
      a1 += (double)v1;
      a2 += (double)v2;
      a3 += (double)v3;
    }
    for (; r < rows; r += DJ_FIN_LANES) a0 += (double)x[r * ld + c];
  }
  red[ty][tx] = a0 + a1 + a2 + a3;
  __syncthreads();
  if (ty < 8) {
    double a = red[ty][tx];
    for (int j = ty + 8; j < DJ_FIN_LANES; j += 8) a += red[j][tx];
    red[ty][tx] = a;
  }
  __syncthreads();
  if (ty == 0 && c < C) {
    double a = red[0][tx];
    for (int j = 1; j < 8; ++j) a += red[j][tx];
    float v = (float)a;
    out[c] = beta ? out[c] + v : v;
  }
}

__global__ __launch_bounds__(DJ_FIN_THREADS) void dj_colsum_direct_kernel(const float* x, long rows, int C, int ld, float* out,
                                                                          int beta) {
  dj_colsum_direct_body(x, rows, C, ld, out, beta);
}

// the same for up to DJ_COLSUM_PARTS tensors in one launch (blockIdx.y = tensor): the twenty-odd bias gradients of the SSD
// head and extra-feature convolutions, each a 5 us launch on the backward chain, become one launch at its end
struct DjColsumParts {
  dj_colsum_part part[DJ_COLSUM_PARTS];
};
__global__ __launch_bounds__(DJ_FIN_THREADS) void dj_colsum_multi_kernel(const DjColsumParts ps) {
  const dj_colsum_part& p = ps.part[blockIdx.y];
  if ((int)blockIdx.x * DJ_FIN_CH >= p.C) return;   // uniform per workgroup: no barrier is skipped by part of it
  dj_colsum_direct_body(p.x, p.rows, p.C, p.ld, p.out, p.beta);
}

extern "C" int dj_colsum_multi(const dj_colsum_part* parts, int n_parts, void* stream) {
  DJ_CHECK_ARG(parts && n_parts >= 1 && n_parts <= DJ_COLSUM_PARTS, "colsum_multi: 1..%d parts", DJ_COLSUM_PARTS);
  DjColsumParts ps;
  int most = 0;
  for (int i = 0; i < n_parts; ++i) {
    const dj_colsum_part& p = parts[i];
    DJ_CHECK_ARG(p.x && p.out && p.rows > 0 && p.C > 0 && p.ld >= p.C, "colsum_multi: bad part %d", i);
    ps.part[i] = p;
    if (p.C > most) most = p.C;
  }
  hipLaunchKernelGGL(dj_colsum_multi_kernel, dim3(dj_cdiv(most, DJ_FIN_CH), n_parts), dim3(DJ_FIN_THREADS), 0,
                     (hipStream_t)stream, ps);
  DJ_CHECK_LAUNCH("dj_colsum_multi");
  return DJ_OK;
}

extern "C" int dj_colsum_direct(const float* x, long rows, int C, int ld, float* out, int beta, void* stream) {
  DJ_CHECK_ARG(x && out && rows > 0 && C > 0 && ld >= C, "colsum_direct: bad arguments");
  hipLaunchKernelGGL(dj_colsum_direct_kernel, dim3(dj_cdiv(C, DJ_FIN_CH)), dim3(DJ_FIN_THREADS), 0, (hipStream_t)stream, x,
                     rows, C, ld, out, beta);
  DJ_CHECK_LAUNCH("dj_colsum_direct");
  return DJ_OK;
}

// ---------------------------------------------------------------------------------
// BatchNormalization, training mode: finalize statistics -> (scale, shift)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(DJ_FIN_THREADS) void dj_bn_train_finalize_kernel(const float* partial, int nrows, double count, const float* conv_bias,
                                            const float* gamma, const float* beta, float eps, float momentum,
                                            float* moving_mean, float* moving_var, float* scale, float* shift,
                                            float* save_mean, float* save_invstd, int C) {
  double s, q;
  dj_partial_sums(partial, nrows, C, 3, s, q);
  int c = blockIdx.x * DJ_FIN_CH + (threadIdx.x & (DJ_FIN_CH - 1));
  if ((threadIdx.x / DJ_FIN_CH) != 0 || c >= C) return;
  double m = s / count;
  double var = q / count - m * m;
  if (var < 0.0) var = 0.0;
  // the partials were taken on the conv accumulator before its bias: shift the mean only
  double mean = m + (conv_bias ? (double)conv_bias[c] : 0.0);
  float invstd = (float)(1.0 / sqrt(var + (double)eps));
  float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  save_mean[c] = (float)mean;
  save_invstd[c] = invstd;
  if (moving_mean) {
    double unbiased = var * (count / (count > 1.0 ? count - 1.0 : 1.0));
    moving_mean[c] = moving_mean[c] * momentum + (float)mean * (1.f - momentum);
    moving_var[c] = moving_var[c] * momentum + (float)unbiased * (1.f - momentum);
  }
}

extern "C" int dj_bn_train_finalize(const float* partial, int nrows, long count, const float* conv_bias,
                                    const float* gamma, const float* beta, float eps, float momentum,
                                    float* moving_mean, float* moving_var, float* scale, float* shift,
                                    float* save_mean, float* save_invstd, int C, void* stream) {
  DJ_CHECK_ARG(partial && gamma && beta && scale && shift && save_mean && save_invstd, "bn_train_finalize: null");
  DJ_CHECK_ARG(nrows > 0 && count > 0 && C > 0, "bn_train_finalize: bad sizes");
  DJ_CHECK_ARG((moving_mean == nullptr) == (moving_var == nullptr), "bn_train_finalize: moving stats come together");
  hipLaunchKernelGGL(dj_bn_train_finalize_kernel, dim3(dj_cdiv(C, DJ_FIN_CH)), dim3(DJ_FIN_THREADS), 0, (hipStream_t)stream, partial,
                     nrows, (double)count, conv_bias, gamma, beta, eps, momentum, moving_mean, moving_var, scale,
                     shift, save_mean, save_invstd, C);
  DJ_CHECK_LAUNCH("dj_bn_train_finalize");
  return DJ_OK;
}

__global__ void dj_bn_infer_coeffs_kernel(const float* gamma, const float* beta, const float* moving_mean,
                                          const float* moving_var, float eps, float* scale, float* shift, int C) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float sc = gamma[c] * rsqrtf(moving_var[c] + eps);
  scale[c] = sc;
  shift[c] = beta[c] - moving_mean[c] * sc;
}

extern "C" int dj_bn_infer_coeffs(const float* gamma, const float* beta, const float* moving_mean,
                                  const float* moving_var, float eps, float* scale, float* shift, int C,
                                  void* stream) {
  DJ_CHECK_ARG(gamma && beta && moving_mean && moving_var && scale && shift && C > 0, "bn_infer_coeffs: bad arguments");
  hipLaunchKernelGGL(dj_bn_infer_coeffs_kernel, dim3(dj_cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, gamma, beta,
                     moving_mean, moving_var, eps, scale, shift, C);
  DJ_CHECK_LAUNCH("dj_bn_infer_coeffs");
  return DJ_OK;
}

// ---------------------------------------------------------------------------------
// y = act(x*scale + shift [+ res*res_scale + res_shift]); scale/shift may be null (identity)
// ---------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void dj_affine_act_kernel(const float* x, int ldx, const float* scale,
                                                             const float* shift, const float* res, int ldres,
                                                             const float* rscale, const float* rshift, float* y,
                                                             int ldy, long rows, int C, int relu) {
  using IO = VecIO<VEC>;
  const int cv = C / VEC;
  long total = rows * cv;
  const float floor_ = relu ? 0.f : -INFINITY;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i / cv;
    int c = (int)(i - r * cv) * VEC;
    f32x4 v = IO::ld(x + r * ldx + c);
    if (scale) v = v * IO::ld(scale + c) + IO::ld(shift + c);
    if (res) {
      f32x4 t = IO::ld(res + r * ldres + c);
      if (rscale) t = t * IO::ld(rscale + c) + IO::ld(rshift + c);
      v += t;
    }
    v.x = fmaxf(v.x, floor_);
    v.y = fmaxf(v.y, floor_);
    v.z = fmaxf(v.z, floor_);
    v.w = fmaxf(v.w, floor_);
    IO::st(y + r * ldy + c, v);
  }
}

static inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

static inline int ew_blocks(long total) {
  long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

extern "C" int dj_affine_act(const float* x, int ldx, const float* scale, const float* shift, const float* res,
                             int ldres, const float* res_scale, const float* res_shift, float* y, int ldy,
                             long rows, int C, int relu, void* stream) {
  DJ_CHECK_ARG(x && y && rows > 0 && C > 0 && ldx >= C && ldy >= C, "affine_act: bad arguments");
  DJ_CHECK_ARG((scale == nullptr) == (shift == nullptr) && (res_scale == nullptr) == (res_shift == nullptr),
               "affine_act: scale/shift come together");
  DJ_CHECK_ARG(res || !res_scale, "affine_act: res_scale without res");
  hipStream_t s = (hipStream_t)stream;
  bool v4 = (C % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && (!res || ldres % 4 == 0) && al16(x) && al16(y) &&
            al16(res) && al16(scale) && al16(shift) && al16(res_scale) && al16(res_shift);
  if (v4) {
    hipLaunchKernelGGL(dj_affine_act_kernel<4>, dim3(ew_blocks(rows * (C / 4))), dim3(256), 0, s, x, ldx, scale, shift,
                       res, ldres, res_scale, res_shift, y, ldy, rows, C, relu);
  } else {
    hipLaunchKernelGGL(dj_affine_act_kernel<1>, dim3(ew_blocks(rows * C)), dim3(256), 0, s, x, ldx, scale, shift, res,
                       ldres, res_scale, res_shift, y, ldy, rows, C, relu);
  }
  DJ_CHECK_LAUNCH("dj_affine_act");
  return DJ_OK;
}

// ---------------------------------------------------------------------------------
// BatchNormalization backward
// ---------------------------------------------------------------------------------
extern "C" int dj_bn_bwd_reduce(const float* dy, int ld_dy, const float* z, int ld_z, const float* y, int ld_y,
                                const float* mean, const float* invstd, const float* scale, const float* shift,
                                int mask_mode, long rows, int C, float* partial, void* stream) {
  DJ_CHECK_ARG(dy && z && mean && invstd && partial && rows > 0 && C > 0, "bn_bwd_reduce: bad arguments");
  DJ_CHECK_ARG(mask_mode >= 0 && mask_mode <= 2, "bn_bwd_reduce: mask_mode");
  DJ_CHECK_ARG(mask_mode != 1 || y, "bn_bwd_reduce: mask_mode 1 needs y");
  DJ_CHECK_ARG(mask_mode != 2 || (scale && shift), "bn_bwd_reduce: mask_mode 2 needs scale/shift");
  BnBwdF f{dy, ld_dy, z, ld_z, y, ld_y, mean, invstd, scale, shift, mask_mode};
  bool v4 = C % 4 == 0 && ld_dy % 4 == 0 && ld_z % 4 == 0 && (mask_mode != 1 || ld_y % 4 == 0) && al16p(dy) && al16p(z) &&
            al16p(y) && al16p(mean) && al16p(invstd) && al16p(scale) && al16p(shift) && al16p(partial);
  return launch_colreduce(f, rows, C, v4, partial, (hipStream_t)stream, "dj_bn_bwd_reduce");
}

// dgamma, dbeta and the coefficients of dz = k0*dy_masked + k1*z + k2
__global__ __launch_bounds__(DJ_FIN_THREADS) void dj_bn_bwd_finalize_kernel(const float* partial, int nrows, double count, const float* gamma,
                                          const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                          float* k0, float* k1, float* k2, int C) {
  double sb, sg;
  dj_partial_sums(partial, nrows, C, 3, sb, sg);
  int c = blockIdx.x * DJ_FIN_CH + (threadIdx.x & (DJ_FIN_CH - 1));
  if ((threadIdx.x / DJ_FIN_CH) != 0 || c >= C) return;
  dbeta[c] = (float)sb;
  dgamma[c] = (float)sg;
  double is = (double)invstd[c], sc = (double)gamma[c] * is;
  double c1 = sb / count, c2 = sg / count;
  k0[c] = (float)sc;
  k1[c] = (float)(-sc * c2 * is);
  k2[c] = (float)(-sc * c1 + sc * c2 * (double)mean[c] * is);
}

extern "C" int dj_bn_bwd_finalize(const float* partial, int nrows, long count, const float* gamma, const float* mean,
                                  const float* invstd, float* dgamma, float* dbeta, float* k0, float* k1, float* k2,
                                  int C, void* stream) {
  DJ_CHECK_ARG(partial && gamma && mean && invstd && dgamma && dbeta && k0 && k1 && k2, "bn_bwd_finalize: null");
  DJ_CHECK_ARG(nrows > 0 && count > 0 && C > 0, "bn_bwd_finalize: bad sizes");
  hipLaunchKernelGGL(dj_bn_bwd_finalize_kernel, dim3(dj_cdiv(C, DJ_FIN_CH)), dim3(DJ_FIN_THREADS), 0, (hipStream_t)stream, partial,
                     nrows, (double)count, gamma, mean, invstd, dgamma, dbeta, k0, k1, k2, C);
  DJ_CHECK_LAUNCH("dj_bn_bwd_finalize");
  return DJ_OK;
}

template <int VEC>
__global__ __launch_bounds__(256) void dj_bn_bwd_apply_kernel(const float* dy, int ld_dy, const float* z, int ld_z,
                                                               const float* y, int ld_y, const float* scale,
                                                               const float* shift, int mask_mode, const float* k0,
                                                               const float* k1, const float* k2, float* dz, int ld_dz,
                                                               long rows, int C, float* dmasked, int ld_dm,
                                                               int dm_beta) {
  using IO = VecIO<VEC>;
  const int cv = C / VEC;
  long total = rows * cv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i / cv;
    int c = (int)(i - r * cv) * VEC;
    f32x4 g = IO::ld(dy + r * ld_dy + c);
    f32x4 zz = IO::ld(z + r * ld_z + c);
    f32x4 m = {1.f, 1.f, 1.f, 1.f};
    if (mask_mode == 1) {
      m = IO::ld(y + r * ld_y + c);
    } else if (mask_mode == 2) {
      m = zz * IO::ld(scale + c) + IO::ld(shift + c);
    }
    g.x = (m.x > 0.f) ? g.x : 0.f;
    g.y = (m.y > 0.f) ? g.y : 0.f;
    g.z = (m.z > 0.f) ? g.z : 0.f;
    g.w = (m.w > 0.f) ? g.w : 0.f;
    IO::st(dz + r * ld_dz + c, IO::ld(k0 + c) * g + IO::ld(k1 + c) * zz + IO::ld(k2 + c));
    if (dmasked) {   // the masked upstream gradient is also the identity shortcut's gradient (Add + ReLU backward)
      float* d = dmasked + r * ld_dm + c;
      IO::st(d, dm_beta ? g + IO::ld(d) : g);
    }
  }
}

extern "C" int dj_bn_bwd_apply(const float* dy, int ld_dy, const float* z, int ld_z, const float* y, int ld_y,
                               const float* scale, const float* shift, int mask_mode, const float* k0,
                               const float* k1, const float* k2, float* dz, int ld_dz, long rows, int C,
                               float* dmasked, int ld_dm, int dm_beta, void* stream) {
  DJ_CHECK_ARG(dy && z && k0 && k1 && k2 && dz && rows > 0 && C > 0, "bn_bwd_apply: bad arguments");
  DJ_CHECK_ARG(!dmasked || ld_dm >= C, "bn_bwd_apply: ld_dm < C");
  DJ_CHECK_ARG(mask_mode >= 0 && mask_mode <= 2, "bn_bwd_apply: mask_mode");
  DJ_CHECK_ARG(mask_mode != 1 || y, "bn_bwd_apply: mask_mode 1 needs y");
  DJ_CHECK_ARG(mask_mode != 2 || (scale && shift), "bn_bwd_apply: mask_mode 2 needs scale/shift");
  hipStream_t s = (hipStream_t)stream;
  bool v4 = (C % 4 == 0) && (ld_dy % 4 == 0) && (ld_z % 4 == 0) && (ld_dz % 4 == 0) && (mask_mode != 1 || ld_y % 4 == 0) &&
            al16(dy) && al16(z) && al16(dz) && al16(y) && al16(scale) && al16(shift) && al16(k0) && al16(k1) && al16(k2) &&
            al16(dmasked) && (!dmasked || ld_dm % 4 == 0);
  if (v4)
    hipLaunchKernelGGL(dj_bn_bwd_apply_kernel<4>, dim3(ew_blocks(rows * (C / 4))), dim3(256), 0, s, dy, ld_dy, z, ld_z,
                       y, ld_y, scale, shift, mask_mode, k0, k1, k2, dz, ld_dz, rows, C, dmasked, ld_dm, dm_beta);
  else
    hipLaunchKernelGGL(dj_bn_bwd_apply_kernel<1>, dim3(ew_blocks(rows * C)), dim3(256), 0, s, dy, ld_dy, z, ld_z, y,
                       ld_y, scale, shift, mask_mode, k0, k1, k2, dz, ld_dz, rows, C, dmasked, ld_dm, dm_beta);
  DJ_CHECK_LAUNCH("dj_bn_bwd_apply");
  return DJ_OK;
}

// dx (+)= dy * [y > 0]
template <int VEC>
__global__ __launch_bounds__(256) void dj_relu_bwd_kernel(const float* dy, int ld_dy, const float* y, int ld_y,
                                                           float* dx, int ld_dx, long rows, int C, int beta) {
  using IO = VecIO<VEC>;
  const int cv = C / VEC;
  long total = rows * cv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i / cv;
    int c = (int)(i - r * cv) * VEC;
    f32x4 g = IO::ld(dy + r * ld_dy + c);
    f32x4 m = IO::ld(y + r * ld_y + c);
    g.x = (m.x > 0.f) ? g.x : 0.f;
    g.y = (m.y > 0.f) ? g.y : 0.f;
    g.z = (m.z > 0.f) ? g.z : 0.f;
    g.w = (m.w > 0.f) ? g.w : 0.f;
    float* d = dx + r * ld_dx + c;
    if (beta) g += IO::ld(d);
    IO::st(d, g);
  }
}

extern "C" int dj_relu_bwd(const float* dy, int ld_dy, const float* y, int ld_y, float* dx, int ld_dx, long rows,
                           int C, int beta, void* stream) {
  DJ_CHECK_ARG(dy && y && dx && rows > 0 && C > 0, "relu_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  bool v4 = (C % 4 == 0) && (ld_dy % 4 == 0) && (ld_y % 4 == 0) && (ld_dx % 4 == 0) && al16(dy) && al16(y) && al16(dx);
  if (v4)
    hipLaunchKernelGGL(dj_relu_bwd_kernel<4>, dim3(ew_blocks(rows * (C / 4))), dim3(256), 0, s, dy, ld_dy, y, ld_y, dx,
                       ld_dx, rows, C, beta);
  else
    hipLaunchKernelGGL(dj_relu_bwd_kernel<1>, dim3(ew_blocks(rows * C)), dim3(256), 0, s, dy, ld_dy, y, ld_y, dx, ld_dx,
                       rows, C, beta);
  DJ_CHECK_LAUNCH("dj_relu_bwd");
  return DJ_OK;
}

// ---------------------------------------------------------------------------------
// The same elementwise passes over tensors that carry their storage type (round 3, BASELINE config 5: activations held as
// fp16 and their gradients as bf16 in HBM inside the backbone, fp32 elsewhere): every tensor operand comes with a type
// code (DJ_F32 / DJ_F16 / DJ_BF16), looked at per access with a wave-uniform branch -- these passes are HBM-bound, the
// branch is free -- and all arithmetic stays fp32.  A 4-element piece of a 16-bit tensor is one 8-byte access.
// ---------------------------------------------------------------------------------
typedef unsigned int dj_u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 dj_h4 __attribute__((ext_vector_type(4)));

template <int VEC>
__device__ __forceinline__ f32x4 dj_ldt(const void* base, long idx, int dt) {
  if (dt == DJ_F32) return VecIO<VEC>::ld(reinterpret_cast<const float*>(base) + idx);
  if (VEC == 4) {
    const dj_u32x2 v = *reinterpret_cast<const dj_u32x2*>(reinterpret_cast<const unsigned short*>(base) + idx);
    if (dt == DJ_F16) {
      const dj_h4 h = __builtin_bit_cast(dj_h4, v);
      return f32x4{(float)h.x, (float)h.y, (float)h.z, (float)h.w};
    }
    return f32x4{__builtin_bit_cast(float, v.x << 16), __builtin_bit_cast(float, v.x & 0xFFFF0000u),
                 __builtin_bit_cast(float, v.y << 16), __builtin_bit_cast(float, v.y & 0xFFFF0000u)};
  }
  const unsigned short u = reinterpret_cast<const unsigned short*>(base)[idx];
  const float f = (dt == DJ_F16) ? (float)__builtin_bit_cast(_Float16, u) : __builtin_bit_cast(float, (unsigned)u << 16);
  return f32x4{f, 0.f, 0.f, 0.f};
}

__device__ __forceinline__ unsigned dj_pack2(float lo, float hi, int dt) {
  if (dt == DJ_F16) {
    const _Float16 a = (_Float16)lo, b = (_Float16)hi;
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
  }
  const __bf16 a = (__bf16)lo, b = (__bf16)hi;   // round to nearest even
  return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}

template <int VEC>
__device__ __forceinline__ void dj_stt(void* base, long idx, int dt, f32x4 v) {
  if (dt == DJ_F32) {
    VecIO<VEC>::st(reinterpret_cast<float*>(base) + idx, v);
  } else if (VEC == 4) {
    *reinterpret_cast<dj_u32x2*>(reinterpret_cast<unsigned short*>(base) + idx) = dj_u32x2{dj_pack2(v.x, v.y, dt), dj_pack2(v.z, v.w, dt)};
  } else {
    reinterpret_cast<unsigned short*>(base)[idx] = (unsigned short)(dj_pack2(v.x, 0.f, dt) & 0xFFFFu);
  }
}

// eight consecutive elements: ONE 16-byte access of a 16-bit tensor (two of a fp32 one) -- twice the bytes in flight per
// thread of the 4-element form, which is what these HBM-bound passes are short of when every tensor is 16 bits wide
typedef unsigned int dj_u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 dj_h8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void dj_ld8(const void* base, long idx, int dt, f32x4& lo, f32x4& hi) {
  if (dt == DJ_F32) {
    lo = VecIO<4>::ld(reinterpret_cast<const float*>(base) + idx);
    hi = VecIO<4>::ld(reinterpret_cast<const float*>(base) + idx + 4);
    return;
  }
  const dj_u32x4 v = *reinterpret_cast<const dj_u32x4*>(reinterpret_cast<const unsigned short*>(base) + idx);
  if (dt == DJ_F16) {
    const dj_h8 h = __builtin_bit_cast(dj_h8, v);
    lo = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    hi = f32x4{(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
  } else {
    lo = f32x4{__builtin_bit_cast(float, v.x << 16), __builtin_bit_cast(float, v.x & 0xFFFF0000u),
               __builtin_bit_cast(float, v.y << 16), __builtin_bit_cast(float, v.y & 0xFFFF0000u)};
    hi = f32x4{__builtin_bit_cast(float, v.z << 16), __builtin_bit_cast(float, v.z & 0xFFFF0000u),
               __builtin_bit_cast(float, v.w << 16), __builtin_bit_cast(float, v.w & 0xFFFF0000u)};
  }
}
__device__ __forceinline__ void dj_st8(void* base, long idx, int dt, f32x4 lo, f32x4 hi) {
  if (dt == DJ_F32) {
    VecIO<4>::st(reinterpret_cast<float*>(base) + idx, lo);
    VecIO<4>::st(reinterpret_cast<float*>(base) + idx + 4, hi);
  } else {
    *reinterpret_cast<dj_u32x4*>(reinterpret_cast<unsigned short*>(base) + idx) =
        dj_u32x4{dj_pack2(lo.x, lo.y, dt), dj_pack2(lo.z, lo.w, dt), dj_pack2(hi.x, hi.y, dt), dj_pack2(hi.z, hi.w, dt)};
  }
}
__device__ __forceinline__ f32x4 dj_mask4(f32x4 g, f32x4 m) {
  g.x = (m.x > 0.f) ? g.x : 0.f;
  g.y = (m.y > 0.f) ? g.y : 0.f;
  g.z = (m.z > 0.f) ? g.z : 0.f;
  g.w = (m.w > 0.f) ? g.w : 0.f;
  return g;
}

static inline bool dt_known(int dt) { return dt == DJ_F32 || dt == DJ_F16 || dt == DJ_BF16; }
// 4-element pieces legal: 16-byte aligned base (8 would do for 16-bit) and a pixel stride that is a multiple of 4 elements
static inline bool vec_ok(const void* p, long ld) { return al16(p) && ld % 4 == 0; }

template <int VEC>
__global__ __launch_bounds__(256) void dj_affine_act_t_kernel(const void* x, int dt_x, int ldx, const float* scale,
                                                               const float* shift, const void* res, int dt_res, int ldres,
                                                               const float* rscale, const float* rshift, void* y, int dt_y,
                                                               int ldy, long rows, int C, int relu) {
  using IO = VecIO<VEC>;
  const int cv = C / VEC;
  long total = rows * cv;
  const float floor_ = relu ? 0.f : -INFINITY;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i / cv;
    int c = (int)(i - r * cv) * VEC;
    f32x4 v = dj_ldt<VEC>(x, r * ldx + c, dt_x);
    if (scale) v = v * IO::ld(scale + c) + IO::ld(shift + c);
    if (res) {
      f32x4 t = dj_ldt<VEC>(res, r * ldres + c, dt_res);
      if (rscale) t = t * IO::ld(rscale + c) + IO::ld(rshift + c);
      v += t;
    }
    v.x = fmaxf(v.x, floor_);
    v.y = fmaxf(v.y, floor_);
    v.z = fmaxf(v.z, floor_);
    v.w = fmaxf(v.w, floor_);
    dj_stt<VEC>(y, r * ldy + c, dt_y, v);
  }
}

extern "C" int dj_affine_act_t(const void* x, int dt_x, int ldx, const float* scale, const float* shift, const void* res,
                               int dt_res, int ldres, const float* res_scale, const float* res_shift, void* y, int dt_y,
                               int ldy, long rows, int C, int relu, void* stream) {
  DJ_CHECK_ARG(x && y && rows > 0 && C > 0 && ldx >= C && ldy >= C, "affine_act: bad arguments");
  DJ_CHECK_ARG(dt_known(dt_x) && dt_known(dt_y) && (!res || dt_known(dt_res)), "affine_act: unknown storage type");
  DJ_CHECK_ARG((scale == nullptr) == (shift == nullptr) && (res_scale == nullptr) == (res_shift == nullptr),
               "affine_act: scale/shift come together");
  DJ_CHECK_ARG(res || !res_scale, "affine_act: res_scale without res");
  hipStream_t s = (hipStream_t)stream;
  bool v4 = (C % 4 == 0) && vec_ok(x, ldx) && vec_ok(y, ldy) && (!res || vec_ok(res, ldres)) && al16(scale) && al16(shift) &&
            al16(res_scale) && al16(res_shift);
  if (v4)
    hipLaunchKernelGGL(dj_affine_act_t_kernel<4>, dim3(ew_blocks(rows * (C / 4))), dim3(256), 0, s, x, dt_x, ldx, scale, shift,
                       res, dt_res, ldres, res_scale, res_shift, y, dt_y, ldy, rows, C, relu);
  else
    hipLaunchKernelGGL(dj_affine_act_t_kernel<1>, dim3(ew_blocks(rows * C)), dim3(256), 0, s, x, dt_x, ldx, scale, shift, res,
                       dt_res, ldres, res_scale, res_shift, y, dt_y, ldy, rows, C, relu);
  DJ_CHECK_LAUNCH("dj_affine_act_t");
  return DJ_OK;
}

// BnBwdF over typed tensors
struct BnBwdTF {
  const void* dy;
  int dt_dy, ld_dy;
  const void* z;
  int dt_z, ld_z;
  const void* y;
  int dt_y, ld_y;
  const float* mean;
  const float* invstd;
  const float* scale;
  const float* shift;
  int mask_mode;
  template <int VEC>
  __device__ __forceinline__ void eval(long r, int c, f32x4& v0, f32x4& v1) const {
    using IO = VecIO<VEC>;
    f32x4 g = dj_ldt<VEC>(dy, r * ld_dy + c, dt_dy);
    f32x4 zz = dj_ldt<VEC>(z, r * ld_z + c, dt_z);
    f32x4 m = {1.f, 1.f, 1.f, 1.f};
    if (mask_mode == 1)
      m = dj_ldt<VEC>(y, r * ld_y + c, dt_y);
    else if (mask_mode == 2)
      m = zz * IO::ld(scale + c) + IO::ld(shift + c);
    g.x = (m.x > 0.f) ? g.x : 0.f;
    g.y = (m.y > 0.f) ? g.y : 0.f;
    g.z = (m.z > 0.f) ? g.z : 0.f;
    g.w = (m.w > 0.f) ? g.w : 0.f;
    v0 = g;
    v1 = g * (zz - IO::ld(mean + c)) * IO::ld(invstd + c);
  }
};

extern "C" int dj_bn_bwd_reduce_t(const void* dy, int dt_dy, int ld_dy, const void* z, int dt_z, int ld_z, const void* y,
                                  int dt_y, int ld_y, const float* mean, const float* invstd, const float* scale,
                                  const float* shift, int mask_mode, long rows, int C, float* partial, void* stream) {
  DJ_CHECK_ARG(dy && z && mean && invstd && partial && rows > 0 && C > 0, "bn_bwd_reduce: bad arguments");
  DJ_CHECK_ARG(dt_known(dt_dy) && dt_known(dt_z) && (mask_mode != 1 || dt_known(dt_y)), "bn_bwd_reduce: unknown storage type");
  DJ_CHECK_ARG(mask_mode >= 0 && mask_mode <= 2, "bn_bwd_reduce: mask_mode");
  DJ_CHECK_ARG(mask_mode != 1 || y, "bn_bwd_reduce: mask_mode 1 needs y");
  DJ_CHECK_ARG(mask_mode != 2 || (scale && shift), "bn_bwd_reduce: mask_mode 2 needs scale/shift");
  BnBwdTF f{dy, dt_dy, ld_dy, z, dt_z, ld_z, y, dt_y, ld_y, mean, invstd, scale, shift, mask_mode};
  bool v4 = C % 4 == 0 && vec_ok(dy, ld_dy) && vec_ok(z, ld_z) && (mask_mode != 1 || vec_ok(y, ld_y)) && al16p(mean) &&
            al16p(invstd) && al16p(scale) && al16p(shift) && al16p(partial);
  return launch_colreduce(f, rows, C, v4, partial, (hipStream_t)stream, "dj_bn_bwd_reduce_t");
}

template <int VEC>
__global__ __launch_bounds__(256) void dj_bn_bwd_apply_t_kernel(const void* dy, int dt_dy, int ld_dy, const void* z, int dt_z,
                                                                 int ld_z, const void* y, int dt_y, int ld_y,
                                                                 const float* scale, const float* shift, int mask_mode,
                                                                 const float* k0, const float* k1, const float* k2, void* dz,
                                                                 int dt_dz, int ld_dz, long rows, int C, void* dmasked,
                                                                 int dt_dm, int ld_dm, int dm_beta) {
  using IO = VecIO<VEC>;
  const int cv = C / VEC;
  long total = rows * cv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i / cv;
    int c = (int)(i - r * cv) * VEC;
    f32x4 g = dj_ldt<VEC>(dy, r * ld_dy + c, dt_dy);
    f32x4 zz = dj_ldt<VEC>(z, r * ld_z + c, dt_z);
    f32x4 m = {1.f, 1.f, 1.f, 1.f};
    if (mask_mode == 1) {
      m = dj_ldt<VEC>(y, r * ld_y + c, dt_y);
    } else if (mask_mode == 2) {
      m = zz * IO::ld(scale + c) + IO::ld(shift + c);
    }
    g.x = (m.x > 0.f) ? g.x : 0.f;
    g.y = (m.y > 0.f) ? g.y : 0.f;
    g.z = (m.z > 0.f) ? g.z : 0.f;
    g.w = (m.w > 0.f) ? g.w : 0.f;
    dj_stt<VEC>(dz, r * ld_dz + c, dt_dz, IO::ld(k0 + c) * g + IO::ld(k1 + c) * zz + IO::ld(k2 + c));
    if (dmasked) {   // the masked upstream gradient is also the identity shortcut's gradient (Add + ReLU backward)
      const long o = r * ld_dm + c;
      dj_stt<VEC>(dmasked, o, dt_dm, dm_beta ? g + dj_ldt<VEC>(dmasked, o, dt_dm) : g);
    }
  }
}

// the 8-elements-per-thread form of the kernel above (C % 8 == 0, pixel strides multiples of 8)
__global__ __launch_bounds__(256) void dj_bn_bwd_apply_t8_kernel(const void* dy, int dt_dy, int ld_dy, const void* z, int dt_z,
                                                                  int ld_z, const void* y, int dt_y, int ld_y,
                                                                  const float* scale, const float* shift, int mask_mode,
                                                                  const float* k0, const float* k1, const float* k2, void* dz,
                                                                  int dt_dz, int ld_dz, long rows, int C, void* dmasked,
                                                                  int dt_dm, int ld_dm, int dm_beta) {
  using IO = VecIO<4>;
  const int cv = C / 8;
  long total = rows * cv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i / cv;
    int c = (int)(i - r * cv) * 8;
    f32x4 g0, g1, z0, z1, m0 = {1.f, 1.f, 1.f, 1.f}, m1 = {1.f, 1.f, 1.f, 1.f};
    dj_ld8(dy, r * ld_dy + c, dt_dy, g0, g1);
    dj_ld8(z, r * ld_z + c, dt_z, z0, z1);
    if (mask_mode == 1) {
      dj_ld8(y, r * ld_y + c, dt_y, m0, m1);
    } else if (mask_mode == 2) {
      m0 = z0 * IO::ld(scale + c) + IO::ld(shift + c);
      m1 = z1 * IO::ld(scale + c + 4) + IO::ld(shift + c + 4);
    }
    g0 = dj_mask4(g0, m0);
    g1 = dj_mask4(g1, m1);
    dj_st8(dz, r * ld_dz + c, dt_dz, IO::ld(k0 + c) * g0 + IO::ld(k1 + c) * z0 + IO::ld(k2 + c),
           IO::ld(k0 + c + 4) * g1 + IO::ld(k1 + c + 4) * z1 + IO::ld(k2 + c + 4));
    if (dmasked) {
      const long o = r * ld_dm + c;
      if (dm_beta) {
        f32x4 a0, a1;
        dj_ld8(dmasked, o, dt_dm, a0, a1);
        g0 += a0;
        g1 += a1;
      }
      dj_st8(dmasked, o, dt_dm, g0, g1);
    }
  }
}

extern "C" int dj_bn_bwd_apply_t(const void* dy, int dt_dy, int ld_dy, const void* z, int dt_z, int ld_z, const void* y,
                                 int dt_y, int ld_y, const float* scale, const float* shift, int mask_mode, const float* k0,
                                 const float* k1, const float* k2, void* dz, int dt_dz, int ld_dz, long rows, int C,
                                 void* dmasked, int dt_dm, int ld_dm, int dm_beta, void* stream) {
  DJ_CHECK_ARG(dy && z && k0 && k1 && k2 && dz && rows > 0 && C > 0, "bn_bwd_apply: bad arguments");
  DJ_CHECK_ARG(dt_known(dt_dy) && dt_known(dt_z) && dt_known(dt_dz) && (mask_mode != 1 || dt_known(dt_y)) &&
                   (!dmasked || dt_known(dt_dm)),
               "bn_bwd_apply: unknown storage type");
  DJ_CHECK_ARG(!dmasked || ld_dm >= C, "bn_bwd_apply: ld_dm < C");
  DJ_CHECK_ARG(mask_mode >= 0 && mask_mode <= 2, "bn_bwd_apply: mask_mode");
  DJ_CHECK_ARG(mask_mode != 1 || y, "bn_bwd_apply: mask_mode 1 needs y");
  DJ_CHECK_ARG(mask_mode != 2 || (scale && shift), "bn_bwd_apply: mask_mode 2 needs scale/shift");
  hipStream_t s = (hipStream_t)stream;
  bool v4 = (C % 4 == 0) && vec_ok(dy, ld_dy) && vec_ok(z, ld_z) && vec_ok(dz, ld_dz) && (mask_mode != 1 || vec_ok(y, ld_y)) &&
            al16(scale) && al16(shift) && al16(k0) && al16(k1) && al16(k2) && (!dmasked || vec_ok(dmasked, ld_dm));
  const bool v8 = v4 && (C % 8 == 0) && (ld_dy % 8 == 0) && (ld_z % 8 == 0) && (ld_dz % 8 == 0) &&
                  (mask_mode != 1 || ld_y % 8 == 0) && (!dmasked || ld_dm % 8 == 0);
  if (v8)
    hipLaunchKernelGGL(dj_bn_bwd_apply_t8_kernel, dim3(ew_blocks(rows * (C / 8))), dim3(256), 0, s, dy, dt_dy, ld_dy, z, dt_z,
                       ld_z, y, dt_y, ld_y, scale, shift, mask_mode, k0, k1, k2, dz, dt_dz, ld_dz, rows, C, dmasked, dt_dm,
                       ld_dm, dm_beta);
  else if (v4)
    hipLaunchKernelGGL(dj_bn_bwd_apply_t_kernel<4>, dim3(ew_blocks(rows * (C / 4))), dim3(256), 0, s, dy, dt_dy, ld_dy, z, dt_z,
                       ld_z, y, dt_y, ld_y, scale, shift, mask_mode, k0, k1, k2, dz, dt_dz, ld_dz, rows, C, dmasked, dt_dm,
                       ld_dm, dm_beta);
  else
    hipLaunchKernelGGL(dj_bn_bwd_apply_t_kernel<1>, dim3(ew_blocks(rows * C)), dim3(256), 0, s, dy, dt_dy, ld_dy, z, dt_z, ld_z,
                       y, dt_y, ld_y, scale, shift, mask_mode, k0, k1, k2, dz, dt_dz, ld_dz, rows, C, dmasked, dt_dm, ld_dm,
                       dm_beta);
  DJ_CHECK_LAUNCH("dj_bn_bwd_apply_t");
  return DJ_OK;
}

template <int VEC>
__global__ __launch_bounds__(256) void dj_relu_bwd_t_kernel(const void* dy, int dt_dy, int ld_dy, const void* y, int dt_y,
                                                             int ld_y, void* dx, int dt_dx, int ld_dx, long rows, int C,
                                                             int beta) {
  const int cv = C / VEC;
  long total = rows * cv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i / cv;
    int c = (int)(i - r * cv) * VEC;
    f32x4 g = dj_ldt<VEC>(dy, r * ld_dy + c, dt_dy);
    f32x4 m = dj_ldt<VEC>(y, r * ld_y + c, dt_y);
    g.x = (m.x > 0.f) ? g.x : 0.f;
    g.y = (m.y > 0.f) ? g.y : 0.f;
    g.z = (m.z > 0.f) ? g.z : 0.f;
    g.w = (m.w > 0.f) ? g.w : 0.f;
    const long o = r * ld_dx + c;
    if (beta) g += dj_ldt<VEC>(dx, o, dt_dx);
    dj_stt<VEC>(dx, o, dt_dx, g);
  }
}

extern "C" int dj_relu_bwd_t(const void* dy, int dt_dy, int ld_dy, const void* y, int dt_y, int ld_y, void* dx, int dt_dx,
                             int ld_dx, long rows, int C, int beta, void* stream) {
  DJ_CHECK_ARG(dy && y && dx && rows > 0 && C > 0, "relu_bwd: bad arguments");
  DJ_CHECK_ARG(dt_known(dt_dy) && dt_known(dt_y) && dt_known(dt_dx), "relu_bwd: unknown storage type");
  hipStream_t s = (hipStream_t)stream;
  bool v4 = (C % 4 == 0) && vec_ok(dy, ld_dy) && vec_ok(y, ld_y) && vec_ok(dx, ld_dx);
  if (v4)
    hipLaunchKernelGGL(dj_relu_bwd_t_kernel<4>, dim3(ew_blocks(rows * (C / 4))), dim3(256), 0, s, dy, dt_dy, ld_dy, y, dt_y,
                       ld_y, dx, dt_dx, ld_dx, rows, C, beta);
  else
    hipLaunchKernelGGL(dj_relu_bwd_t_kernel<1>, dim3(ew_blocks(rows * C)), dim3(256), 0, s, dy, dt_dy, ld_dy, y, dt_y, ld_y, dx,
                       dt_dx, ld_dx, rows, C, beta);
  DJ_CHECK_LAUNCH("dj_relu_bwd_t");
  return DJ_OK;
}

// dst[r][c] (+)= src[r][c] with a change of storage type on the way (a 16-bit backbone tensor handed to a layer that
// works on fp32, the gradient coming back)
template <int VEC>
__global__ __launch_bounds__(256) void dj_copy2d_t_kernel(const void* src, int dt_src, long lds, void* dst, int dt_dst,
                                                           long ldd, long rows, long cols, int beta) {
  const long cv = cols / VEC;
  long total = rows * cv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i / cv;
    long c = (i - r * cv) * VEC;
    f32x4 v = dj_ldt<VEC>(src, r * lds + c, dt_src);
    const long o = r * ldd + c;
    if (beta) v += dj_ldt<VEC>(dst, o, dt_dst);
    dj_stt<VEC>(dst, o, dt_dst, v);
  }
}

extern "C" int dj_copy2d_t(const void* src, int dt_src, long ld_src, void* dst, int dt_dst, long ld_dst, long rows, long cols,
                           int beta, void* stream) {
  DJ_CHECK_ARG(src && dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= cols, "copy2d: bad arguments");
  DJ_CHECK_ARG(dt_known(dt_src) && dt_known(dt_dst), "copy2d: unknown storage type");
  hipStream_t s = (hipStream_t)stream;
  bool v4 = (cols % 4 == 0) && vec_ok(src, ld_src) && vec_ok(dst, ld_dst);
  if (v4)
    hipLaunchKernelGGL(dj_copy2d_t_kernel<4>, dim3(ew_blocks(rows * (cols / 4))), dim3(256), 0, s, src, dt_src, ld_src, dst,
                       dt_dst, ld_dst, rows, cols, beta);
  else
    hipLaunchKernelGGL(dj_copy2d_t_kernel<1>, dim3(ew_blocks(rows * cols)), dim3(256), 0, s, src, dt_src, ld_src, dst, dt_dst,
                       ld_dst, rows, cols, beta);
  DJ_CHECK_LAUNCH("dj_copy2d_t");
  return DJ_OK;
}

// w16[i] = fp16(w[i]), wbf[i] = bf16(w[i]): the per-step 16-bit shadows of the fp32 master weights that the reduced-
// precision GEMMs read as their B operand (fp16 in the forward pass, bf16 in the input gradient).  One pass over the flat
// weight buffer at the head of the forward list: 4 bytes read, 4 written per parameter.
__global__ __launch_bounds__(256) void dj_shadow_weights_kernel(const float* w, void* w16, void* wbf, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 v = VecIO<4>::ld(w + 4 * i);
    if (w16) dj_stt<4>(w16, 4 * i, DJ_F16, v);
    if (wbf) dj_stt<4>(wbf, 4 * i, DJ_BF16, v);
  }
}

extern "C" int dj_shadow_weights(const float* w, void* w16, void* wbf, long n, void* stream) {
  DJ_CHECK_ARG(w && (w16 || wbf) && n > 0 && n % 4 == 0, "shadow_weights: bad arguments (n must be a multiple of 4)");
  DJ_CHECK_ARG(al16(w) && al16(w16) && al16(wbf), "shadow_weights: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(dj_shadow_weights_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, w, w16, wbf, n / 4);
  DJ_CHECK_LAUNCH("dj_shadow_weights");
  return DJ_OK;
}

// dst[r][c] (+)= src[r][c]   (Concatenate / its gradient / Reshape+Concatenate(axis=1))
template <int VEC>
__global__ __launch_bounds__(256) void dj_copy2d_kernel(const float* src, long lds, float* dst, long ldd, long rows,
                                                         long cols, int beta) {
  using IO = VecIO<VEC>;
  const long cv = cols / VEC;
  long total = rows * cv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i / cv;
    long c = (i - r * cv) * VEC;
    f32x4 v = IO::ld(src + r * lds + c);
    float* d = dst + r * ldd + c;
    if (beta) v += IO::ld(d);
    IO::st(d, v);
  }
}

extern "C" int dj_copy2d(const float* src, long ld_src, float* dst, long ld_dst, long rows, long cols, int beta,
                         void* stream) {
  DJ_CHECK_ARG(src && dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= cols, "copy2d: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  bool v4 = (cols % 4 == 0) && (ld_src % 4 == 0) && (ld_dst % 4 == 0) && al16(src) && al16(dst);
  if (v4)
    hipLaunchKernelGGL(dj_copy2d_kernel<4>, dim3(ew_blocks(rows * (cols / 4))), dim3(256), 0, s, src, ld_src, dst,
                       ld_dst, rows, cols, beta);
  else
    hipLaunchKernelGGL(dj_copy2d_kernel<1>, dim3(ew_blocks(rows * cols)), dim3(256), 0, s, src, ld_src, dst, ld_dst,
                       rows, cols, beta);
  DJ_CHECK_LAUNCH("dj_copy2d");
  return DJ_OK;
}

// Up to DJ_COPY_PARTS strided 2-D copies in ONE launch (blockIdx.y = part): Concatenate of several tensors and its
// gradient are a handful of tiny copies each, whose cost is launch latency, not bytes.
struct DjCopyParts {
  dj_copy_part part[DJ_COPY_PARTS];
};

__global__ __launch_bounds__(256) void dj_copy2d_multi_kernel(DjCopyParts ps) {
  const dj_copy_part p = ps.part[blockIdx.y];
  const bool v4 = (p.cols % 4 == 0) && (p.ld_src % 4 == 0) && (p.ld_dst % 4 == 0) && ((((uintptr_t)p.src) & 15) == 0) &&
                  ((((uintptr_t)p.dst) & 15) == 0);
  if (v4) {
    const long cv = p.cols / 4, total = p.rows * cv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
      long r = i / cv;
      long c = (i - r * cv) * 4;
      f32x4 v = *reinterpret_cast<const f32x4*>(p.src + r * p.ld_src + c);
      f32x4* d = reinterpret_cast<f32x4*>(p.dst + r * p.ld_dst + c);
      *d = p.beta ? v + *d : v;
    }
  } else {
    const long total = p.rows * p.cols;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
      long r = i / p.cols;
      long c = i - r * p.cols;
      float v = p.src[r * p.ld_src + c];
      float* d = p.dst + r * p.ld_dst + c;
      *d = p.beta ? v + *d : v;
    }
  }
}

extern "C" int dj_copy2d_multi(const dj_copy_part* parts, int n_parts, void* stream) {
  DJ_CHECK_ARG(parts && n_parts >= 1 && n_parts <= DJ_COPY_PARTS, "copy2d_multi: 1..%d parts", DJ_COPY_PARTS);
  DjCopyParts ps;
  long most = 0;
  for (int i = 0; i < n_parts; ++i) {
    const dj_copy_part& p = parts[i];
    DJ_CHECK_ARG(p.src && p.dst && p.rows > 0 && p.cols > 0 && p.ld_src >= p.cols && p.ld_dst >= p.cols,
                 "copy2d_multi: bad part %d", i);
    ps.part[i] = p;
    long n = p.rows * p.cols;
    if (n > most) most = n;
  }
  hipLaunchKernelGGL(dj_copy2d_multi_kernel, dim3(ew_blocks((most + 3) / 4), n_parts), dim3(256), 0, (hipStream_t)stream,
                     ps);
  DJ_CHECK_LAUNCH("dj_copy2d_multi");
  return DJ_OK;
}

// UpSampling2D() nearest x2: y[b, 2i+a, 2j+c, :] = x[b, i, j, :]
__global__ __launch_bounds__(256) void dj_upsample2x_kernel(const float* x, int ldx, float* y, int ldy, int B, int H,
                                                             int W, int C) {
  long total = (long)B * 2 * H * 2 * W * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long p = i / C;
    int ow = (int)(p % (2 * W));
    long q = p / (2 * W);
    int oh = (int)(q % (2 * H));
    int b = (int)(q / (2 * H));
    y[p * ldy + c] = x[((long)(b * H + oh / 2) * W + ow / 2) * ldx + c];
  }
}

extern "C" int dj_upsample2x(const float* x, int ldx, float* y, int ldy, int B, int H, int W, int C, void* stream) {
  DJ_CHECK_ARG(x && y && B > 0 && H > 0 && W > 0 && C > 0 && ldx >= C && ldy >= C, "upsample2x: bad arguments");
  hipLaunchKernelGGL(dj_upsample2x_kernel, dim3(ew_blocks((long)B * 4 * H * W * C)), dim3(256), 0, (hipStream_t)stream,
                     x, ldx, y, ldy, B, H, W, C);
  DJ_CHECK_LAUNCH("dj_upsample2x");
  return DJ_OK;
}

// ---------------------------------------------------------------------------------
// L2Normalization: y = x * rsqrt(max(sum_c x^2, 1e-12)) * gamma ; one wave per pixel
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dj_l2norm_fwd_kernel(const float* x, int ldx, const float* gamma, float* y,
                                                             int ldy, float* rnorm, long rows, int C) {
  long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + row * ldx;
  float ss = 0.f;
  for (int c = lane; c < C; c += 64) ss += xr[c] * xr[c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  float rn = rsqrtf(fmaxf(ss, 1e-12f));
  if (lane == 0 && rnorm) rnorm[row] = rn;
  float* yr = y + row * ldy;
  for (int c = lane; c < C; c += 64) yr[c] = xr[c] * rn * gamma[c];
}

extern "C" int dj_l2norm_fwd(const float* x, int ldx, const float* gamma, float* y, int ldy, float* rnorm, long rows,
                             int C, void* stream) {
  DJ_CHECK_ARG(x && gamma && y && rows > 0 && C > 0, "l2norm_fwd: bad arguments");
  hipLaunchKernelGGL(dj_l2norm_fwd_kernel, dim3((unsigned)dj_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                     gamma, y, ldy, rnorm, rows, C);
  DJ_CHECK_LAUNCH("dj_l2norm_fwd");
  return DJ_OK;
}

// dx (+)= rn * (g - xhat * sum_c(g*xhat)), g = dy*gamma, xhat = x*rn  (clamped rows: dx = g*rn)
__global__ __launch_bounds__(256) void dj_l2norm_bwd_kernel(const float* dy, int ld_dy, const float* x, int ldx,
                                                             const float* gamma, const float* rnorm, float* dx,
                                                             int ld_dx, long rows, int C, int beta) {
  long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + row * ldx;
  const float* gr = dy + row * ld_dy;
  float rn = rnorm[row];
  float dot = 0.f;
  for (int c = lane; c < C; c += 64) dot += gr[c] * gamma[c] * xr[c] * rn;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
  bool clamped = rn >= 1e6f;  // sum(x^2) <= 1e-12: the norm is the constant 1e-6
  float* dr = dx + row * ld_dx;
  for (int c = lane; c < C; c += 64) {
    float g = gr[c] * gamma[c];
    float v = clamped ? g * rn : rn * (g - xr[c] * rn * dot);
    dr[c] = beta ? dr[c] + v : v;
  }
}

extern "C" int dj_l2norm_bwd(const float* dy, int ld_dy, const float* x, int ldx, const float* gamma,
                             const float* rnorm, float* dx, int ld_dx, float* dgamma_partial, long rows, int C,
                             int beta, void* stream) {
  DJ_CHECK_ARG(dy && x && gamma && rnorm && rows > 0 && C > 0, "l2norm_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (dx) {
    hipLaunchKernelGGL(dj_l2norm_bwd_kernel, dim3((unsigned)dj_cdiv(rows, 4)), dim3(256), 0, s, dy, ld_dy, x, ldx, gamma,
                       rnorm, dx, ld_dx, rows, C, beta);
    DJ_CHECK_LAUNCH("dj_l2norm_bwd");
  }
  if (dgamma_partial) {
    L2DgF f{dy, x, rnorm, ld_dy, ldx};
    bool v4 = C % 4 == 0 && ld_dy % 4 == 0 && ldx % 4 == 0 && al16p(dy) && al16p(x) && al16p(dgamma_partial);
    return launch_colreduce(f, rows, C, v4, dgamma_partial, s, "dj_l2norm_bwd(dgamma)");
  }
  return DJ_OK;
}

// ---------------------------------------------------------------------------------
// MaxPooling2D: k x k window, stride s, explicit leading pads.  pad_zero = 0: padding never wins (TF 'same'
// / 'valid'); pad_zero = 1: out-of-range taps are zeros that take part in the max (a preceding ZeroPadding2D).
// `pool5_ssd` is (3,3)/1/'same' (keras_ssd300_dct_j2d_resnet.py:481); the ResNet50RGB stem is ZeroPadding2D(1)
// + (3,3)/2 (resnet_dct.py:165-314).
// ---------------------------------------------------------------------------------
struct PoolGeom {
  int B, H, W, C, OH, OW, kh, kw, sh, sw, pt, pl, pad_zero;
};

__device__ __forceinline__ float pool_window(const float* x, const PoolGeom& g, int b, int oh, int ow, int c, int* ah,
                                             int* aw) {
  float m = -INFINITY;
  int bh = -2, bw = -2;  // -2: nothing yet, -1: a zero pad tap won
  for (int i = 0; i < g.kh; ++i)
    for (int j = 0; j < g.kw; ++j) {
      int hh = oh * g.sh + i - g.pt, ww = ow * g.sw + j - g.pl;
      bool in = (unsigned)hh < (unsigned)g.H && (unsigned)ww < (unsigned)g.W;
      if (!in && !g.pad_zero) continue;
      float v = in ? x[((long)(b * g.H + hh) * g.W + ww) * g.C + c] : 0.f;
      if (v > m) {  // first maximum in row-major window order
        m = v;
        bh = in ? hh : -1;
        bw = in ? ww : -1;
      }
    }
  *ah = bh;
  *aw = bw;
  return m;
}

__global__ __launch_bounds__(256) void dj_maxpool2d_fwd_kernel(const float* x, float* y, unsigned char* argmax,
                                                                PoolGeom g) {
  long total = (long)g.B * g.OH * g.OW * g.C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % g.C);
    long p = i / g.C;
    int ow = (int)(p % g.OW);
    long q = p / g.OW;
    int oh = (int)(q % g.OH);
    int b = (int)(q / g.OH);
    int ah, aw;
    y[i] = pool_window(x, g, b, oh, ow, c, &ah, &aw);
    // window-relative tap of the winner (255: a zero pad tap), so that backward need not rescan the windows
    if (argmax) argmax[i] = ah < 0 ? 255 : (unsigned char)((ah - (oh * g.sh - g.pt)) * g.kw + (aw - (ow * g.sw - g.pl)));
  }
}

// gather form: each input element sums the gradients of the windows whose (first) maximum it is
template <bool SAVED>
__global__ __launch_bounds__(256) void dj_maxpool2d_bwd_kernel(const float* x, const unsigned char* argmax, const float* dy,
                                                                float* dx, PoolGeom g, int beta) {
  long total = (long)g.B * g.H * g.W * g.C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % g.C);
    long p = i / g.C;
    int w = (int)(p % g.W);
    long q = p / g.W;
    int h = (int)(q % g.H);
    int b = (int)(q / g.H);
    float acc = 0.f;
    // windows (oh, ow) with oh*sh - pt <= h < oh*sh - pt + kh
    int oh_lo = (h + g.pt - g.kh + g.sh) / g.sh;  // ceil((h + pt - kh + 1) / sh)
    if (h + g.pt - g.kh + 1 <= 0) oh_lo = 0;
    int oh_hi = (h + g.pt) / g.sh;
    int ow_lo = (w + g.pl - g.kw + g.sw) / g.sw;
    if (w + g.pl - g.kw + 1 <= 0) ow_lo = 0;
    int ow_hi = (w + g.pl) / g.sw;
    if (oh_hi >= g.OH) oh_hi = g.OH - 1;
    if (ow_hi >= g.OW) ow_hi = g.OW - 1;
    for (int oh = oh_lo; oh <= oh_hi; ++oh)
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        long o = ((long)(b * g.OH + oh) * g.OW + ow) * g.C + c;
        bool mine;
        if (SAVED) {
          mine = (int)argmax[o] == (h - (oh * g.sh - g.pt)) * g.kw + (w - (ow * g.sw - g.pl));
        } else {
          int ah, aw;
          pool_window(x, g, b, oh, ow, c, &ah, &aw);
          mine = ah == h && aw == w;
        }
        if (mine) acc += dy[o];
      }
    dx[i] = beta ? dx[i] + acc : acc;
  }
}

// Four channels per thread (C % 4 == 0, 16-byte aligned tensors): one float4 per tap, the four tap codes as one 32-bit
// store, and the index arithmetic (runtime divisions) paid once per four elements.  Same comparisons in the same window
// order as the scalar kernels, channel by channel.
__global__ __launch_bounds__(256) void dj_maxpool2d_fwd4_kernel(const float* x, float* y, unsigned char* argmax,
                                                                 PoolGeom g) {
  const int cv = g.C / 4;
  const long total = (long)g.B * g.OH * g.OW * cv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * 4;
    const int p = (int)(i / cv);
    const int ow = p % g.OW;
    const int q = p / g.OW;
    const int oh = q % g.OH;
    const int b = q / g.OH;
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    unsigned code[4] = {255u, 255u, 255u, 255u};
    for (int ki = 0; ki < g.kh; ++ki)
      for (int kj = 0; kj < g.kw; ++kj) {
        const int hh = oh * g.sh + ki - g.pt, ww = ow * g.sw + kj - g.pl;
        const bool in = (unsigned)hh < (unsigned)g.H && (unsigned)ww < (unsigned)g.W;
        if (!in && !g.pad_zero) continue;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (in) v = VecIO<4>::ld(x + ((long)(b * g.H + hh) * g.W + ww) * g.C + c);
        const float vv[4] = {v.x, v.y, v.z, v.w};
        const unsigned tap = in ? (unsigned)(ki * g.kw + kj) : 255u;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (vv[j] > m[j]) {
            m[j] = vv[j];
            code[j] = tap;
          }
      }
    const long o = (long)p * g.C + c;
    VecIO<4>::st(y + o, f32x4{m[0], m[1], m[2], m[3]});
    if (argmax)
      *reinterpret_cast<unsigned*>(argmax + o) = code[0] | (code[1] << 8) | (code[2] << 16) | (code[3] << 24);
  }
}

__global__ __launch_bounds__(256) void dj_maxpool2d_bwd4_kernel(const unsigned char* argmax, const float* dy, float* dx,
                                                                 PoolGeom g, int beta) {
  const int cv = g.C / 4;
  const long total = (long)g.B * g.H * g.W * cv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * 4;
    const int p = (int)(i / cv);
    const int w = p % g.W;
    const int q = p / g.W;
    const int h = q % g.H;
    const int b = q / g.H;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int oh_lo = (h + g.pt - g.kh + g.sh) / g.sh;
    if (h + g.pt - g.kh + 1 <= 0) oh_lo = 0;
    int oh_hi = (h + g.pt) / g.sh;
    int ow_lo = (w + g.pl - g.kw + g.sw) / g.sw;
    if (w + g.pl - g.kw + 1 <= 0) ow_lo = 0;
    int ow_hi = (w + g.pl) / g.sw;
    if (oh_hi >= g.OH) oh_hi = g.OH - 1;
    if (ow_hi >= g.OW) ow_hi = g.OW - 1;
    for (int oh = oh_lo; oh <= oh_hi; ++oh)
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        const long o = ((long)(b * g.OH + oh) * g.OW + ow) * g.C + c;
        const unsigned codes = *reinterpret_cast<const unsigned*>(argmax + o);
        const unsigned mine = (unsigned)((h - (oh * g.sh - g.pt)) * g.kw + (w - (ow * g.sw - g.pl)));
        if (((codes & 255u) != mine) && (((codes >> 8) & 255u) != mine) && (((codes >> 16) & 255u) != mine) &&
            ((codes >> 24) != mine))
          continue;
        const f32x4 d = VecIO<4>::ld(dy + o);
        if ((codes & 255u) == mine) acc[0] += d.x;
        if (((codes >> 8) & 255u) == mine) acc[1] += d.y;
        if (((codes >> 16) & 255u) == mine) acc[2] += d.z;
        if ((codes >> 24) == mine) acc[3] += d.w;
      }
    float* dst = dx + (long)p * g.C + c;
    f32x4 r = {acc[0], acc[1], acc[2], acc[3]};
    if (beta) r = r + VecIO<4>::ld(dst);
    VecIO<4>::st(dst, r);
  }
}

static int pool_geom(PoolGeom* g, int B, int H, int W, int C, int OH, int OW, int kh, int kw, int sh, int sw, int pt,
                     int pl, int pad_zero) {
  DJ_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && OH > 0 && OW > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0 && pt >= 0 &&
                   pl >= 0,
               "maxpool: bad geometry");
  DJ_CHECK_ARG((long)(OH - 1) * sh - pt < H && (long)(OW - 1) * sw - pl < W, "maxpool: output grid too large");
  *g = PoolGeom{B, H, W, C, OH, OW, kh, kw, sh, sw, pt, pl, pad_zero};
  return DJ_OK;
}

extern "C" int dj_maxpool2d_fwd(const float* x, float* y, int B, int H, int W, int C, int OH, int OW, int kh, int kw,
                                int sh, int sw, int pt, int pl, int pad_zero, unsigned char* argmax, void* stream) {
  DJ_CHECK_ARG(x && y, "maxpool fwd: null tensor");
  DJ_CHECK_ARG(!argmax || kh * kw < 255, "maxpool fwd: window too large for the saved arg-max encoding");
  PoolGeom g;
  if (int rc = pool_geom(&g, B, H, W, C, OH, OW, kh, kw, sh, sw, pt, pl, pad_zero)) return rc;
  const bool small = (long)B * H * W < (1L << 31) / 4 && (long)B * OH * OW < (1L << 31) / 4;   // 32-bit pixel indices
  if (C % 4 == 0 && small && al16(x) && al16(y) && (((uintptr_t)argmax) & 3) == 0)
    hipLaunchKernelGGL(dj_maxpool2d_fwd4_kernel, dim3(ew_blocks((long)B * OH * OW * (C / 4))), dim3(256), 0,
                       (hipStream_t)stream, x, y, argmax, g);
  else
    hipLaunchKernelGGL(dj_maxpool2d_fwd_kernel, dim3(ew_blocks((long)B * OH * OW * C)), dim3(256), 0, (hipStream_t)stream,
                       x, y, argmax, g);
  DJ_CHECK_LAUNCH("dj_maxpool2d_fwd");
  return DJ_OK;
}

extern "C" int dj_maxpool2d_bwd(const float* x, const float* dy, float* dx, int B, int H, int W, int C, int OH, int OW,
                                int kh, int kw, int sh, int sw, int pt, int pl, int pad_zero, int beta,
                                const unsigned char* argmax, void* stream) {
  DJ_CHECK_ARG((x || argmax) && dy && dx, "maxpool bwd: null tensor");
  PoolGeom g;
  if (int rc = pool_geom(&g, B, H, W, C, OH, OW, kh, kw, sh, sw, pt, pl, pad_zero)) return rc;
  const bool small = (long)B * H * W < (1L << 31) / 4 && (long)B * OH * OW < (1L << 31) / 4;
  if (argmax && C % 4 == 0 && small && al16(dy) && al16(dx) && (((uintptr_t)argmax) & 3) == 0)
    hipLaunchKernelGGL(dj_maxpool2d_bwd4_kernel, dim3(ew_blocks((long)B * H * W * (C / 4))), dim3(256), 0,
                       (hipStream_t)stream, argmax, dy, dx, g, beta);
  else if (argmax)
    hipLaunchKernelGGL(dj_maxpool2d_bwd_kernel<true>, dim3(ew_blocks((long)B * H * W * C)), dim3(256), 0,
                       (hipStream_t)stream, x, argmax, dy, dx, g, beta);
  else
    hipLaunchKernelGGL(dj_maxpool2d_bwd_kernel<false>, dim3(ew_blocks((long)B * H * W * C)), dim3(256), 0,
                       (hipStream_t)stream, x, argmax, dy, dx, g, beta);
  DJ_CHECK_LAUNCH("dj_maxpool2d_bwd");
  return DJ_OK;
}
