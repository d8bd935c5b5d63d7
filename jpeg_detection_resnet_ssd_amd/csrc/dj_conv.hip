// Host launchers of the implicit-GEMM convolution (see dj_igemm.h for the kernel).
#include "../../include/dj_hip.h"
#include "dj_conv_launch.h"
#include <string.h>
#include <map>
#include <mutex>
#include <atomic>
#include <array>
#include <algorithm>

static thread_local char g_err[512] = "";
void dj_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* dj_last_error(void) { return g_err; }
extern "C" int dj_abi_version(void) { return DJ_ABI_VERSION; }

// ---- the library's only mutable state: three settings, each safe to touch from any thread ----
// test switch: atomic, read once per launch
std::atomic<bool> g_dj_allow_fast{true};
extern "C" void dj_set_fast_path(int enable) { g_dj_allow_fast.store(enable != 0, std::memory_order_relaxed); }

// Arithmetic mode.  0: fp32 MFMA everywhere (default).  1: forward GEMMs round their operands to fp16, gradient GEMMs
// (dgrad, wgrad) to bf16 (gradients need the exponent range); 2: bf16 everywhere; 3 ("float32x3"): fp32 tensors, every
// product as three bf16 MFMAs on hi / lo split operands (~2^-17 relative per product); 4 ("float32x6"): six MFMAs on
// hi / mid / lo pieces (2^-24: fp32 results).  Mode 0 takes, per geometry, the fp32 MFMA kernel or the mode-4 kernel its
// tuning entry names (both fp32 results); 5: fp32 MFMA kernels only.  fp32 accumulation in all modes.
// A process-wide default (atomic) that a thread can override for its own launches: a plan lowered under one
// mode keeps running in it whatever other models or threads of the process select (engine.Plan sets the override before
// it launches).
static std::atomic<int> g_default_compute_mode{0};
static thread_local int tl_compute_mode = -1;
int dj_compute_mode() { return tl_compute_mode >= 0 ? tl_compute_mode : g_default_compute_mode.load(std::memory_order_relaxed); }
extern "C" int dj_set_compute_mode(int mode) {
  int prev = g_default_compute_mode.load(std::memory_order_relaxed);
  if (mode >= 0 && mode <= 5) g_default_compute_mode.store(mode, std::memory_order_relaxed);
  return prev;
}
extern "C" int dj_set_thread_compute_mode(int mode) {
  int prev = tl_compute_mode;
  if (mode >= -1 && mode <= 5) tl_compute_mode = mode;
  return prev;
}
extern "C" int dj_get_compute_mode(void) { return dj_compute_mode(); }

// instantiated in dj_conv_i00.hip (fwd), dj_conv_i01.hip (strided 1x1 dgrad as GEMM), dj_conv_i11.hip (dgrad),
// dj_conv_i20.hip (wgrad)
extern template int dj_launch_cfg<0, 0>(int, const DjIgemmParams&, int, hipStream_t);
extern template int dj_launch_cfg<0, 1>(int, const DjIgemmParams&, int, hipStream_t);
extern template int dj_launch_cfg<1, 1>(int, const DjIgemmParams&, int, hipStream_t);
extern template int dj_launch_cfg<2, 0>(int, const DjIgemmParams&, int, hipStream_t);

// Pick a tile shape: smallest padded-N waste first, then enough workgroups to fill
// 256 CUs (2 resident workgroups each for the large tiles).
static int choose_cfg(long M, long N, long K, bool allow_split, int* splits_out) {
  int bn;
  if (N > 96) {
    // 128 vs 64: both pad N to the same multiple of 64 or better with 64
    long pad128 = (long)dj_cdiv(N, 128) * 128, pad64 = (long)dj_cdiv(N, 64) * 64;
    bn = (pad64 < pad128) ? 64 : 128;
  } else if (N > 32 && (long)dj_cdiv(N, 64) * 64 <= (long)dj_cdiv(N, 32) * 32) {
    bn = 64;
  } else {
    bn = 32;
  }
  int cfg;
  auto tiles = [&](int bm, int b_n) { return (long)dj_cdiv(M, bm) * dj_cdiv(N, b_n); };
  if (bn == 128) {
    if (tiles(128, 128) >= 384)
      cfg = CFG_128x128;
    else if (tiles(128, 64) >= 256)
      cfg = CFG_128x64;
    else
      cfg = CFG_64x64;
  } else if (bn == 64) {
    cfg = (tiles(128, 64) >= 384) ? CFG_128x64 : CFG_64x64;
  } else {
    cfg = CFG_128x32;
  }
  int splits = 1;
  if (allow_split) {
    long t = tiles(kCfgs[cfg].bm, kCfgs[cfg].bn);
    if (t < 256) {
      long want = (512 + t - 1) / t;
      long maxs = K / 256;  // keep >= 8 K-steps per split
      if (maxs < 1) maxs = 1;
      splits = (int)(want < maxs ? want : maxs);
      if (splits < 1) splits = 1;
    }
  }
  *splits_out = splits;
  return cfg;
}


// ---------------------------------------------------------------------------------
// Per-geometry launch overrides (tile configuration, split-K factor) recorded by the plan-time
// autotuner (engine.Plan.autotune): key = direction (+4 when BN statistics are taken) + geometry.
// ---------------------------------------------------------------------------------
typedef std::array<int, 16> TuneKey;
static std::map<TuneKey, std::pair<int, int>> g_tune;
static std::mutex g_tune_mu;

// (the arithmetic mode of the calling thread is part of the key: the fp32 and the reduced-precision kernels have tables
// of their own, and two models of different modes in one process do not overwrite each other's choices)
static TuneKey tune_key(int dir, const dj_conv2d_desc* d) {
  return TuneKey{dir | (dj_compute_mode() << 4), d->batch, d->in_h, d->in_w, d->in_c, d->out_h, d->out_w, d->out_c, d->kernel_h, d->kernel_w,
                 d->stride_h, d->stride_w, d->dilation_h, d->dilation_w, d->pad_top, d->pad_left};
}

static bool tune_lookup(int dir, const dj_conv2d_desc* d, int* cfg, int* splits) {
  std::lock_guard<std::mutex> g(g_tune_mu);
  auto it = g_tune.find(tune_key(dir, d));
  if (it == g_tune.end()) return false;
  *cfg = it->second.first;
  *splits = it->second.second;
  return true;
}

// mode 0: [0, N_CFG) fp32 MFMA variants, [N_CFG, 2 N_CFG) the split-bf16 (float32x6) variants of the same indices
extern "C" int dj_conv2d_tune_configs(void) { return dj_compute_mode() == 0 ? 2 * N_CFG : N_CFG; }

// dir: 0 fwd, 1 dgrad, 2 wgrad, +4 when the forward takes BN statistics, 9 = input gradient that takes the
// BatchNormalization backward statistics (dj_conv2d_nhwc_dgrad_bnbwd).  cfg < 0 removes the override.
extern "C" int dj_conv2d_tune_set(int dir, const dj_conv2d_desc* d, int cfg, int splits) {
  DJ_CHECK_ARG(d != nullptr && cfg < (dj_compute_mode() == 0 ? 2 * N_CFG : N_CFG) && splits >= 1, "tune_set: bad arguments");
  std::lock_guard<std::mutex> g(g_tune_mu);
  if (cfg < 0)
    g_tune.erase(tune_key(dir, d));
  else
    g_tune[tune_key(dir, d)] = std::make_pair(cfg, splits);
  return DJ_OK;
}

// Default choice the launcher would make (for the tuner to seed its candidate list): writes cfg and splits.
extern "C" int dj_conv2d_default_config(int dir, const dj_conv2d_desc* d, int* cfg, int* splits);

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
// a thread's 4-element piece: 16 bytes of a fp32 tensor, 8 bytes of a 16-bit one
static inline bool aligned_dt(const void* p, int dt) { return (((uintptr_t)p) & (dt == 0 ? 15 : 7)) == 0; }

static int check_desc(const dj_conv2d_desc* d) {
  DJ_CHECK_ARG(d != nullptr, "conv desc is null");
  DJ_CHECK_ARG(d->batch > 0 && d->in_h > 0 && d->in_w > 0 && d->in_c > 0 && d->out_h > 0 && d->out_w > 0 &&
                   d->out_c > 0,
               "conv desc: non-positive dimension");
  DJ_CHECK_ARG(d->kernel_h > 0 && d->kernel_w > 0 && d->stride_h > 0 && d->stride_w > 0 && d->dilation_h > 0 &&
                   d->dilation_w > 0,
               "conv desc: non-positive kernel/stride/dilation");
  DJ_CHECK_ARG(d->pad_top >= 0 && d->pad_left >= 0, "conv desc: negative padding");
  DJ_CHECK_ARG(d->ld_x >= d->in_c && d->ld_y >= d->out_c, "conv desc: ld_x/ld_y smaller than channel count");
  // every output pixel must read inside [-pad, in + pad_after): guaranteed if the last tap start is valid
  long last_h = (long)(d->out_h - 1) * d->stride_h - d->pad_top;
  long last_w = (long)(d->out_w - 1) * d->stride_w - d->pad_left;
  DJ_CHECK_ARG(last_h < d->in_h && last_w < d->in_w, "conv desc: output grid larger than the input allows");
  DJ_CHECK_ARG((long)d->batch * d->in_h * d->in_w * (long)d->ld_x < (1L << 31) &&
                   (long)d->batch * d->out_h * d->out_w * (long)d->ld_y < (1L << 31),
               "conv desc: tensor exceeds 2^31 elements");
  return DJ_OK;
}

__global__ void dj_relu_rows_kernel(float* y, long rows, int cols, int ld) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = rows * cols;
  for (; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i / cols;
    int c = (int)(i - r * cols);
    float* p = y + r * ld + c;
    *p = fmaxf(*p, 0.f);
  }
}

static int extent_bytes(long pixels, long ld, long c, int es = 4) {
  long b = ((pixels - 1) * ld + c) * es;
  return (b > 0 && b < 0x7FFFFFF0L) ? (int)b : 0;
}

// storage types of the tensors of a launch (DJ_F32 / DJ_F16 / DJ_BF16); the float entry points pass all zeros
static inline int dt_size(int dt) { return dt == DJ_F32 ? 4 : 2; }
static inline bool dt_ok(int dt) { return dt == DJ_F32 || dt == DJ_F16 || dt == DJ_BF16; }

// reciprocals of the row grid (p.rowH, p.rowW must be set): the kernels decompose a GEMM row into (image, h, w) with a
// float multiply and a one-step correction instead of integer divisions
static void set_row_recip(DjIgemmParams& p) {
  p.inv_rowHW = 1.0f / (float)(p.rowH * p.rowW);
  p.inv_rowW = 1.0f / (float)p.rowW;
}

static void fill_geom(DjIgemmParams& p, const dj_conv2d_desc* d) {
  p.KH = d->kernel_h;
  p.KW = d->kernel_w;
  p.sH = d->stride_h;
  p.sW = d->stride_w;
  p.dH = d->dilation_h;
  p.dW = d->dilation_w;
  p.pT = d->pad_top;
  p.pL = d->pad_left;
  set_row_recip(p);
}

extern "C" int dj_conv2d_fwd_stats_rows(const dj_conv2d_desc* d) {
  if (check_desc(d) != DJ_OK) return DJ_ERR_ARG;
  // one partial row per 64 output pixels, independent of the tile configuration the launcher picks
  return dj_cdiv((long)d->batch * d->out_h * d->out_w, 64);
}

// Second half of a split-K forward convolution without atomics: the K chunks left their partial tiles in `nsplit` slabs
// [rows][N]; this pass adds them IN A FIXED ORDER (bit-reproducible, unlike fp32 atomics in arrival order), takes the
// BatchNormalization column statistics of the sum (per 64 rows: [ceil(rows/64)][2][N], before the bias, as the GEMM
// epilogue does), then adds bias / applies ReLU and writes y.  One workgroup per 64 rows x (16 * VEC) columns.
template <int VEC>
__global__ __launch_bounds__(256) void dj_splitk_reduce_kernel(const float* slabs, int nsplit, long slab_stride, long rows,
                                                               int N, const float* bias, int relu, float* y, int ld_y,
                                                               float* stats) {
  // 16 column chunks (16 * VEC columns) x 16 row lanes; a thread owns rows ty, ty + 16, ty + 32, ty + 48 of the group
  __shared__ float red[16][16][2 * VEC];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = (blockIdx.y * 16 + tx) * VEC;
  const long r0 = (long)blockIdx.x * 64;
  float s[VEC], q[VEC], b[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    s[v] = 0.f;
    q[v] = 0.f;
    b[v] = (bias && c + v < N) ? bias[c + v] : 0.f;
  }
  if (c < N) {
    float a[4][VEC];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int v = 0; v < VEC; ++v) a[j][v] = 0.f;
    for (int k = 0; k < nsplit; ++k) {      // slabs in index order: the sum does not depend on who finished first
      const float* slab = slabs + (long)k * slab_stride + c;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long r = r0 + ty + 16 * j;
        if (r < rows) {
          if (VEC == 4) {
            f32x4 t = *reinterpret_cast<const f32x4*>(slab + r * N);
            a[j][0] += t.x;
            a[j][1 % VEC] += t.y;
            a[j][2 % VEC] += t.z;
            a[j][3 % VEC] += t.w;
          } else {
            a[j][0] += slab[r * N];
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long r = r0 + ty + 16 * j;
      if (r >= rows) continue;
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        s[v] += a[j][v];
        q[v] += a[j][v] * a[j][v];
        float o = a[j][v] + b[v];
        a[j][v] = relu ? fmaxf(o, 0.f) : o;
      }
      float* dst = y + r * ld_y + c;
      if (VEC == 4) *reinterpret_cast<f32x4*>(dst) = f32x4{a[j][0], a[j][1 % VEC], a[j][2 % VEC], a[j][3 % VEC]};
      else dst[0] = a[j][0];
    }
  }
  if (!stats) return;
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    red[ty][tx][v] = s[v];
    red[ty][tx][VEC + v] = q[v];
  }
  __syncthreads();
  if (ty == 0 && c < N) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float ss = 0.f, qq = 0.f;
      for (int l = 0; l < 16; ++l) {     // fixed order over the row lanes
        ss += red[l][tx][v];
        qq += red[l][tx][VEC + v];
      }
      stats[((long)blockIdx.x * 2 + 0) * N + c + v] = ss;
      stats[((long)blockIdx.x * 2 + 1) * N + c + v] = qq;
    }
  }
}

static int launch_splitk_reduce(const float* slabs, int nsplit, long slab_stride, long rows, int N, const float* bias,
                                int relu, float* y, int ld_y, float* stats, hipStream_t s) {
  const bool v4 = (N % 4 == 0) && (ld_y % 4 == 0) && aligned16(slabs) && aligned16(y) && (slab_stride % 4 == 0);
  const int cols_per_block = v4 ? 64 : 16;
  dim3 grid((unsigned)((rows + 63) / 64), (unsigned)((N + cols_per_block - 1) / cols_per_block));
  if (v4)
    hipLaunchKernelGGL(dj_splitk_reduce_kernel<4>, grid, dim3(256), 0, s, slabs, nsplit, slab_stride, rows, N, bias, relu, y,
                       ld_y, stats);
  else
    hipLaunchKernelGGL(dj_splitk_reduce_kernel<1>, grid, dim3(256), 0, s, slabs, nsplit, slab_stride, rows, N, bias, relu, y,
                       ld_y, stats);
  DJ_CHECK_LAUNCH("dj_splitk_reduce_kernel");
  return DJ_OK;
}

struct FwdResidual {
  const dj_bn_train* bn = nullptr;   // finish the following BatchNormalization inside the launch
  const float* res = nullptr;
  int ld_res = 0;
  const float* res_scale = nullptr;
  const float* res_shift = nullptr;
  float* sum_out = nullptr;
  int ld_sum = 0;
};

struct FwdTypes {   // how x (and the residual operand), the weights, y and the stored residual sum are held in HBM
  int x = DJ_F32, w = DJ_F32, y = DJ_F32, sum = DJ_F32;
};

static int conv_fwd_impl(const dj_conv2d_desc* d, const float* x, const float* w, const float* bias, float* y,
                         const float* pro_scale, const float* pro_shift, int pro_relu, int relu, float* stats,
                         const FwdResidual& rz, void* stream, float* ws = nullptr, long ws_floats = 0,
                         const FwdTypes& ty = FwdTypes()) {
  if (int rc = check_desc(d)) return rc;
  DJ_CHECK_ARG(x && w && y, "conv fwd: null tensor");
  const bool io16 = ty.x != DJ_F32 || ty.y != DJ_F32 || ty.sum != DJ_F32 || ty.w != DJ_F32;
  DJ_CHECK_ARG((ty.x == DJ_F32 || ty.x == DJ_F16) && (ty.y == DJ_F32 || ty.y == DJ_F16) && (ty.sum == DJ_F32 || ty.sum == DJ_F16) &&
                   (ty.w == DJ_F32 || ty.w == DJ_F16),
               "conv fwd: activations and the weight shadow are held as fp32 or fp16");
  DJ_CHECK_ARG(!io16 || (dj_compute_mode() == 1 && !rz.bn),
               "conv fwd: 16-bit tensors need arithmetic mode 1 (float16) and no in-kernel BatchNormalization finalize");
  DJ_CHECK_ARG((pro_scale == nullptr) == (pro_shift == nullptr), "conv fwd: pro_scale/pro_shift must come together");
  hipStream_t s = (hipStream_t)stream;
  DjIgemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = x;
  p.B = w;
  p.C = y;
  p.bias = bias;
  p.pro_scale = pro_scale;
  p.pro_shift = pro_shift;
  p.pro_relu = pro_relu;
  p.stats = stats;
  p.M = d->batch * d->out_h * d->out_w;
  p.N = d->out_c;
  p.K = d->kernel_h * d->kernel_w * d->in_c;
  p.rowH = d->out_h;
  p.rowW = d->out_w;
  p.srcH = d->in_h;
  p.srcW = d->in_w;
  p.srcC = d->in_c;
  p.ldsrc = d->ld_x;
  fill_geom(p, d);
  p.ldb = d->out_c;
  p.ldc = d->ld_y;
  p.cmap = 0;
  const bool y_zeroed = (relu & DJ_CONV_Y_ZEROED) != 0;
  const bool stats_may_split = (relu & DJ_CONV_STATS_MAY_SPLIT) != 0 && stats != nullptr && ws != nullptr && !rz.bn;
  relu &= DJ_CONV_RELU;
  p.relu = relu;
  p.vecA = (d->in_c % 4 == 0) && (d->ld_x % 4 == 0) && aligned_dt(x, ty.x) &&
           (!pro_scale || (aligned16(pro_scale) && aligned16(pro_shift)));
  p.vecB = (d->out_c % 4 == 0) && aligned_dt(w, ty.w);
  p.a_bytes = extent_bytes((long)d->batch * d->in_h * d->in_w, d->ld_x, d->in_c, dt_size(ty.x));
  p.a_dt = ty.x;
  p.c_dt = ty.y;
  p.sum_dt = ty.sum;
  if (rz.res) {
    p.A2 = rz.res;
    p.ldsrc2 = rz.ld_res;
    p.pro_scale2 = rz.res_scale;
    p.pro_shift2 = rz.res_shift;
    p.sum_out = rz.sum_out;
    p.ld_sum = rz.ld_sum;
    p.a2_bytes = extent_bytes((long)d->batch * d->in_h * d->in_w, rz.ld_res, d->in_c, dt_size(ty.x));
    p.sum_bytes = rz.sum_out ? extent_bytes((long)d->batch * d->in_h * d->in_w, rz.ld_sum, d->in_c, dt_size(ty.sum)) : 0;
  }
  p.b_bytes = extent_bytes((long)p.K, d->out_c, d->out_c, dt_size(ty.w));
  p.b_dt = ty.w;
  if (rz.bn) {
    const dj_bn_train& b = *rz.bn;
    DJ_CHECK_ARG(b.acc && b.ticket && b.gamma && b.beta && b.scale && b.shift && b.save_mean && b.save_invstd,
                 "conv fwd (BN): null pointer in dj_bn_train");
    DJ_CHECK_ARG((b.moving_mean == nullptr) == (b.moving_var == nullptr), "conv fwd (BN): moving stats come together");
    DJ_CHECK_ARG(stats == nullptr && !relu, "conv fwd (BN): no partial statistics / fused ReLU together with the fused finalize");
    p.bn_acc = b.acc;
    // one copy of the accumulators per 8192 GEMM rows: the big-M layers are the ones whose hundreds of workgroups
    // would otherwise queue on the same 2*N addresses
    p.bn_replicas = (int)std::min<long>(DJ_BN_ACC_REPLICAS, std::max<long>(1, p.M / 8192));
    p.bn_ticket = b.ticket;
    p.bn_gamma = b.gamma;
    p.bn_beta = b.beta;
    p.bn_moving_mean = b.moving_mean;
    p.bn_moving_var = b.moving_var;
    p.bn_scale = b.scale;
    p.bn_shift = b.shift;
    p.bn_save_mean = b.save_mean;
    p.bn_save_invstd = b.save_invstd;
    p.bn_eps = b.eps;
    p.bn_momentum = b.momentum;
    p.bn_count = (double)p.M;
  }
  // statistics come out of the GEMM epilogue (one K range per tile) -- or, for a caller that allows it and supplies a
  // workspace, out of the slab reduction of a split launch (tuned like a launch without statistics)
  const bool wants_stats = (stats != nullptr || rz.bn != nullptr) && !stats_may_split;
  int splits = 1;
  int cfg = choose_cfg(p.M, p.N, p.K, !wants_stats, &splits);
  tune_lookup(wants_stats ? 4 : 0, d, &cfg, &splits);
  if (wants_stats) splits = 1;
  if (ty.y != DJ_F32) splits = 1;   // a 16-bit result is written by one K range (no slabs, no atomics)
  p.kchunk = dj_cdiv(dj_cdiv(p.K, splits), DJ_BK) * DJ_BK;
  splits = dj_cdiv(p.K, p.kchunk);
  // split-K through slabs in the caller's workspace + a fixed-order reduction (deterministic), when they fit
  bool slabs = splits > 1 && ws != nullptr && !rz.bn && (long)splits * p.M * p.N <= ws_floats && aligned16(ws);
  if (splits > 1 && !slabs && stats_may_split) {   // statistics cannot come from atomics: one K range after all
    splits = 1;
    p.kchunk = dj_cdiv(p.K, DJ_BK) * DJ_BK;
  }
  if (slabs) {
    p.C = ws;
    p.ldc = p.N;
    p.slab_stride = (long)p.M * p.N;
    p.bias = nullptr;
    p.relu = 0;
    p.stats = nullptr;
  } else if (splits > 1) {
    p.atomic = 1;
    p.relu = 0;
    if (!y_zeroed) {
      hipError_t e = hipMemset2DAsync(y, (size_t)d->ld_y * 4, 0, (size_t)d->out_c * 4, (size_t)p.M, s);
      if (e != hipSuccess) {
        dj_set_error("conv fwd: memset: %s", hipGetErrorString(e));
        return DJ_ERR_HIP;
      }
    }
  }
  if (rz.res) {
    DJ_CHECK_ARG(dj_fast_mode_fwd(p) == 3, "conv fwd (residual add): the branch-free kernel's preconditions do not hold "
                                           "(channels %% 32, 16-byte aligned tensors)");
  }
  DJ_CHECK_ARG(!io16 || dj_fast_mode_fwd(p) != 0, "conv fwd: 16-bit tensors need the branch-free kernel's preconditions "
                                                  "(in_c %% 32 == 0, out_c %% 4 == 0, 16-byte aligned tensors)");
  if (int rc = dj_launch_cfg<0, 0>(cfg, p, splits, s)) return rc;
  if (slabs) return launch_splitk_reduce(ws, splits, p.slab_stride, p.M, p.N, bias, relu, y, d->ld_y, stats, s);
  if (splits > 1 && relu) {
    long total = (long)p.M * p.N;
    int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(dj_relu_rows_kernel, dim3(blocks), dim3(256), 0, s, y, (long)p.M, p.N, d->ld_y);
    DJ_CHECK_LAUNCH("dj_relu_rows_kernel");
  }
  return DJ_OK;
}

extern "C" int dj_conv2d_nhwc_fwd(const dj_conv2d_desc* d, const float* x, const float* w, const float* bias,
                                  float* y, const float* pro_scale, const float* pro_shift, int pro_relu, int relu,
                                  float* stats, void* stream) {
  return conv_fwd_impl(d, x, w, bias, y, pro_scale, pro_shift, pro_relu, relu, stats, FwdResidual(), stream);
}

// As dj_conv2d_nhwc_fwd, with a caller-owned workspace of `workspace_floats` floats (dj_conv2d_fwd_workspace_floats): a
// split-K launch then writes its partial tiles there and a fixed-order reduction produces y -- bit-reproducible, where
// the version without workspace accumulates with fp32 atomics in arrival order.  With DJ_CONV_STATS_MAY_SPLIT in `relu`
// a launch with `stats` may be split too (the reduction takes the statistics).
extern "C" int dj_conv2d_nhwc_fwd_ws(const dj_conv2d_desc* d, const float* x, const float* w, const float* bias,
                                     float* y, const float* pro_scale, const float* pro_shift, int pro_relu, int relu,
                                     float* stats, float* workspace, long workspace_floats, void* stream) {
  DJ_CHECK_ARG(workspace_floats >= 0 && (workspace != nullptr || workspace_floats == 0), "conv fwd: bad workspace");
  return conv_fwd_impl(d, x, w, bias, y, pro_scale, pro_shift, pro_relu, relu, stats, FwdResidual(), stream, workspace,
                       workspace_floats);
}

// Floats of workspace with which dj_conv2d_nhwc_fwd_ws runs this geometry without atomics under the CURRENT tuning
// choice (0: the launch is not split).  may_split_stats: as the flag of the same name.
extern "C" long dj_conv2d_fwd_workspace_floats(const dj_conv2d_desc* d, int may_split_stats) {
  if (check_desc(d)) return -1;
  const int M = d->batch * d->out_h * d->out_w, N = d->out_c, K = d->kernel_h * d->kernel_w * d->in_c;
  int splits = 1;
  int cfg = choose_cfg(M, N, K, true, &splits);
  tune_lookup(0, d, &cfg, &splits);
  (void)may_split_stats;
  const int kchunk = dj_cdiv(dj_cdiv(K, splits), DJ_BK) * DJ_BK;
  splits = dj_cdiv(K, kchunk);
  return splits > 1 ? (long)splits * M * N : 0;
}

// Forward conv whose only consumer is a training-mode BatchNormalization: that layer's statistics, scale/shift and
// moving averages are produced by the same launch (res == NULL: plain input; else the residual-add prologue above).
extern "C" int dj_conv2d_nhwc_fwd_bn(const dj_conv2d_desc* d, const float* x, const float* w, const float* bias, float* y,
                                     const float* pro_scale, const float* pro_shift, int pro_relu, const float* res,
                                     int ld_res, const float* res_scale, const float* res_shift, float* sum_out,
                                     int ld_sum, const dj_bn_train* bn, void* stream) {
  DJ_CHECK_ARG(bn != nullptr, "conv fwd (BN): dj_bn_train is required");
  FwdResidual rz;
  rz.bn = bn;
  if (res) {
    DJ_CHECK_ARG(d && dj_conv2d_fwd_addrelu_supported(d), "conv fwd (BN, residual add): needs a 1x1 stride-1 unpadded conv "
                                                           "with in_c %% 32 == 0");
    DJ_CHECK_ARG(pro_scale && pro_shift, "conv fwd (BN, residual add): pro_scale and pro_shift are required");
    DJ_CHECK_ARG((res_scale == nullptr) == (res_shift == nullptr), "conv fwd (BN, residual add): res_scale/res_shift");
    DJ_CHECK_ARG(ld_res >= d->in_c && ld_res % 4 == 0 && (!sum_out || (ld_sum >= d->in_c && ld_sum % 4 == 0)),
                 "conv fwd (BN, residual add): bad ld_res / ld_sum");
    DJ_CHECK_ARG(aligned16(res) && aligned16(sum_out) && aligned16(res_scale) && aligned16(res_shift),
                 "conv fwd (BN, residual add): tensors must be 16-byte aligned");
    rz.res = res;
    rz.ld_res = ld_res;
    rz.res_scale = res_scale;
    rz.res_shift = res_shift;
    rz.sum_out = sum_out;
    rz.ld_sum = ld_sum;
    pro_relu = 1;
  }
  return conv_fwd_impl(d, x, w, bias, y, pro_scale, pro_shift, pro_relu, 0, nullptr, rz, stream);
}

extern "C" int dj_conv2d_fwd_addrelu_supported(const dj_conv2d_desc* d) {
  if (check_desc(d)) return 0;
  return d->kernel_h == 1 && d->kernel_w == 1 && d->stride_h == 1 && d->stride_w == 1 && d->pad_top == 0 &&
         d->pad_left == 0 && d->in_h == d->out_h && d->in_w == d->out_w && d->in_c % 32 == 0 && d->ld_x % 4 == 0 &&
         d->out_c % 4 == 0;
}

static int conv_fwd_addrelu_impl(const dj_conv2d_desc* d, const float* x, const float* w, const float* bias,
                                 float* y, const float* pro_scale, const float* pro_shift, const float* res,
                                 int ld_res, const float* res_scale, const float* res_shift, float* sum_out,
                                 int ld_sum, int relu, float* stats, float* ws, long ws_floats, void* stream,
                                 const FwdTypes& ty = FwdTypes()) {
  DJ_CHECK_ARG(d && dj_conv2d_fwd_addrelu_supported(d), "conv fwd (residual add): needs a 1x1 stride-1 unpadded conv with "
                                                         "in_c %% 32 == 0");
  DJ_CHECK_ARG(res && pro_scale && pro_shift, "conv fwd (residual add): res, pro_scale and pro_shift are required");
  DJ_CHECK_ARG((res_scale == nullptr) == (res_shift == nullptr), "conv fwd (residual add): res_scale/res_shift come together");
  DJ_CHECK_ARG(ld_res >= d->in_c && ld_res % 4 == 0 && (!sum_out || (ld_sum >= d->in_c && ld_sum % 4 == 0)),
               "conv fwd (residual add): bad ld_res / ld_sum");
  DJ_CHECK_ARG(aligned16(res) && aligned16(sum_out) && aligned16(res_scale) && aligned16(res_shift),
               "conv fwd (residual add): tensors must be 16-byte aligned");
  FwdResidual rz;
  rz.res = res;
  rz.ld_res = ld_res;
  rz.res_scale = res_scale;
  rz.res_shift = res_shift;
  rz.sum_out = sum_out;
  rz.ld_sum = ld_sum;
  return conv_fwd_impl(d, x, w, bias, y, pro_scale, pro_shift, 1, relu, stats, rz, stream, ws, ws_floats, ty);
}

// Forward convolution over tensors that carry their storage type (include/dj_hip.h): the superset of dj_conv2d_nhwc_fwd_ws
// (res == NULL) and dj_conv2d_nhwc_fwd_addrelu_ws.
extern "C" int dj_conv2d_nhwc_fwd_t(const dj_conv2d_desc* d, const void* x, int dt_x, const void* w, int dt_w, const float* bias,
                                    void* y, int dt_y, const float* pro_scale, const float* pro_shift, int pro_relu, int relu,
                                    float* stats, const void* res, int ld_res, const float* res_scale,
                                    const float* res_shift, void* sum_out, int ld_sum, int dt_sum, float* workspace,
                                    long workspace_floats, void* stream) {
  DJ_CHECK_ARG(dt_ok(dt_x) && dt_ok(dt_w) && dt_ok(dt_y) && dt_ok(dt_sum), "conv fwd: unknown storage type");
  DJ_CHECK_ARG(workspace_floats >= 0 && (workspace != nullptr || workspace_floats == 0), "conv fwd: bad workspace");
  FwdTypes ty;
  ty.x = dt_x;
  ty.w = dt_w;
  ty.y = dt_y;
  ty.sum = sum_out ? dt_sum : DJ_F32;
  if (res)
    return conv_fwd_addrelu_impl(d, (const float*)x, (const float*)w, bias, (float*)y, pro_scale, pro_shift, (const float*)res, ld_res,
                                 res_scale, res_shift, (float*)sum_out, ld_sum, relu, stats, workspace, workspace_floats,
                                 stream, ty);
  DJ_CHECK_ARG(!sum_out, "conv fwd: sum_out without res");
  return conv_fwd_impl(d, (const float*)x, (const float*)w, bias, (float*)y, pro_scale, pro_shift, pro_relu, relu, stats,
                       FwdResidual(), stream, workspace, workspace_floats, ty);
}

extern "C" int dj_conv2d_nhwc_fwd_addrelu(const dj_conv2d_desc* d, const float* x, const float* w, const float* bias,
                                          float* y, const float* pro_scale, const float* pro_shift, const float* res,
                                          int ld_res, const float* res_scale, const float* res_shift, float* sum_out,
                                          int ld_sum, int relu, float* stats, void* stream) {
  return conv_fwd_addrelu_impl(d, x, w, bias, y, pro_scale, pro_shift, res, ld_res, res_scale, res_shift, sum_out, ld_sum,
                               relu, stats, nullptr, 0, stream);
}

// dj_conv2d_nhwc_fwd_addrelu with the split-K workspace of dj_conv2d_nhwc_fwd_ws (same rules, same size query)
extern "C" int dj_conv2d_nhwc_fwd_addrelu_ws(const dj_conv2d_desc* d, const float* x, const float* w, const float* bias,
                                             float* y, const float* pro_scale, const float* pro_shift, const float* res,
                                             int ld_res, const float* res_scale, const float* res_shift, float* sum_out,
                                             int ld_sum, int relu, float* stats, float* workspace, long workspace_floats,
                                             void* stream) {
  DJ_CHECK_ARG(workspace_floats >= 0 && (workspace != nullptr || workspace_floats == 0), "conv fwd (residual add): bad workspace");
  return conv_fwd_addrelu_impl(d, x, w, bias, y, pro_scale, pro_shift, res, ld_res, res_scale, res_shift, sum_out, ld_sum,
                               relu, stats, workspace, workspace_floats, stream);
}

struct DgradBnBwd {   // dj_conv2d_nhwc_dgrad_bnbwd: see include/dj_hip.h
  const float* z;
  int dt_z;
  int ld_z;
  const float* mean;
  const float* invstd;
  const float* scale;
  const float* shift;
  float* partial;
};

static int conv_dgrad_impl(const dj_conv2d_desc* d, const float* dy, const float* w, const float* bias, float* dx, int beta,
                           const DgradBnBwd* bnb, void* stream, int dt_dy = DJ_F32, int dt_dx = DJ_F32, int dt_w = DJ_F32) {
  if (int rc = check_desc(d)) return rc;
  DJ_CHECK_ARG(dy && w && dx, "conv dgrad: null tensor");
  const bool io16 = dt_dy != DJ_F32 || dt_dx != DJ_F32 || dt_w != DJ_F32 || (bnb && bnb->dt_z != DJ_F32);
  DJ_CHECK_ARG((dt_dy == DJ_F32 || dt_dy == DJ_BF16) && (dt_w == DJ_F32 || dt_w == DJ_BF16) &&
                   (!bnb || bnb->dt_z == DJ_F32 || bnb->dt_z == DJ_F16),
               "conv dgrad: gradients and the weight shadow are held as fp32 or bf16, the BatchNormalization input as fp32 or fp16");
  DJ_CHECK_ARG(!io16 || dj_compute_mode() == 1, "conv dgrad: 16-bit tensors need arithmetic mode 1 (float16)");
  const bool one_k_range = (beta & DJ_DGRAD_NO_SPLIT) != 0;   // no split-K: no arrival-order arithmetic
  beta &= 1;
  hipStream_t s = (hipStream_t)stream;
  DjIgemmParams p;
  memset(&p, 0, sizeof(p));
  if (bnb) {
    p.bnb_z = bnb->z;
    p.bnb_ldz = bnb->ld_z;
    p.bnb_zbytes = extent_bytes((long)d->batch * d->in_h * d->in_w, bnb->ld_z, d->in_c, dt_size(bnb->dt_z));
    p.bnb_zdt = bnb->dt_z;
    DJ_CHECK_ARG(p.bnb_zbytes > 0, "conv dgrad + BN backward statistics: z of 2 GiB or more");
    p.bnb_mean = bnb->mean;
    p.bnb_invstd = bnb->invstd;
    p.bnb_scale = bnb->scale;
    p.bnb_shift = bnb->shift;
    p.stats = bnb->partial;
  }
  p.A = dy;
  p.B = w;
  p.C = dx;
  p.bias = bias;
  p.N = d->in_c;
  p.srcC = d->out_c;
  p.ldsrc = d->ld_y;
  p.ldb = d->out_c;
  p.bTapStride = (long)d->in_c * d->out_c;
  p.ldc = d->ld_x;
  p.beta = beta;
  p.vecA = (d->out_c % 4 == 0) && (d->ld_y % 4 == 0) && aligned_dt(dy, dt_dy);
  p.vecB = (d->out_c % 4 == 0) && aligned_dt(w, dt_w);
  p.a_bytes = extent_bytes((long)d->batch * d->out_h * d->out_w, d->ld_y, d->out_c, dt_size(dt_dy));
  p.b_bytes = extent_bytes((long)d->kernel_h * d->kernel_w * d->in_c, d->out_c, d->out_c, dt_size(dt_w));
  p.b_dt = dt_w;
  p.a_dt = dt_dy;
  p.c_dt = dt_dx;
  const int es_dx = dt_size(dt_dx);
  const long in_pixels = (long)d->batch * d->in_h * d->in_w;
  bool strided_1x1 = d->kernel_h == 1 && d->kernel_w == 1 && d->pad_top == 0 && d->pad_left == 0 &&
                     (d->stride_h > 1 || d->stride_w > 1) && d->stride_h == d->stride_w && bias == nullptr;
  int splits = 1;
  DJ_CHECK_ARG(!(bnb && strided_1x1), "conv dgrad + BN backward statistics: not for strided 1x1 convolutions");
  if (strided_1x1) {
    // compact GEMM over the output grid, rows scattered to the strided input pixels
    p.M = d->batch * d->out_h * d->out_w;
    p.K = d->out_c;
    p.rowH = d->out_h;
    p.rowW = d->out_w;
    p.srcH = d->out_h;
    p.srcW = d->out_w;
    p.KH = p.KW = 1;
    p.sH = p.sW = p.dH = p.dW = 1;
    p.pT = p.pL = 0;
    set_row_recip(p);
    p.cmap = 1;
    p.cgH = d->out_h;
    p.cgW = d->out_w;
    p.cH = d->in_h;
    p.cW = d->in_w;
    p.cS = d->stride_h;
    if (!beta) {
      hipError_t e = hipMemset2DAsync(dx, (size_t)d->ld_x * es_dx, 0, (size_t)d->in_c * es_dx, (size_t)in_pixels, s);
      if (e != hipSuccess) {
        dj_set_error("conv dgrad: memset: %s", hipGetErrorString(e));
        return DJ_ERR_HIP;
      }
    }
    int cfg = choose_cfg(p.M, p.N, p.K, false, &splits);
    tune_lookup(1, d, &cfg, &splits);
    p.kchunk = dj_cdiv(p.K, DJ_BK) * DJ_BK;
    DJ_CHECK_ARG(!io16 || (fast_mode<0, 1>(p)) != 0, "conv dgrad (strided 1x1): 16-bit tensors need the branch-free kernel's "
                                                   "preconditions (channels %% 32, 16-byte aligned tensors)");
    return dj_launch_cfg<0, 1>(cfg, p, 1, s);
  }
  p.M = (int)in_pixels;
  p.K = d->kernel_h * d->kernel_w * d->out_c;
  p.rowH = d->in_h;
  p.rowW = d->in_w;
  p.srcH = d->out_h;
  p.srcW = d->out_w;
  fill_geom(p, d);
  p.cmap = 0;
  int cfg = choose_cfg(p.M, p.N, p.K, true, &splits);
  // the launch that also takes BatchNormalization backward statistics runs the EPI twin and is never split: a tuner
  // entry of its own (direction 9), falling back to the plain input gradient's tile variant
  if (!(bnb && tune_lookup(9, d, &cfg, &splits))) tune_lookup(1, d, &cfg, &splits);
  if (one_k_range || dt_dx != DJ_F32) splits = 1;   // (a 16-bit result is written by one K range: no atomics)
  p.kchunk = dj_cdiv(dj_cdiv(p.K, splits), DJ_BK) * DJ_BK;
  splits = dj_cdiv(p.K, p.kchunk);
  DJ_CHECK_ARG(!io16 || (fast_mode<1, 1>(p)) != 0, "conv dgrad: 16-bit tensors need the branch-free kernel's preconditions "
                                                 "(stride 1, channels %% 32, 16-byte aligned tensors)");
  if (splits > 1) {
    p.atomic = 1;
    if (!beta) {
      hipError_t e = hipMemset2DAsync(dx, (size_t)d->ld_x * 4, 0, (size_t)d->in_c * 4, (size_t)in_pixels, s);
      if (e != hipSuccess) {
        dj_set_error("conv dgrad: memset: %s", hipGetErrorString(e));
        return DJ_ERR_HIP;
      }
    }
    p.beta = 0;
  }
  return dj_launch_cfg<1, 1>(cfg, p, splits, s);
}

extern "C" int dj_conv2d_nhwc_dgrad(const dj_conv2d_desc* d, const float* dy, const float* w, const float* bias,
                                    float* dx, int beta, void* stream) {
  return conv_dgrad_impl(d, dy, w, bias, dx, beta, nullptr, stream);
}

extern "C" int dj_conv2d_nhwc_dgrad_bnbwd(const dj_conv2d_desc* d, const float* dy, const float* w, float* dx, const float* z,
                                          int ld_z, const float* mean, const float* invstd, const float* scale,
                                          const float* shift, float* partial, void* stream) {
  DJ_CHECK_ARG(z && mean && invstd && partial, "conv dgrad + BN backward statistics: null tensor");
  DJ_CHECK_ARG((scale == nullptr) == (shift == nullptr), "conv dgrad + BN backward statistics: scale/shift must come together");
  DJ_CHECK_ARG(d && ld_z >= d->in_c, "conv dgrad + BN backward statistics: ld_z < in_c");
  DgradBnBwd b{z, DJ_F32, ld_z, mean, invstd, scale, shift, partial};
  // one K range per tile, no accumulation: the accumulator of a tile IS the gradient the statistics are taken of
  return conv_dgrad_impl(d, dy, w, nullptr, dx, DJ_DGRAD_NO_SPLIT, &b, stream);
}

// Input gradient over tensors that carry their storage type: the superset of dj_conv2d_nhwc_dgrad (z == NULL) and
// dj_conv2d_nhwc_dgrad_bnbwd.
extern "C" int dj_conv2d_nhwc_dgrad_t(const dj_conv2d_desc* d, const void* dy, int dt_dy, const void* w, int dt_w,
                                      const float* bias, void* dx, int dt_dx, int beta, const void* z, int ld_z, int dt_z, const float* mean,
                                      const float* invstd, const float* scale, const float* shift, float* partial,
                                      void* stream) {
  DJ_CHECK_ARG(dt_ok(dt_dy) && dt_ok(dt_w) && dt_ok(dt_dx) && dt_ok(dt_z), "conv dgrad: unknown storage type");
  if (!z)
    return conv_dgrad_impl(d, (const float*)dy, (const float*)w, bias, (float*)dx, beta, nullptr, stream, dt_dy, dt_dx, dt_w);
  DJ_CHECK_ARG(mean && invstd && partial, "conv dgrad + BN backward statistics: null tensor");
  DJ_CHECK_ARG((scale == nullptr) == (shift == nullptr), "conv dgrad + BN backward statistics: scale/shift must come together");
  DJ_CHECK_ARG(d && ld_z >= d->in_c && !bias && !(beta & 1), "conv dgrad + BN backward statistics: ld_z < in_c, or a bias / "
                                                              "an accumulating launch");
  DgradBnBwd b{(const float*)z, dt_z, ld_z, mean, invstd, scale, shift, partial};
  return conv_dgrad_impl(d, (const float*)dy, (const float*)w, nullptr, (float*)dx, DJ_DGRAD_NO_SPLIT, &b, stream, dt_dy, dt_dx,
                         dt_w);
}

static int conv_wgrad_impl(const dj_conv2d_desc* d, const float* x, const float* dy, float* dw, const float* pro_scale,
                           const float* pro_shift, int pro_relu, int dw_zeroed, void* stream, int dt_x, int dt_dy) {
  if (int rc = check_desc(d)) return rc;
  DJ_CHECK_ARG(x && dy && dw, "conv wgrad: null tensor");
  const bool io16 = dt_x != DJ_F32 || dt_dy != DJ_F32;
  DJ_CHECK_ARG((dt_x == DJ_F32 || dt_x == DJ_F16) && (dt_dy == DJ_F32 || dt_dy == DJ_BF16),
               "conv wgrad: activations are held as fp32 or fp16, gradients as fp32 or bf16");
  DJ_CHECK_ARG(!io16 || dj_compute_mode() == 1, "conv wgrad: 16-bit tensors need arithmetic mode 1 (float16)");
  DJ_CHECK_ARG((pro_scale == nullptr) == (pro_shift == nullptr), "conv wgrad: pro_scale/pro_shift must come together");
  hipStream_t s = (hipStream_t)stream;
  DjIgemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = x;
  p.B = dy;
  p.C = dw;
  p.pro_scale = pro_scale;
  p.pro_shift = pro_shift;
  p.pro_relu = pro_relu;
  p.M = d->kernel_h * d->kernel_w * d->in_c;
  p.N = d->out_c;
  p.K = d->batch * d->out_h * d->out_w;
  p.rowH = d->out_h;
  p.rowW = d->out_w;
  p.srcH = d->in_h;
  p.srcW = d->in_w;
  p.srcC = d->in_c;
  p.ldsrc = d->ld_x;
  fill_geom(p, d);
  p.ldb = d->ld_y;
  p.ldc = d->out_c;
  p.cmap = 0;
  p.vecA = (d->in_c % 4 == 0) && (d->ld_x % 4 == 0) && aligned_dt(x, dt_x) &&
           (!pro_scale || (aligned16(pro_scale) && aligned16(pro_shift)));
  p.vecB = (d->out_c % 4 == 0) && (d->ld_y % 4 == 0) && aligned_dt(dy, dt_dy);
  p.a_bytes = extent_bytes((long)d->batch * d->in_h * d->in_w, d->ld_x, d->in_c, dt_size(dt_x));
  p.b_bytes = extent_bytes((long)d->batch * d->out_h * d->out_w, d->ld_y, d->out_c, dt_size(dt_dy));
  p.a_dt = dt_x;
  p.b_dt = dt_dy;
  // tile by the (taps*Cin) x Cout extent only -- the pixel reduction is split over blockIdx.y to fill the chip
  int splits = 1;
  int cfg;
  if (p.N > 96 && (long)dj_cdiv(p.N, 128) * 128 <= (long)dj_cdiv(p.N, 64) * 64)
    cfg = (p.M >= 128) ? CFG_128x128 : CFG_64x64;
  else if (p.N > 32)
    cfg = (p.M >= 128) ? CFG_128x64 : CFG_64x64;
  else
    cfg = CFG_128x32;
  long t = (long)dj_cdiv(p.M, kCfgs[cfg].bm) * dj_cdiv(p.N, kCfgs[cfg].bn);
  long want = (768 + t - 1) / t;
  long maxs = p.K / 128;
  if (maxs < 1) maxs = 1;
  splits = (int)(want < maxs ? want : maxs);
  if (splits < 1) splits = 1;
  tune_lookup(2, d, &cfg, &splits);
  p.kchunk = dj_cdiv(dj_cdiv(p.K, splits), DJ_BK) * DJ_BK;
  splits = dj_cdiv(p.K, p.kchunk);
  if (splits > 1) {
    p.atomic = 1;
    if (!dw_zeroed) {
      hipError_t e = hipMemsetAsync(dw, 0, (size_t)p.M * p.N * 4, s);
      if (e != hipSuccess) {
        dj_set_error("conv wgrad: memset: %s", hipGetErrorString(e));
        return DJ_ERR_HIP;
      }
    }
  }
  DJ_CHECK_ARG(!io16 || (fast_mode<2, 0>(p)) != 0, "conv wgrad: 16-bit tensors need the branch-free kernel's preconditions "
                                                 "(channels %% 4, 16-byte aligned tensors)");
  return dj_launch_cfg<2, 0>(cfg, p, splits, s);
}

extern "C" int dj_conv2d_nhwc_wgrad(const dj_conv2d_desc* d, const float* x, const float* dy, float* dw,
                                    const float* pro_scale, const float* pro_shift, int pro_relu, int dw_zeroed,
                                    void* stream) {
  return conv_wgrad_impl(d, x, dy, dw, pro_scale, pro_shift, pro_relu, dw_zeroed, stream, DJ_F32, DJ_F32);
}

// Weight gradient over tensors that carry their storage type (dw is always the fp32 gradient of the master weights).
extern "C" int dj_conv2d_nhwc_wgrad_t(const dj_conv2d_desc* d, const void* x, int dt_x, const void* dy, int dt_dy, float* dw,
                                      const float* pro_scale, const float* pro_shift, int pro_relu, int dw_zeroed,
                                      void* stream) {
  DJ_CHECK_ARG(dt_ok(dt_x) && dt_ok(dt_dy), "conv wgrad: unknown storage type");
  return conv_wgrad_impl(d, (const float*)x, (const float*)dy, dw, pro_scale, pro_shift, pro_relu, dw_zeroed, stream, dt_x,
                         dt_dy);
}

extern "C" int dj_conv2d_default_config(int dir, const dj_conv2d_desc* d, int* cfg, int* splits) {
  if (int rc = check_desc(d)) return rc;
  DJ_CHECK_ARG(cfg && splits, "default_config: null output");
  int base = dir & 3;
  *splits = 1;
  if (base == 0) {
    *cfg = choose_cfg((long)d->batch * d->out_h * d->out_w, d->out_c, (long)d->kernel_h * d->kernel_w * d->in_c,
                      (dir & 4) == 0, splits);
  } else if (base == 1) {
    bool strided_1x1 = d->kernel_h == 1 && d->kernel_w == 1 && d->pad_top == 0 && d->pad_left == 0 &&
                       (d->stride_h > 1 || d->stride_w > 1) && d->stride_h == d->stride_w;
    if (strided_1x1)
      *cfg = choose_cfg((long)d->batch * d->out_h * d->out_w, d->in_c, d->out_c, false, splits);
    else
      *cfg = choose_cfg((long)d->batch * d->in_h * d->in_w, d->in_c, (long)d->kernel_h * d->kernel_w * d->out_c, true,
                        splits);
  } else {
    long M = (long)d->kernel_h * d->kernel_w * d->in_c, N = d->out_c, K = (long)d->batch * d->out_h * d->out_w;
    int c;
    if (N > 96 && (long)dj_cdiv(N, 128) * 128 <= (long)dj_cdiv(N, 64) * 64)
      c = (M >= 128) ? CFG_128x128 : CFG_64x64;
    else if (N > 32)
      c = (M >= 128) ? CFG_128x64 : CFG_64x64;
    else
      c = CFG_128x32;
    long t = (long)dj_cdiv(M, kCfgs[c].bm) * dj_cdiv(N, kCfgs[c].bn);
    long want = (768 + t - 1) / t, maxs = K / 128;
    if (maxs < 1) maxs = 1;
    *cfg = c;
    *splits = (int)(want < maxs ? want : maxs);
    if (*splits < 1) *splits = 1;
  }
  return DJ_OK;
}
