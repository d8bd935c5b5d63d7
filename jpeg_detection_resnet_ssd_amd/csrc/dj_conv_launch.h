// Tile variants of the implicit-GEMM convolution and their launchers.  dj_launch_cfg<AM, BMD> is instantiated once per
// GEMM role in its own translation unit (dj_conv_i*.hip) so that the ~100 kernel instantiations compile in parallel.
#pragma once
#include "../../include/dj_hip.h"
#include "dj_igemm.h"
#include "dj_igemm_fast.h"
#include <stdlib.h>
#include <atomic>

// ---------------------------------------------------------------------------------
// tile configurations
// ---------------------------------------------------------------------------------
struct TileCfg {
  int bm, bn;
};
static const TileCfg kCfgs[] = {{128, 128}, {128, 64}, {64, 64}, {128, 32}, {128, 128}, {128, 64}, {64, 64}, {64, 64}, {128, 64}, {128, 128}, {128, 64}, {64, 64}, {64, 64}, {128, 64}};
// *_S1: same tile with a single LDS stage; *_S1P: single stage with loads two K-steps ahead; *_P: two stages with
// the pinned-load / read-ahead schedule (fast kernel only; the generic kernel ignores the distinction)
enum {
  CFG_128x128 = 0,
  CFG_128x64,
  CFG_64x64,
  CFG_128x32,
  CFG_128x128_S1,
  CFG_128x64_S1,
  CFG_64x64_S1,
  CFG_64x64_S1P,
  CFG_128x64_S1P,
  CFG_128x128_P,
  CFG_128x64_P,
  CFG_64x64_P,
  CFG_64x64_PK2,
  CFG_128x64_PK2,
  N_CFG
};

template <typename KernT>
static int launch_kernel(KernT kern, int smem_bytes, int bm, int bn, const DjIgemmParams& p, int splits, hipStream_t s,
                         std::atomic<bool>* attr_done, int threads = 256) {
  // (first launch of an instantiation: two threads may both get here -- the call is idempotent, the flag is atomic so
  // that the header's "callable from any thread" holds by the letter as well)
  if (!attr_done->load(std::memory_order_relaxed)) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
    if (e != hipSuccess) {
      dj_set_error("hipFuncSetAttribute(%d B LDS): %s", smem_bytes, hipGetErrorString(e));
      return DJ_ERR_HIP;
    }
    attr_done->store(true, std::memory_order_relaxed);
  }
  int tiles_m = dj_cdiv(p.M, bm), tiles_n = dj_cdiv(p.N, bn);
  dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)splits);
  hipLaunchKernelGGL(kern, grid, dim3(threads), smem_bytes, s, p);
  DJ_CHECK_LAUNCH("dj_igemm_kernel");
  return DJ_OK;
}

// One launch site per kernel instantiation; an input-gradient launch that asks for BatchNormalization backward statistics
// (p.bnb_z) takes the EPI = 1 twin of the same variant.
template <int BM, int BN, int WM, int WN, int AM, int BMD, int PRO, int NSTAGE, int PREC, int KS, int NP>
static int launch_fast(int smem_bytes, const DjIgemmParams& p, int splits, hipStream_t s, int threads = 256) {
  if constexpr (AM == 1 && BMD == 1 && PRO == 0) {   // (the input-gradient GEMM has no prologue)
    if (p.bnb_z) {
      static std::atomic<bool> done1{false};
      return launch_kernel(dj_igemm_fast_kernel<BM, BN, WM, WN, AM, BMD, PRO, NSTAGE, PREC, KS, NP, 1>, smem_bytes, BM, BN, p,
                           splits, s, &done1, threads);
    }
  }
  static std::atomic<bool> done0{false};
  return launch_kernel(dj_igemm_fast_kernel<BM, BN, WM, WN, AM, BMD, PRO, NSTAGE, PREC, KS, NP, 0>, smem_bytes, BM, BN, p, splits,
                       s, &done0, threads);
}

// fast = 0: generic kernel; 1: branch-free kernel; 2: branch-free kernel with the affine prologue
template <int BM, int BN, int WM, int WN, int AM, int BMD, int NSTAGE>
static int launch_one(const DjIgemmParams& p, int splits, hipStream_t s, int fast) {
  using Cfg = DjIgemmCfg<BM, BN, WM, WN, AM, BMD>;
  static std::atomic<bool> done[1] = {{false}};
  const int smem_fast = Cfg::SMEM_BYTES / 2 * ((NSTAGE == 2 || NSTAGE == 4) ? 2 : 1);
  {
    // 1x1 kernels without padding: the variant that does no per-K-step bounds arithmetic (NP, see dj_igemm_fast.h); for
    // the weight gradient also stride 1 (input pixel == output pixel)
    static const bool np_off = getenv("DJ_NO_NP") != nullptr;
    const bool same_grid = (AM != 2) || (p.sH == 1 && p.sW == 1 && p.rowH == p.srcH && p.rowW == p.srcW);
    if ((fast == 1 || fast == 2) && p.KH == 1 && p.KW == 1 && p.pT == 0 && p.pL == 0 && same_grid && !np_off) {
      if (fast == 1)
        return launch_fast<BM, BN, WM, WN, AM, BMD, 0, NSTAGE, 0, 1, 1>(smem_fast, p, splits, s);
      return launch_fast<BM, BN, WM, WN, AM, BMD, 1, NSTAGE, 0, 1, 1>(smem_fast, p, splits, s);
    }
    if constexpr (AM == 2) {
      // weight gradient with taps / padding / stride whose tiles lie under one tap each: incremental pixel walk (NP 3)
      static const bool walk_off = getenv("DJ_NO_WALK") != nullptr;
      if ((fast == 1 || fast == 2) && !np_off && !walk_off && p.srcC % BM == 0) {
        if (fast == 1)
          return launch_fast<BM, BN, WM, WN, AM, BMD, 0, NSTAGE, 0, 1, 3>(smem_fast, p, splits, s);
        return launch_fast<BM, BN, WM, WN, AM, BMD, 1, NSTAGE, 0, 1, 3>(smem_fast, p, splits, s);
      }
    }
    if constexpr (AM != 2) {
      // kernels with taps / padding: per-tap row offsets cached across the K-steps of a tap (NP 2)
      static const bool ht_off = getenv("DJ_NO_TAPCACHE") != nullptr;
      if ((fast == 1 || fast == 2) && !np_off && !ht_off) {
        if (fast == 1)
          return launch_fast<BM, BN, WM, WN, AM, BMD, 0, NSTAGE, 0, 1, 2>(smem_fast, p, splits, s);
        return launch_fast<BM, BN, WM, WN, AM, BMD, 1, NSTAGE, 0, 1, 2>(smem_fast, p, splits, s);
      }
    }
  }
  if (fast == 1)
    return launch_fast<BM, BN, WM, WN, AM, BMD, 0, NSTAGE, 0, 1, 0>(smem_fast, p, splits, s);
  if (fast == 2)
    return launch_fast<BM, BN, WM, WN, AM, BMD, 1, NSTAGE, 0, 1, 0>(smem_fast, p, splits, s);
  if (fast == 3) {   // residual-add prologue: forward GEMM only
    if constexpr (AM == 0 && BMD == 0) {
      return launch_fast<BM, BN, WM, WN, AM, BMD, 3, NSTAGE, 0, 1, 0>(smem_fast, p, splits, s);
    } else {
      dj_set_error("residual-add prologue outside the forward GEMM");
      return DJ_ERR_ARG;
    }
  }
  return launch_kernel(dj_igemm_kernel<BM, BN, WM, WN, AM, BMD>, Cfg::SMEM_BYTES, BM, BN, p, splits, s, &done[0]);
}

// two K groups per workgroup: pipelined schedule, 512 threads, two LDS rings
template <int BM, int BN, int AM, int BMD>
static int launch_k2(const DjIgemmParams& p, int splits, hipStream_t s, int fast) {
  using Cfg = DjIgemmCfg<BM, BN, 2, 2, AM, BMD>;
  if (fast == 1)
    return launch_fast<BM, BN, 2, 2, AM, BMD, 0, 4, 0, 2, 0>(2 * Cfg::SMEM_BYTES, p, splits, s, 512);
  if (fast == 3) {
    if constexpr (AM == 0 && BMD == 0) {
      return launch_fast<BM, BN, 2, 2, AM, BMD, 3, 4, 0, 2, 0>(2 * Cfg::SMEM_BYTES, p, splits, s, 512);
    } else {
      dj_set_error("residual-add prologue outside the forward GEMM");
      return DJ_ERR_ARG;
    }
  }
  return launch_fast<BM, BN, 2, 2, AM, BMD, 1, 4, 0, 2, 0>(2 * Cfg::SMEM_BYTES, p, splits, s, 512);
}

extern std::atomic<bool> g_dj_allow_fast;   // dj_conv.hip

// 0: fp32 results (default; fp32 MFMA or split-bf16 kernel per tuning entry), 5: fp32 MFMA only.  1: forward GEMMs round
// their operands to fp16, gradient GEMMs (dgrad, wgrad) to bf16 (gradients need the exponent range); 2: bf16 everywhere;
// 3 / 4: fp32 products as three / six bf16 MFMAs on split operands.  fp32 accumulation in all modes.
int dj_compute_mode();   // dj_conv.hip: the calling thread's override, else the process default

// Preconditions of dj_igemm_fast_kernel (see its header comment).
template <int AM, int BMD>
static int fast_mode(const DjIgemmParams& p) {
  if (!g_dj_allow_fast.load(std::memory_order_relaxed) || !p.vecA || !p.vecB) return 0;
  if (p.a_bytes <= 0 || p.b_bytes <= 0) return 0;  // operand >= 2 GiB (extent overflowed int)
  if (AM != 2 && p.srcC % 32 != 0) return 0;
  if (AM != 2 && p.M >= (1 << 22)) return 0;   // dj_row_decompose: float-reciprocal row decomposition
  if (AM == 1 && (p.sH != 1 || p.sW != 1)) return 0;
  if (AM == 2 && (p.srcC % 4 != 0 || p.K >= (1 << 24))) return 0;
  if (BMD == 0 && (p.N % 4 != 0 || p.ldb % 4 != 0)) return 0;
  if (BMD == 1 && p.srcC % 32 != 0) return 0;
  if (p.A2) return (AM == 0 && BMD == 0 && p.pro_scale) ? 3 : 0;
  return p.pro_scale ? 2 : 1;
}

// reduced-precision MFMA variants exist for the two-stage 128x128 / 128x64 / 64x64 tiles of the fast kernel
static inline int dj_fast_mode_fwd(const DjIgemmParams& p) { return fast_mode<0, 0>(p); }

// Reduced-precision (16-bit-tile) variants: defined in dj_conv_launch_h16.h and instantiated in translation units of their
// own (dj_conv_h*.hip), so that they compile beside the fp32 kernels instead of behind them
template <int AM, int BMD>
int dj_launch_lowp(int cfg, const DjIgemmParams& p, int splits, hipStream_t s, int fast, int mode);

template <int AM, int BMD>
int dj_launch_cfg(int cfg, const DjIgemmParams& p, int splits, hipStream_t s) {
  const int fast = fast_mode<AM, BMD>(p);
  int mode = dj_compute_mode();
  if (mode == 5) mode = 0;                      // fp32 MFMA kernels only: its tables hold indices below N_CFG
  if (cfg >= N_CFG) {
    // mode 0, upper half of the index range: the split-bf16 kernel (fp32 tensors, fp32 results; dj_igemm_h16.h PREC 4)
    // of that index -- where the branch-free kernels' preconditions do not hold, the fp32 variant of the same index
    cfg -= N_CFG;
    if (fast && mode == 0 && cfg < N_CFG) return dj_launch_lowp<AM, BMD>(cfg, p, splits, s, fast, 4);
  }
  if (fast && mode != 0 && cfg >= 0 && cfg < N_CFG) {
    // A-mode 0 with B-mode 0 is the forward GEMM; everything else carries gradients
    return dj_launch_lowp<AM, BMD>(cfg, p, splits, s, fast, mode);
  }
  switch (cfg) {
    case CFG_128x128: return launch_one<128, 128, 2, 2, AM, BMD, 2>(p, splits, s, fast);
    case CFG_128x64: return launch_one<128, 64, 2, 2, AM, BMD, 2>(p, splits, s, fast);
    case CFG_64x64: return launch_one<64, 64, 2, 2, AM, BMD, 2>(p, splits, s, fast);
    case CFG_128x32: return launch_one<128, 32, 4, 1, AM, BMD, 2>(p, splits, s, fast);
    case CFG_128x128_S1:
      return fast ? launch_one<128, 128, 2, 2, AM, BMD, 1>(p, splits, s, fast)
                  : launch_one<128, 128, 2, 2, AM, BMD, 2>(p, splits, s, fast);
    case CFG_128x64_S1:
      return fast ? launch_one<128, 64, 2, 2, AM, BMD, 1>(p, splits, s, fast)
                  : launch_one<128, 64, 2, 2, AM, BMD, 2>(p, splits, s, fast);
    case CFG_64x64_S1:
      return fast ? launch_one<64, 64, 2, 2, AM, BMD, 1>(p, splits, s, fast)
                  : launch_one<64, 64, 2, 2, AM, BMD, 2>(p, splits, s, fast);
    case CFG_64x64_S1P:
      return fast ? launch_one<64, 64, 2, 2, AM, BMD, 3>(p, splits, s, fast)
                  : launch_one<64, 64, 2, 2, AM, BMD, 2>(p, splits, s, fast);
    case CFG_128x64_S1P:
      return fast ? launch_one<128, 64, 2, 2, AM, BMD, 3>(p, splits, s, fast)
                  : launch_one<128, 64, 2, 2, AM, BMD, 2>(p, splits, s, fast);
    case CFG_128x128_P:
      return fast ? launch_one<128, 128, 2, 2, AM, BMD, 4>(p, splits, s, fast)
                  : launch_one<128, 128, 2, 2, AM, BMD, 2>(p, splits, s, fast);
    case CFG_128x64_P:
      return fast ? launch_one<128, 64, 2, 2, AM, BMD, 4>(p, splits, s, fast)
                  : launch_one<128, 64, 2, 2, AM, BMD, 2>(p, splits, s, fast);
    case CFG_64x64_P:
      return fast ? launch_one<64, 64, 2, 2, AM, BMD, 4>(p, splits, s, fast)
                  : launch_one<64, 64, 2, 2, AM, BMD, 2>(p, splits, s, fast);
    case CFG_64x64_PK2:
      return fast ? launch_k2<64, 64, AM, BMD>(p, splits, s, fast) : launch_one<64, 64, 2, 2, AM, BMD, 2>(p, splits, s, fast);
    case CFG_128x64_PK2:
      return fast ? launch_k2<128, 64, AM, BMD>(p, splits, s, fast)
                  : launch_one<128, 64, 2, 2, AM, BMD, 2>(p, splits, s, fast);
  }
  dj_set_error("bad tile cfg %d", cfg);
  return DJ_ERR_ARG;
}

