// Launchers of the reduced-precision (16-bit-tile) convolution kernels, dj_igemm_h16.h.  Included by dj_conv_h*.hip only:
// one translation unit per GEMM role, beside the fp32 units dj_conv_i*.hip.
#pragma once
#include "dj_conv_launch.h"
#include "dj_igemm_h16.h"

// an input-gradient launch that asks for BatchNormalization backward statistics (p.bnb_z) takes the EPI = 1 twin.
// AT / BT: how the operands A (and A2) / B are stored in HBM (0 fp32, 1 fp16, 2 bf16), see dj_igemm_h16.h.
template <int BM, int BN, int AM, int BMD, int PRO, int PREC, int BK, int PF, int AT, int BT, int NP>
static int launch_h16_np(int smem_bytes, const DjIgemmParams& p, int splits, hipStream_t s) {
  if constexpr (AM == 1 && BMD == 1 && PRO == 0) {
    if (p.bnb_z) {
      static std::atomic<bool> done1{false};
      return launch_kernel(dj_igemm_h16_kernel<BM, BN, AM, BMD, PRO, PREC, BK, PF, 1, AT, BT, NP>, smem_bytes, BM, BN, p, splits,
                           s, &done1, 256);
    }
  }
  static std::atomic<bool> done0{false};
  return launch_kernel(dj_igemm_h16_kernel<BM, BN, AM, BMD, PRO, PREC, BK, PF, 0, AT, BT, NP>, smem_bytes, BM, BN, p, splits, s,
                       &done0, 256);
}

template <int BM, int BN, int AM, int BMD, int PRO, int PREC, int BK, int PF, int AT, int BT>
static int launch_h16_kernel(int smem_bytes, const DjIgemmParams& p, int splits, hipStream_t s) {
  // 1x1 kernels without padding (for the weight gradient also stride 1, same pixel grid): the variant without per-K-step
  // bounds arithmetic (NP of dj_igemm_h16.h); DJ_NO_NP=1 switches it off
  if constexpr (PRO != 3) {
    static const bool np_off = getenv("DJ_NO_NP") != nullptr;
    const bool same_grid = (AM != 2) || (p.sH == 1 && p.sW == 1 && p.rowH == p.srcH && p.rowW == p.srcW);
    if (!np_off && p.KH == 1 && p.KW == 1 && p.pT == 0 && p.pL == 0 && same_grid)
      return launch_h16_np<BM, BN, AM, BMD, PRO, PREC, BK, PF, AT, BT, 1>(smem_bytes, p, splits, s);
  }
  return launch_h16_np<BM, BN, AM, BMD, PRO, PREC, BK, PF, AT, BT, 0>(smem_bytes, p, splits, s);
}

// BK: K-step depth of the 16-bit-tile kernel (64: forward / input gradient), PF: K-steps of register prefetch -- see
// dj_igemm_h16.h
template <int BM, int BN, int AM, int BMD, int PREC, int BK, int PF, int AT, int BT>
static int launch_h16(const DjIgemmParams& p, int splits, hipStream_t s, int fast) {
  // float32x3 / float32x6: two / three bf16 images per operand
  constexpr int SMEM_BYTES = DjH16Cfg<BM, BN, AM, BMD, BK>::SMEM_BYTES * (PREC == 4 ? 3 : PREC == 3 ? 2 : 1);
  static_assert(SMEM_BYTES <= 160 * 1024, "LDS of a CU");
  if (fast == 3) {
    if constexpr (AM == 0 && BMD == 0) {
      return launch_h16_kernel<BM, BN, AM, BMD, 3, PREC, BK, PF, AT, BT>(SMEM_BYTES, p, splits, s);
    } else {
      dj_set_error("residual-add prologue outside the forward GEMM");
      return DJ_ERR_ARG;
    }
  }
  if (fast == 1)
    return launch_h16_kernel<BM, BN, AM, BMD, 0, PREC, BK, PF, AT, BT>(SMEM_BYTES, p, splits, s);
  if constexpr (AM != 1) {  // the input-gradient GEMM has no prologue
    return launch_h16_kernel<BM, BN, AM, BMD, 1, PREC, BK, PF, AT, BT>(SMEM_BYTES, p, splits, s);
  } else {
    dj_set_error("prologue on the input-gradient GEMM");
    return DJ_ERR_ARG;
  }
}

template <int BM, int BN, int AM, int BMD, int PREC, int PF, int AT, int BT>
static int launch_h16_depth(const DjIgemmParams& p, int splits, hipStream_t s, int fast, bool deep) {
  static const bool bk32 = getenv("DJ_H16_BK32") != nullptr;   // DJ_H16_BK32=1: 32-deep K-steps only
  if constexpr (PREC >= 3 && BM == 128) {
    // float32x6, 128-row tiles: three images of a 64-deep stage pair exceed the 160 KB of a CU, and those of a 32-deep one
    // (92-120 KB) leave room for ONE workgroup = one wave per SIMD, with nobody to issue while it waits (counters: matrix
    // pipe 50 % busy, 44 % of the wave's cycles waiting).  The "deep" indices name 16-deep K-steps instead: 56-74 KB, two
    // workgroups per CU -- all three GEMM roles.  float32x3 likewise: its 64-deep pair is 110-147 KB, and even the 80 KB of
    // the 32-deep 128x128 pair runs one workgroup per CU (counters: 0.87 waves per SIMD, matrix pipe 34 % busy); 16-deep: 45 KB.
    if (deep && !bk32 && p.srcC % 16 == 0 && p.kchunk % 16 == 0)
      return launch_h16<BM, BN, AM, BMD, PREC, 16, PF, AT, BT>(p, splits, s, fast);
  } else if constexpr (AM != 2) {
    // 64-deep K-steps where a step stays inside one filter tap and every K chunk is whole
    constexpr int IMGS = (PREC == 4 ? 3 : PREC == 3 ? 2 : 1);
    if constexpr (DjH16Cfg<BM, BN, AM, BMD, 64>::SMEM_BYTES * IMGS <= 160 * 1024) {
      if (deep && !bk32 && p.srcC % 64 == 0 && p.kchunk % 64 == 0)
        return launch_h16<BM, BN, AM, BMD, PREC, 64, PF, AT, BT>(p, splits, s, fast);
    }
  }
  return launch_h16<BM, BN, AM, BMD, PREC, 32, PF, AT, BT>(p, splits, s, fast);
}

template <int BM, int BN, int AM, int BMD, int PREC, int AT, int BT>
static int launch_lowp(const DjIgemmParams& p, int splits, hipStream_t s, int fast, bool deep, bool pf2) {
  // 16-bit tiles in LDS + 16-deep MFMA (dj_igemm_h16.h).  (The round-1 path -- fp32 tiles, fragments rounded at read time,
  // 8-deep MFMA: PREC != 0 of dj_igemm_fast_kernel -- is no longer instantiated; it was kept for A/B runs until the end
  // of round 2: 16.63 ms against 15.04 ms per fp16 step when the 16-bit tiles arrived.)
  // (round 2: "the 128x128 tile has no registers to spare for a second prefetch set" -- with 16-bit operands a staging set is
  // half the registers, and the tuner decides: 128x128 with two prefetch sets are configurations 4 and 13)
  static const bool pf1 = getenv("DJ_H16_PF1") != nullptr;
  if (pf2 && !pf1) return launch_h16_depth<BM, BN, AM, BMD, PREC, 2, AT, BT>(p, splits, s, fast, deep);
  return launch_h16_depth<BM, BN, AM, BMD, PREC, 1, AT, BT>(p, splits, s, fast, deep);
}

// Reduced-precision variants behind the fourteen configuration indices of the tuner (the schedule variants of the fp32
// kernel do not exist here; their indices select K-step depth and prefetch depth of the 16-bit-tile kernel instead):
//   tile 128x128: 0 = 32-deep; 4 = 32-deep, two prefetch sets; 9 = 64-deep; 13 = 64-deep, two prefetch sets
//   tile 128x64:  1 = 32-deep; 5 = 32-deep, two prefetch sets; 10 = 64-deep; 8 = 64-deep, two prefetch sets
//   tile 64x64:   2 = 32-deep; 3, 6 = 32-deep, two prefetch sets; 11 = 64-deep; 7, 12 = 64-deep, two prefetch sets
// (the weight gradient, whose reduction runs over pixels, has 32-deep K-steps only)
template <int AM, int BMD, int PREC, int AT, int BT>
static int launch_lowp_cfg(int cfg, const DjIgemmParams& p, int splits, hipStream_t s, int fast) {
  int bm = kCfgs[cfg].bm, bn = kCfgs[cfg].bn;
  if (cfg == CFG_128x64_PK2) bn = 128;   // (index 13 was a twin of 8: it now names the 128x128 tile with two prefetch sets)
  const bool deep = cfg >= CFG_64x64_S1P;
  const bool pf2 = cfg == CFG_128x32 || cfg == CFG_128x64_S1 || cfg == CFG_64x64_S1 || cfg == CFG_64x64_S1P ||
                   cfg == CFG_128x64_S1P || cfg == CFG_64x64_PK2 || cfg == CFG_128x64_PK2 || cfg == CFG_128x128_S1;
  if (bm == 128 && bn == 128) return launch_lowp<128, 128, AM, BMD, PREC, AT, BT>(p, splits, s, fast, deep, pf2);
  if (bm == 128 && bn == 64) return launch_lowp<128, 64, AM, BMD, PREC, AT, BT>(p, splits, s, fast, deep, pf2);
  return launch_lowp<64, 64, AM, BMD, PREC, AT, BT>(p, splits, s, fast, deep, pf2);   // 64x64 and 128x32 requests
}

// fp32 tensors, split-bf16 arithmetic: instantiated per GEMM role in dj_conv_x*.hip
template <int AM, int BMD>
int dj_launch_split(int cfg, const DjIgemmParams& p, int splits, hipStream_t s, int fast, int mode);
#ifdef DJ_SPLIT_UNIT
template <int AM, int BMD>
int dj_launch_split(int cfg, const DjIgemmParams& p, int splits, hipStream_t s, int fast, int mode) {
  if (mode == 4) return launch_lowp_cfg<AM, BMD, 4, 0, 0>(cfg, p, splits, s, fast);
  return launch_lowp_cfg<AM, BMD, 3, 0, 0>(cfg, p, splits, s, fast);
}
#endif

// One instantiation per (GEMM role, storage types of A and B), each in a translation unit of its own (dj_conv_h*.hip).
// Storage types that exist: activations fp16 (A of the forward GEMM and of the weight gradient), gradients bf16 (A of the
// input gradients, B of the weight gradient); with 16-bit operands only the mixed mode 1 (fp16 forward / bf16 gradients).
template <int AM, int BMD, int AT, int BT>
int dj_launch_lowp_io(int cfg, const DjIgemmParams& p, int splits, hipStream_t s, int fast, int mode) {
  // A-mode 0 with B-mode 0 is the forward GEMM; everything else carries gradients (bf16: they need the exponent range)
  if (mode >= 3) {   // float32x3 / float32x6: fp32 tensors, products as three / six bf16 MFMAs (units dj_conv_x*.hip)
    if constexpr (AT == 0 && BT == 0) {
      return dj_launch_split<AM, BMD>(cfg, p, splits, s, fast, mode);
    } else {
      dj_set_error("arithmetic modes 3 / 4 (float32x3 / float32x6) work on fp32 tensors");
      return DJ_ERR_ARG;
    }
  }
  if constexpr (AM == 0 && BMD == 0) {   // (fp16 variants are instantiated for the forward GEMM only)
    if (mode == 1) return launch_lowp_cfg<AM, BMD, 1, AT, BT>(cfg, p, splits, s, fast);
  }
  if constexpr (AM == 0 && BMD == 0 && (AT != 0 || BT != 0)) {
    dj_set_error("16-bit activations / weight shadows in HBM need arithmetic mode 1 (float16)");
    return DJ_ERR_ARG;
  } else {
    return launch_lowp_cfg<AM, BMD, 2, AT, BT>(cfg, p, splits, s, fast);
  }
}
