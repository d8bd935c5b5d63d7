// Reduced-precision convolution launches: (GEMM role, storage types of the operands in HBM) -> the translation unit that
// holds those kernels (dj_conv_h*.hip, one per combination so that they compile side by side).
#include "dj_conv_launch.h"

template <int AM, int BMD, int AT, int BT>
int dj_launch_lowp_io(int cfg, const DjIgemmParams& p, int splits, hipStream_t s, int fast, int mode);   // dj_conv_launch_h16.h

template <int AM, int BMD>
int dj_launch_lowp(int cfg, const DjIgemmParams& p, int splits, hipStream_t s, int fast, int mode) {
  const int at = p.a_dt, bt = p.b_dt;
  if constexpr (AM == 0 && BMD == 0) {          // forward: x fp32 | fp16, weights fp32 | fp16 shadow
    if (at == 0 && bt == 0) return dj_launch_lowp_io<0, 0, 0, 0>(cfg, p, splits, s, fast, mode);
    if (at == 1 && bt == 0) return dj_launch_lowp_io<0, 0, 1, 0>(cfg, p, splits, s, fast, mode);
    if (at == 0 && bt == 1) return dj_launch_lowp_io<0, 0, 0, 1>(cfg, p, splits, s, fast, mode);
    if (at == 1 && bt == 1) return dj_launch_lowp_io<0, 0, 1, 1>(cfg, p, splits, s, fast, mode);
  } else if constexpr (AM == 2) {               // weight gradient: x fp32 | fp16, dy fp32 | bf16
    if (at == 0 && bt == 0) return dj_launch_lowp_io<2, 0, 0, 0>(cfg, p, splits, s, fast, mode);
    if (at == 1 && bt == 0) return dj_launch_lowp_io<2, 0, 1, 0>(cfg, p, splits, s, fast, mode);
    if (at == 0 && bt == 2) return dj_launch_lowp_io<2, 0, 0, 2>(cfg, p, splits, s, fast, mode);
    if (at == 1 && bt == 2) return dj_launch_lowp_io<2, 0, 1, 2>(cfg, p, splits, s, fast, mode);
  } else {                                      // input gradients (plain and strided 1x1): dy fp32 | bf16, weights fp32 | bf16 shadow
    if (at == 0 && bt == 0) return dj_launch_lowp_io<AM, BMD, 0, 0>(cfg, p, splits, s, fast, mode);
    if (at == 2 && bt == 0) return dj_launch_lowp_io<AM, BMD, 2, 0>(cfg, p, splits, s, fast, mode);
    if (at == 0 && bt == 2) return dj_launch_lowp_io<AM, BMD, 0, 2>(cfg, p, splits, s, fast, mode);
    if (at == 2 && bt == 2) return dj_launch_lowp_io<AM, BMD, 2, 2>(cfg, p, splits, s, fast, mode);
  }
  dj_set_error("no reduced-precision kernel for operand storage types A=%d B=%d in this GEMM role (activations are held as "
               "fp16, gradients as bf16, weights as fp32 or as the fp16 / bf16 shadow of their direction)", at, bt);
  return DJ_ERR_ARG;
}

template int dj_launch_lowp<0, 0>(int, const DjIgemmParams&, int, hipStream_t, int, int);
template int dj_launch_lowp<0, 1>(int, const DjIgemmParams&, int, hipStream_t, int, int);
template int dj_launch_lowp<1, 1>(int, const DjIgemmParams&, int, hipStream_t, int, int);
template int dj_launch_lowp<2, 0>(int, const DjIgemmParams&, int, hipStream_t, int, int);
