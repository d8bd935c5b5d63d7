// Reduced-precision kernel instantiations: strided 1x1 input gradient (compact GEMM + scatter), bf16 dy in HBM.
#include "dj_conv_launch_h16.h"

template int dj_launch_lowp_io<0, 1, 2, 0>(int, const DjIgemmParams&, int, hipStream_t, int, int);
