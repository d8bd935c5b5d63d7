// Reduced-precision kernel instantiations: weight-gradient GEMM, fp16 x, fp32 dy.
#include "dj_conv_launch_h16.h"

template int dj_launch_lowp_io<2, 0, 1, 0>(int, const DjIgemmParams&, int, hipStream_t, int, int);
