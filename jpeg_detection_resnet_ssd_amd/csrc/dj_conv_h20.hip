// Reduced-precision kernel instantiations: weight-gradient GEMM, fp32 x and dy.
#include "dj_conv_launch_h16.h"

template int dj_launch_lowp_io<2, 0, 0, 0>(int, const DjIgemmParams&, int, hipStream_t, int, int);
