// Reduced-precision kernel instantiations of the weight-gradient GEMM.
#include "dj_conv_launch_h16.h"

template int dj_launch_lowp<2, 0>(int, const DjIgemmParams&, int, hipStream_t, int, int);
