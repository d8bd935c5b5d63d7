// Reduced-precision kernel instantiations of the strided 1x1 input gradient (compact GEMM).
#include "dj_conv_launch_h16.h"

template int dj_launch_lowp<0, 1>(int, const DjIgemmParams&, int, hipStream_t, int, int);
