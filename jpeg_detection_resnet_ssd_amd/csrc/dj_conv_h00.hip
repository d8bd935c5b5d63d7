// Reduced-precision kernel instantiations of the forward GEMM.
#include "dj_conv_launch_h16.h"

template int dj_launch_lowp<0, 0>(int, const DjIgemmParams&, int, hipStream_t, int, int);
