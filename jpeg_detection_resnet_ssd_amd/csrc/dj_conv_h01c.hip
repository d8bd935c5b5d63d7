// Reduced-precision kernel instantiations: strided 1x1 input gradient, bf16 dy, bf16 weight shadow.
#include "dj_conv_launch_h16.h"

template int dj_launch_lowp_io<0, 1, 2, 2>(int, const DjIgemmParams&, int, hipStream_t, int, int);
