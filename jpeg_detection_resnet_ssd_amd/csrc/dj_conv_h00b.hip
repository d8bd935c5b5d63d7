// Reduced-precision kernel instantiations: forward GEMM, fp32 activations, fp16 weight shadow.
#include "dj_conv_launch_h16.h"

template int dj_launch_lowp_io<0, 0, 0, 1>(int, const DjIgemmParams&, int, hipStream_t, int, int);
