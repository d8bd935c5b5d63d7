// Reduced-precision kernel instantiations: input-gradient GEMM, fp32 dy, bf16 weight shadow.
#include "dj_conv_launch_h16.h"

template int dj_launch_lowp_io<1, 1, 0, 2>(int, const DjIgemmParams&, int, hipStream_t, int, int);
