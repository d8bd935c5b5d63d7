// Reduced-precision kernel instantiations: forward GEMM, fp16 activations in HBM.
#include "dj_conv_launch_h16.h"

template int dj_launch_lowp_io<0, 0, 1, 0>(int, const DjIgemmParams&, int, hipStream_t, int, int);
