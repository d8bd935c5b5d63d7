// Kernel instantiations of the weight-gradient GEMM (A m-contiguous, K' = pixels).
#include "dj_conv_launch.h"

template int dj_launch_cfg<2, 0>(int, const DjIgemmParams&, int, hipStream_t);
