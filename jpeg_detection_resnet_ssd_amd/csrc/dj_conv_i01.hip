// Kernel instantiations of the stride-2 1x1 input gradient run as a compact GEMM.
#include "dj_conv_launch.h"

template int dj_launch_cfg<0, 1>(int, const DjIgemmParams&, int, hipStream_t);
