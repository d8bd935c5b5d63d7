// Kernel instantiations of the input-gradient GEMM (A gather of dy, B = W^T per tap).
#include "dj_conv_launch.h"

template int dj_launch_cfg<1, 1>(int, const DjIgemmParams&, int, hipStream_t);
