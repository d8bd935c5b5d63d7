// Kernel instantiations of the forward GEMM (A k-contiguous gather of x, B = W as [K][N]).
#include "dj_conv_launch.h"

template int dj_launch_cfg<0, 0>(int, const DjIgemmParams&, int, hipStream_t);
