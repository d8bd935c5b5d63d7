// Split-bf16 fp32 arithmetic (float32x3 / float32x6, dj_igemm_h16.h PREC 3 / 4) of the GEMM role <A-mode 0, B-mode 0>.
#define DJ_SPLIT_UNIT 1
#include "dj_conv_launch_h16.h"

template int dj_launch_split<0, 0>(int, const DjIgemmParams&, int, hipStream_t, int, int);
