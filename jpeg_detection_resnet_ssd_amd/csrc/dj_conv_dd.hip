// Instantiations and launcher of the LDS-free input-gradient kernel (dj_dgrad_direct.h).
#include "dj_conv_launch.h"
#include "dj_dgrad_direct.h"

bool dj_dgrad_direct_ok(const DjIgemmParams& p) {
  return p.vecA && p.vecB && p.srcC % 32 == 0 && p.sH == 1 && p.sW == 1 && p.a_bytes > 0 && p.b_bytes > 0 &&
         p.bias == nullptr && p.cmap == 0 && p.atomic == 0 && p.pro_scale == nullptr;
}

template <int TM, int TN>
static int launch_dd(const DjIgemmParams& p, hipStream_t s) {
  const int tiles = dj_cdiv(p.M, 32 * TM) * dj_cdiv(p.N, 32 * TN);
  hipLaunchKernelGGL((dj_dgrad_direct_kernel<TM, TN, 2>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, s, p);
  DJ_CHECK_LAUNCH("dj_dgrad_direct_kernel");
  return DJ_OK;
}

int dj_launch_dgrad_direct(int cfg, const DjIgemmParams& p, hipStream_t s) {
  switch (cfg) {
    case CFG_DD_2x2: return launch_dd<2, 2>(p, s);
    case CFG_DD_2x4: return launch_dd<2, 4>(p, s);
    case CFG_DD_4x2: return launch_dd<4, 2>(p, s);
    case CFG_DD_4x4: return launch_dd<4, 4>(p, s);
    case CFG_DD_1x2: return launch_dd<1, 2>(p, s);
  }
  dj_set_error("bad input-gradient variant %d", cfg);
  return DJ_ERR_ARG;
}
