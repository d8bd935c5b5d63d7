// Instantiations and launcher of the LDS-free weight-gradient kernel (dj_wgrad_direct.h).
#include "dj_conv_launch.h"
#include "dj_wgrad_direct.h"
#include <stdlib.h>

bool dj_wgrad_direct_ok(const DjIgemmParams& p) {
  // x rows are read with 16-byte (8-byte) buffer loads at channel offsets that are multiples of TM; dy with dword loads
  // (any out_c, any ld_y); 32-bit byte offsets; the pixel counter advances by two per step with at most one row wrap
  return p.vecA && p.srcC % 4 == 0 && p.ldsrc % 4 == 0 && p.a_bytes > 0 && p.b_bytes > 0 && p.rowW >= 2 &&
         p.K < (1 << 24) && p.M % 4 == 0;
}

// everything about a pixel pair is wave-uniform and affine in the pair index (see dj_wgrad_direct_lin_kernel)
template <int TM>
static bool wd_linear(const DjIgemmParams& p) {
  return p.sH == 1 && p.sW == 1 && p.rowH == p.srcH && p.rowW == p.srcW && p.srcC % (32 * TM) == 0 &&
         getenv("DJ_WD_GENERIC") == nullptr;
}

template <int TM, int TN, int U>
static int launch_wd(const DjIgemmParams& p, int splits, hipStream_t s) {
  const int tiles = dj_cdiv(p.M, 32 * TM) * dj_cdiv(p.N, 32 * TN);
  const int groups = (tiles + 3) / 4;
  dim3 grid((unsigned)(groups * splits));
  if (wd_linear<TM>(p)) {
    const bool pad = !(p.KH == 1 && p.KW == 1 && p.pT == 0 && p.pL == 0);
    if (p.pro_scale) {
      if (pad)
        hipLaunchKernelGGL((dj_wgrad_direct_lin_kernel<TM, TN, 1, 1, U>), grid, dim3(256), 0, s, p);
      else
        hipLaunchKernelGGL((dj_wgrad_direct_lin_kernel<TM, TN, 1, 0, U>), grid, dim3(256), 0, s, p);
    } else {
      if (pad)
        hipLaunchKernelGGL((dj_wgrad_direct_lin_kernel<TM, TN, 0, 1, U>), grid, dim3(256), 0, s, p);
      else
        hipLaunchKernelGGL((dj_wgrad_direct_lin_kernel<TM, TN, 0, 0, U>), grid, dim3(256), 0, s, p);
    }
    DJ_CHECK_LAUNCH("dj_wgrad_direct_lin_kernel");
    return DJ_OK;
  }
  if (p.pro_scale)
    hipLaunchKernelGGL((dj_wgrad_direct_kernel<TM, TN, 1, U>), grid, dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL((dj_wgrad_direct_kernel<TM, TN, 0, U>), grid, dim3(256), 0, s, p);
  DJ_CHECK_LAUNCH("dj_wgrad_direct_kernel");
  return DJ_OK;
}

int dj_launch_wgrad_direct(int cfg, const DjIgemmParams& p, int splits, hipStream_t s) {
  switch (cfg) {
    case CFG_WD_4x2: return launch_wd<4, 2, 4>(p, splits, s);
    case CFG_WD_4x4: return launch_wd<4, 4, 4>(p, splits, s);
    case CFG_WD_2x2: return launch_wd<2, 2, 4>(p, splits, s);
    case CFG_WD_2x4: return launch_wd<2, 4, 4>(p, splits, s);
    case CFG_WD_4x1: return launch_wd<4, 1, 4>(p, splits, s);
  }
  dj_set_error("bad weight-gradient variant %d", cfg);
  return DJ_ERR_ARG;
}
