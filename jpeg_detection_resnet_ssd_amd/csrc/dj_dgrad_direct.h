// Input-gradient GEMM without LDS (stride-1 convolutions): dx[(img,ih,iw)][ci] = sum over taps and co of
// dy[img, ih + pad - kh*d, iw + pad - kw*d][co] * W[kh][kw][ci][co]   (TF Conv2DBackpropInput; reference call sites: the
// Conv2D layers of localisation_part/models/keras_ssd300_dct_j2d_resnet.py:77-96,128-160,483-545,562-675).
//
// Both operands are k-contiguous in HBM (dy rows over co; HWIO weights read as W[tap][ci][co], i.e. rows over co), which
// is exactly the order v_mfma_f32_32x32x2_f32 wants them in: lane (row or column i = lane % 32, half h = lane / 32) loads
// ONE 16-byte piece [k0 + 4h, k0 + 4h + 4) of its own row and owns the operands of four consecutive MFMAs.  So the
// tiles are loaded straight into registers -- no LDS staging, no barrier, no shared state between the waves of a
// workgroup -- and the inner loop over the channels of one filter tap has NO vector-ALU instruction at all: the per-lane
// byte offset (pixel row incl. the tap shift, or an out-of-range value for padding) is a loop constant, the channel
// position goes through the scalar offset operand and the immediate.  On gfx950 that matters more than the extra L1
// traffic of row-gathering loads: the fp32 MFMA and the vector ALU do not overlap (tools/micro/mfma_valu.hip), every
// VALU instruction of a wave is ~5 cycles added to the 64 of an MFMA, and a barrier-synchronised LDS pipeline stalls
// whenever fewer than ~4 workgroups share a CU (the 19x19 / 10x10 layers at batch 32).
// Preconditions (host-checked): stride 1, out_c % 32 == 0, 16-byte aligned dy / W, ld_y % 4 == 0, no split-K, no bias.
#pragma once
#include "dj_igemm.h"
#include "dj_igemm_fast.h"

// out-of-range marker that stays out of range when an immediate / scalar offset of a few KB is added to it (operands are
// smaller than 2 GiB); DJ_OOB (0xFFFFFFF0) would wrap around
#define DJ_OOB_MID 0x80000000u

// TM x TN 32x32 accumulators per wave (rows = input pixels, columns = input channels); U = k groups (8 channels each)
// per software-pipeline stage, two stages in flight.
template <int TM, int TN, int U>
__global__ __launch_bounds__(256, (TM * TN > 8) ? 1 : ((TM * TN > 4) ? 2 : 3)) void dj_dgrad_direct_kernel(const DjIgemmParams p) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, lh = lane >> 5;

  const int tiles_m = (p.M + 32 * TM - 1) / (32 * TM), tiles_n = (p.N + 32 * TN - 1) / (32 * TN);
  int logical;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, slot = b >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int wt = logical * 4 + wave;          // the four waves of a workgroup: adjacent column tiles of the same rows
  if (wt >= tiles_m * tiles_n) return;        // no barrier anywhere
  const int tile_m = wt / tiles_n, tile_n = wt - tile_m * tiles_n;
  const int m0 = tile_m * 32 * TM, n0 = tile_n * 32 * TN;

  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.b_bytes, 0x00020000);
  const int ldy4 = p.ldsrc * 4;

  // this lane's rows (input pixels) and columns (input channels)
  int r_base[TM], r_h[TM], r_w[TM];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int m = m0 + 32 * t + l31;
    if (m < p.M) {
      const int img = m / (p.rowH * p.rowW);
      const int rem = m - img * (p.rowH * p.rowW);
      const int h = rem / p.rowW;
      r_base[t] = img * p.srcH * p.srcW;
      r_h[t] = h + p.pT;
      r_w[t] = rem - h * p.rowW + p.pL;
    } else {
      r_base[t] = 0;
      r_h[t] = -(1 << 28);     // never inside the dy grid
      r_w[t] = -(1 << 28);
    }
  }
  unsigned voffB[TN];
#pragma unroll
  for (int u = 0; u < TN; ++u) {
    const int n = n0 + 32 * u + l31;
    voffB[u] = (n < p.N) ? (unsigned)(n * p.ldb * 4 + lh * 16) : DJ_OOB_MID;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int u = 0; u < TN; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

  struct Stage {
    f32x4 a[U][TM], b[U][TN];
  };
  Stage s0, s1;
  unsigned voffA[TM];

  // channel position c (multiple of 8) of the stage's first group: scalar offset; the groups of a stage: immediates
  auto issue = [&](Stage& S, int soffA, int soffB) {
#pragma unroll
    for (int j = 0; j < U; ++j) {
#pragma unroll
      for (int t = 0; t < TM; ++t)
        S.a[j][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, (int)voffA[t] + 32 * j, soffA, 0));
#pragma unroll
      for (int u = 0; u < TN; ++u)
        S.b[j][u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rB, (int)voffB[u] + 32 * j, soffB, 0));
    }
  };
  auto compute = [&](const Stage& S) {
#pragma unroll
    for (int j = 0; j < U; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
          for (int u = 0; u < TN; ++u)
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(S.a[j][t][e], S.b[j][u][e], acc[t][u], 0, 0, 0);
  };

  const int cbytes = p.srcC * 4;              // bytes of one tap's channel range in a dy row / a W row
  const int step = 2 * U * 32;                // bytes of channels one loop iteration consumes (two stages)
  int kh = 0, kw = 0;
  for (int tap = 0; tap < p.KH * p.KW; ++tap) {
    // rows under this tap: dy pixel (ih + pad - kh*d, iw + pad - kw*d) or, outside the dy grid, an out-of-range offset
    const int dh = kh * p.dH, dw = kw * p.dW;
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int oh = r_h[t] - dh, ow = r_w[t] - dw;
      const bool ok = ((unsigned)oh < (unsigned)p.srcH) & ((unsigned)ow < (unsigned)p.srcW);
      voffA[t] = ok ? (unsigned)((r_base[t] + oh * p.srcW + ow) * ldy4 + lh * 16) : DJ_OOB_MID;
    }
    const int tapB = (int)(tap * p.bTapStride * 4);
    const int last = cbytes - U * 32;         // scalar offset of the tap's last stage: over-issued prefetches re-read it
    issue(s0, 0, tapB);
    for (int c = 0; c < cbytes; c += step) {
      issue(s1, c + U * 32, tapB + c + U * 32);
      compute(s0);
      const int cn = min(c + step, last);
      issue(s0, cn, tapB + cn);
      compute(s1);
    }
    kw += 1;
    if (kw == p.KW) {
      kw = 0;
      kh += 1;
    }
  }

  // epilogue: register r of tile (t, u) is row (r & 3) + 8 (r >> 2) + 4 h, column lane % 32
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m >= p.M) continue;
      float* row = p.C + (size_t)m * p.ldc;
#pragma unroll
      for (int u = 0; u < TN; ++u) {
        const int n = n0 + 32 * u + l31;
        if (n >= p.N) continue;
        float v = acc[t][u][r];
        if (p.beta) v += row[n];
        row[n] = v;
      }
    }
}
