// Weight-gradient GEMM without LDS: dW[(kh,kw,ci)][co] = sum over output pixels p of x'[p + tap][ci] * dy[p][co]
// (keras.layers.Conv2D kernel gradient; reference call sites: every Conv2D of
// localisation_part/models/keras_ssd300_dct_j2d_resnet.py:77-96,128-160,483-545,562-675).
//
// Both operands of this GEMM are PIXEL-major in HBM (x: [pixel][ci], dy: [pixel][co]) and the reduction runs over
// pixels, i.e. for one pixel the 32 rows (ci) and the 32 columns (co) an MFMA wants sit next to each other in memory.
// v_mfma_f32_32x32x2_f32 takes A[i][k] from lane (i = lane % 32, k = lane / 32) and B[k][j] from lane (k, j), so a lane
// half owns one pixel and
//   * ONE 16-byte load per lane, x[p][m0 + 4*i .. +3], is the A operand of FOUR row tiles (tile t = rows m0 + 4*i + t:
//     an interleaved row set is as good a GEMM tile as a contiguous one), 512 contiguous bytes per lane half;
//   * the B operand of column tile u is one dword load dy[p][n0 + 32*u + j], 128 contiguous bytes per lane half
//     (columns stay contiguous per tile so that each accumulator register is written / added as two 128-byte rows);
// -> per pixel pair 1 + TN loads feed 4*TN MFMAs straight from registers: no LDS staging, no transposition, no barrier,
// and each wave owns its tile and its share of the pixels outright (nothing to share, hence nothing to synchronise).
// The generic kernel (dj_igemm_fast.h, A-mode 2) needs 16 ds_read_b32 per 16 MFMAs for the same operands.
//
// Grid: one wave per (TM*32) x (TN*32) tile of dW and per pixel chunk; the four waves of a workgroup take adjacent
// column tiles (same x rows: L1 hits).  Workgroups are renumbered so that every XCD owns whole pixel chunks: the x / dy
// rows of a chunk are read from HBM once into that XCD's L2 and re-read by its other tiles from there.
// Preconditions (host-checked): in_c % 4 == 0, ld_x % 4 == 0, x 16-byte aligned, out_w >= 2, operands < 2 GiB.
#pragma once
#include "dj_igemm.h"

typedef float dj_f32x2 __attribute__((ext_vector_type(2)));

template <int TM>
struct DjVecOf;
template <>
struct DjVecOf<4> {
  typedef f32x4 type;
};
template <>
struct DjVecOf<2> {
  typedef dj_f32x2 type;
};

template <int TM>
__device__ __forceinline__ typename DjVecOf<TM>::type dj_buf_ldv(__amdgpu_buffer_rsrc_t r, unsigned off);
template <>
__device__ __forceinline__ f32x4 dj_buf_ldv<4>(__amdgpu_buffer_rsrc_t r, unsigned off) {
  typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}
template <>
__device__ __forceinline__ dj_f32x2 dj_buf_ldv<2>(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(dj_f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0));
}
__device__ __forceinline__ float dj_buf_ld1(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}

#define DJ_WD_OOB 0xFFFFFFF0u

// TM in {2, 4}: row tiles per wave (rows interleaved by TM), TN in {1, 2, 4}: column tiles per wave.
// PRO: 0 = x as it is, 1 = relu?(x * scale[ci] + shift[ci]) on in-bounds pixels (the BatchNormalization(+ReLU) that the
// forward pass folded into the consumer's load).  U = pixel pairs per software-pipeline stage.
template <int TM, int TN, int PRO, int U>
__global__ __launch_bounds__(256, (TM * TN > 8) ? 1 : 2) void dj_wgrad_direct_kernel(const DjIgemmParams p) {
  typedef typename DjVecOf<TM>::type avec;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lh = lane >> 5;

  // ---- which tile, which pixel chunk ----
  const int tiles_m = (p.M + 32 * TM - 1) / (32 * TM), tiles_n = (p.N + 32 * TN - 1) / (32 * TN);
  const int groups = (tiles_m * tiles_n + 3) >> 2;   // workgroups per pixel chunk
  int logical;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, slot = b >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int chunk = logical / groups;
  const int wt = (logical - chunk * groups) * 4 + wave;
  if (wt >= tiles_m * tiles_n) return;               // no barriers anywhere: a wave may leave
  const int tile_m = wt / tiles_n, tile_n = wt - tile_m * tiles_n;
  const int kbeg = chunk * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const int npairs = (kend - kbeg + 1) >> 1;

  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
  // dy rows past this chunk's last pixel are out of range of the descriptor: the hardware returns zeros for them, so
  // the pairs beyond the chunk need no test of their own (their products are x * 0)
  const long b_rows = (long)kend * p.ldb * 4;
  const __amdgpu_buffer_rsrc_t rB =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)(b_rows < (long)p.b_bytes ? b_rows : (long)p.b_bytes), 0x00020000);

  // ---- per-lane constants ----
  const int m_lane = tile_m * 32 * TM + TM * l31;   // first of this lane's TM consecutive rows (same tap: in_c % TM == 0)
  const bool a_ok = m_lane < p.M;
  const int tap = m_lane / p.srcC;
  const int c_lane = m_lane - tap * p.srcC;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int dh = kh * p.dH - p.pT, dw = kw * p.dW - p.pL;
  const int c4 = c_lane * 4, ldx4 = p.ldsrc * 4, ldy4 = p.ldb * 4;
  avec sc, sh;
  if (PRO) {
    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)p.pro_scale, 0, p.srcC * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc((void*)p.pro_shift, 0, p.srcC * 4, 0x00020000);
    sc = dj_buf_ldv<TM>(rS, a_ok ? (unsigned)c4 : DJ_WD_OOB);
    sh = dj_buf_ldv<TM>(rT, a_ok ? (unsigned)c4 : DJ_WD_OOB);
  }
  const float relu_floor = p.pro_relu ? 0.f : -INFINITY;
  // byte offset of this lane's column inside a dy row; columns past N get an offset beyond any 2 GiB operand, i.e. an
  // out-of-range (zero) load without a per-load test
  unsigned n4[TN];
#pragma unroll
  for (int u = 0; u < TN; ++u) {
    const int n = tile_n * 32 * TN + 32 * u + l31;
    n4[u] = (n < p.N) ? (unsigned)n * 4u : 0x80000000u;
  }

  // ---- this lane's pixel: p = kbeg + 2*pair + lh, decomposed once, then advanced by 2 per pair ----
  int pix = kbeg + lh;
  int img, oh, ow;
  {
    const int hw = p.rowH * p.rowW;
    img = pix / hw;
    const int rem = pix - img * hw;
    oh = rem / p.rowW;
    ow = rem - oh * p.rowW;
  }

  struct Stage {
    avec a[U];
    float b[U][TN];
    unsigned okmask;   // bit j: pair j's x pixel is in bounds (PRO needs it: the affine of a padding zero is not zero)
  };
  Stage s0, s1;

  auto issue = [&](Stage& S) {
    S.okmask = 0;
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int h = oh * p.sH + dh, w = ow * p.sW + dw;
      const bool live = pix < kend;
      const bool ok = live & a_ok & ((unsigned)h < (unsigned)p.srcH) & ((unsigned)w < (unsigned)p.srcW);
      const unsigned offA = (unsigned)(((img * p.srcH + h) * p.srcW + w) * ldx4 + c4);
      S.a[j] = dj_buf_ldv<TM>(rA, ok ? offA : DJ_WD_OOB);
      S.okmask |= ok ? (1u << j) : 0u;
      const unsigned rowB = (unsigned)(pix * ldy4);
#pragma unroll
      for (int u = 0; u < TN; ++u) S.b[j][u] = dj_buf_ld1(rB, rowB + n4[u]);
      // next pair of this lane half: two pixels further (out_w >= 2: at most one wrap each)
      pix += 2;
      ow += 2;
      const bool wrap = ow >= p.rowW;
      ow -= wrap ? p.rowW : 0;
      oh += wrap ? 1 : 0;
      const bool wrap2 = oh >= p.rowH;
      oh -= wrap2 ? p.rowH : 0;
      img += wrap2 ? 1 : 0;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int u = 0; u < TN; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

  auto compute = [&](Stage& S) {
#pragma unroll
    for (int j = 0; j < U; ++j) {
      avec a = S.a[j];
      if (PRO) {
        const bool ok = (S.okmask >> j) & 1u;
#pragma unroll
        for (int t = 0; t < TM; ++t) a[t] = ok ? fmaxf(a[t] * sc[t] + sh[t], relu_floor) : 0.f;
      }
#pragma unroll
      for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], S.b[j][u], acc[t][u], 0, 0, 0);
    }
  };

  // two stages in flight: the loads of stage s^1 are issued before the MFMAs of stage s (pairs past the chunk's end are
  // out-of-range loads: zeros, no memory traffic)
  issue(s0);
  for (int it = 0; it < npairs; it += 2 * U) {
    issue(s1);
    compute(s0);
    issue(s0);
    compute(s1);
  }

  // ---- epilogue: row i = (r & 3) + 8 (r >> 2) + 4 lh of tile t is dW row m0 + TM*i + t; lanes are consecutive columns ----
  const int m0 = tile_m * 32 * TM;
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int m = m0 + TM * i + t;
      if (m >= p.M) continue;
      float* row = p.C + (size_t)m * p.ldc;
#pragma unroll
      for (int u = 0; u < TN; ++u) {
        const int n = tile_n * 32 * TN + 32 * u + l31;
        if (n >= p.N) continue;
        if (p.atomic)
          unsafeAtomicAdd(row + n, acc[t][u][r]);
        else
          row[n] = acc[t][u][r];
      }
    }
}
