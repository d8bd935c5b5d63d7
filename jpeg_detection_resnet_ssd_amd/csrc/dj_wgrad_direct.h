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


// ---------------------------------------------------------------------------------------------------------------
// The same GEMM for the geometries where everything about a pixel pair is WAVE-UNIFORM and affine in the pair index:
// stride 1 with out size == in size (every 1x1 and every 'same' k x k convolution of the graphs) and one filter tap
// per wave tile (in_c % (32*TM) == 0).  Measured on gfx950 (tools/micro/mfma_valu.hip): v_mfma_f32_32x32x2_f32 and the
// vector ALU do not overlap -- every VALU instruction of a wave adds its ~5 issue cycles to the 64 of an MFMA -- while
// scalar instructions and loads are free.  So this variant spends its per-pair work on the scalar unit:
//   * x / dy byte offsets advance by one v_add each per pair (the per-lane parts are loop constants; the column tiles
//     of dy use the instruction's immediate offset);
//   * the pixel coordinates of the pair are tracked in SGPRs; for padded taps the validity of the two pixels becomes a
//     64-bit lane mask (low half = first pixel) that one v_cndmask applies to the x offset;
//   * the BatchNormalization(+ReLU) prologue is the only other vector work: 2 (+1 when padded) VALU per x element.
// PAD = 0: no tap of the tile can leave the image (1x1 convolutions): no validity at all.
// ---------------------------------------------------------------------------------------------------------------
template <int TM, int TN, int PRO, int PAD, int U>
__global__ __launch_bounds__(256, (TM * TN > 8) ? 1 : 2) void dj_wgrad_direct_lin_kernel(const DjIgemmParams p) {
  typedef typename DjVecOf<TM>::type avec;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably uniform from here on
  const int l31 = lane & 31, lh = lane >> 5;

  const int tiles_m = (p.M + 32 * TM - 1) / (32 * TM), tiles_n = (p.N + 32 * TN - 1) / (32 * TN);
  const int groups = (tiles_m * tiles_n + 3) >> 2;
  int logical;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, slot = b >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int chunk = logical / groups;
  const int wt = (logical - chunk * groups) * 4 + wave;
  if (wt >= tiles_m * tiles_n) return;
  const int tile_m = wt / tiles_n, tile_n = wt - tile_m * tiles_n;
  const int kbeg = chunk * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const int npairs = (kend - kbeg + 1) >> 1;

  // one tap per wave tile
  const int m0 = tile_m * 32 * TM;
  const int tap = m0 / p.srcC;
  const int c0 = m0 - tap * p.srcC;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int dh = kh * p.dH - p.pT, dw = kw * p.dW - p.pL;
  const int ldx4 = p.ldsrc * 4, ldy4 = p.ldb * 4;

  // x: the whole tensor; a lane's byte offset (first pixel of the chunk + tap shift + channel) may be negative for taps
  // above the first image row: as an unsigned offset that is out of range, which is what such a pixel must read anyway.
  // dy: based at this chunk's first pixel and ending with its last one, so pairs past the chunk read zeros.  Range
  // checks see the VGPR offset + immediate only: nothing per-pair is passed through the scalar offset operand.
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.B + (long)kbeg * ldy4), 0,
                                                                      (int)((long)(kend - kbeg) * ldy4), 0x00020000);

  const int c_lane = c0 + TM * l31;
  const bool a_ok = m0 + TM * l31 < p.M;
  // (the x offset of a lane that has no row stays out of range for the whole chunk: 2^31 + anything a chunk adds)
  // signed arithmetic (no wrap-around the compiler must respect): constant parts fold into the immediate offsets
  int offA = a_ok ? (kbeg + lh + dh * p.srcW + dw) * ldx4 + c_lane * 4 : (int)0x80000000;
  int offB = lh * ldy4 + (tile_n * 32 * TN + l31) * 4;
  const int stepA = 2 * ldx4, stepB = 2 * ldy4;
  avec sc, sh;
  if (PRO) {
    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)p.pro_scale, 0, p.srcC * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc((void*)p.pro_shift, 0, p.srcC * 4, 0x00020000);
    sc = dj_buf_ldv<TM>(rS, a_ok ? (unsigned)c_lane * 4u : DJ_WD_OOB);
    sh = dj_buf_ldv<TM>(rT, a_ok ? (unsigned)c_lane * 4u : DJ_WD_OOB);
  }
  const float relu_floor = p.pro_relu ? 0.f : -INFINITY;

  // scalar pixel state of the pair's FIRST pixel
  int s_pix = kbeg, s_oh, s_ow;
  {
    const int hw = p.rowH * p.rowW;
    const int img = s_pix / hw;
    const int rem = s_pix - img * hw;
    s_oh = rem / p.rowW;
    s_ow = rem - s_oh * p.rowW;
  }

  struct Stage {
    avec a[U];
    float b[U][TN];
    unsigned ok0, ok1;   // bit j: first / second pixel of pair j lies inside the image under this tile's tap (PAD)
  };
  Stage s0, s1;

  auto lane_mask = [](bool ok0, bool ok1) -> unsigned long long {
    return (ok0 ? 0x00000000FFFFFFFFull : 0ull) | (ok1 ? 0xFFFFFFFF00000000ull : 0ull);
  };
  auto issue = [&](Stage& S) {
    S.ok0 = S.ok1 = 0;
#pragma unroll
    for (int j = 0; j < U; ++j) {
      int oa = offA;
      if (PAD) {
        // second pixel of the pair: one step right, wrapping to the next row (out_w >= 2); selects, not branches: the
        // loop body must stay one basic block for the MFMA / load pipeline
        const int w1 = (s_ow + 1 >= p.rowW) ? 1 : 0;
        const int ow1 = w1 ? 0 : s_ow + 1;
        const int oh1 = (s_oh + w1 >= p.rowH) ? 0 : s_oh + w1;
        const bool ok0 = ((unsigned)(s_oh + dh) < (unsigned)p.srcH) & ((unsigned)(s_ow + dw) < (unsigned)p.srcW) & (s_pix < kend);
        const bool ok1 = ((unsigned)(oh1 + dh) < (unsigned)p.srcH) & ((unsigned)(ow1 + dw) < (unsigned)p.srcW) & (s_pix + 1 < kend);
        S.ok0 |= ok0 ? (1u << j) : 0u;
        S.ok1 |= ok1 ? (1u << j) : 0u;
        asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(oa) : "v"((int)DJ_WD_OOB), "v"(offA), "s"(lane_mask(ok0, ok1)));
        const int w2 = (s_ow + 2 >= p.rowW) ? 1 : 0;
        s_ow = s_ow + 2 - (w2 ? p.rowW : 0);
        s_oh = (s_oh + w2 >= p.rowH) ? 0 : s_oh + w2;
      }
      S.a[j] = dj_buf_ldv<TM>(rA, (unsigned)oa);
#pragma unroll
      for (int u = 0; u < TN; ++u) S.b[j][u] = dj_buf_ld1(rB, (unsigned)(offB + 128 * u));
      // one vector add per operand and pair, kept opaque so that the compiler does not trade it for an induction
      // variable per load (each of those is another VALU instruction per pair)
      asm volatile("v_add_u32 %0, %1, %0" : "+v"(offA) : "s"(stepA));
      asm volatile("v_add_u32 %0, %1, %0" : "+v"(offB) : "s"(stepB));
      s_pix += 2;
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
    // (the validity bits are wave-uniform; said explicitly, or they travel through the loop in vector registers)
    const unsigned k0 = PAD ? __builtin_amdgcn_readfirstlane(S.ok0) : 0u, k1 = PAD ? __builtin_amdgcn_readfirstlane(S.ok1) : 0u;
#pragma unroll
    for (int j = 0; j < U; ++j) {
      avec a = S.a[j];
      if (PRO) {
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          float v = fmaxf(a[t] * sc[t] + sh[t], relu_floor);
          if (PAD) asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(v) : "v"(v), "s"(lane_mask((k0 >> j) & 1u, (k1 >> j) & 1u)));
          a[t] = v;
        }
      }
#pragma unroll
      for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], S.b[j][u], acc[t][u], 0, 0, 0);
    }
  };

  issue(s0);
  for (int it = 0; it < npairs; it += 2 * U) {
    issue(s1);
    compute(s0);
    issue(s0);
    compute(s1);
  }

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
