// Branch-free fast path of the implicit-GEMM convolution (same math and LDS layout as dj_igemm.h).
//
// What differs from the generic kernel:
//   * every global read is a raw BUFFER load (128-bit descriptor, 32-bit byte offset): halo / tail
//     elements get the offset 0xFFFFFFF0, the hardware range check returns zeros -- no branches
//     around loads, so the whole K-step is ONE basic block the scheduler can interleave with MFMAs;
//   * the tap decomposition (kh, kw, c0) of a K-step is wave-uniform (requires srcC % 32 == 0 for the
//     k-contiguous operands) and is advanced incrementally in scalar registers -- no per-step division;
//   * per-row gather offsets are precomputed once; a K-step adds one uniform delta per row;
//   * the two LDS stages are addressed with compile-time offsets (loop unrolled by 2), so the LDS
//     writes of stage s^1 can be proven independent of the fragment reads of stage s;
//   * sched_group_barrier directives spread the staging instructions over the MFMA issue slots: a
//     v_mfma_f32_32x32x2_f32 occupies the matrix pipe for 64 cycles and the wave can issue ~14 other
//     instructions in its shadow (MI355X_MICROARCH.md, per-instruction cycle constants).
//
// Preconditions (checked by the host launcher, which otherwise falls back to dj_igemm_kernel):
//   A-mode 0/1: srcC % 32 == 0, stride 1 for mode 1; A-mode 2: srcC % 4 == 0, K' < 2^24;
//   B-mode 0: N % 4 == 0, ldb % 4 == 0; B-mode 1: srcC % 32 == 0;
//   16-byte aligned bases, every operand smaller than 2 GiB.
#pragma once
#include "dj_igemm.h"
#include <type_traits>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define DJ_OOB 0xFFFFFFF0u

__device__ __forceinline__ f32x4 dj_buf_ld4(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}

typedef _Float16 dj_half4 __attribute__((ext_vector_type(4)));
typedef short dj_short4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ dj_half4 dj_to_half4(f32x4 v) {
  return dj_half4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};   // round-to-nearest-even
}
__device__ __forceinline__ dj_short4 dj_to_bf16x4(f32x4 v) {
  __bf16 a = (__bf16)v.x, b = (__bf16)v.y, c = (__bf16)v.z, d = (__bf16)v.w;       // v_cvt_pk_bf16_f32 (RNE) on gfx950
  return dj_short4{__builtin_bit_cast(short, a), __builtin_bit_cast(short, b), __builtin_bit_cast(short, c),
                   __builtin_bit_cast(short, d)};
}

// PRO: 0 = plain A, 1 = A*scale[c]+shift[c] (then max(., floor) with floor = 0 or -inf) on in-bounds elements,
//      3 = residual add: relu(A*scale+shift + A2*scale2+shift2) with a second gathered tensor A2 (1x1 stride-1 forward
//          only); the workgroups of column tile 0 also store that sum -- the Add()+Activation('relu') output -- to
//          p.sum_out, so the separate elementwise pass (read, read, write, then read again by this conv) disappears
// PREC: 0 = exact fp32 MFMA (v_mfma_f32_32x32x2_f32); 1 = operands rounded to fp16, 2 = to bf16 when the fragments
//       are read, one v_mfma_f32_32x32x8_{f16,bf16} per 8-deep k group (same lane <-> k mapping as the four fp32
//       MFMAs it replaces), fp32 accumulation.  HBM and LDS contents stay fp32 ("fp32 master" tensors).
// KS: 1, or 2 = two 256-thread groups per workgroup, each with its own LDS ring, take alternate K-steps of the SAME
//       output tile and add their accumulators through LDS before the epilogue: twice the waves per SIMD for launches
//       that have too few tiles to fill the CUs (19x19 and 10x10 maps at batch 32), no atomics, BN statistics intact.
// NP: 1 = the gathered operand can never leave its tensor (1x1 kernel, no padding; A-modes 0/1, PRO 0/1, KS 1): the row
//       validity is a per-row constant, so the K-step spends no vector instruction on coordinates, bounds tests and
//       offset selects -- on gfx950 the fp32 MFMA and the vector ALU do not overlap (tools/micro/mfma_valu.hip: every
//       VALU instruction adds its ~5 cycles to the 64 of a v_mfma_f32_32x32x2_f32), and two thirds of the convolutions
//       of a bottleneck block are 1x1.  Rows past M carry an offset that stays out of range whatever is added to it.
//       Weight gradient (A-mode 2, stride 1): the input pixel IS the output pixel, so a thread's x row advances by a
//       constant per K-step -- no pixel decomposition, no coordinate tests (pixels past the chunk meet zero dy rows).
//   2 = kernels with taps / padding (A-modes 0/1, PRO 0/1, KS 1): a row's offset under the CURRENT filter tap (or an
//       out-of-range marker where the tap leaves the image) is kept in a register and recomputed only when the prefetch
//       moves on to the next tap -- a wave-uniform branch taken once per in_c / 32 K-steps; the other K-steps add the
//       channel position to it and nothing else.
//   3 = weight gradient with taps / padding / stride (A-mode 2) when in_c is a multiple of BM: the tile's BM rows lie
//       under ONE filter tap, and a thread's pixel moves by the same number of pixels every K-step.  The pixel is
//       decomposed once; afterwards (ow, oh) and the byte offset of the tapped input pixel advance by wave-uniform
//       constants with two carries (row end, image end) -- ~15 vector instructions per K-step where the float-reciprocal
//       decomposition, the per-chunk tap arithmetic and its three integer multiplies took ~75 (64x64 tile: 134 -> ~75
//       per 16 MFMAs, and the multiplies are quarter rate).
// EPI: 1 = the epilogue also takes BatchNormalization backward statistics (DjIgemmParams::bnb_z; input-gradient GEMM only).
//      An instantiation of its own: that code needs registers (eight z values in flight beside the accumulators) and the
//      launches that do not ask for it keep their occupancy.
template <int BM, int BN, int WM, int WN, int AM, int BMD, int PRO, int NSTAGE = 2, int PREC = 0, int KS = 1, int NP = 0,
          int EPI = 0>
__global__ __launch_bounds__(256 * KS) void dj_igemm_fast_kernel(const DjIgemmParams p) {
  static_assert(EPI == 0 || (AM == 1 && BMD == 1), "BatchNormalization backward statistics: input-gradient GEMM only");
  static_assert(NP == 0 || (PRO != 3 && KS == 1), "NP: no residual-add prologue, no in-workgroup K split");
  static_assert(NP != 2 || AM != 2, "NP 2 (per-tap offsets) is for the k-contiguous A operand");
  static_assert(NP != 3 || AM == 2, "NP 3 (pixel walk) is for the weight gradient's gathered operand");
  using Cfg = DjIgemmCfg<BM, BN, WM, WN, AM, BMD>;
  constexpr int TM = Cfg::TM, TN = Cfg::TN, NA = Cfg::NA, NB = Cfg::NB;
  constexpr int LDA_S = Cfg::LDA_S, LDB_S = Cfg::LDB_S;
  constexpr int STAGE = Cfg::STAGE_FLOATS, AFL = Cfg::A_FLOATS;
  extern __shared__ __attribute__((aligned(16))) float smem_base[];
  static_assert(KS == 1 || NSTAGE == 4, "the K-split groups use the pipelined two-stage schedule");

  const int tid = threadIdx.x & 255;
  const int kg = (KS > 1) ? (int)(threadIdx.x >> 8) : 0;   // K group of this thread
  float* const smem = smem_base + kg * (2 * STAGE);        // each group owns one two-stage ring
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lh = lane >> 5;

  // XCD-aware order of (tile, K chunk): dj_tile_of_workgroup (dj_igemm.h)
  const int tiles_n = (p.N + BN - 1) / BN;
  int tile_id, ky;
  dj_tile_of_workgroup(tile_id, ky);
  const int tile_m = tile_id / tiles_n;
  const int tile_n = tile_id - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kbeg0 = ky * p.kchunk;
  const int kend = min(p.K, kbeg0 + p.kchunk);
  const int nk_all = (kend - kbeg0 + DJ_BK - 1) / DJ_BK;
  // group kg takes K-steps kg, kg + KS, ...; every group runs the same number of loop iterations (barriers match),
  // iterations past a group's last tile load nothing (out-of-range offsets) and multiply zeros
  const int kbeg = kbeg0 + kg * DJ_BK;
  constexpr int KSTEP = KS * DJ_BK;
  const int nk_live = (nk_all + KS - 1 - kg) / KS;
  const int nk = (nk_all + KS - 1) / KS;

  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.b_bytes, 0x00020000);
  // PRO 1: the per-channel arrays through buffer descriptors (one VGPR offset per load).  PRO 3 has four such arrays and
  // would keep eight 4-SGPR descriptors live in the K-loop: its variants spilled 35-56 SGPRs into vector lanes
  // (v_readlane / v_writelane in the loop, 36 B of scratch in two of them); there the arrays, whose channel offsets are
  // always in range, are read by plain global loads from their uniform base pointers (2 SGPRs each): <= 10 spills, no
  // scratch.  (For PRO 1 the plain loads cost more vector instructions than they save: 26.34 -> 26.46 ms per step.)
  const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)p.pro_scale, 0, PRO == 1 ? p.srcC * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc((void*)p.pro_shift, 0, PRO == 1 ? p.srcC * 4 : 0, 0x00020000);
  const float relu_floor = (p.pro_relu || PRO == 3) ? 0.f : -INFINITY;
  const __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A2, 0, PRO == 3 ? p.a2_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc((void*)p.sum_out, 0, (PRO == 3 && p.sum_out) ? p.sum_bytes : 0, 0x00020000);
  const bool store_sum = (PRO == 3) && p.sum_out != nullptr && tile_n == 0;

  // ---------------- per-thread staging state ----------------
  const int ac = tid & 7, ar0 = tid >> 3;              // A k-contiguous: (row ar0+32j, chunk ac)
  constexpr int BKSTEP = 1024 / BN;
  const int bcn = tid % (BN / 4), bkr0 = tid / (BN / 4);  // B n-contiguous
  const int bc = tid & 7, br0 = tid >> 3;                  // B k-contiguous

  int a_off[NA], a_rh[NA], a_rw[NA];   // modes 0/1: byte offset of (row, tap 0, c 0) + chunk; row coordinates
  int a2_off[PRO == 3 ? NA : 1], y_off[PRO == 3 ? NA : 1];   // PRO 3: same pixel in A2 / sum_out
  // mode 2: thread -> (pixel row ar0 of the K-step, m' chunks ac + 8 i): ONE pixel decomposition per K-step
  int a2_c[NA], a2_dh[NA], a2_dw[NA];
  bool a2_ok[NA];
  f32x4 a2_sc[NA], a2_sh[NA];
  unsigned row_valid = 0;                 // NP: bit j = row j of this thread exists
  if (AM != 2) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      int m = m0 + ar0 + 32 * j;
      row_valid |= (m < p.M) ? (1u << j) : 0u;
      if (m < p.M) {
        int img, h, w;
        dj_row_decompose(p, m, img, h, w);
        int rh = (AM == 0) ? h * p.sH - p.pT : h + p.pT;
        int rw = (AM == 0) ? w * p.sW - p.pL : w + p.pL;
        a_rh[j] = rh;
        a_rw[j] = rw;
        a_off[j] = ((img * p.srcH * p.srcW + rh * p.srcW + rw) * p.ldsrc + 4 * ac) * 4;
        if (PRO == 3) {
          a2_off[j] = (m * p.ldsrc2 + 4 * ac) * 4;
          y_off[j] = (m * p.ld_sum + 4 * ac) * 4;
        }
      } else {
        a_rh[j] = -(1 << 28);
        a_rw[j] = -(1 << 28);
        a_off[j] = NP ? (int)0x80000000 : 0;
        if (PRO == 3) {
          a2_off[j] = 0;
          y_off[j] = 0;
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int mm = m0 + 4 * (ac + 8 * i);
      a2_ok[i] = mm < p.M;
      int tap = mm / p.srcC;
      a2_c[i] = mm - tap * p.srcC;
      int kh = tap / p.KW;
      int kw = tap - kh * p.KW;
      a2_dh[i] = kh * p.dH - p.pT;
      a2_dw[i] = kw * p.dW - p.pL;
      if (PRO) {
        a2_sc[i] = dj_buf_ld4(rS, a2_ok[i] ? (unsigned)a2_c[i] * 4u : DJ_OOB);
        a2_sh[i] = dj_buf_ld4(rT, a2_ok[i] ? (unsigned)a2_c[i] * 4u : DJ_OOB);
      }
      row_valid |= a2_ok[i] ? (1u << i) : 0u;
      if (NP == 1) a2_c[i] = a2_ok[i] ? a2_c[i] * 4 : (int)0x80000000;    // byte offset of the chunk inside its pixel
    }
  }
  // NP, A-mode 2: byte offset of this thread's pixel row (pixel kbeg + ar0 of the first K-step), advanced per K-step
  int np_row = (NP == 1 && AM == 2) ? (kbeg + ar0) * p.ldsrc * 4 : 0;
  const int np_step = DJ_BK * p.ldsrc * 4;
  // NP 3: this thread's output pixel (walk_ow, walk_oh), its index walk_kp, and the byte offset walk_off of chunk 0 at
  // the tapped input pixel; walk_c0/1/2: what a K-step, a row carry and an image carry add to that offset
  int walk_ow = 0, walk_oh = 0, walk_kp = 0, walk_off = 0;
  int walk_sw = 0, walk_sh = 0, walk_c0 = 0, walk_c1 = 0, walk_c2 = 0, walk_dh = 0, walk_dw = 0;
  if (NP == 3) {
    const int tap = m0 / p.srcC, cb = m0 - tap * p.srcC;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    walk_dh = kh * p.dH - p.pT;
    walk_dw = kw * p.dW - p.pL;
    const int hw = p.rowH * p.rowW;
    const int s_img = KSTEP / hw, r = KSTEP - s_img * hw;
    walk_sh = r / p.rowW;
    walk_sw = r - walk_sh * p.rowW;
    const int pix4 = p.ldsrc * 4;
    walk_c0 = ((s_img * p.srcH + walk_sh * p.sH) * p.srcW + walk_sw * p.sW) * pix4;
    walk_c1 = (p.sH * p.srcW - p.rowW * p.sW) * pix4;
    walk_c2 = (p.srcH - p.rowH * p.sH) * p.srcW * pix4;
    walk_kp = kbeg + ar0;
    const int img = walk_kp / hw, rem = walk_kp - img * hw;
    walk_oh = rem / p.rowW;
    walk_ow = rem - walk_oh * p.rowW;
    walk_off = ((img * p.srcH + walk_oh * p.sH + walk_dh) * p.srcW + walk_ow * p.sW + walk_dw) * pix4 + (cb + 4 * ac) * 4;
  }
  int b_off[NB];
  bool b_ok[NB];
  if (BMD == 0) {
    int n = n0 + 4 * bcn;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      b_ok[j] = n < p.N;  // N % 4 == 0: chunks are all-or-nothing
      b_off[j] = ((bkr0 + BKSTEP * j) * p.ldb + n) * 4;
    }
  } else {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      int n = n0 + br0 + 32 * j;
      b_ok[j] = n < p.N;
      b_off[j] = (n * p.ldb + 4 * bc) * 4;
    }
  }

  // wave-uniform tap state of the k-contiguous operands (A modes 0/1, B mode 1), advanced per K-step
  int t_c0 = 0, t_kh = 0, t_kw = 0, t_tap = 0;
  if (AM != 2 || BMD == 1) {
    t_tap = kbeg / p.srcC;
    t_c0 = kbeg - t_tap * p.srcC;
    t_kh = t_tap / p.KW;
    t_kw = t_tap - t_kh * p.KW;
  }

  // NP 2: per-row offsets under the tap the NEXT issue_loads will read
  int a_tap[NP == 2 ? NA : 1];
  unsigned tap_valid = 0;
  auto recompute_tap = [&]() {
    if (NP == 2) {
      const int dh = t_kh * p.dH, dw = t_kw * p.dW;
      const int tapdelta = (AM == 0) ? (dh * p.srcW + dw) * p.ldsrc * 4 : -(dh * p.srcW + dw) * p.ldsrc * 4;
      tap_valid = 0;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const int h = (AM == 0) ? a_rh[j] + dh : a_rh[j] - dh;
        const int w = (AM == 0) ? a_rw[j] + dw : a_rw[j] - dw;
        const bool ok = ((unsigned)h < (unsigned)p.srcH) & ((unsigned)w < (unsigned)p.srcW);
        a_tap[j] = ok ? a_off[j] + tapdelta : (int)0x80000000;   // stays out of range with any channel offset added
        tap_valid |= ok ? (1u << j) : 0u;
      }
    }
  };
  recompute_tap();

  // one K-step's operands in flight: registers between the buffer loads and the LDS writes
  struct Regs {
    f32x4 ra[NA], rb[NB];
    unsigned a_valid;  // bit j: row j of the prefetched A tile is in bounds (needed when PRO)
    f32x4 psc, psh;
    f32x4 ra2[PRO == 3 ? NA : 1], psc2, psh2;   // PRO 3: the residual operand and its affine
    int c0;                                       // PRO 3: first channel of this K-step (for the sum_out store)
  };
  Regs r0, r1;
  r0.a_valid = r1.a_valid = 0;
  r0.psc = r1.psc = f32x4{1.f, 1.f, 1.f, 1.f};
  r0.psh = r1.psh = f32x4{0.f, 0.f, 0.f, 0.f};

  auto issue_loads = [&](Regs& R, int kcur, bool live) {
    f32x4(&ra)[NA] = R.ra;
    f32x4(&rb)[NB] = R.rb;
    unsigned& a_valid = R.a_valid;
    f32x4& psc = R.psc;
    f32x4& psh = R.psh;
    // ---- A ----
    if (AM != 2) {
      const int dh = t_kh * p.dH, dw = t_kw * p.dW;
      const int delta = (AM == 0) ? ((dh * p.srcW + dw) * p.ldsrc + t_c0) * 4 : (-(dh * p.srcW + dw) * p.ldsrc + t_c0) * 4;
      if (PRO) {
        if (PRO == 3) {
          psc = dj_ld4(p.pro_scale + (t_c0 + 4 * ac));
          psh = dj_ld4(p.pro_shift + (t_c0 + 4 * ac));
        } else {
          psc = dj_buf_ld4(rS, (unsigned)(t_c0 + 4 * ac) * 4u);
          psh = dj_buf_ld4(rT, (unsigned)(t_c0 + 4 * ac) * 4u);
        }
      }
      if (PRO == 3) {
        R.c0 = t_c0;
        R.psc2 = f32x4{1.f, 1.f, 1.f, 1.f};
        R.psh2 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.pro_scale2) {
          R.psc2 = dj_ld4(p.pro_scale2 + (t_c0 + 4 * ac));
          R.psh2 = dj_ld4(p.pro_shift2 + (t_c0 + 4 * ac));
        }
      }
      a_valid = (NP == 1) ? row_valid : ((NP == 2) ? tap_valid : 0u);
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        if (NP == 2) {
          ra[j] = dj_buf_ld4(rA, (unsigned)(a_tap[j] + t_c0 * 4));
          continue;
        }
        if (NP) {
          // (a prefetch past the last K-step reads other, finite, data of the same tensor or nothing: never consumed)
          ra[j] = dj_buf_ld4(rA, (unsigned)(a_off[j] + delta));
          continue;
        }
        int h = (AM == 0) ? a_rh[j] + dh : a_rh[j] - dh;
        int w = (AM == 0) ? a_rw[j] + dw : a_rw[j] - dw;
        bool ok = live && (unsigned)h < (unsigned)p.srcH && (unsigned)w < (unsigned)p.srcW;
        ra[j] = dj_buf_ld4(rA, ok ? (unsigned)(a_off[j] + delta) : DJ_OOB);
        if (PRO == 3) R.ra2[j] = dj_buf_ld4(rA2, ok ? (unsigned)(a2_off[j] + t_c0 * 4) : DJ_OOB);
        a_valid |= ok ? (1u << j) : 0u;
      }
    } else if (NP == 1) {
      a_valid = row_valid;
#pragma unroll
      for (int i = 0; i < NA; ++i) ra[i] = dj_buf_ld4(rA, (unsigned)(np_row + a2_c[i]));
      np_row += np_step;
    } else if (NP == 3) {
      // 24-bit multiplies (full rate; v_mul_lo_u32 is quarter rate): coordinates and strides are far below 2^23
      const int ih = __mul24(walk_oh, p.sH) + walk_dh, iw = __mul24(walk_ow, p.sW) + walk_dw;
      const bool ok = live & (walk_kp < kend) & ((unsigned)ih < (unsigned)p.srcH) & ((unsigned)iw < (unsigned)p.srcW);
      const unsigned voff = ok ? (unsigned)walk_off : 0x80000000u;   // stays out of range with the chunk offsets added
      a_valid = ok ? 0xFFFFFFFFu : 0u;
#pragma unroll
      for (int i = 0; i < NA; ++i) ra[i] = dj_buf_ld4(rA, voff + 128u * i);
      // on to the pixel KSTEP further: s_w < W and s_h < H, so each carry is at most one
      walk_kp += KSTEP;
      walk_ow += walk_sw;
      const bool c1 = walk_ow >= p.rowW;
      walk_ow -= c1 ? p.rowW : 0;
      walk_oh += walk_sh + (c1 ? 1 : 0);
      const bool c2 = walk_oh >= p.rowH;
      walk_oh -= c2 ? p.rowH : 0;
      walk_off += walk_c0 + (c1 ? walk_c1 : 0) + (c2 ? walk_c2 : 0);
    } else {
      a_valid = 0;
      const int kp = kcur + ar0;
      const int hw = p.rowH * p.rowW;
      int img = (int)((float)kp * p.inv_rowHW);
      int rem = kp - img * hw;
      int adj = (rem < 0) ? -1 : ((rem >= hw) ? 1 : 0);  // float reciprocal is within one of the quotient
      img += adj;
      rem -= adj * hw;
      int oh = (int)((float)rem * p.inv_rowW);
      int ow = rem - oh * p.rowW;
      int adj2 = (ow < 0) ? -1 : ((ow >= p.rowW) ? 1 : 0);
      oh += adj2;
      ow -= adj2 * p.rowW;
      const bool rowok = live && kp < kend;
      const int h0 = oh * p.sH, w0 = ow * p.sW, pb = img * p.srcH * p.srcW;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        int h = h0 + a2_dh[i], w = w0 + a2_dw[i];
        bool ok = rowok && a2_ok[i] && (unsigned)h < (unsigned)p.srcH && (unsigned)w < (unsigned)p.srcW;
        unsigned off = (unsigned)((pb + h * p.srcW + w) * p.ldsrc + a2_c[i]) * 4u;
        ra[i] = dj_buf_ld4(rA, ok ? off : DJ_OOB);
        a_valid |= ok ? (1u << i) : 0u;
      }
    }
    // ---- B ----
    if (BMD == 0) {
      const int base = kcur * p.ldb * 4;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        bool ok = live && b_ok[j] && (AM != 2 || kcur + bkr0 + BKSTEP * j < kend);
        rb[j] = dj_buf_ld4(rB, ok ? (unsigned)(b_off[j] + base) : DJ_OOB);
      }
    } else {
      const unsigned base = (unsigned)(t_tap * p.bTapStride + t_c0) * 4u;
#pragma unroll
      for (int j = 0; j < NB; ++j) rb[j] = dj_buf_ld4(rB, (live && b_ok[j]) ? (unsigned)b_off[j] + base : DJ_OOB);
    }
    // advance the uniform tap state to this group's next K-step
    if (AM != 2 || BMD == 1) {
#pragma unroll
      for (int a = 0; a < KS; ++a) {
        t_c0 += DJ_BK;
        const int wrap = (t_c0 >= p.srcC) ? 1 : 0;
        t_c0 = wrap ? 0 : t_c0;
        t_tap += wrap;
        t_kw += wrap;
        const int wrap2 = (t_kw >= p.KW) ? 1 : 0;
        t_kw = wrap2 ? 0 : t_kw;
        t_kh += wrap2;
        if (NP == 2 && wrap) recompute_tap();     // wave-uniform: the next K-step starts a new tap
      }
    }
  };

  auto transform = [&](Regs& R) {
    f32x4(&ra)[NA] = R.ra;
    const unsigned a_valid = R.a_valid;
    const f32x4 psc = R.psc, psh = R.psh;
    if (PRO) {
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        f32x4 sc = (AM == 2) ? a2_sc[j] : psc, sh = (AM == 2) ? a2_sh[j] : psh;
        f32x4 v = ra[j] * sc + sh;
        if (PRO == 3) v += R.ra2[j] * R.psc2 + R.psh2;
        // NP weight gradient: nothing to zero -- a chunk past M has zero scale AND shift (out-of-range loads), a pixel
        // past the chunk meets a zero dy row
        // ReLU and the zeroing of out-of-bounds chunks in ONE v_med3 per value: clamp to [floor, +inf] in bounds, to
        // [0, 0] outside (max + select were two of the three vector instructions per value here, and on the fp32
        // matrix pipe every one of them is exposed)
        bool ok = (NP == 1 && AM == 2) ? true : (bool)((a_valid >> j) & 1u);
        const float lo = ok ? relu_floor : 0.f, hi = ok ? INFINITY : 0.f;
        v.x = __builtin_amdgcn_fmed3f(v.x, lo, hi);
        v.y = __builtin_amdgcn_fmed3f(v.y, lo, hi);
        v.z = __builtin_amdgcn_fmed3f(v.z, lo, hi);
        v.w = __builtin_amdgcn_fmed3f(v.w, lo, hi);
        ra[j] = v;
        if (PRO == 3 && store_sum)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rY,
                                                 ok ? (int)(y_off[j] + R.c0 * 4) : (int)DJ_OOB, 0, 0);
      }
    }
  };

  auto store_tiles = [&](const Regs& R, float* sA, float* sB) {
    const f32x4(&ra)[NA] = R.ra;
    const f32x4(&rb)[NB] = R.rb;
    if (AM != 2) {
#pragma unroll
      for (int j = 0; j < NA; ++j) *reinterpret_cast<f32x4*>(sA + (ar0 + 32 * j) * LDA_S + 4 * ac) = ra[j];
    } else {
#pragma unroll
      for (int j = 0; j < NA; ++j) *reinterpret_cast<f32x4*>(sA + ar0 * LDA_S + 4 * (ac + 8 * j)) = ra[j];
    }
    if (BMD == 0) {
#pragma unroll
      for (int j = 0; j < NB; ++j) *reinterpret_cast<f32x4*>(sB + (bkr0 + BKSTEP * j) * LDB_S + 4 * bcn) = rb[j];
    } else {
#pragma unroll
      for (int j = 0; j < NB; ++j) *reinterpret_cast<f32x4*>(sB + (br0 + 32 * j) * LDB_S + 4 * bc) = rb[j];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // a wave with a single 32x32 output tile would issue one dependent MFMA chain; a second accumulator for the odd
  // k pairs gives the matrix pipe two independent chains (summed before the epilogue)
  constexpr bool DUAL = (TM * TN == 1);
  f32x16 acc_odd;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc_odd[r] = 0.f;

  // one 8-deep k group = one fragment set (TM + TN b128 registers) feeding 4*TM*TN MFMAs
  struct Frag {
    f32x4 a[TM], b[TN];
  };
  auto read_frag = [&](Frag& f, const float* sA, const float* sB, int kk) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      int row = (wm * TM + i) * 32 + l31;
      if (Cfg::A_KC) {
        f.a[i] = *reinterpret_cast<const f32x4*>(sA + row * LDA_S + kk * 8 + lh * 4);
      } else {
        const float* q = sA + (kk * 8 + lh * 4) * LDA_S + row;
        f.a[i].x = q[0];
        f.a[i].y = q[LDA_S];
        f.a[i].z = q[2 * LDA_S];
        f.a[i].w = q[3 * LDA_S];
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      int col = (wn * TN + j) * 32 + l31;
      if (Cfg::B_KC) {
        f.b[j] = *reinterpret_cast<const f32x4*>(sB + col * LDB_S + kk * 8 + lh * 4);
      } else {
        const float* q = sB + (kk * 8 + lh * 4) * LDB_S + col;
        f.b[j].x = q[0];
        f.b[j].y = q[LDB_S];
        f.b[j].z = q[2 * LDB_S];
        f.b[j].w = q[3 * LDB_S];
      }
    }
  };
  auto mma = [&](const Frag& f) {
    if (PREC == 1) {
      dj_half4 ah[TM], bh[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) ah[i] = dj_to_half4(f.a[i]);
#pragma unroll
      for (int j = 0; j < TN; ++j) bh[j] = dj_to_half4(f.b[j]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x8f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    } else if (PREC == 2) {
      dj_short4 ah[TM], bh[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) ah[i] = dj_to_bf16x4(f.a[i]);
#pragma unroll
      for (int j = 0; j < TN; ++j) bh[j] = dj_to_bf16x4(f.b[j]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ah[i], bh[j], acc[i][j], 0, 0, 0);
    } else if (DUAL) {
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[0][e], f.b[0][e], acc[0][0], 0, 0, 0);
        acc_odd = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[0][e + 1], f.b[0][e + 1], acc_odd, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][e], f.b[j][e], acc[i][j], 0, 0, 0);
    }
  };
  auto compute_kk = [&](const float* sA, const float* sB, int kk) {
    Frag f;
    read_frag(f, sA, sB, kk);
    mma(f);
  };

  // One K-step on compile-time LDS stage `S` (two-stage ring), two schedules:
  // NSTAGE == 2: prefetch tile kt+1 while the MFMAs of tile kt run; the compiler places the loads (it sinks them
  //   towards their use: fewest registers, most resident workgroups).
  // NSTAGE == 4 ("pipelined"): entry: F0 holds the kk = 0 fragments of stage S.  Loads of tile kt+1 pinned at the top
  //   (a whole step of MFMAs to land); fragments read one k group ahead of the MFMAs that use them; tile kt+1 goes to
  //   the other stage, barrier, and the FIRST fragments of that stage are read BEFORE the last k group's MFMAs are
  //   issued, so the LDS latency across the step boundary hides behind them.  Wins when few workgroups share a CU.
  Frag F0, F1;
  auto kstep = [&](auto stage_tag, int kt) {
    constexpr int S = decltype(stage_tag)::value;
    float* curA = smem + S * STAGE;
    float* curB = curA + AFL;
    float* nxtA = smem + (S ^ 1) * STAGE;
    float* nxtB = nxtA + AFL;
    // the loads of a non-existent next tile are issued with out-of-range offsets (they return zeros
    // without touching memory) so that the K-step stays one straight-line block
    issue_loads(r0, kbeg + (kt + 1) * KSTEP, kt + 1 < nk_live);
    if (NSTAGE == 4) {
      __builtin_amdgcn_sched_barrier(0);
      read_frag(F1, curA, curB, 1);
      mma(F0);
      read_frag(F0, curA, curB, 2);
      mma(F1);
      read_frag(F1, curA, curB, 3);
      mma(F0);
      transform(r0);
      store_tiles(r0, nxtA, nxtB);
      __syncthreads();
      read_frag(F0, nxtA, nxtB, 0);
      __builtin_amdgcn_sched_barrier(0);
      mma(F1);
    } else {
      compute_kk(curA, curB, 0);
      compute_kk(curA, curB, 1);
      compute_kk(curA, curB, 2);
      transform(r0);
      store_tiles(r0, nxtA, nxtB);
      compute_kk(curA, curB, 3);
      __syncthreads();
    }
  };

  issue_loads(r0, kbeg, nk_live > 0);
  transform(r0);
  store_tiles(r0, smem, smem + AFL);
  __syncthreads();
  if (NSTAGE == 2 || NSTAGE == 4) {
    if (NSTAGE == 4) read_frag(F0, smem, smem + AFL, 0);
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      kstep(std::integral_constant<int, 0>{}, kt);
      kstep(std::integral_constant<int, 1>{}, kt + 1);
    }
    if (kt < nk) kstep(std::integral_constant<int, 0>{}, kt);
  } else {
    // single LDS stage (half the LDS, twice the resident workgroups), two barriers per K-step
    if (NSTAGE == 1) {
      // the next tile waits in registers while this one is consumed; few registers, up to 8 workgroups per CU hide
      // the load latency for each other
      for (int kt = 0; kt < nk; ++kt) {
        issue_loads(r0, kbeg + (kt + 1) * DJ_BK, kt + 1 < nk);
        compute_kk(smem, smem + AFL, 0);
        compute_kk(smem, smem + AFL, 1);
        compute_kk(smem, smem + AFL, 2);
        compute_kk(smem, smem + AFL, 3);
        transform(r0);
        __syncthreads();
        store_tiles(r0, smem, smem + AFL);
        __syncthreads();
      }
    } else {
      // NSTAGE == 3: as above, but the loads run TWO K-steps ahead (pinned above the MFMAs): a load has 2 x 16 x 64
      // MFMA cycles of a 32x32 wave tile to land.  For launches with too few workgroups per CU to hide latency by
      // occupancy; costs ~30 VGPRs.
      auto s1_step = [&](Regs& cur, Regs& nxt, int kt) {
        issue_loads(nxt, kbeg + (kt + 2) * DJ_BK, kt + 2 < nk);
        __builtin_amdgcn_sched_barrier(0);
        compute_kk(smem, smem + AFL, 0);
        compute_kk(smem, smem + AFL, 1);
        compute_kk(smem, smem + AFL, 2);
        compute_kk(smem, smem + AFL, 3);
        transform(cur);
        __syncthreads();
        store_tiles(cur, smem, smem + AFL);
        __syncthreads();
      };
      issue_loads(r0, kbeg + DJ_BK, 1 < nk);
      int kt = 0;
      for (; kt + 1 < nk; kt += 2) {
        s1_step(r0, r1, kt);
        s1_step(r1, r0, kt + 1);
      }
      if (kt < nk) s1_step(r0, r1, kt);
    }
  }

  if (DUAL) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] += acc_odd[r];
  }
  if (KS > 1) {
    // group 1 hands its partial tile to group 0 through its own (now idle) LDS ring: [value][thread], conflict-free
    __syncthreads();
    float* red = smem_base + 2 * STAGE;
    static_assert(KS == 1 || TM * TN * 16 * 256 <= 2 * STAGE, "partial tile does not fit the idle LDS ring");
    if (kg == 1) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[((i * TN + j) * 16 + r) * 256 + tid] = acc[i][j][r];
    }
    __syncthreads();
    if (kg == 1) {
      if (p.stats || p.bn_acc) __syncthreads();   // matches the barrier of the statistics epilogue
      if (p.bn_acc) {                              // ... and the two of the in-kernel BatchNormalization finalize
        __syncthreads();
        __syncthreads();
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] += red[((i * TN + j) * 16 + r) * 256 + tid];
  }
  dj_igemm_epilogue<BM, BN, WM, WN, EPI == 1>(p, acc, smem_base, tile_m, m0, n0, ky);
}
