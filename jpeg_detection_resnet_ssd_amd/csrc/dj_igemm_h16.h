// Reduced-precision implicit-GEMM convolution (BASELINE config 5, "fp16 MFMA"; K.set_floatx('float16' / 'bfloat16')):
// the same three GEMMs as dj_igemm_fast.h (forward, input gradient, weight gradient of keras.layers.Conv2D, reference
// call sites localisation_part/models/keras_ssd300_dct_j2d_resnet.py:77-96,128-160,483-545,562-675), with the operands
// rounded to fp16 (forward) or bf16 (gradients) ONCE, when a thread stores its piece of the tile to LDS, and fed to
// v_mfma_f32_32x32x16_{f16,bf16} -- the 16-deep CDNA4 instruction, fp32 accumulation.  HBM tensors stay fp32.
//
// What changed against the PREC != 0 path of dj_igemm_fast.h (fp32 tiles in LDS, every wave converting the fragments it
// reads, 8-deep MFMA): half the LDS bytes, each element converted once instead of once per reading wave, and half the
// MFMA instructions per K-step.
//
// LDS images (16-bit elements), one K-step = BK k (32, or 64 for forward / input gradient: see BK below):
//   * k-contiguous operand (A of forward / input gradient, B of the input gradient): [rows][BK k], pitch BK + 8 elements
//     (80 / 144 B: 16-byte aligned rows, conflict-free for the 16-lane groups of ds_read_b128).  Lane (row r, half h) reads
//     k = 16 s + 8 h .. + 7 of MFMA step s with one ds_read_b128: exactly its operand of v_mfma_*_32x32x16.
//   * the operand that is contiguous along the OTHER GEMM dimension (HWIO weights in the forward pass; both operands of the
//     weight gradient, whose reduction runs over pixels): stored as it arrives, [BK k][cols], pitch cols + 32 elements,
//     and read with ds_read_b64_tr_b16, gfx950's transposing LDS read: a 16-lane group hands in 4 row (k) addresses x
//     16 columns and every lane receives the 4 k values of ITS column -- two of them are a lane's 8-deep operand.  The
//     pitch puts the 4 rows x 2 column groups of a 32-lane half on 64 different banks.
// The global-load side (buffer loads with out-of-range offsets for halo / tails, wave-uniform tap state, the
// BatchNormalization(+ReLU) prologue in fp32 before the rounding, XCD-aware tile order, split-K) and the epilogue are
// those of the fp32 kernel.
//
// PREC 4 ("float32x6", arithmetic mode 4): as PREC 3 below with THREE bf16 pieces per operand (24 significant bits: all of an
// fp32 value) and six MFMAs per product -- hi*hi, hi*mid, mid*hi, hi*lo, mid*mid, lo*hi; the dropped terms are <= 2^-24 of
// the product, the size of ONE fp32 rounding -- i.e. fp32 results on the bf16 matrix pipe at 6/16 of the fp32 MFMA's time.
// Three LDS images per operand: K-steps of 32 only for the 128-row tiles.
//
// PREC 3 ("float32x3", arithmetic mode 3; round 3): fp32 tensors, fp32 RESULTS to ~2^-17, on the bf16 matrix pipe.  The fp32
// MFMA of gfx950 runs at 157 TFLOP/s, the bf16 MFMA at 2500: an fp32 operand is split, when it goes to LDS, into the bf16
// nearest to it (hi) and the bf16 nearest to what is left (lo) -- |x - hi - lo| <= 2^-18 |x| -- and a product is three MFMAs,
// hi*hi + hi*lo + lo*hi, accumulated in fp32 (the dropped terms lo*lo and the two residuals are each <= 2^-18 of the
// product).  Per-product error ~1e-5 against the 1e-3 bar of the parity tests (exact fp32: 1e-7), at 3/16 of the fp32
// MFMA's matrix-pipe time.  LDS holds two images per operand (the same bytes as fp32 tiles); everything else -- loads,
// prologues in fp32 before the split, tap state, XCD order, split-K, epilogues -- is the code of the other modes.
//
// Operands in HBM (round 3, BASELINE config 5 proper): AT / BT name how the gathered operand A (and the residual operand
// A2) and the operand B are STORED -- 0 fp32, 1 fp16, 2 bf16 (DJ_F32 / DJ_F16 / DJ_BF16 of include/dj_hip.h).  Activations
// are kept as fp16 and their gradients as bf16 inside the backbone: a thread's 4-element piece is then ONE 8-byte buffer
// load, stays in two registers until it goes to LDS (half the staging registers, half the bytes in flight per element),
// is widened to fp32 only where a prologue has arithmetic to do, and goes to LDS untouched when it already has the
// MFMA's type and there is no prologue (block sums in the forward pass, dy in both gradient GEMMs).  Weights are the fp32
// master copy or its per-step 16-bit shadow (fp16 for the forward GEMM, bf16 for the input gradient: dj_shadow_weights) --
// the shadow halves the bytes every tile pulls through L2 for its B operand and takes the conversion out of the tiles.  The result C, the stored residual sum and the BatchNormalization input z of the statistics
// epilogue carry their types at run time (DjIgemmParams::c_dt / sum_dt / bnb_zdt): they are touched once per tile.
//
// BK: with 16-bit MFMAs a 32-deep K-step is 2-8 matrix instructions per wave (64-256 cycles) between two barriers, and
// the loads of the next step have just that long to land: a workgroup's K-loop is a chain of global-load latencies
// (19x19 3x3 256->256 forward, 64x64 tiles: 72 steps of ~0.9 us = the 68 us the launch takes).  BK = 64 halves the
// number of links in that chain: twice the bytes in flight per thread and twice the MFMAs per barrier (forward and input
// gradient, channel counts that are multiples of 64; the weight gradient, whose reduction runs over pixels, keeps 32).
#pragma once
#include "dj_igemm.h"
#include "dj_igemm_fast.h"

typedef short dj_tr4 __attribute__((__vector_size__(4 * sizeof(short))));   // what the transposing read returns
typedef short dj_s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 dj_half8 __attribute__((ext_vector_type(8)));
typedef __bf16 dj_bf16x8 __attribute__((ext_vector_type(8)));
typedef short __attribute__((address_space(3))) dj_lds_short;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// A thread's 4-element piece of an operand between its buffer load and its LDS store, as the operand is stored in HBM
template <int DT>
struct DjRaw {
  using type = f32x4;
};
template <>
struct DjRaw<1> {
  using type = u32x2;
};
template <>
struct DjRaw<2> {
  using type = u32x2;
};

template <int DT>
__device__ __forceinline__ typename DjRaw<DT>::type dj_buf_ldraw(__amdgpu_buffer_rsrc_t r, unsigned off) {
  if constexpr (DT == 0) {
    return dj_buf_ld4(r, off);
  } else {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0));
  }
}

template <int DT>
__device__ __forceinline__ f32x4 dj_raw_to_f32(typename DjRaw<DT>::type v) {
  if constexpr (DT == 0) {
    return v;
  } else if constexpr (DT == 1) {
    const dj_half4 h = __builtin_bit_cast(dj_half4, v);
    return f32x4{(float)h.x, (float)h.y, (float)h.z, (float)h.w};
  } else {   // bf16: the upper half of the fp32 word
    return f32x4{__builtin_bit_cast(float, v.x << 16), __builtin_bit_cast(float, v.x & 0xFFFF0000u),
                 __builtin_bit_cast(float, v.y << 16), __builtin_bit_cast(float, v.y & 0xFFFF0000u)};
  }
}

template <int BM, int BN, int AM, int BMD, int BK = 32>
struct DjH16Cfg {
  static_assert(BK == 32 || BK == 16 || (BK == 64 && AM != 2), "64-deep K-steps: forward / input gradient only");
  static constexpr int TM = BM / 64, TN = BN / 64;   // 4 waves as 2 x 2
  static constexpr bool A_KC = (AM != 2), B_KC = (BMD == 1);
  // k-contiguous operands: a row of a K-step is KCH 16-byte chunks of four fp32; the 256 threads cover RPP rows per pass
  static constexpr int KCH = BK / 4, RPP = 256 / KCH;
  // the other layout ([k][cols], four columns per thread): 1024 / cols k rows per pass
  static constexpr int NA = A_KC ? BM / RPP : BK * BM / 1024;   // 16-byte loads per thread and K-step
  static constexpr int NB = B_KC ? BN / RPP : BK * BN / 1024;
  static constexpr int PA = A_KC ? BK + 8 : BM + 32, PB = B_KC ? BK + 8 : BN + 32;   // pitches, in 16-bit elements
  static constexpr int A_H = (A_KC ? BM : BK) * PA, B_H = (B_KC ? BN : BK) * PB;
  static constexpr int STAGE_H = A_H + B_H;
  static constexpr int SMEM_BYTES = 2 * STAGE_H * 2;
};

template <int PREC>
__device__ __forceinline__ dj_short4 dj_round4(f32x4 v) {
  if (PREC == 1) return __builtin_bit_cast(dj_short4, dj_to_half4(v));
  return dj_to_bf16x4(v);
}

// float32x3: x -> (hi, lo) with hi = bf16(x), lo = bf16(x - hi); hi to `dst`, lo `lo_off` elements behind it
__device__ __forceinline__ void dj_split_store(short* dst, f32x4 v, int lo_off) {
  const dj_short4 hi = dj_to_bf16x4(v);
  const f32x4 back = {__builtin_bit_cast(float, (unsigned)(unsigned short)hi.x << 16),
                      __builtin_bit_cast(float, (unsigned)(unsigned short)hi.y << 16),
                      __builtin_bit_cast(float, (unsigned)(unsigned short)hi.z << 16),
                      __builtin_bit_cast(float, (unsigned)(unsigned short)hi.w << 16)};
  *reinterpret_cast<dj_short4*>(dst) = hi;
  *reinterpret_cast<dj_short4*>(dst + lo_off) = dj_to_bf16x4(v - back);
}

__device__ __forceinline__ f32x4 dj_bf16x4_to_f32(dj_short4 h) {
  return f32x4{__builtin_bit_cast(float, (unsigned)(unsigned short)h.x << 16), __builtin_bit_cast(float, (unsigned)(unsigned short)h.y << 16),
               __builtin_bit_cast(float, (unsigned)(unsigned short)h.z << 16), __builtin_bit_cast(float, (unsigned)(unsigned short)h.w << 16)};
}
// float32x6: x = hi + mid + lo, each the bf16 nearest to what the ones before it leave (3 x 8 significant bits: what is left
// after lo is <= 2^-24 |x|); the images sit lo_off elements apart
__device__ __forceinline__ void dj_split_store3(short* dst, f32x4 v, int lo_off) {
  const dj_short4 hi = dj_to_bf16x4(v);
  const f32x4 r1 = v - dj_bf16x4_to_f32(hi);          // exact: hi agrees with v in its leading bits
  const dj_short4 mid = dj_to_bf16x4(r1);
  const f32x4 r2 = r1 - dj_bf16x4_to_f32(mid);        // exact
  *reinterpret_cast<dj_short4*>(dst) = hi;
  *reinterpret_cast<dj_short4*>(dst + lo_off) = mid;
  *reinterpret_cast<dj_short4*>(dst + 2 * lo_off) = dj_to_bf16x4(r2);
}

// PRO: 0 plain A, 1 A*scale[c]+shift[c] (+ReLU) on in-bounds elements, 3 the residual-add prologue of the forward 1x1
// convolutions, relu(A*scale+shift + A2*scale2+shift2), whose column-tile-0 workgroups also store that sum (fp32) to
// p.sum_out (see dj_igemm_fast.h).  PREC: 1 fp16, 2 bf16.
// PF: K-steps of register prefetch.  1: the loads of tile kt+1 are issued at the top of step kt and stored to LDS in its
// middle.  2: two register sets -- the loads of tile kt+2 are issued at the top of step kt, tile kt+1 (issued a whole
// step earlier) goes to LDS: a load has a full K-step longer to land (small tiles, whose steps are short; costs one
// more set of staging registers).
// EPI: 1 = the epilogue also takes BatchNormalization backward statistics (see dj_igemm_fast.h), input gradient only.
// AT, BT: storage type of A (and A2) / of B in HBM, see the header comment.
// NP: 1 = the gathered operand can never leave its tensor (1x1 kernel, no padding; for the weight gradient also stride 1,
//     input pixel == output pixel): the K-step has no coordinate adds, bounds tests or offset selects -- a row's validity is
//     a constant of the tile, a row past M carries an offset that stays out of range whatever is added to it.  Round 2 did
//     this for the fp32 kernels (NP 1 of dj_igemm_fast.h); here the counters say the kernels are bound by instruction ISSUE
//     (rocprofv3, 1x1 256->1024 @38x38 forward: instructions issuing in 83 % of a SIMD's cycles, matrix pipe busy 18 %, 37 %
//     of a wave's life waiting for data), and two thirds of the convolutions of a bottleneck block are 1x1.
template <int BM, int BN, int AM, int BMD, int PRO, int PREC, int BK = 32, int PF = 1, int EPI = 0, int AT = 0, int BT = 0, int NP = 0>
__global__ __launch_bounds__(256) void dj_igemm_h16_kernel(const DjIgemmParams p) {
  static_assert(NP == 0 || PRO != 3, "NP: not with the residual-add prologue (which keeps row indices of its own)");
  static_assert(EPI == 0 || (AM == 1 && BMD == 1), "BatchNormalization backward statistics: input-gradient GEMM only");
  static_assert(PREC < 3 || (AT == 0 && BT == 0), "float32x3 / float32x6: fp32 tensors");
  constexpr int EA = AT ? 2 : 4, EB = BT ? 2 : 4;   // bytes per stored element
  using ARaw = typename DjRaw<AT>::type;
  using BRaw = typename DjRaw<BT>::type;
  // the piece goes from HBM to LDS as it is: stored in the MFMA's type, no prologue
  constexpr bool A_COPY = (AT == PREC) && PRO == 0 && PREC < 3, B_COPY = (BT == PREC) && PREC < 3;
  using Cfg = DjH16Cfg<BM, BN, AM, BMD, BK>;
  constexpr int IMGS = (PREC == 4) ? 3 : (PREC == 3) ? 2 : 1;   // LDS images per operand: the bf16 pieces of PREC 3 / 4
  constexpr int STAGE = Cfg::STAGE_H * IMGS;          // elements per LDS stage; the lo images sit Cfg::STAGE_H behind the hi ones
  constexpr int TM = Cfg::TM, TN = Cfg::TN, NA = Cfg::NA, NB = Cfg::NB;
  constexpr int PA = Cfg::PA, PB = Cfg::PB, KCH = Cfg::KCH, RPP = Cfg::RPP;
  constexpr int WN = 2;
  extern __shared__ __attribute__((aligned(16))) float smem_base[];
  short* const smem = reinterpret_cast<short*>(smem_base);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lh = lane >> 5;

  const int tiles_n = (p.N + BN - 1) / BN;
  int tile_id, ky;
  dj_tile_of_workgroup(tile_id, ky);   // XCD-aware order of (tile, K chunk), dj_igemm.h
  const int tile_m = tile_id / tiles_n;
  const int tile_n = tile_id - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kbeg = ky * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const int nk = (kend - kbeg + BK - 1) / BK;

  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, p.b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)p.pro_scale, 0, PRO ? p.srcC * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc((void*)p.pro_shift, 0, PRO ? p.srcC * 4 : 0, 0x00020000);
  const float relu_floor = (p.pro_relu || PRO == 3) ? 0.f : -INFINITY;
  const __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A2, 0, PRO == 3 ? p.a2_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc((void*)p.sum_out, 0, (PRO == 3 && p.sum_out) ? p.sum_bytes : 0, 0x00020000);
  const bool store_sum = (PRO == 3) && p.sum_out != nullptr && tile_n == 0;

  // ---------------- per-thread staging state (as in dj_igemm_fast.h) ----------------
  // (weight gradient: a thread holds pixel row tid / ACG of the K-step and the 4-column chunks ac, ac + ACG, ...)
  constexpr int ACG = (AM != 2) ? KCH : 256 / BK;
  const int ac = tid % ACG, ar0 = tid / ACG;
  constexpr int ARPP = (AM != 2) ? RPP : 32;   // A rows between a thread's loads
  constexpr int BKSTEP = 1024 / BN;
  const int bcn = tid % (BN / 4), bkr0 = tid / (BN / 4);
  const int bc = tid % KCH, br0 = tid / KCH;

  int a_off[NA], a_rh[NA], a_rw[NA];
  unsigned row_valid = 0;   // NP: bit j = row j of this thread exists
  // PRO 3 (1x1, stride 1, unpadded: the input pixel IS the GEMM row): only the row index is kept (-1 past M) and the three
  // byte offsets of a row -- x, the residual operand, the stored sum -- are one 24-bit multiply-add each where they are
  // used; with offsets and coordinates held per row and operand the 64-deep variants needed 268-323 registers (one wave
  // per SIMD), without them 128x128 fits two
  int pm[PRO == 3 ? NA : 1];
  int a2_c[NA], a2_dh[NA], a2_dw[NA];
  bool a2_ok[NA];
  f32x4 a2_sc[NA], a2_sh[NA];
  if (PRO == 3) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int m = m0 + ar0 + ARPP * j;
      pm[j] = (m < p.M) ? m : -1;
    }
  } else if (AM != 2) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      int m = m0 + ar0 + ARPP * j;
      if (m < p.M) {
        int img, h, w;
        dj_row_decompose(p, m, img, h, w);
        int rh = (AM == 0) ? h * p.sH - p.pT : h + p.pT;
        int rw = (AM == 0) ? w * p.sW - p.pL : w + p.pL;
        a_rh[j] = rh;
        a_rw[j] = rw;
        a_off[j] = ((img * p.srcH * p.srcW + rh * p.srcW + rw) * p.ldsrc + 4 * ac) * EA;
        row_valid |= 1u << j;
      } else {
        a_rh[j] = -(1 << 28);
        a_rw[j] = -(1 << 28);
        a_off[j] = NP ? (int)0x80000000 : 0;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int mm = m0 + 4 * (ac + ACG * i);
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
      if (NP) a2_c[i] = a2_ok[i] ? a2_c[i] * EA : (int)0x80000000;   // byte offset of the chunk inside its pixel
    }
  }
  // NP, weight gradient: byte offset of this thread's pixel row (pixel kbeg + ar0 of the first K-step), advanced per K-step;
  // a pixel past the chunk reads other (finite) rows of x or nothing and meets a zero dy row
  int np_row = (NP && AM == 2) ? (kbeg + ar0) * p.ldsrc * EA : 0;
  const int np_step = BK * p.ldsrc * EA;
  int b_off[NB];
  bool b_ok[NB];
  if (BMD == 0) {
    int n = n0 + 4 * bcn;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      b_ok[j] = n < p.N;
      b_off[j] = ((bkr0 + BKSTEP * j) * p.ldb + n) * EB;
    }
  } else {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      int n = n0 + br0 + RPP * j;
      b_ok[j] = n < p.N;
      b_off[j] = (n * p.ldb + 4 * bc) * EB;
    }
  }

  int t_c0 = 0, t_kh = 0, t_kw = 0, t_tap = 0;
  if (AM != 2 || BMD == 1) {
    t_tap = kbeg / p.srcC;
    t_c0 = kbeg - t_tap * p.srcC;
    t_kh = t_tap / p.KW;
    t_kw = t_tap - t_kh * p.KW;
  }

  // one K-step's operands between the buffer loads and the LDS stores
  struct Regs {
    ARaw ra[NA];
    BRaw rb[NB];
    unsigned a_valid;
    f32x4 psc, psh;
    ARaw ra2[PRO == 3 ? NA : 1];
    f32x4 psc2, psh2;
    int pro_c0;
  };
  Regs r0, r1;
  r0.a_valid = r1.a_valid = 0;
  r0.pro_c0 = r1.pro_c0 = 0;
  r0.psc = r1.psc = r0.psc2 = r1.psc2 = f32x4{1.f, 1.f, 1.f, 1.f};
  r0.psh = r1.psh = r0.psh2 = r1.psh2 = f32x4{0.f, 0.f, 0.f, 0.f};

  auto issue_loads = [&](Regs& R, int kcur, bool live) {
    ARaw(&ra)[NA] = R.ra;
    BRaw(&rb)[NB] = R.rb;
    ARaw(&ra2)[PRO == 3 ? NA : 1] = R.ra2;
    unsigned& a_valid = R.a_valid;
    f32x4 &psc = R.psc, &psh = R.psh, &psc2 = R.psc2, &psh2 = R.psh2;
    int& pro_c0 = R.pro_c0;
    if (AM != 2) {
      const int dh = t_kh * p.dH, dw = t_kw * p.dW;
      const int delta = (AM == 0) ? ((dh * p.srcW + dw) * p.ldsrc + t_c0) * EA : (-(dh * p.srcW + dw) * p.ldsrc + t_c0) * EA;
      if (PRO == 1) {
        psc = dj_buf_ld4(rS, (unsigned)(t_c0 + 4 * ac) * 4u);
        psh = dj_buf_ld4(rT, (unsigned)(t_c0 + 4 * ac) * 4u);
      }
      if (PRO == 3) {   // plain loads from uniform bases: fewer descriptors live in the loop (see dj_igemm_fast.h)
        pro_c0 = t_c0;
        psc = dj_ld4(p.pro_scale + (t_c0 + 4 * ac));
        psh = dj_ld4(p.pro_shift + (t_c0 + 4 * ac));
        psc2 = f32x4{1.f, 1.f, 1.f, 1.f};
        psh2 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.pro_scale2) {
          psc2 = dj_ld4(p.pro_scale2 + (t_c0 + 4 * ac));
          psh2 = dj_ld4(p.pro_shift2 + (t_c0 + 4 * ac));
        }
      }
      a_valid = 0;
      if (PRO == 3) {
        const unsigned cb = (unsigned)(4 * ac + t_c0);
#pragma unroll
        for (int j = 0; j < NA; ++j) {
          const bool ok = live && pm[j] >= 0;
          const unsigned m = (unsigned)pm[j];
          ra[j] = dj_buf_ldraw<AT>(rA, ok ? (__umul24(m, (unsigned)p.ldsrc) + cb) * (unsigned)EA : DJ_OOB);
          ra2[j] = dj_buf_ldraw<AT>(rA2, ok ? (__umul24(m, (unsigned)p.ldsrc2) + cb) * (unsigned)EA : DJ_OOB);
          a_valid |= ok ? (1u << j) : 0u;
        }
      } else if (NP) {
        // the prefetch past the last K-step (never consumed) is sent out of range by the wave-uniform part of the offset:
        // on these short-K launches one wasted K-step of loads is an eighth of the traffic
        a_valid = row_valid;
        const int udelta = live ? delta : (int)0x80000000;
#pragma unroll
        for (int j = 0; j < NA; ++j) ra[j] = dj_buf_ldraw<AT>(rA, (unsigned)(a_off[j] + udelta));
      } else {
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        int h = (AM == 0) ? a_rh[j] + dh : a_rh[j] - dh;
        int w = (AM == 0) ? a_rw[j] + dw : a_rw[j] - dw;
        bool ok = live && (unsigned)h < (unsigned)p.srcH && (unsigned)w < (unsigned)p.srcW;
        ra[j] = dj_buf_ldraw<AT>(rA, ok ? (unsigned)(a_off[j] + delta) : DJ_OOB);
        a_valid |= ok ? (1u << j) : 0u;
      }
      }
    } else if (NP) {
      a_valid = row_valid;
      const int urow = live ? np_row : (int)0x80000000;
#pragma unroll
      for (int i = 0; i < NA; ++i) ra[i] = dj_buf_ldraw<AT>(rA, (unsigned)(urow + a2_c[i]));
      np_row += np_step;
    } else {
      a_valid = 0;
      const int kp = kcur + ar0;
      const int hw = p.rowH * p.rowW;
      int img = (int)((float)kp * p.inv_rowHW);
      int rem = kp - img * hw;
      int adj = (rem < 0) ? -1 : ((rem >= hw) ? 1 : 0);
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
        unsigned off = (unsigned)((pb + h * p.srcW + w) * p.ldsrc + a2_c[i]) * (unsigned)EA;
        ra[i] = dj_buf_ldraw<AT>(rA, ok ? off : DJ_OOB);
        a_valid |= ok ? (1u << i) : 0u;
      }
    }
    if (BMD == 0) {
      const int base = kcur * p.ldb * EB;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        bool ok = live && b_ok[j] && (AM != 2 || kcur + bkr0 + BKSTEP * j < kend);
        rb[j] = dj_buf_ldraw<BT>(rB, ok ? (unsigned)(b_off[j] + base) : DJ_OOB);
      }
    } else {
      const unsigned base = (unsigned)(t_tap * p.bTapStride + t_c0) * (unsigned)EB;
#pragma unroll
      for (int j = 0; j < NB; ++j) rb[j] = dj_buf_ldraw<BT>(rB, (live && b_ok[j]) ? (unsigned)b_off[j] + base : DJ_OOB);
    }
    if (AM != 2 || BMD == 1) {
      t_c0 += BK;
      const int wrap = (t_c0 >= p.srcC) ? 1 : 0;
      t_c0 = wrap ? 0 : t_c0;
      t_tap += wrap;
      t_kw += wrap;
      const int wrap2 = (t_kw >= p.KW) ? 1 : 0;
      t_kw = wrap2 ? 0 : t_kw;
      t_kh += wrap2;
    }
  };

  auto store_a = [&](const Regs& R, short* sA, int j) {
    const ARaw(&ra)[NA] = R.ra;
    const ARaw(&ra2)[PRO == 3 ? NA : 1] = R.ra2;
    const unsigned a_valid = R.a_valid;
    const f32x4 psc = R.psc, psh = R.psh, psc2 = R.psc2, psh2 = R.psh2;
    const int pro_c0 = R.pro_c0;
    {
      short* dst = (AM != 2) ? sA + (ar0 + ARPP * j) * PA + 4 * ac : sA + ar0 * PA + 4 * (ac + ACG * j);
      if constexpr (A_COPY) {   // stored in the MFMA's type, no prologue: out-of-range pieces arrived as zeros
        *reinterpret_cast<u32x2*>(dst) = ra[j];
        return;
      }
      f32x4 v = dj_raw_to_f32<AT>(ra[j]);
      if (PRO) {
        f32x4 sc = (AM == 2) ? a2_sc[j] : psc, sh = (AM == 2) ? a2_sh[j] : psh;
        v = v * sc + sh;
        if (PRO == 3) v += dj_raw_to_f32<AT>(ra2[j]) * psc2 + psh2;
        // (NP weight gradient: nothing to zero -- a chunk past M has zero scale AND shift, a pixel past the chunk meets a
        // zero dy row)
        bool ok = (NP && AM == 2) ? true : (bool)((a_valid >> j) & 1u);
        const float lo = ok ? relu_floor : 0.f, hi = ok ? INFINITY : 0.f;   // ReLU + out-of-bounds zero: one v_med3
        v.x = __builtin_amdgcn_fmed3f(v.x, lo, hi);
        v.y = __builtin_amdgcn_fmed3f(v.y, lo, hi);
        v.z = __builtin_amdgcn_fmed3f(v.z, lo, hi);
        v.w = __builtin_amdgcn_fmed3f(v.w, lo, hi);
        if (PRO == 3 && store_sum) {
          // the Add + ReLU output for its other readers, in the type that tensor has in HBM (wave-uniform)
          const unsigned e = __umul24((unsigned)pm[j], (unsigned)p.ld_sum) + (unsigned)(4 * ac + pro_c0);
          if (p.sum_dt == 0)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rY, ok ? (int)(e * 4u) : (int)DJ_OOB, 0, 0);
          else
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, p.sum_dt == 1 ? dj_round4<1>(v) : dj_round4<2>(v)),
                                                  rY, ok ? (int)(e * 2u) : (int)DJ_OOB, 0, 0);
        }
      }
      if constexpr (PREC == 4) {
        dj_split_store3(dst, v, Cfg::STAGE_H);
      } else if constexpr (PREC == 3) {
        dj_split_store(dst, v, Cfg::STAGE_H);
      } else {
        *reinterpret_cast<dj_short4*>(dst) = dj_round4<PREC>(v);
      }
    }
  };
  auto store_b = [&](const Regs& R, short* sB, int j) {
    const BRaw(&rb)[NB] = R.rb;
    {
      short* dst = (BMD == 0) ? sB + (bkr0 + BKSTEP * j) * PB + 4 * bcn : sB + (br0 + RPP * j) * PB + 4 * bc;
      if constexpr (B_COPY)
        *reinterpret_cast<u32x2*>(dst) = rb[j];
      else if constexpr (PREC == 4)
        dj_split_store3(dst, dj_raw_to_f32<BT>(rb[j]), Cfg::STAGE_H);
      else if constexpr (PREC == 3)
        dj_split_store(dst, dj_raw_to_f32<BT>(rb[j]), Cfg::STAGE_H);
      else
        *reinterpret_cast<dj_short4*>(dst) = dj_round4<PREC>(dj_raw_to_f32<BT>(rb[j]));
    }
  };
  // prologue in fp32, then ONE rounding per element, 8-byte LDS stores
  auto store_tiles = [&](const Regs& R, short* sA, short* sB) {
#pragma unroll
    for (int j = 0; j < NA; ++j) store_a(R, sA, j);
#pragma unroll
    for (int j = 0; j < NB; ++j) store_b(R, sB, j);
  };

  // No accumulator clears: the first MFMA of every accumulator takes the constant 0 as its C operand (an inline constant
  // of the instruction), in a peeled first K-step below.  The clears were 128 v_accvgpr_write per wave of a 128x128 tile --
  // a tenth of ALL vector instructions of a K = 256 tile, in kernels whose vector pipe is busier than their matrix pipe.
  f32x16 acc[TM][TN];

  // lane constants of the fragment reads
  //  k-contiguous image: element offset of (row l31, k = 8 h) inside a 32-row tile
  //  transposed image: lane 4q + p of a 16-lane group addresses row (k) q, columns 4p .. 4p+3; the group's columns are
  //  16 * ((lane >> 4) & 1) .. + 15 of the 32-column tile; the lane half picks the k half (8 h)
  const int kc_a = l31 * PA + 8 * lh, kc_b = l31 * PB + 8 * lh;
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  const int tr_a = (8 * lh + tq) * PA + 16 * tg + 4 * tp, tr_b = (8 * lh + tq) * PB + 16 * tg + 4 * tp;

  auto frag = [&](const short* img, bool kc, int kc_off, int tr_off, int pitch, int tile, int s) -> dj_s16x8 {
    if (kc) return *reinterpret_cast<const dj_s16x8*>(img + tile * 32 * pitch + kc_off + 16 * s);
    const short* q = img + tr_off + tile * 32 + 16 * s * pitch;
    dj_tr4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((dj_tr4 __attribute__((address_space(3)))*)(dj_lds_short*)q);
    dj_tr4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((dj_tr4 __attribute__((address_space(3)))*)(dj_lds_short*)(q + 4 * pitch));
    return dj_s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };
  struct Frags {
    dj_s16x8 a[TM], b[TN], a_lo[PREC >= 3 ? TM : 1], b_lo[PREC >= 3 ? TN : 1], a_l2[PREC == 4 ? TM : 1], b_l2[PREC == 4 ? TN : 1];
  };
  auto load_frags = [&](Frags& f, const short* sA, const short* sB, int s) {
#pragma unroll
    for (int i = 0; i < TM; ++i) f.a[i] = frag(sA, Cfg::A_KC, kc_a, tr_a, PA, wm * TM + i, s);
#pragma unroll
    for (int j = 0; j < TN; ++j) f.b[j] = frag(sB, Cfg::B_KC, kc_b, tr_b, PB, wn * TN + j, s);
    if constexpr (PREC >= 3) {
#pragma unroll
      for (int i = 0; i < TM; ++i) f.a_lo[i] = frag(sA + Cfg::STAGE_H, Cfg::A_KC, kc_a, tr_a, PA, wm * TM + i, s);
#pragma unroll
      for (int j = 0; j < TN; ++j) f.b_lo[j] = frag(sB + Cfg::STAGE_H, Cfg::B_KC, kc_b, tr_b, PB, wn * TN + j, s);
    }
    if constexpr (PREC == 4) {
#pragma unroll
      for (int i = 0; i < TM; ++i) f.a_l2[i] = frag(sA + 2 * Cfg::STAGE_H, Cfg::A_KC, kc_a, tr_a, PA, wm * TM + i, s);
#pragma unroll
      for (int j = 0; j < TN; ++j) f.b_l2[j] = frag(sB + 2 * Cfg::STAGE_H, Cfg::B_KC, kc_b, tr_b, PB, wn * TN + j, s);
    }
  };
  // one accumulator's share of a 16-deep slice: one MFMA, three for the split operands of PREC 3
  auto mfma_unit = [&](bool first, const Frags& f, int i, int j) {
    f32x16 c = acc[i][j];
    if (first) {
#pragma unroll
      for (int r = 0; r < 16; ++r) c[r] = 0.f;
    }
    if constexpr (PREC == 4) {
      // hi*hi + (hi*mid + mid*hi) + (hi*lo + mid*mid + lo*hi): every term down to 2^-16 of the product, smallest first;
      // mid*lo, lo*mid (2^-24) and lo*lo are dropped -- what fp32 rounding loses in one accumulation
      auto mm = [](dj_s16x8 x, dj_s16x8 y, f32x16 acc_) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dj_bf16x8, x), __builtin_bit_cast(dj_bf16x8, y), acc_, 0, 0, 0);
      };
      c = mm(f.a_l2[i], f.b[j], c);
      c = mm(f.a[i], f.b_l2[j], c);
      c = mm(f.a_lo[i], f.b_lo[j], c);
      c = mm(f.a_lo[i], f.b[j], c);
      c = mm(f.a[i], f.b_lo[j], c);
      acc[i][j] = mm(f.a[i], f.b[j], c);
    } else if constexpr (PREC == 3) {
      // the two small products first, the large one last
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dj_bf16x8, f.a_lo[i]), __builtin_bit_cast(dj_bf16x8, f.b[j]), c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dj_bf16x8, f.a[i]), __builtin_bit_cast(dj_bf16x8, f.b_lo[j]), c, 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dj_bf16x8, f.a[i]), __builtin_bit_cast(dj_bf16x8, f.b[j]), c, 0, 0, 0);
    } else if constexpr (PREC == 1) {
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(dj_half8, f.a[i]), __builtin_bit_cast(dj_half8, f.b[j]), c, 0, 0, 0);
    } else {
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dj_bf16x8, f.a[i]), __builtin_bit_cast(dj_bf16x8, f.b[j]), c, 0, 0, 0);
    }
  };
  // two LDS stages: tile kt+1 is stored while tile kt is multiplied
  using First = std::integral_constant<bool, true>;
  using Later = std::integral_constant<bool, false>;
  auto kstep = [&](auto first_tag, Regs& load_into, const Regs& store_from, int kt, int k_load, bool live) {
    constexpr bool FIRST = decltype(first_tag)::value;   // the tile's first K-step: its first MFMAs start the accumulators
    short* cur = smem + (kt & 1) * STAGE;
    short* nxt = smem + ((kt + 1) & 1) * STAGE;
    issue_loads(load_into, k_load, live);
    if (PF == 2) __builtin_amdgcn_sched_barrier(0);   // the loads stay up here, a whole step ahead of their LDS stores
    {
      // Interleaved K-step.  The matrix pipe takes 32 cycles per MFMA while the wave is free to issue 4-cycle vector
      // instructions, but the compiler's schedule keeps the MFMAs of a slice together and the next tile's prologue /
      // rounding / LDS stores in one long vector stretch during which the matrix pipe idles (and with one or two waves per
      // SIMD nobody else fills it).  Here the order is pinned: after each accumulator's MFMA(s) comes a share of the next
      // tile's pieces, the following slice's fragments are read a slice ahead.
      constexpr int S = BK / 16, U = TM * TN, P = NA + NB;
      constexpr int G0 = (PF == 2) ? 0 : (S * U) / 4;        // PF 1: the pieces were requested at the top of this step
      Frags f[2];
      load_frags(f[0], cur, cur + Cfg::A_H, 0);
#pragma unroll
      for (int s = 0; s < S; ++s) {
        if (s + 1 < S) load_frags(f[(s + 1) & 1], cur, cur + Cfg::A_H, s + 1);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          mfma_unit(FIRST && s == 0, f[s & 1], u / TN, u % TN);
          const int g = s * U + u;
          if (g >= G0) {
            const int p0 = (g - G0) * P / (S * U - G0), p1 = (g - G0 + 1) * P / (S * U - G0);
#pragma unroll
            for (int q = p0; q < p1; ++q) {
              if (q < NA) store_a(store_from, nxt, q);
              else store_b(store_from, nxt + Cfg::A_H, q - NA);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    __syncthreads();
  };
  issue_loads(r0, kbeg, nk > 0);
  if (PF == 2) issue_loads(r1, kbeg + BK, nk > 1);
  store_tiles(r0, smem, smem + Cfg::A_H);
  __syncthreads();
  // every K chunk of a launch is non-empty (the launchers derive the number of chunks from the chunk length), so there
  // is a first K-step that starts the accumulators
  __builtin_assume(nk > 0);
  if (PF == 2) {
    // step kt: tile kt+2 -> the register set tile kt just left; tile kt+1 (other set) -> LDS
    kstep(First{}, r0, r1, 0, kbeg + 2 * BK, 2 < nk);
    int kt = 1;
    if (nk > 1) {
      kstep(Later{}, r1, r0, 1, kbeg + 3 * BK, 3 < nk);
      kt = 2;
    }
    for (; kt + 1 < nk; kt += 2) {
      kstep(Later{}, r0, r1, kt, kbeg + (kt + 2) * BK, kt + 2 < nk);
      kstep(Later{}, r1, r0, kt + 1, kbeg + (kt + 3) * BK, kt + 3 < nk);
    }
    if (kt < nk) kstep(Later{}, r0, r1, kt, kbeg + (kt + 2) * BK, false);
  } else {
    kstep(First{}, r0, r0, 0, kbeg + BK, 1 < nk);
    for (int kt = 1; kt < nk; ++kt) kstep(Later{}, r0, r0, kt, kbeg + (kt + 1) * BK, kt + 1 < nk);
  }
  dj_igemm_epilogue<BM, BN, 2, 2, EPI == 1, true, Cfg::SMEM_BYTES * IMGS>(p, acc, smem_base, tile_m, m0, n0, ky);
}
