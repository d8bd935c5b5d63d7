// Implicit-GEMM convolution core for gfx950 (MI355X): exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32), NHWC activations x HWIO weights, LDS-staged tiles.
//
// One templated kernel covers the three directions of keras.layers.Conv2D as the
// reference uses it (localisation_part/models/keras_ssd300_dct_j2d_resnet.py:77-96,
// 128-160, 483-545, 562-675) plus Conv2DTranspose (:1709-1711):
//
//   A-mode 0 (forward gather)  rows m = (img, oh, ow), k = (kh, kw, ci)
//                              A[m][k] = src[img, oh*s + kh*d - pad, ow*s + kw*d - pad, ci]
//   A-mode 1 (dgrad gather)    rows m = (img, ih, iw), k = (kh, kw, co)
//                              A[m][k] = src[img, (ih + pad - kh*d)/s, (iw + pad - kw*d)/s, co]
//   A-mode 2 (wgrad gather)    rows m' = (kh, kw, ci), k' = (img, oh, ow)   (A stored m-contiguous)
//                              A[m'][k'] = src[img, oh*s + kh*d - pad, ow*s + kw*d - pad, ci]
//   B-mode 0 (KN)              B[k][n] = Bp[k*ldb + n]                    (HWIO weights; dy rows)
//   B-mode 1 (NK, per tap)     B[k=(tap,c)][n] = Bp[tap*bTapStride + n*ldb + c]   (HWIO read as W^T)
//
// The k order inside one 8-wide MFMA group is permuted identically for A and B
// (lane half h takes k = 8*kk + 4*h + e), which lets k-contiguous operands be read
// from LDS with one ds_read_b128 per 4 MFMAs.
#pragma once
#include "dj_common.h"

#define DJ_BK 32
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

struct DjIgemmParams {
  const float* A;
  const float* B;
  float* C;
  const float* bias;       // [N] or null; added in the epilogue
  const float* pro_scale;  // [srcC] or null: A element -> A*scale[c] + shift[c] (in-bounds only)
  const float* pro_shift;
  float* stats;            // [ceil(M/64)][2][N] column sum / sum-of-squares of the raw accumulator per 64 rows, or null
  int M, N, K;
  int kchunk;              // K range handled per blockIdx.y (multiple of DJ_BK)
  // gather geometry
  int rowH, rowW;          // pixel grid indexed by m (A-mode 0/1) or by k' (A-mode 2)
  int srcH, srcW, srcC;    // gathered tensor
  int ldsrc;               // floats between consecutive pixels of the gathered tensor
  int KH, KW, sH, sW, dH, dW, pT, pL;
  int ldb;
  long bTapStride;
  // epilogue
  int ldc;                 // floats between consecutive C rows
  int cmap;                // 0: row m -> m*ldc ; 1: m=(img,h,w) on (cgH,cgW) -> ((img*cH + h*cS)*cW + w*cS)*ldc
  int cgH, cgW, cH, cW, cS;
  int pro_relu;            // prologue ReLU after the affine
  int relu;                // epilogue ReLU
  int beta;                // 1: C = acc + C
  int atomic;              // 1: atomicAdd into C (split-K)
  long slab_stride;        // != 0: split-K WITHOUT atomics -- K-chunk blockIdx.y stores its partial tile to C + blockIdx.y * slab_stride
  int vecA, vecB;          // 16-byte loads legal for A / B
  float inv_rowHW, inv_rowW;  // 1/(rowH*rowW), 1/rowW for the pixel decomposition of the fast wgrad path
  int a_bytes, b_bytes;       // byte extents of A and B for the buffer descriptors of the fast path
  // residual-add prologue of the fast forward kernel (1x1, stride 1): A = relu(A*scale+shift + A2*scale2+shift2),
  // optionally stored to sum_out by the workgroups of column tile 0 (this conv then IS the Add + ReLU pass)
  const float* A2;
  const float* pro_scale2;  // null: A2 is taken as is
  const float* pro_shift2;
  float* sum_out;
  int ldsrc2, ld_sum, a2_bytes, sum_bytes;
  // BatchNormalization (training) finished inside the producing convolution: every workgroup adds its column sums to
  // bn_acc with fp64 atomics, takes a ticket, and the last one turns the totals into scale / shift / saved and moving
  // statistics and clears bn_acc / bn_ticket again -- no separate finalize launch between this conv and its consumer
  double* bn_acc;           // [bn_replicas][2][N], zero on entry
  int bn_replicas;          // workgroups spread their atomics over this many copies (same-address contention)
  unsigned* bn_ticket;      // zero on entry
  const float* bn_gamma;
  const float* bn_beta;
  float* bn_moving_mean;    // may be null (then bn_moving_var is null too)
  float* bn_moving_var;
  float* bn_scale;
  float* bn_shift;
  float* bn_save_mean;
  float* bn_save_invstd;
  float bn_eps, bn_momentum;
  double bn_count;
  // BatchNormalization BACKWARD statistics taken where the incoming gradient is produced: this GEMM is the input
  // gradient g = dL/d relu(bn(z)) of the (only) consumer of that BatchNormalization, and its epilogue writes, per 64 rows
  // and column, sum(g_m) and sum(g_m * (z - mean) * invstd) with g_m = g where z*scale+shift > 0 (else 0) to `stats` --
  // the partial rows dj_bn_bwd_finalize reads; the separate pass over g and z (dj_bn_bwd_reduce) disappears
  const float* bnb_z;       // [M][bnb_ldz]: input of that BatchNormalization (raw conv output); null: off
  int bnb_ldz;
  int bnb_zbytes;           // byte extent of z, ((M - 1) * bnb_ldz + N) * 4 < 2 GiB, for the buffer descriptor
  const float* bnb_mean;    // [N]
  const float* bnb_invstd;
  const float* bnb_scale;   // [N] or null (no ReLU behind the BatchNormalization: nothing is masked)
  const float* bnb_shift;
  // Storage types of the tensors in HBM, 0 fp32 / 1 fp16 / 2 bf16 (dj_igemm_h16.h; the fp32 kernels only know 0): A (and
  // A2) and B select a kernel instantiation, the others are looked at once per tile.  With a 16-bit operand the pointer
  // fields above are reinterpreted; ld* stay in elements, *_bytes are real byte extents.
  int a_dt, b_dt, c_dt, sum_dt, bnb_zdt;
};

template <int BM, int BN, int WM, int WN, int AM, int BMD>
struct DjIgemmCfg {
  static constexpr int TM = BM / (32 * WM);
  static constexpr int TN = BN / (32 * WN);
  static constexpr int NA = BM / 32;  // float4 loads per thread per K-step for A
  static constexpr int NB = BN / 32;
  static constexpr bool A_KC = (AM != 2);
  static constexpr bool B_KC = (BMD == 1);
  static constexpr int LDA_S = A_KC ? (DJ_BK + 4) : BM;
  static constexpr int A_ROWS = A_KC ? BM : DJ_BK;
  static constexpr int LDB_S = B_KC ? (DJ_BK + 4) : BN;
  static constexpr int B_ROWS = B_KC ? BN : DJ_BK;
  static constexpr int A_FLOATS = A_ROWS * LDA_S;
  static constexpr int B_FLOATS = B_ROWS * LDB_S;
  static constexpr int STAGE_FLOATS = A_FLOATS + B_FLOATS;
  static constexpr int SMEM_BYTES = 2 * STAGE_FLOATS * 4;
};

__device__ __forceinline__ f32x4 dj_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// Workgroup -> (tile, K chunk).  Workgroups are dealt round-robin over the 8 XCDs in launch order (x fastest, then y),
// and each XCD has its own L2.  Order the work chunk-major (all tiles of K chunk 0, then chunk 1, ...; inside a chunk the
// column tiles of one row tile are adjacent) and give every XCD a CONTIGUOUS run of that list, taken in slot order: tiles
// that read the same operand rows run on one L2 at about the same time.  For the weight gradient the chunk is a pixel
// range whose x and dy rows are read by EVERY tile of the chunk: with the tiles of a chunk spread over the XCDs (what
// `blockIdx.x % 8` gives when the split is on blockIdx.y) each L2 fetches them again -- measured on 1x1 256->1024 @38x38:
// x fetched 4x, dy 2x (rocprofv3 FETCH_SIZE per launch, tools/pmc_layers.py), 8.6x on a 3x3 128->128.  Bijective for any grid.
__device__ __forceinline__ void dj_tile_of_workgroup(int& tile_id, int& ky) {
  const int gx = gridDim.x, total = gx * gridDim.y;
  const int L = blockIdx.x + blockIdx.y * gx;
  const int q = total >> 3, r = total & 7, xcd = L & 7, slot = L >> 3;
  const int item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  if (gridDim.y == 1) {
    ky = 0;
    tile_id = item;
  } else {
    ky = item / gx;
    tile_id = item - ky * gx;
  }
}

// GEMM row m -> (image, h, w) on the rowH x rowW pixel grid.  gfx950 has no integer division: `m / d` expands to ~25
// vector instructions, four of them quarter-rate multiplies, and on the fp32 matrix pipe every vector instruction of
// a tile's prologue is paid in full (tools/micro/mfma_valu.hip).  Below 2^22 (the fast kernels' launcher checks M) the
// float reciprocal lands within one of the quotient; a compare on the remainder corrects it (full-rate instructions only).
__device__ __forceinline__ void dj_row_decompose(const DjIgemmParams& p, int m, int& img, int& h, int& w) {
  const int hw = p.rowH * p.rowW;
  img = (int)((float)m * p.inv_rowHW);
  int rem = m - __mul24(img, hw);
  img += (rem < 0) ? -1 : ((rem >= hw) ? 1 : 0);
  rem += (rem < 0) ? hw : ((rem >= hw) ? -hw : 0);
  h = (int)((float)rem * p.inv_rowW);
  w = rem - __mul24(h, p.rowW);
  h += (w < 0) ? -1 : ((w >= p.rowW) ? 1 : 0);
  w += (w < 0) ? p.rowW : ((w >= p.rowW) ? -p.rowW : 0);
}

// Scalar fall-back of one gathered A element for A-modes 0/1 (used when 16-byte
// loads are not legal, e.g. Cin = 3).
template <int AM>
__device__ __forceinline__ float dj_gather_elem(const DjIgemmParams& p, int pixbase, int rh, int rw, int k) {
  if (k >= p.K) return 0.f;
  int tap = k / p.srcC;
  int c = k - tap * p.srcC;
  int kh = tap / p.KW;
  int kw = tap - kh * p.KW;
  int h, w;
  bool ok;
  if (AM == 0) {
    h = rh + kh * p.dH;
    w = rw + kw * p.dW;
    ok = (unsigned)h < (unsigned)p.srcH && (unsigned)w < (unsigned)p.srcW;
  } else {
    int th = rh - kh * p.dH, tw = rw - kw * p.dW;
    ok = th >= 0 && tw >= 0;
    h = th / p.sH;
    w = tw / p.sW;
    ok = ok && (h * p.sH == th) && (w * p.sW == tw) && h < p.srcH && w < p.srcW;
  }
  if (!ok) return 0.f;
  float v = p.A[(size_t)(pixbase + h * p.srcW + w) * p.ldsrc + c];
  if (p.pro_scale) {
    v = v * p.pro_scale[c] + p.pro_shift[c];
    if (p.pro_relu) v = fmaxf(v, 0.f);
  }
  return v;
}

// Store loop of a tile that lies wholly inside C with the plain row map: no per-element tests.  The general loop below
// decides bias / accumulate / ReLU / atomic per element with wave-uniform branches, ~5 taken branches per value; on a
// 128x128 tile that is ~9000 cycles per wave in which the SIMD issues next to nothing (measured on a 2888-tile 1x1
// convolution: 65 us with the epilogue, 15 us without, whatever K; the stores themselves are 1-10 us of it).
// Here the flags are tested once per four rows.  One code path on purpose: with one specialised
// copy per flag combination the compiler hoists the accumulator reads they share above the dispatch and holds the
// whole tile in VGPRs (128x128 variants: 78 -> 201 VGPRs, three waves per SIMD -> one, K-loop 33 % slower).
template <int TM, int TN>
__device__ __forceinline__ void dj_store_full_tile(float* ubase, unsigned lane_byte, const f32x16 (&acc)[TM][TN],
                                                   const float (&bv)[TN], int ldc, bool beta, bool relu, bool atomic) {
  // ubase: the wave's first row and column (wave-uniform: scalar registers), lane_byte: this lane's byte offset from
  // it -- every access is `scalar base + one 32-bit vector offset`, no per-row vector address arithmetic
  const size_t row_bytes = (size_t)ldc * 4;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v[4][TN];
      char* rowb = reinterpret_cast<char*>(ubase) + (size_t)(i * 32 + 8 * g) * row_bytes;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < TN; ++j) v[q][j] = acc[i][j][4 * g + q] + bv[j];
      if (beta) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < TN; ++j) v[q][j] += *reinterpret_cast<const float*>(rowb + q * row_bytes + j * 128 + lane_byte);
      }
      if (relu) {
        // one v_max each; fmaxf() adds a canonicalising v_max(v, v) in front, and like it this maps NaN to 0
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < TN; ++j) asm("v_max_f32_e32 %0, 0, %1" : "=v"(v[q][j]) : "v"(v[q][j]));
      }
      if (atomic) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < TN; ++j) atomicAdd(reinterpret_cast<float*>(rowb + q * row_bytes + j * 128 + lane_byte), v[q][j]);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < TN; ++j) *reinterpret_cast<float*>(rowb + q * row_bytes + j * 128 + lane_byte) = v[q][j];
      }
      __builtin_amdgcn_sched_barrier(0);   // one group's values live at a time
    }
}

// Full-tile store loop of the kernels that may be handed a C tensor held in 16 bits (dj_igemm_h16.h; dt 0 fp32, 1 fp16,
// 2 bf16; a 16-bit result is never atomic: it is written by one K range).  ONE loop over the row groups with the type
// tested inside a group, after the group's accumulators have been read: a second copy of the loop behind a type test made
// the compiler read the whole tile's accumulators above the test (128x128 forward kernels 164 -> 228 registers, three
// waves per SIMD -> two), as dj_store_full_tile's comment describes for the flag combinations.
// 16-bit form: a lane holds one column of four consecutive rows; neighbouring lanes (columns n, n + 1) swap half of
// their rounded values so that each stores two packed 32-bit words -- the even lane rows 0-1, the odd lane rows 2-3 of
// the group -- instead of four 2-byte pieces.
__device__ __forceinline__ unsigned dj_pack_f16(float lo, float hi) {
  const _Float16 a = (_Float16)lo, b = (_Float16)hi;
  return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}
__device__ __forceinline__ unsigned dj_pack_bf16(float lo, float hi) {
  const __bf16 a = (__bf16)lo, b = (__bf16)hi;
  return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}
__device__ __forceinline__ unsigned dj_pack16(float lo, float hi, int dt) {
  return dt == 1 ? dj_pack_f16(lo, hi) : dj_pack_bf16(lo, hi);
}
__device__ __forceinline__ float dj_widen16(unsigned short u, int dt) {
  return dt == 1 ? (float)__builtin_bit_cast(_Float16, u) : __builtin_bit_cast(float, (unsigned)u << 16);
}

template <int TM, int TN>
__device__ __forceinline__ void dj_store_full_tile_io(char* ubase, unsigned lane_elem, int l31, const f32x16 (&acc)[TM][TN],
                                                      const float (&bv)[TN], int ldc, bool beta, bool relu, bool atomic,
                                                      int dt) {
  // ubase: the wave's first row and column (wave-uniform), lane_elem: this lane's ELEMENT offset from it
  const unsigned es = dt ? 2u : 4u;
  const size_t row_bytes = (size_t)ldc * es;
  const unsigned lane_byte = lane_elem * es;
  const bool odd = (l31 & 1) != 0;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v[4][TN];
      char* rowb = ubase + (size_t)(i * 32 + 8 * g) * row_bytes;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < TN; ++j) v[q][j] = acc[i][j][4 * g + q] + bv[j];
      if (beta) {
        if (dt == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < TN; ++j) v[q][j] += *reinterpret_cast<const float*>(rowb + q * row_bytes + j * 128 + lane_byte);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              v[q][j] += dj_widen16(*reinterpret_cast<const unsigned short*>(rowb + q * row_bytes + j * 64 + lane_byte), dt);
        }
      }
      if (relu) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < TN; ++j) asm("v_max_f32_e32 %0, 0, %1" : "=v"(v[q][j]) : "v"(v[q][j]));
      }
      if (dt == 0) {
        if (atomic) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < TN; ++j) atomicAdd(reinterpret_cast<float*>(rowb + q * row_bytes + j * 128 + lane_byte), v[q][j]);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < TN; ++j) *reinterpret_cast<float*>(rowb + q * row_bytes + j * 128 + lane_byte) = v[q][j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          unsigned p01, p23;
          if (dt == 1) {
            p01 = dj_pack_f16(v[0][j], v[1][j]);
            p23 = dj_pack_f16(v[2][j], v[3][j]);
          } else {
            p01 = dj_pack_bf16(v[0][j], v[1][j]);
            p23 = dj_pack_bf16(v[2][j], v[3][j]);
          }
          // even lane: keeps rows 0-1, hands rows 2-3 to its odd neighbour and gets that lane's rows 0-1
          const unsigned got = (unsigned)__shfl_xor((int)(odd ? p01 : p23), 1);
          const unsigned mine = odd ? p23 : p01;
          // word of a row: (column n, column n + 1) = (even lane's value, odd lane's value)
          const unsigned w0 = odd ? ((got & 0xFFFFu) | (mine << 16)) : ((mine & 0xFFFFu) | (got << 16));
          const unsigned w1 = odd ? ((got >> 16) | (mine & 0xFFFF0000u)) : ((mine >> 16) | (got & 0xFFFF0000u));
          char* base = rowb + (odd ? 2 : 0) * row_bytes + j * 64 + (lane_byte & ~3u);
          *reinterpret_cast<unsigned*>(base) = w0;
          *reinterpret_cast<unsigned*>(base + row_bytes) = w1;
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // one group's values live at a time
    }
}

// 16-bit C, plain store (no accumulate): the tile goes through LDS so that every lane stores 16 contiguous bytes.
// In the accumulator layout a lane owns single elements of 32 columns x 2 row groups: stored from there, a 128x128 tile is
// 32 four-byte stores per lane even with the neighbour-lane packing above, and the stores of a short-K tile cost more than
// its K-loop -- scalar-width stores are issue-bound (MI355X_MICROARCH.md: a dword store costs ~6x a dwordx4 store per byte).
// Each wave writes its TM*32 x TN*32 sub-tile, rounded, into a row-major LDS image of its own (64 ds_write_b16 with
// immediate offsets for a 64x64 sub-tile; pitch + 8 elements keeps the two row groups of a wave on different banks), reads
// it back 8 elements per lane and issues TM*TN*2 global_store_dwordx4.  The K-loop's LDS is free at this point; the
// first 4 KB are left to the statistics reduction that other waves may still be reading.
template <int TM, int TN, int DT>
__device__ __forceinline__ void dj_tile16_to_lds(short* wl, const f32x16 (&acc)[TM][TN], const float (&bv)[TN], bool relu) {
  constexpr int PITCH = TN * 32 + 8;
  const float floor_ = relu ? 0.f : -INFINITY;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const float v = fmaxf(acc[i][j][4 * g + q] + bv[j], floor_);
          unsigned short h;
          if (DT == 1) {
            const _Float16 t = (_Float16)v;
            h = __builtin_bit_cast(unsigned short, t);
          } else {
            const __bf16 t = (__bf16)v;
            h = __builtin_bit_cast(unsigned short, t);
          }
          wl[(i * 32 + 8 * g + q) * PITCH + j * 32] = (short)h;
        }
      __builtin_amdgcn_sched_barrier(0);   // one group's values live at a time
    }
}

template <int TM, int TN>
__device__ __forceinline__ void dj_store_tile16_via_lds(short* wave_lds, char* ubase, int lane, int l31, int lh,
                                                        const f32x16 (&acc)[TM][TN], const float (&bv)[TN], int ldc, bool relu,
                                                        int dt) {
  constexpr int PITCH = TN * 32 + 8;                      // elements
  constexpr int CH = TN * 4, RPP = 64 / CH;               // 16-byte chunks per row, rows per read pass
  short* const wl = wave_lds + (4 * lh) * PITCH + l31;
  // (the type test outside the element loop: inside it the compiler computed both roundings and selected)
  if (dt == 1)
    dj_tile16_to_lds<TM, TN, 1>(wl, acc, bv, relu);
  else
    dj_tile16_to_lds<TM, TN, 2>(wl, acc, bv, relu);
  // the wave's own image: written and read by the same wave (LDS operations of a wave complete in order)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int rr = lane / CH, cc = lane - rr * CH;
  const size_t row_bytes = (size_t)ldc * 2;
#pragma unroll
  for (int ps = 0; ps < TM * 32 / RPP; ++ps) {
    const int row = rr + ps * RPP;
    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(wave_lds + row * PITCH + cc * 8);
    *reinterpret_cast<u32x4_t*>(ubase + (size_t)row * row_bytes + cc * 16) = v;
  }
}

// Shared epilogue: optional per-tile BatchNormalization statistics of the raw accumulator, then
// bias / accumulate / ReLU / (atomic) store with an optional strided-pixel row map.
// BNB: the kernel may be asked for BatchNormalization backward statistics (DjIgemmParams::bnb_z) -- only the
// input-gradient GEMM is; compiled into every kernel the extra code and its six parameters cost the others registers
// (128x128 forward variants 160 -> 180 VGPRs, i.e. three waves per SIMD -> two; the residual-add variants 26-42 SGPR spills)
// IO16: the kernel may be handed 16-bit tensors (c_dt / bnb_zdt: the reduced-precision kernels of dj_igemm_h16.h only)
// LDSB: bytes of dynamic LDS the kernel owns (0: unknown -- no LDS-staged stores)
template <int BM, int BN, int WM, int WN, bool BNB = false, bool IO16 = false, int LDSB = 0>
__device__ __forceinline__ void dj_igemm_epilogue(const DjIgemmParams& p, f32x16 (&acc)[BM / (32 * WM)][BN / (32 * WN)],
                                                  float* smem, int tile_m, int m0, int n0, int ky) {
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lh = lane >> 5;
  if (p.stats || p.bn_acc) {
    // per-column sum and sum of squares of the raw accumulator over this tile's rows
    float* red = smem;  // [2][WM][BN]
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float s = 0.f, q = 0.f;
      if (BNB && p.bnb_z) {
        // BatchNormalization backward statistics of the gradient tile (see DjIgemmParams::bnb_z); rows / columns past
        // the GEMM carry zero accumulators and are not read
        const int n = n0 + (wn * TN + j) * 32 + l31;
        const bool nok = n < p.N;
        const float mu = nok ? p.bnb_mean[n] : 0.f, is = nok ? p.bnb_invstd[n] : 0.f;
        const bool masked = p.bnb_scale != nullptr;
        const float sc = (masked && nok) ? p.bnb_scale[n] : 0.f, sh = (masked && nok) ? p.bnb_shift[n] : 1.f;
        // z through a buffer descriptor, no guards around the loads (a guarded scalar load is an exec-mask save / restore
        // and, as the compiler schedules it, one exposed memory latency per VALUE: the first version of this epilogue cost
        // 28 us on a 2888-tile launch): rows past M lie past the descriptor's extent and read as zero; a column past N
        // reads some other finite element of z into a lane whose accumulators are zero and whose sums are not stored
        const __amdgpu_buffer_rsrc_t rZ = __builtin_amdgcn_make_buffer_rsrc((void*)p.bnb_z, 0, p.bnb_zbytes, 0x00020000);
        const bool z16 = IO16 && p.bnb_zdt != 0;                  // z held as fp16 (wave-uniform)
        const unsigned zes = z16 ? 2u : 4u;
        const unsigned row_b = (unsigned)p.bnb_ldz * zes;
        // eight values at a time: left to itself the compiler issues all 16 * TM * TN loads of the tile first, and the
        // registers that takes cost the whole kernel a wave per SIMD (128x128: 160 -> 184 VGPRs)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const unsigned lane_b = (unsigned)(m0 + (wm * TM + i) * 32 + 4 * lh) * row_b + (unsigned)n * zes;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            float zz[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              const int zoff = (int)(lane_b + (unsigned)((r & 3) + 8 * (2 * h + (r >> 2))) * row_b);
              if (z16)
                zz[r] = (float)__builtin_bit_cast(_Float16, (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rZ, zoff, 0, 0));
              else
                zz[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rZ, zoff, 0, 0));
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              const float g = (zz[r] * sc + sh > 0.f) ? acc[i][j][8 * h + r] : 0.f;
              s += g;
              q += g * (zz[r] - mu) * is;
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else {
        // two values per instruction (v_pk_add_f32 / v_pk_fma_f32): the even and the odd rows of a lane are summed apart
        // and joined at the end -- 64 accumulators cost 64 vector instructions instead of 128
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const f32x2 v = {acc[i][j][r], acc[i][j][r + 1]};
            s2 += v;
            q2 += v * v;
          }
        s = s2.x + s2.y;
        q = q2.x + q2.y;
      }
      s += __shfl_xor(s, 32);
      q += __shfl_xor(q, 32);
      if (lh == 0) {
        int col = (wn * TN + j) * 32 + l31;
        red[(0 * WM + wm) * BN + col] = s;
        red[(1 * WM + wm) * BN + col] = q;
      }
    }
    __syncthreads();
    if (p.bn_acc) {
      for (int col = tid; col < BN; col += 256) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          s += red[(0 * WM + w) * BN + col];
          q += red[(1 * WM + w) * BN + col];
        }
        int n = n0 + col;
        if (n < p.N) {
          double* acc = p.bn_acc + (size_t)(blockIdx.x % (unsigned)p.bn_replicas) * 2 * p.N;
          unsafeAtomicAdd(acc + n, (double)s);
          unsafeAtomicAdd(acc + p.N + n, (double)q);
        }
      }
    }
    // one partial row per 64 GEMM rows, whatever the tile shape: [ceil(M/64)][2][N]
    constexpr int GROUPS = BM / 64;          // 64-row groups in this tile
    constexpr int WPG = WM / GROUPS;         // wave rows per group
    for (int idx = tid; p.stats && idx < GROUPS * BN; idx += 256) {
      int g = idx / BN, col = idx - g * BN;
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < WPG; ++w) {
        s += red[(0 * WM + g * WPG + w) * BN + col];
        q += red[(1 * WM + g * WPG + w) * BN + col];
      }
      int n = n0 + col;
      if (n < p.N && m0 + 64 * g < p.M) {
        size_t row = (size_t)(m0 / 64 + g);
        p.stats[(row * 2 + 0) * p.N + n] = s;
        p.stats[(row * 2 + 1) * p.N + n] = q;
      }
    }
  }

  float* const Cb = p.C + (size_t)ky * p.slab_stride;   // ky: this workgroup's K chunk
  if (IO16) {
    // the reduced-precision kernels: C in fp32 (possibly atomic / slabs) or in 16 bits (one K range, never atomic)
    const int dt = p.c_dt;
    const unsigned es = dt ? 2u : 4u;
    char* const Cc = reinterpret_cast<char*>(p.C) + (size_t)ky * p.slab_stride * 4;
    if (p.cmap == 0 && m0 + BM <= p.M && n0 + BN <= p.N && (dt == 0 || (p.ldc & 1) == 0)) {
      float bv[TN];
      const bool add_bias = p.bias && (!p.atomic || ky == 0);
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[j] = add_bias ? p.bias[n0 + (wn * TN + j) * 32 + l31] : 0.f;
      const int uwave = __builtin_amdgcn_readfirstlane(wave);
      const int uwm = uwave / WN, uwn = uwave % WN;
      char* ubase = Cc + ((size_t)(m0 + uwm * TM * 32) * p.ldc + (n0 + uwn * TN * 32)) * es;
      const unsigned lane_elem = (unsigned)(4 * lh * p.ldc + l31);
      constexpr int WAVE_H = TM * 32 * (TN * 32 + 8);          // elements of a wave's LDS image
      constexpr bool LDS_FITS = LDSB >= 4096 + WM * WN * WAVE_H * 2;
      if (LDS_FITS && dt != 0 && !p.beta && (p.ldc & 7) == 0 && ((((size_t)p.C) | ((size_t)n0 * 2)) & 15) == 0) {
        short* wave_lds = reinterpret_cast<short*>(smem) + 2048 + uwave * WAVE_H;
        dj_store_tile16_via_lds<TM, TN>(wave_lds, ubase, lane, l31, lh, acc, bv, p.ldc, p.relu != 0, dt);
      } else {
        dj_store_full_tile_io<TM, TN>(ubase, lane_elem, l31, acc, bv, p.ldc, p.beta != 0 && !p.atomic, p.relu != 0 && !p.atomic,
                                      p.atomic != 0, dt);
      }
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m >= p.M) continue;
          size_t rowoff;
          if (p.cmap == 0) {
            rowoff = (size_t)m * p.ldc;
          } else {
            int img = m / (p.cgH * p.cgW);
            int rem = m - img * (p.cgH * p.cgW);
            int h = rem / p.cgW;
            int w = rem - h * p.cgW;
            rowoff = (size_t)((img * p.cH + h * p.cS) * p.cW + w * p.cS) * p.ldc;
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            int n = n0 + (wn * TN + j) * 32 + l31;
            if (n >= p.N) continue;
            float v = acc[i][j][r];
            if (dt == 0) {
              float* dst = reinterpret_cast<float*>(Cc) + rowoff + n;
              if (p.atomic) {
                if (p.bias && ky == 0) v += p.bias[n];
                atomicAdd(dst, v);
              } else {
                if (p.bias) v += p.bias[n];
                if (p.beta) v += *dst;
                if (p.relu) v = fmaxf(v, 0.f);
                *dst = v;
              }
            } else {
              unsigned short* dst = reinterpret_cast<unsigned short*>(Cc + (rowoff + n) * 2);
              if (p.bias) v += p.bias[n];
              if (p.beta) v += dj_widen16(*dst, dt);
              if (p.relu) v = fmaxf(v, 0.f);
              *dst = (unsigned short)(dj_pack16(v, 0.f, dt) & 0xFFFFu);
            }
          }
        }
      }
    }
  } else if (p.cmap == 0 && m0 + BM <= p.M && n0 + BN <= p.N) {
    float bv[TN];
    const bool add_bias = p.bias && (!p.atomic || ky == 0);
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[j] = add_bias ? p.bias[n0 + (wn * TN + j) * 32 + l31] : 0.f;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const int uwm = uwave / WN, uwn = uwave % WN;
    float* ubase = Cb + (size_t)(m0 + uwm * TM * 32) * p.ldc + (n0 + uwn * TN * 32);
    const unsigned lane_byte = (unsigned)(4 * lh * p.ldc + l31) * 4u;
    dj_store_full_tile<TM, TN>(ubase, lane_byte, acc, bv, p.ldc, p.beta != 0 && !p.atomic, p.relu != 0 && !p.atomic, p.atomic != 0);
  } else {
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m >= p.M) continue;
      size_t rowoff;
      if (p.cmap == 0) {
        rowoff = (size_t)m * p.ldc;
      } else {
        int img = m / (p.cgH * p.cgW);
        int rem = m - img * (p.cgH * p.cgW);
        int h = rem / p.cgW;
        int w = rem - h * p.cgW;
        rowoff = (size_t)((img * p.cH + h * p.cS) * p.cW + w * p.cS) * p.ldc;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        int n = n0 + (wn * TN + j) * 32 + l31;
        if (n >= p.N) continue;
        float v = acc[i][j][r];
        float* dst = Cb + rowoff + n;
        if (p.atomic) {
          if (p.bias && ky == 0) v += p.bias[n];
          atomicAdd(dst, v);
        } else {
          if (p.bias) v += p.bias[n];
          if (p.beta) v += *dst;
          if (p.relu) v = fmaxf(v, 0.f);
          *dst = v;
        }
      }
    }
  }
  }
  if (p.bn_acc) {
    // Take a ticket once our atomics have been performed; the workgroup that takes the last one sees every other
    // workgroup's sums.  Not __threadfence(): on gfx950 an agent-scope release is `buffer_wbl2 sc1` + `buffer_inv sc1`,
    // i.e. every workgroup would write back and invalidate its XCD's L2 (measured: 1172 -> 880 img/s on the step).
    // Only the accumulators need ordering, they are touched by device-scope atomics / sc1 accesses alone, and those
    // are complete when vmcnt reaches zero.
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      unsigned t = atomicAdd(p.bn_ticket, 1u);
      s_last = (t == gridDim.x * gridDim.y - 1u) ? 1 : 0;
    }
    __syncthreads();
    if (s_last) {
      for (int c = tid; c < p.N; c += 256) {
        double S = 0.0, Q = 0.0;
        for (int r = 0; r < p.bn_replicas; ++r) {
          double* acc = p.bn_acc + (size_t)r * 2 * p.N;
          S += __hip_atomic_load(acc + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          Q += __hip_atomic_load(acc + p.N + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(acc + c, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(acc + p.N + c, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        double m = S / p.bn_count;
        double var = Q / p.bn_count - m * m;
        if (var < 0.0) var = 0.0;
        // the sums were taken on the accumulator before the conv bias: it shifts the mean only
        double mean = m + (p.bias ? (double)p.bias[c] : 0.0);
        float invstd = (float)(1.0 / sqrt(var + (double)p.bn_eps));
        float sc = p.bn_gamma[c] * invstd;
        p.bn_scale[c] = sc;
        p.bn_shift[c] = p.bn_beta[c] - (float)mean * sc;
        p.bn_save_mean[c] = (float)mean;
        p.bn_save_invstd[c] = invstd;
        if (p.bn_moving_mean) {
          double unbiased = var * (p.bn_count / (p.bn_count > 1.0 ? p.bn_count - 1.0 : 1.0));
          p.bn_moving_mean[c] = p.bn_moving_mean[c] * p.bn_momentum + (float)mean * (1.f - p.bn_momentum);
          p.bn_moving_var[c] = p.bn_moving_var[c] * p.bn_momentum + (float)unbiased * (1.f - p.bn_momentum);
        }
      }
      if (tid == 0) __hip_atomic_store(p.bn_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int BM, int BN, int WM, int WN, int AM, int BMD>
__global__ __launch_bounds__(256) void dj_igemm_kernel(const DjIgemmParams p) {
  using Cfg = DjIgemmCfg<BM, BN, WM, WN, AM, BMD>;
  constexpr int TM = Cfg::TM, TN = Cfg::TN, NA = Cfg::NA, NB = Cfg::NB;
  constexpr int LDA_S = Cfg::LDA_S, LDB_S = Cfg::LDB_S;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lh = lane >> 5;

  const int tiles_n = (p.N + BN - 1) / BN;
  const int tile_m = blockIdx.x / tiles_n;
  const int tile_n = blockIdx.x - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kbeg = blockIdx.y * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const int nk = (kend - kbeg + DJ_BK - 1) / DJ_BK;

  // ---------------- per-thread staging state ----------------
  // A, k-contiguous tiles (modes 0/1): thread -> (row r0 + 32 j, 16-byte chunk ac)
  const int ac = tid & 7, ar0 = tid >> 3;
  int a_pix[NA], a_rh[NA], a_rw[NA];
  // A, m-contiguous tiles (mode 2): thread -> (k row akr0 + AKSTEP j, chunk acm)
  constexpr int AKSTEP = 1024 / BM;
  const int acm = tid % (BM / 4), akr0 = tid / (BM / 4);
  int a2_c = 0, a2_dh = 0, a2_dw = 0;
  bool a2_ok = false;
  f32x4 a2_sc = {1.f, 1.f, 1.f, 1.f}, a2_sh = {0.f, 0.f, 0.f, 0.f};
  if (AM != 2) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      int m = m0 + ar0 + 32 * j;
      if (m < p.M) {
        int img = m / (p.rowH * p.rowW);
        int rem = m - img * (p.rowH * p.rowW);
        int h = rem / p.rowW;
        int w = rem - h * p.rowW;
        a_pix[j] = img * p.srcH * p.srcW;
        if (AM == 0) {
          a_rh[j] = h * p.sH - p.pT;
          a_rw[j] = w * p.sW - p.pL;
        } else {
          a_rh[j] = h + p.pT;
          a_rw[j] = w + p.pL;
        }
      } else {
        a_pix[j] = 0;
        a_rh[j] = -(1 << 28);
        a_rw[j] = -(1 << 28);
      }
    }
  } else {
    int mm = m0 + 4 * acm;
    a2_ok = mm < p.M;
    int tap = mm / p.srcC;
    a2_c = mm - tap * p.srcC;
    int kh = tap / p.KW;
    int kw = tap - kh * p.KW;
    a2_dh = kh * p.dH - p.pT;
    a2_dw = kw * p.dW - p.pL;
    if (a2_ok && p.pro_scale && p.vecA) {
      a2_sc = dj_ld4(p.pro_scale + a2_c);
      a2_sh = dj_ld4(p.pro_shift + a2_c);
    }
  }
  // B, n-contiguous tiles (mode 0): thread -> (k row bkr0 + BKSTEP j, chunk bcn)
  constexpr int BKSTEP = 1024 / BN;
  const int bcn = tid % (BN / 4), bkr0 = tid / (BN / 4);
  // B, k-contiguous tiles (mode 1): thread -> (row n = br0 + 32 j, chunk bc)
  const int bc = tid & 7, br0 = tid >> 3;

  f32x4 ra[NA], rb[NB];

  auto load_tiles = [&](int kcur) {
    // ---- A ----
    if (AM != 2) {
      int k = kcur + 4 * ac;
      if (p.vecA) {
        bool kok = k < kend;
        int tap = k / p.srcC;
        int c = k - tap * p.srcC;
        int kh = tap / p.KW;
        int kw = tap - kh * p.KW;
        int dh = kh * p.dH, dw = kw * p.dW;
        f32x4 sc, sh;
        if (p.pro_scale && kok) {
          sc = dj_ld4(p.pro_scale + c);
          sh = dj_ld4(p.pro_shift + c);
        }
#pragma unroll
        for (int j = 0; j < NA; ++j) {
          int h, w;
          bool ok;
          if (AM == 0) {
            h = a_rh[j] + dh;
            w = a_rw[j] + dw;
            ok = kok && (unsigned)h < (unsigned)p.srcH && (unsigned)w < (unsigned)p.srcW;
          } else {
            int th = a_rh[j] - dh, tw = a_rw[j] - dw;
            ok = kok && th >= 0 && tw >= 0;
            if (p.sH == 1 && p.sW == 1) {
              h = th;
              w = tw;
            } else {
              h = th / p.sH;
              w = tw / p.sW;
              ok = ok && (h * p.sH == th) && (w * p.sW == tw);
            }
            ok = ok && h < p.srcH && w < p.srcW;
          }
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (ok) {
            v = dj_ld4(p.A + (size_t)(a_pix[j] + h * p.srcW + w) * p.ldsrc + c);
            if (p.pro_scale) {
              v = v * sc + sh;
              if (p.pro_relu) {
                v.x = fmaxf(v.x, 0.f);
                v.y = fmaxf(v.y, 0.f);
                v.z = fmaxf(v.z, 0.f);
                v.w = fmaxf(v.w, 0.f);
              }
            }
          }
          ra[j] = v;
        }
      } else {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
          f32x4 v;
          v.x = (k + 0 < kend) ? dj_gather_elem<AM>(p, a_pix[j], a_rh[j], a_rw[j], k + 0) : 0.f;
          v.y = (k + 1 < kend) ? dj_gather_elem<AM>(p, a_pix[j], a_rh[j], a_rw[j], k + 1) : 0.f;
          v.z = (k + 2 < kend) ? dj_gather_elem<AM>(p, a_pix[j], a_rh[j], a_rw[j], k + 2) : 0.f;
          v.w = (k + 3 < kend) ? dj_gather_elem<AM>(p, a_pix[j], a_rh[j], a_rw[j], k + 3) : 0.f;
          ra[j] = v;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        int kp = kcur + akr0 + AKSTEP * j;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (kp < kend && a2_ok) {
          int img = kp / (p.rowH * p.rowW);
          int rem = kp - img * (p.rowH * p.rowW);
          int oh = rem / p.rowW;
          int ow = rem - oh * p.rowW;
          int h = oh * p.sH + a2_dh, w = ow * p.sW + a2_dw;
          if (!p.vecA || ((unsigned)h < (unsigned)p.srcH && (unsigned)w < (unsigned)p.srcW)) {
            if (p.vecA) {
              const float* src = p.A + (size_t)((img * p.srcH + h) * p.srcW + w) * p.ldsrc;
              v = dj_ld4(src + a2_c);
              if (p.pro_scale) {
                v = v * a2_sc + a2_sh;
                if (p.pro_relu) {
                  v.x = fmaxf(v.x, 0.f);
                  v.y = fmaxf(v.y, 0.f);
                  v.z = fmaxf(v.z, 0.f);
                  v.w = fmaxf(v.w, 0.f);
                }
              }
            } else {
              // element-wise: each of the 4 rows m'+e has its own (tap, c)
              float t[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                int mm = m0 + 4 * acm + e;
                float x = 0.f;
                if (mm < p.M) {
                  int tap = mm / p.srcC;
                  int c = mm - tap * p.srcC;
                  int kh = tap / p.KW;
                  int kw = tap - kh * p.KW;
                  int hh = oh * p.sH + kh * p.dH - p.pT, ww = ow * p.sW + kw * p.dW - p.pL;
                  if ((unsigned)hh < (unsigned)p.srcH && (unsigned)ww < (unsigned)p.srcW) {
                    x = p.A[(size_t)((img * p.srcH + hh) * p.srcW + ww) * p.ldsrc + c];
                    if (p.pro_scale) {
                      x = x * p.pro_scale[c] + p.pro_shift[c];
                      if (p.pro_relu) x = fmaxf(x, 0.f);
                    }
                  }
                }
                t[e] = x;
              }
              v.x = t[0];
              v.y = t[1];
              v.z = t[2];
              v.w = t[3];
            }
          }
        }
        ra[j] = v;
      }
    }
    // ---- B ----
    if (BMD == 0) {
      int n = n0 + 4 * bcn;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        int k = kcur + bkr0 + BKSTEP * j;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < kend) {
          const float* src = p.B + (size_t)k * p.ldb + n;
          if (p.vecB && n + 3 < p.N) {
            v = dj_ld4(src);
          } else {
            if (n + 0 < p.N) v.x = src[0];
            if (n + 1 < p.N) v.y = src[1];
            if (n + 2 < p.N) v.z = src[2];
            if (n + 3 < p.N) v.w = src[3];
          }
        }
        rb[j] = v;
      }
    } else {
      int k = kcur + 4 * bc;
      int tap = k / p.srcC;
      int c = k - tap * p.srcC;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        int n = n0 + br0 + 32 * j;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n < p.N) {
          if (p.vecB) {
            if (k < kend) v = dj_ld4(p.B + (size_t)tap * p.bTapStride + (size_t)n * p.ldb + c);
          } else {
            float t[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              int ke = k + e;
              float x = 0.f;
              if (ke < kend) {
                int tp = ke / p.srcC;
                int ce = ke - tp * p.srcC;
                x = p.B[(size_t)tp * p.bTapStride + (size_t)n * p.ldb + ce];
              }
              t[e] = x;
            }
            v.x = t[0];
            v.y = t[1];
            v.z = t[2];
            v.w = t[3];
          }
        }
        rb[j] = v;
      }
    }
  };

  auto store_tiles = [&](int buf) {
    float* sA = smem + buf * Cfg::STAGE_FLOATS;
    float* sB = sA + Cfg::A_FLOATS;
    if (AM != 2) {
#pragma unroll
      for (int j = 0; j < NA; ++j) *reinterpret_cast<f32x4*>(sA + (ar0 + 32 * j) * LDA_S + 4 * ac) = ra[j];
    } else {
#pragma unroll
      for (int j = 0; j < NA; ++j) *reinterpret_cast<f32x4*>(sA + (akr0 + AKSTEP * j) * LDA_S + 4 * acm) = ra[j];
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

  auto compute = [&](int buf) {
    const float* sA = smem + buf * Cfg::STAGE_FLOATS;
    const float* sB = sA + Cfg::A_FLOATS;
#pragma unroll
    for (int kk = 0; kk < DJ_BK / 8; ++kk) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        int row = (wm * TM + i) * 32 + l31;
        if (Cfg::A_KC) {
          af[i] = *reinterpret_cast<const f32x4*>(sA + row * LDA_S + kk * 8 + lh * 4);
        } else {
          const float* q = sA + (kk * 8 + lh * 4) * LDA_S + row;
          af[i].x = q[0];
          af[i].y = q[LDA_S];
          af[i].z = q[2 * LDA_S];
          af[i].w = q[3 * LDA_S];
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        int col = (wn * TN + j) * 32 + l31;
        if (Cfg::B_KC) {
          bf[j] = *reinterpret_cast<const f32x4*>(sB + col * LDB_S + kk * 8 + lh * 4);
        } else {
          const float* q = sB + (kk * 8 + lh * 4) * LDB_S + col;
          bf[j].x = q[0];
          bf[j].y = q[LDB_S];
          bf[j].z = q[2 * LDB_S];
          bf[j].w = q[3 * LDB_S];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
    }
  };

  // ---------------- main loop: register prefetch + LDS double buffer ----------------
  if (nk > 0) {
    load_tiles(kbeg);
    store_tiles(0);
  }
  __syncthreads();
  int buf = 0;
  for (int kt = 0; kt < nk; ++kt) {
    bool more = kt + 1 < nk;
    if (more) load_tiles(kbeg + (kt + 1) * DJ_BK);
    compute(buf);
    if (more) store_tiles(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  dj_igemm_epilogue<BM, BN, WM, WN, (AM == 1 && BMD == 1)>(p, acc, smem, tile_m, m0, n0, (int)blockIdx.y);
}
