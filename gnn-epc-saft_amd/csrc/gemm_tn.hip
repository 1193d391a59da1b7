// Weight-gradient GEMM ("TN"):  dW[n][k] = sum_m dY[m][n] * A[m][k]
// for the backward of every Linear on the path (autograd through
// /root/reference/gnnepcsaft/train/models.py:105-135).
//
// The contraction runs over the ROWS m (nodes / edges / graphs: 1e4..1e6) and the output is a small
// [n_out, k] weight matrix, so the rows are cut into chunks of kTnChunk and every workgroup produces
// one 64x64 output tile for one chunk into a slab [chunk][n_out][k]; a second kernel sums the slabs in
// chunk order.  No atomics: bitwise reproducible.  Both operands are staged in their natural
// row-major layout (row = m); the 32x32x2 f32 MFMA reads its fragments column-wise (ds_read_b32,
// conflict-free: consecutive lanes read consecutive n / k).
//
// The A operand is virtual, as in gemm.hip:
//   TnPlain  : a row-major matrix (optionally ReLU'd: extra pre/post layers)
//   (the PNAConv update input cat[x_i, A, A*amp_i, A*att_i] has its own kernel, k_gemm_tn_postfold)
//   TnOneHot : concatenated one-hot rows of categorical columns (embedding tables: dE = OneHot^T dX)
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "x6.hpp"

namespace gs {

constexpr int kTnChunk = 256;   // rows per slab (small: the output tiles alone cannot fill 256 CUs)
constexpr int kTnBK = 32;       // rows per LDS stage
constexpr int kTnTile = 64;     // output tile (n and k)
constexpr int kTnLd = kTnTile + 4;

struct TnPlain {
  const float *a;
  int64_t lda;
  int relu;
  int k;
  __device__ __forceinline__ f32x4 load(int64_t m, int kk) const {
    f32x4 v = gs_ld4(a + m * lda + (kk < k ? kk : 0));
    if (relu) {
      v.x = fmaxf(v.x, 0.f);
      v.y = fmaxf(v.y, 0.f);
      v.z = fmaxf(v.z, 0.f);
      v.w = fmaxf(v.w, 0.f);
    }
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    return kk < k ? v : zero;
  }
};

struct TnOneHot {
  const int64_t *idx;  // [N, ncol]
  int ncol;
  int32_t off[GNNSAFT_MAX_TABLES + 1];  // first concatenated row of every table
  __device__ __forceinline__ f32x4 load(int64_t m, int kk) const {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < ncol; ++t) {
      int64_t r = idx[m * ncol + t];
      const int dim = off[t + 1] - off[t];
      r = (r < 0 || r >= dim) ? 0 : r;  // as the forward clamps
      const int c = off[t] + (int)r - kk;
      v.x += c == 0 ? 1.f : 0.f;
      v.y += c == 1 ? 1.f : 0.f;
      v.z += c == 2 ? 1.f : 0.f;
      v.w += c == 3 ? 1.f : 0.f;
    }
    return v;
  }
};

// Y_CLASS: the left operand is not a matrix but the one-hot encoding of an int32 class id per row
// (`dy` reinterpreted): out[c][:] = sum of the A rows of class c, i.e. a segmented reduction by an arbitrary key on
// the matrix cores, in the fixed summation order of the slab scheme (no atomics).
// `direct` (single slab, i.e. few rows: edge classes, small batches): the tile goes straight to its destination
// (SlabOut row blocks, leading dimension ld_out) -- no slab, no k_sum_slabs launch.
struct TnDirect {
  SlabOut so;
  int64_t ld_out;
  int on, accumulate;
};
template <class AProv, bool Y_CLASS>
__global__ __launch_bounds__(256) void k_gemm_tn(const float *__restrict__ dy, int64_t ldy, AProv ap, int64_t m,
                                                 int n_out, int k, float *__restrict__ slabs, int64_t rows_per_z,
                                                 TnDirect direct) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * kTnBK * kTnLd];  // [buf][dy | a][32][68]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;  // 2 x 2 waves, 32 x 32 each
  const int n0 = blockIdx.y * kTnTile, k0 = blockIdx.x * kTnTile;
  const int64_t m_beg = (int64_t)blockIdx.z * rows_per_z;
  int64_t m_end = m_beg + rows_per_z;
  if (m_end > m) m_end = m;

  // staging: 32 rows x 16 float4; thread -> (row tid>>4 (+16), float4 column tid&15)
  const int sc = (tid & 15) * 4;
  const int sr = tid >> 4;
  f32x4 ry[2], ra[2];
  auto fetch = [&](int64_t mrow0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int64_t mm = mrow0 + sr + 16 * j;
      const bool ok = mm < m_end;
      mm = ok ? mm : m_end - 1;
      const int nn = n0 + sc;
      f32x4 vy;
      if (Y_CLASS) {
        const int c = reinterpret_cast<const int32_t *>(dy)[mm] - nn;
        vy = f32x4{c == 0 ? 1.f : 0.f, c == 1 ? 1.f : 0.f, c == 2 ? 1.f : 0.f, c == 3 ? 1.f : 0.f};
      } else {
        vy = gs_ld4(dy + mm * ldy + (nn < n_out ? nn : 0));
      }
      f32x4 va = ap.load(mm, k0 + sc);
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      ry[j] = (ok && nn < n_out) ? vy : zero;  // rows past the chunk contribute nothing
      ra[j] = ok ? va : zero;
    }
  };
  auto stash = [&](int buf) {
    float *ys = lds + buf * 2 * kTnBK * kTnLd;
    float *as = ys + kTnBK * kTnLd;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      gs_st4(ys + (sr + 16 * j) * kTnLd + sc, ry[j]);
      gs_st4(as + (sr + 16 * j) * kTnLd + sc, ra[j]);
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  const int64_t steps = (m_end - m_beg + kTnBK - 1) / kTnBK;
  fetch(m_beg);
  stash(0);
  __syncthreads();
  for (int64_t s = 0; s < steps; ++s) {
    // unconditional: rows past the slab come back as zeros (see gemm.hip on branches in the staging path)
    fetch(m_beg + (s + 1) * kTnBK);
    const float *ys = lds + (s & 1) * 2 * kTnBK * kTnLd + wn * 32 + (lane & 31);
    const float *as = ys + kTnBK * kTnLd - wn * 32 + wk * 32;
#pragma unroll
    for (int q = 0; q < kTnBK / 2; ++q) {
      const int row = 2 * q + (lane >> 5);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ys[row * kTnLd], as[row * kTnLd], acc, 0, 0, 0);
    }
    stash((s + 1) & 1);
    __syncthreads();
  }
  // C/D: col = lane & 31 (k), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (n)
  float *slab = slabs + (int64_t)blockIdx.z * n_out * (int64_t)k;
  const int kc = k0 + wk * 32 + (lane & 31);
  if (direct.on) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int nr = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (nr < n_out && kc < k) {
        const int64_t blk = nr / direct.so.rows_per_block;
        float *o = direct.so.base[blk] + (nr - blk * direct.so.rows_per_block) * direct.ld_out + kc;
        *o = direct.accumulate ? *o + acc[r] : acc[r];
      }
    }
    return;
  }
  if (n0 + kTnTile <= n_out && k0 + kTnTile <= k) {  // full tile: unguarded stores, issued back to back (gemm.hip)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int nr = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      slab[(int64_t)nr * k + kc] = acc[r];
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int nr = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (nr < n_out && kc < k) slab[(int64_t)nr * k + kc] = acc[r];
    }
  }
}

// ---- wide variant: WN x WK waves, every wave owns a 64 x 64 block of the output (2 x 2 MFMA tiles, four independent
// accumulators), so a workgroup covers 64 WN x 64 WK of dW and every staged operand element feeds 64 WK (dY) or
// 64 WN (A) multiply-adds instead of 64: with 4 x 4 waves (the register file of a CU) a [256, 256] gradient is one
// workgroup tile and both operands are read exactly once.  Measured (tools/tn_tune.py, MI355X): 4 x 4 wins where the
// problem is large -- C3 (164 k rows): [256,256] 267 -> 229 us, [1024,256] 964 -> 788 us (94 / 109 TFLOP/s) -- and
// every wide grid LOSES on the 10-20 k row problems of C2 / C5 (fewer, longer waves per CU hide less latency:
// [128,128] 27 -> 33 us), so launch_tn uses it for large problems only.
template <int WN, int WK, class AProv, bool Y_CLASS>
__global__ __launch_bounds__(64 * WN * WK) void k_gemm_tn_wide(const float *__restrict__ dy, int64_t ldy, AProv ap,
                                                               int64_t m, int n_out, int k, float *__restrict__ slabs,
                                                               int64_t rows_per_z, TnDirect direct) {
  constexpr int TN_ = 64 * WN, TK_ = 64 * WK;
  constexpr int LDY = TN_ + 4, LDA = TK_ + 4;
  constexpr int YL = 8 / WK, AL = 8 / WN;              // float4 loads per thread and stage
  constexpr int STAGE = kTnBK * (LDY + LDA);
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [2][ dy 32 x LDY | a 32 x LDA ]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wn = wave / WK, wk = wave % WK;
  const int n0 = blockIdx.y * TN_, k0 = blockIdx.x * TK_;
  const int64_t m_beg = (int64_t)blockIdx.z * rows_per_z;
  int64_t m_end = m_beg + rows_per_z;
  if (m_end > m) m_end = m;

  const int yr = tid / (16 * WN), yc = (tid % (16 * WN)) * 4;   // + j * 4 WK rows
  const int ar = tid / (16 * WK), ac = (tid % (16 * WK)) * 4;   // + j * 4 WN rows
  f32x4 ry[YL], ra[AL];
  auto fetch = [&](int64_t mrow0) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < YL; ++j) {
      int64_t mm = mrow0 + yr + j * 4 * WK;
      const bool ok = mm < m_end;
      mm = ok ? mm : m_end - 1;
      const int nn = n0 + yc;
      f32x4 vy;
      if (Y_CLASS) {
        const int c = reinterpret_cast<const int32_t *>(dy)[mm] - nn;
        vy = f32x4{c == 0 ? 1.f : 0.f, c == 1 ? 1.f : 0.f, c == 2 ? 1.f : 0.f, c == 3 ? 1.f : 0.f};
      } else {
        vy = gs_ld4(dy + mm * ldy + (nn < n_out ? nn : 0));
      }
      ry[j] = (ok && nn < n_out) ? vy : zero;  // rows past the chunk contribute nothing
    }
#pragma unroll
    for (int j = 0; j < AL; ++j) {
      int64_t mm = mrow0 + ar + j * 4 * WN;
      const bool ok = mm < m_end;
      mm = ok ? mm : m_end - 1;
      const f32x4 va = ap.load(mm, k0 + ac);
      ra[j] = ok ? va : zero;
    }
  };
  auto stash = [&](int buf) {
    float *ys = lds + buf * STAGE;
    float *as = ys + kTnBK * LDY;
#pragma unroll
    for (int j = 0; j < YL; ++j) gs_st4(ys + (yr + j * 4 * WK) * LDY + yc, ry[j]);
#pragma unroll
    for (int j = 0; j < AL; ++j) gs_st4(as + (ar + j * 4 * WN) * LDA + ac, ra[j]);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int64_t steps = (m_end - m_beg + kTnBK - 1) / kTnBK;
  fetch(m_beg);
  stash(0);
  __syncthreads();
  for (int64_t s = 0; s < steps; ++s) {
    fetch(m_beg + (s + 1) * kTnBK);  // unconditional: rows past the slab come back as zeros
    const float *ys = lds + (s & 1) * STAGE + wn * 64 + (lane & 31);
    const float *as = lds + (s & 1) * STAGE + kTnBK * LDY + wk * 64 + (lane & 31);
#pragma unroll
    for (int q = 0; q < kTnBK / 2; ++q) {
      const int row = 2 * q + (lane >> 5);
      const float y0 = ys[row * LDY], y1 = ys[row * LDY + 32];
      const float a0 = as[row * LDA], a1 = as[row * LDA + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(y0, a0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(y0, a1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(y1, a0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(y1, a1, acc[1][1], 0, 0, 0);
    }
    stash((s + 1) & 1);
    __syncthreads();
  }
  // C/D: col = lane & 31 (k), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (n)
  float *slab = slabs + (int64_t)blockIdx.z * n_out * (int64_t)k;
  const bool full = n0 + TN_ <= n_out && k0 + TK_ <= k;  // block-uniform: unguarded stores, issued back to back
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kc = k0 + wk * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int nr = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (direct.on) {
          if (nr < n_out && kc < k) {
            const int64_t blk = nr / direct.so.rows_per_block;
            float *o = direct.so.base[blk] + (nr - blk * direct.so.rows_per_block) * direct.ld_out + kc;
            *o = direct.accumulate ? *o + acc[i][j][r] : acc[i][j][r];
          }
        } else if (full) {
          slab[(int64_t)nr * k + kc] = acc[i][j][r];
        } else if (nr < n_out && kc < k) {
          slab[(int64_t)nr * k + kc] = acc[i][j][r];
        }
      }
    }
}

// ---- split-bf16 variant (x6.hpp): the contraction runs over the ROWS, so the bf16 fragments of both operands are
// k-strided in the row-major matrices.  The staging pass transposes in registers: a staging thread loads the SAME
// float4 column group of FOUR consecutive rows, splits the 16 values into (hi, mid, lo) and writes, per column and
// plane, the four rows' bf16 as one 8-byte word into an LDS image [column][16 rows] (48-byte rows as in gemm.hip, so
// the fragments are plain ds_read_b128).  A stage is 16 rows (one MFMA k step), double-buffered.  4 x 4 waves of
// 64 x 64 (the 256 x 256 tile of the wide f32 kernel): the first TN_ threads stage dY, the next TK_ the operand (waves
// 0-7; two of the four waves of every SIMD).  Per 32 rows and 64 x 64 of output a wave issues 48 bf16 MFMAs of 8
// passes where the f32 kernel issues 64 of 16 passes.
// Measured (tools/tn_tune.py, C3: [163 907, 256]^T [., 256] and [., 1024]^T [., 256]): f32 230 / 793 us, this kernel
// 180 / 567 us.  16 waves leave 128 registers per lane (64 of them accumulators).  Tried and dropped: the two halves
// of the workgroup staging alternate stages (loads two stages ahead of their use) -- 196 / 650 us with the turn taken
// under a condition inside one loop, 214 / 710 us with one straight-line loop per half; a second register set for the
// same two-stage distance -- spills, 228 / 760 us; 8 waves of 64 x 128 (no spills, every thread stages) --
// 234 / 706 us; a 64 x 64-tile variant (six MFMAs per barrier) -- slower than the f32 64 x 64 kernel.
constexpr int kTnX6BK = 16;
template <int WN, int WK, int TIN, int TIK, class AProv, bool Y_CLASS>
__global__ __launch_bounds__(64 * WN * WK) void k_gemm_tn_x6(const float *__restrict__ dy, int64_t ldy, AProv ap,
                                                             int64_t m, int n_out, int k, float *__restrict__ slabs,
                                                             int64_t rows_per_z, TnDirect direct) {
  constexpr int NT = 64 * WN * WK;
  constexpr int TN_ = 32 * TIN * WN, TK_ = 32 * TIK * WK;
  constexpr int PLANE = (TN_ + TK_) * kX6RowBytes;   // bytes per bf16 plane: dY columns, then operand columns
  constexpr int STAGE = 3 * PLANE;
  static_assert(TN_ + TK_ <= NT && (TN_ % 64) == 0 && (TK_ % 64) == 0, "one staging thread per 4 columns x 4 rows");
  extern __shared__ __attribute__((aligned(16))) char lds_x6[];   // [2][3 planes][TN_ + TK_][48 B]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wn = wave / WK, wk = wave % WK;
  const int n0 = blockIdx.y * TN_, k0 = blockIdx.x * TK_;
  const int64_t m_beg = (int64_t)blockIdx.z * rows_per_z;
  int64_t m_end = m_beg + rows_per_z;
  if (m_end > m) m_end = m;

  // staging role (wave-uniform): 0 = dY, 1 = operand, 2 = none
  const int role = tid < TN_ ? 0 : (tid < TN_ + TK_ ? 1 : 2);
  const int su = role == 0 ? tid : tid - TN_;
  const int sq = su & 3;          // rows 4 sq .. 4 sq + 3 of the stage
  const int sc4 = su >> 2;        // float4 column group of the tile
  f32x4 rr[4];
  auto fetch = [&](int64_t mrow0) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    if (role == 0) {
      const int nn = n0 + sc4 * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int64_t mm = mrow0 + 4 * sq + j;
        const bool ok = mm < m_end;
        mm = ok ? mm : m_end - 1;
        f32x4 vy;
        if (Y_CLASS) {
          const int c = reinterpret_cast<const int32_t *>(dy)[mm] - nn;
          vy = f32x4{c == 0 ? 1.f : 0.f, c == 1 ? 1.f : 0.f, c == 2 ? 1.f : 0.f, c == 3 ? 1.f : 0.f};
        } else {
          vy = gs_ld4(dy + mm * ldy + (nn < n_out ? nn : 0));
        }
        rr[j] = (ok && nn < n_out) ? vy : zero;   // rows past the chunk contribute nothing
      }
    } else if (role == 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int64_t mm = mrow0 + 4 * sq + j;
        const bool ok = mm < m_end;
        mm = ok ? mm : m_end - 1;
        const f32x4 va = ap.load(mm, k0 + sc4 * 4);
        rr[j] = ok ? va : zero;
      }
    }
  };
  auto stash = [&](int buf) {
    if (role == 2) return;
    char *base = lds_x6 + buf * STAGE + ((role == 0 ? 0 : TN_) + sc4 * 4) * kX6RowBytes + sq * 8;
#pragma unroll
    for (int e = 0; e < 4; ++e) {   // column 4 sc4 + e: its four rows as one 8-byte word per plane
      uint32_t h[4], md[4], l[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) gs_split3(rr[j][e], h[j], md[j], l[j]);
      char *p = base + e * kX6RowBytes;
      *reinterpret_cast<uint2 *>(p) = uint2{gs_pack_hi16(h[0], h[1]), gs_pack_hi16(h[2], h[3])};
      *reinterpret_cast<uint2 *>(p + PLANE) = uint2{gs_pack_hi16(md[0], md[1]), gs_pack_hi16(md[2], md[3])};
      *reinterpret_cast<uint2 *>(p + 2 * PLANE) = uint2{gs_pack_hi16(l[0], l[1]), gs_pack_hi16(l[2], l[3])};
    }
  };

  f32x16 acc[TIN][TIK];
#pragma unroll
  for (int i = 0; i < TIN; ++i)
#pragma unroll
    for (int j = 0; j < TIK; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int64_t steps = (m_end - m_beg + kTnX6BK - 1) / kTnX6BK;
  fetch(m_beg);
  stash(0);
  __syncthreads();
  const int frag = (lane & 31) * kX6RowBytes + (lane >> 5) * 16;
  for (int64_t s = 0; s < steps; ++s) {
    fetch(m_beg + (s + 1) * kTnX6BK);  // unconditional: rows past the slab come back as zeros
    const char *xs = lds_x6 + (s & 1) * STAGE;
    const char *ya = xs + (wn * 32 * TIN) * kX6RowBytes + frag;
    const char *aa = xs + (TN_ + wk * 32 * TIK) * kX6RowBytes + frag;
    bf16x8 yf[TIN][3], af[TIK][3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int i = 0; i < TIN; ++i)
        yf[i][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(ya + p * PLANE + i * 32 * kX6RowBytes));
#pragma unroll
      for (int j = 0; j < TIK; ++j)
        af[j][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(aa + p * PLANE + j * 32 * kX6RowBytes));
    }
    constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};   // smallest products first
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < TIN; ++i)
#pragma unroll
        for (int j = 0; j < TIK; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[i][pa[t]], af[j][pb[t]], acc[i][j], 0, 0, 0);
    stash((s + 1) & 1);
    __syncthreads();
  }
  // C/D: col = lane & 31 (k), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (n)
  float *slab = slabs + (int64_t)blockIdx.z * n_out * (int64_t)k;
  const bool full = n0 + TN_ <= n_out && k0 + TK_ <= k;  // block-uniform: unguarded stores, issued back to back
#pragma unroll
  for (int i = 0; i < TIN; ++i)
#pragma unroll
    for (int j = 0; j < TIK; ++j) {
      const int kc = k0 + wk * 32 * TIK + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int nr = n0 + wn * 32 * TIN + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (direct.on) {
          if (nr < n_out && kc < k) {
            const int64_t blk = nr / direct.so.rows_per_block;
            float *o = direct.so.base[blk] + (nr - blk * direct.so.rows_per_block) * direct.ld_out + kc;
            *o = direct.accumulate ? *o + acc[i][j][r] : acc[i][j][r];
          }
        } else if (full) {
          slab[(int64_t)nr * k + kc] = acc[i][j][r];
        } else if (nr < n_out && kc < k) {
          slab[(int64_t)nr * k + kc] = acc[i][j][r];
        }
      }
    }
}

// ---- per-class row sums on the bf16 matrix cores: out[c][:] = sum of the rows of A whose class id is c.
// (The backward's dR = OneHot(class)^T dm behind the edge-class tables, PNAConv edge_encoder + the edge columns of
// pre_nns, /root/reference/gnnepcsaft/train/models.py:59,128.)  The one-hot operand is exact in bf16 and has no mid /
// lo part, so only THREE of the six split products exist -- (1, hi) (1, mid) (1, lo) -- at 16x the f32 matrix-core rate:
// the sum is then a streaming read of A (1 GB per layer at BASELINE config 3; the f32 64 x 64 kernel spent 412-438 us on
// it = 2.4 TB/s, matrix-core-bound).  A workgroup of 4 waves owns 64 classes x 256 columns and a chunk of rows, two
// workgroups per CU (80 KB of LDS each); a stage is 16 rows (one MFMA k16 step), every thread stages 4 rows x 4 columns
// of A (one 1-KiB row piece per wave instruction) TWO stages ahead (64 KB in flight per CU), transposing in registers
// as k_gemm_tn_x6 does; the one-hot image is built from the class ids (one 16-byte load per thread).  Rows in slab
// order, slabs summed in order (k_sum_slabs_batched): bitwise reproducible, no atomics.
constexpr int kClsX3Cols = 256, kClsX3Rows = 16, kClsX3Classes = 64;
__global__ __launch_bounds__(256, 2) void k_class_sum_x3(const int32_t *__restrict__ cls, const float *__restrict__ a,
                                                         int64_t lda, int64_t m, int n_out, int k,
                                                         float *__restrict__ slabs, int64_t rows_per_z) {
  constexpr int YIMG = kClsX3Classes * kX6RowBytes;        // one-hot image: [class][16 rows] bf16, hi plane only
  constexpr int XPLANE = kClsX3Cols * kX6RowBytes;         // one plane of the A image: [col][16 rows]
  constexpr int STAGE = YIMG + 3 * XPLANE;
  extern __shared__ __attribute__((aligned(16))) char lds_cs[];   // [2 buffers][Y | X hi | X mid | X lo]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;                 // class tile (32 classes), column tiles 4 wk .. 4 wk + 3
  const int k0 = blockIdx.x * kClsX3Cols;
  const int64_t m_beg = (int64_t)blockIdx.z * rows_per_z;
  int64_t m_end = m_beg + rows_per_z;
  if (m_end > m) m_end = m;

  // A staging: column group cg (4 columns), row group rg (4 rows) of the 16-row stage
  const int cg = tid & 63, rg = tid >> 6;
  const float *acol = a + k0 + 4 * cg;
  // one-hot staging: class yc, rows 4 yq .. 4 yq + 3
  const int yc = tid >> 2, yq = tid & 3;
  f32x4 rr[2][4];
  typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
  i32x4 rc[2];
  auto fetch = [&](int64_t mrow0, f32x4(&r)[4], i32x4 &c) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int64_t mm = mrow0 + 4 * rg + j;
      const bool ok = mm < m_end;
      mm = ok ? mm : m_end - 1;
      const f32x4 v = gs_ld4(acol + mm * lda);
      r[j] = ok ? v : zero;                                // rows past the chunk contribute nothing
    }
    const int64_t y0 = mrow0 + 4 * yq;                     // (m_beg and the stage are multiples of 4: aligned)
    const i32x4 none = {-1, -1, -1, -1};
    if (y0 + 3 < m_end) {
      c = *reinterpret_cast<const i32x4 *>(cls + y0);
    } else {
      c = none;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (y0 + j < m_end) c[j] = cls[y0 + j];
    }
  };
  auto pack = [](uint32_t x0, uint32_t x1) { return __builtin_amdgcn_perm(x1, x0, 0x07060302u); };   // upper halves, x0 low
  auto stash = [&](int buf, const f32x4(&r)[4], const i32x4 &c) {
    char *st = lds_cs + buf * STAGE;
    {
      char *base = st + YIMG + (4 * cg) * kX6RowBytes + rg * 8;
#pragma unroll
      for (int e = 0; e < 4; ++e) {   // column 4 cg + e: its four rows as one 8-byte word per plane
        uint32_t h[4], md[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) gs_split3(r[j][e], h[j], md[j], l[j]);
        char *p = base + e * kX6RowBytes;
        *reinterpret_cast<uint2 *>(p) = uint2{pack(h[0], h[1]), pack(h[2], h[3])};
        *reinterpret_cast<uint2 *>(p + XPLANE) = uint2{pack(md[0], md[1]), pack(md[2], md[3])};
        *reinterpret_cast<uint2 *>(p + 2 * XPLANE) = uint2{pack(l[0], l[1]), pack(l[2], l[3])};
      }
    }
    {
      const uint32_t one = 0x3f80u;   // bf16 1.0
      const uint32_t w0 = (c[0] == yc ? one : 0u) | (c[1] == yc ? one << 16 : 0u);
      const uint32_t w1 = (c[2] == yc ? one : 0u) | (c[3] == yc ? one << 16 : 0u);
      *reinterpret_cast<uint2 *>(st + yc * kX6RowBytes + yq * 8) = uint2{w0, w1};
    }
  };

  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int frag = (lane & 31) * kX6RowBytes + (lane >> 5) * 16;
  auto compute = [&](int buf) {
    const char *xs = lds_cs + buf * STAGE;
    const bf16x8 yf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(xs + (wn * 32) * kX6RowBytes + frag));
    bf16x8 af[4][3];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        af[j][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(xs + YIMG + p * XPLANE + (wk * 128 + j * 32) * kX6RowBytes + frag));
#pragma unroll
    for (int p = 2; p >= 0; --p)   // smallest products first: (1, lo) (1, mid) (1, hi)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf, af[j][p], acc[j], 0, 0, 0);
  };

  const int64_t steps = (m_end - m_beg + kClsX3Rows - 1) / kClsX3Rows;
  fetch(m_beg, rr[0], rc[0]);
  fetch(m_beg + kClsX3Rows, rr[1], rc[1]);
  stash(0, rr[0], rc[0]);
  __syncthreads();
  for (int64_t s = 0; s < steps; s += 2) {   // loads two stages ahead of their stash (unconditional: zeros past the end)
    fetch(m_beg + (s + 2) * kClsX3Rows, rr[0], rc[0]);
    compute(0);
    stash(1, rr[1], rc[1]);
    __syncthreads();
    fetch(m_beg + (s + 3) * kClsX3Rows, rr[1], rc[1]);
    if (s + 1 < steps) compute(1);
    stash(0, rr[0], rc[0]);
    __syncthreads();
  }
  // C/D: col = lane & 31 (column of A), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (class)
  float *slab = slabs + (int64_t)blockIdx.z * n_out * (int64_t)k;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int kc = k0 + wk * 128 + j * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (c < n_out) slab[(int64_t)c * k + kc] = acc[j][r];
    }
  }
}

// ---- post_nns weight gradient through the degree tiles (the backward twin of the degree-folded update).
// dW_t = du_t^T cat[x, A_t, amp A_t, att A_t] has K = 13F, but amp / att are constant over a degree tile, so a
// workgroup contracts only [x | A_t] (K = 5F) and, after every tile (rows of ONE in-degree), folds the tile's
// partial product into three accumulators:  id += S,  amp += amp(d) S,  att += att(d) S  (48 VALU fmas per lane
// per tile against 32+ MFMAs) -- 10 N F^2 FLOP per layer instead of 26 N F^2.  Rows are addressed through the
// degree permutation; grid.z walks groups of `tiles_per_z` tiles, grid.y = (n tile, tower).
struct TnFoldArgs {
  const float *du;        // [N, F]  (tower t uses columns t F/2 ..)
  const float *x;         // [N, F]
  const float *agg;       // [N, 2, 4F]
  const int32_t *perm;    // slot -> node
  const int32_t *tiles;   // [num_tiles][4] = degree, first slot, rows, -
  const int32_t *num_tiles;
  const float *avg;       // device [1]
  int f;
  int tile_rows;
  int tiles_per_z;
};

__global__ __launch_bounds__(256) void k_gemm_tn_postfold(TnFoldArgs a, float *__restrict__ slabs) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * kTnBK * kTnLd];  // [buf][du | operand][32][68]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;  // 2 x 2 waves, 32 x 32 each
  const int f = a.f, n_out = f / 2, kfold = 5 * f, kfull = 13 * f;
  const int n_tiles_n = (n_out + kTnTile - 1) / kTnTile;
  const int tower = blockIdx.y / n_tiles_n;
  const int n0 = (blockIdx.y - tower * n_tiles_n) * kTnTile, k0 = blockIdx.x * kTnTile;
  const bool agg_part = k0 >= f;  // a 64-wide k tile lies wholly in the x block or in the aggregate block (F % 64 == 0)
  const int nt = a.num_tiles[0];
  const int t_beg = blockIdx.z * a.tiles_per_z;
  int t_end = t_beg + a.tiles_per_z;
  if (t_end > nt) t_end = nt;

  const int sc = (tid & 15) * 4;
  const int sr = tid >> 4;
  const int nn = n0 + sc;
  const float *dy_col = a.du + tower * n_out + (nn < n_out ? nn : 0);
  const float *op_col = agg_part ? a.agg + tower * 4 * f + (k0 - f) + sc : a.x + k0 + sc;
  const int64_t op_ld = agg_part ? 8 * (int64_t)f : f;
  const float avgv = a.avg[0];

  f32x16 acc, acc_id, acc_amp, acc_att;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = acc_id[r] = acc_amp[r] = acc_att[r] = 0.f;

  f32x4 ry[2], ra[2];
  for (int t = t_beg; t < t_end; ++t) {
    const int deg = a.tiles[4 * t + 0], slot0 = a.tiles[4 * t + 1], count = a.tiles[4 * t + 2];
    auto fetch = [&](int row0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int r = row0 + sr + 16 * j;
        const bool ok = r < count;
        r = ok ? r : count - 1;
        const int64_t node = a.perm[slot0 + r];
        const f32x4 vy = gs_ld4(dy_col + node * f);
        const f32x4 va = gs_ld4(op_col + node * op_ld);
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        ry[j] = (ok && nn < n_out) ? vy : zero;
        ra[j] = ok ? va : zero;
      }
    };
    auto stash = [&](int buf) {
      float *ys = lds + buf * 2 * kTnBK * kTnLd;
      float *as = ys + kTnBK * kTnLd;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        gs_st4(ys + (sr + 16 * j) * kTnLd + sc, ry[j]);
        gs_st4(as + (sr + 16 * j) * kTnLd + sc, ra[j]);
      }
    };
    const int steps = (count + kTnBK - 1) / kTnBK;
    __syncthreads();  // the previous tile's last stage has been consumed by every wave
    fetch(0);
    stash(0);
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
      fetch((s + 1) * kTnBK);  // unconditional: rows past the tile come back as zeros
      const float *ys = lds + (s & 1) * 2 * kTnBK * kTnLd + wn * 32 + (lane & 31);
      const float *as = ys + kTnBK * kTnLd - wn * 32 + wk * 32;
#pragma unroll
      for (int q = 0; q < kTnBK / 2; ++q) {
        const int row = 2 * q + (lane >> 5);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ys[row * kTnLd], as[row * kTnLd], acc, 0, 0, 0);
      }
      stash((s + 1) & 1);
      __syncthreads();
    }
    if (agg_part) {  // fold this degree's partial product
      float amp, att;
      degree_scalers(deg, avgv, amp, att);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc_id[r] += acc[r];
        acc_amp[r] += amp * acc[r];
        acc_att[r] += att * acc[r];
        acc[r] = 0.f;
      }
    }
  }
  // slab layout = the unfolded gradient [2 towers][F/2][13F], so the ordinary slab sum finishes the job.
  // C/D: col = lane & 31 (k), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (n)
  float *slab = slabs + ((int64_t)blockIdx.z * 2 + tower) * n_out * (int64_t)kfull;
  const int kc = k0 + wk * 32 + (lane & 31);
  const bool full = n0 + kTnTile <= n_out;  // kfold = 5F is a multiple of the 64-wide k tile
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int nr = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    float *o = slab + (int64_t)(full || nr < n_out ? nr : 0) * kfull;
    if (full) {  // block-uniform: unguarded stores, issued back to back (gemm.hip)
      if (!agg_part) {
        o[kc] = acc[r];
      } else {
        o[kc] = acc_id[r];
        o[kc + 4 * f] = acc_amp[r];
        o[kc + 8 * f] = acc_att[r];
      }
    } else if (nr < n_out && kc < kfold) {
      if (!agg_part) {
        o[kc] = acc[r];
      } else {
        o[kc] = acc_id[r];
        o[kc + 4 * f] = acc_amp[r];
        o[kc + 8 * f] = acc_att[r];
      }
    }
  }
}

// ---- the same gradient for LARGE batches, split-bf16 and wide (x6.hpp, k_gemm_tn_x6): a workgroup owns ONE chunk of
// rows of ONE in-degree (the permutation is sorted by degree; chunk z -> (degree, first slot, rows) is derived from
// the degree histogram by every workgroup itself) and writes the chunk's plain partial product S_z = du_t^T [x | A_t]
// ([F/2, 5F] per tower) to its slab.  The degree scalers are applied when the slabs are summed (k_sum_slabs_fold):
// id += S_z, amp += amp(d_z) S_z, att += att(d_z) S_z -- no fold accumulators in the GEMM, 5 F^2 floats per slab
// instead of 13 F^2 and ~100 slabs instead of ~330 at C3.
struct TnFoldWideArgs {
  const float *du;        // [N, F]
  const float *x;         // [N, F]
  const float *agg;       // [N, 2, 4F]
  const int32_t *perm;    // slot -> node, sorted by in-degree
  const int32_t *hist;    // [kDegreeBuckets] nodes per in-degree
  const float *avg;       // device [1]
  int f;
  int chunk_rows;
};

// chunk z -> its degree, first slot of the permutation and row count; false past the last chunk
__device__ __forceinline__ bool fold_chunk(const int32_t *__restrict__ hist, int chunk_rows, int z, int &deg,
                                           int &slot0, int &count) {
  int slot = 0;
  for (int d = 0; d < kDegreeBuckets; ++d) {
    const int c = hist[d];
    const int nc = (c + chunk_rows - 1) / chunk_rows;
    if (z < nc) {
      deg = d;
      slot0 = slot + z * chunk_rows;
      const int left = c - z * chunk_rows;
      count = left < chunk_rows ? left : chunk_rows;
      return true;
    }
    z -= nc;
    slot += c;
  }
  return false;
}

template <int WN, int WK, int TIN, int TIK>
__global__ __launch_bounds__(64 * WN * WK) void k_gemm_tn_postfold_x6(TnFoldWideArgs a, float *__restrict__ slabs) {
  constexpr int NT = 64 * WN * WK;
  constexpr int TN_ = 32 * TIN * WN, TK_ = 32 * TIK * WK;
  constexpr int PLANE = (TN_ + TK_) * kX6RowBytes;
  constexpr int STAGE = 3 * PLANE;
  static_assert(TN_ + TK_ <= NT && (TN_ % 64) == 0 && (TK_ % 64) == 0, "one staging thread per 4 columns x 4 rows");
  extern __shared__ __attribute__((aligned(16))) char lds_x6[];   // [2][3 planes][TN_ + TK_][48 B]
  int deg, slot0, count;
  if (!fold_chunk(a.hist, a.chunk_rows, (int)blockIdx.z, deg, slot0, count)) return;   // (block-uniform)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wn = wave / WK, wk = wave % WK;
  const int f = a.f, n_out = f / 2, kfold = 5 * f;
  const int n_tiles_n = n_out / TN_;
  const int tower = blockIdx.y / n_tiles_n;
  const int n0 = (blockIdx.y - tower * n_tiles_n) * TN_, k0 = blockIdx.x * TK_;
  const bool agg_part = k0 >= f;   // a k tile lies wholly in the x block or in the aggregate block (F % TK_ == 0)

  const int role = tid < TN_ ? 0 : (tid < TN_ + TK_ ? 1 : 2);   // wave-uniform: dY | operand | none
  const int su = role == 0 ? tid : tid - TN_;
  const int sq = su & 3;
  const int sc4 = su >> 2;
  const float *col = role == 0 ? a.du + tower * n_out + n0 + sc4 * 4
                               : (agg_part ? a.agg + tower * 4 * f + (k0 - f) + sc4 * 4 : a.x + k0 + sc4 * 4);
  const int64_t ld = role == 0 ? f : (agg_part ? 8 * (int64_t)f : f);
  const int32_t *perm = a.perm + slot0;
  int nid[4];   // the nodes of this thread's four rows of the NEXT stage to fetch
  auto fetch_ids = [&](int row0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = row0 + 4 * sq + j;
      nid[j] = perm[r < count ? r : count - 1];
    }
  };
  f32x4 rr[4];
  auto fetch = [&](int row0) {   // rows row0 .. row0 + 15 of the chunk (their node ids are in nid)
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    if (role != 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 v = gs_ld4(col + (int64_t)nid[j] * ld);
        rr[j] = row0 + 4 * sq + j < count ? v : zero;
      }
    }
  };
  auto stash = [&](int buf) {
    if (role == 2) return;
    char *base = lds_x6 + buf * STAGE + ((role == 0 ? 0 : TN_) + sc4 * 4) * kX6RowBytes + sq * 8;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      uint32_t h[4], md[4], l[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) gs_split3(rr[j][e], h[j], md[j], l[j]);
      char *p = base + e * kX6RowBytes;
      *reinterpret_cast<uint2 *>(p) = uint2{gs_pack_hi16(h[0], h[1]), gs_pack_hi16(h[2], h[3])};
      *reinterpret_cast<uint2 *>(p + PLANE) = uint2{gs_pack_hi16(md[0], md[1]), gs_pack_hi16(md[2], md[3])};
      *reinterpret_cast<uint2 *>(p + 2 * PLANE) = uint2{gs_pack_hi16(l[0], l[1]), gs_pack_hi16(l[2], l[3])};
    }
  };

  f32x16 acc[TIN][TIK];
#pragma unroll
  for (int i = 0; i < TIN; ++i)
#pragma unroll
    for (int j = 0; j < TIK; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int steps = (count + kTnX6BK - 1) / kTnX6BK;
  fetch_ids(0);
  fetch(0);
  fetch_ids(kTnX6BK);
  stash(0);
  __syncthreads();
  const int frag = (lane & 31) * kX6RowBytes + (lane >> 5) * 16;
  for (int s = 0; s < steps; ++s) {
    fetch((s + 1) * kTnX6BK);        // (node ids loaded one stage earlier; rows past the chunk come back as zeros)
    fetch_ids((s + 2) * kTnX6BK);
    const char *xs = lds_x6 + (s & 1) * STAGE;
    const char *ya = xs + (wn * 32 * TIN) * kX6RowBytes + frag;
    const char *aa = xs + (TN_ + wk * 32 * TIK) * kX6RowBytes + frag;
    bf16x8 yf[TIN][3], af[TIK][3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int i = 0; i < TIN; ++i)
        yf[i][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(ya + p * PLANE + i * 32 * kX6RowBytes));
#pragma unroll
      for (int j = 0; j < TIK; ++j)
        af[j][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(aa + p * PLANE + j * 32 * kX6RowBytes));
    }
    constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};   // smallest products first
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < TIN; ++i)
#pragma unroll
        for (int j = 0; j < TIK; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[i][pa[t]], af[j][pb[t]], acc[i][j], 0, 0, 0);
    stash((s + 1) & 1);
    __syncthreads();
  }
  // slab z: [2 towers][F/2][5F].  C/D: col = lane & 31 (k), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (n)
  float *slab = slabs + ((int64_t)blockIdx.z * 2 + tower) * n_out * (int64_t)kfold;
#pragma unroll
  for (int i = 0; i < TIN; ++i)
#pragma unroll
    for (int j = 0; j < TIK; ++j) {
      const int kc = k0 + wk * 32 * TIK + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int nr = n0 + wn * 32 * TIN + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        slab[(int64_t)nr * kfold + kc] = acc[i][j][r];   // (tiles are full: F/2 % TN_ == 0, 5F % TK_ == 0)
      }
    }
}

// dW_t[n][0:F] = sum_z S_z[.][0:F];  dW_t[n][F + j | 5F + j | 9F + j] = sum_z (1 | amp(d_z) | att(d_z)) S_z[.][F + j]
// over the chunks in ascending order (fixed: bitwise reproducible).  One thread per float4 of the folded [2][F/2][5F].
constexpr int kFoldSumMaxChunks = 1024;
__global__ __launch_bounds__(256) void k_sum_slabs_fold(const float *__restrict__ slabs, const int32_t *__restrict__ hist,
                                                        const float *__restrict__ avg, int chunk_rows, int f,
                                                        float *__restrict__ dw0, float *__restrict__ dw1) {
  __shared__ float s_amp[kFoldSumMaxChunks], s_att[kFoldSumMaxChunks];
  __shared__ int s_total;
  if (threadIdx.x == 0) {
    const float avgv = avg[0];
    int z = 0;
    for (int d = 0; d < kDegreeBuckets; ++d) {
      const int nc = (hist[d] + chunk_rows - 1) / chunk_rows;
      float amp, att;
      degree_scalers(d, avgv, amp, att);
      for (int c = 0; c < nc && z < kFoldSumMaxChunks; ++c, ++z) {
        s_amp[z] = amp;
        s_att[z] = att;
      }
    }
    s_total = z;
  }
  __syncthreads();
  const int total = s_total;
  const int n_out = f / 2, kfold = 5 * f, per_row4 = kfold / 4;
  const int64_t per_slab = 2 * (int64_t)n_out * kfold;
  const int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i4 >= per_slab / 4) return;
  const int64_t row = i4 / per_row4;            // tower * n_out + n
  const int k = (int)(i4 - row * per_row4) * 4;
  const float *src = slabs + i4 * 4;
  f32x4 sid = {0.f, 0.f, 0.f, 0.f}, samp = sid, satt = sid;
  const bool agg_part = k >= f;
  for (int z0 = 0; z0 < total; z0 += 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int z = z0 + u < total ? z0 + u : total - 1;
      v[u] = gs_ld4(src + (int64_t)z * per_slab);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (z0 + u < total) {
        sid += v[u];
        if (agg_part) {
          const float wa = s_amp[z0 + u], wt = s_att[z0 + u];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            samp[e] = fmaf(wa, v[u][e], samp[e]);
            satt[e] = fmaf(wt, v[u][e], satt[e]);
          }
        }
      }
  }
  const int tower = (int)(row / n_out);
  float *o = (tower == 0 ? dw0 : dw1) + (row - (int64_t)tower * n_out) * (13 * (int64_t)f) + k;
  gs_st4(o, sid);
  if (agg_part) {
    gs_st4(o + 4 * f, samp);
    gs_st4(o + 8 * f, satt);
  }
}

// out[i] (+)= sum over the slabs; 8 slab lanes per output float4, combined in a fixed order
// The dense result [rows, cols] may be scattered by row blocks: block b = row / rows_per_block goes to base[b]
// (several weight gradients that share one TN GEMM, e.g. the four [F,F] blocks of the message weights).
__global__ __launch_bounds__(256) void k_sum_slabs(const float *__restrict__ slabs, int64_t per_slab, int64_t chunks,
                                                   SlabOut so, int64_t ld_out, int cols, int accumulate,
                                                   int64_t slab_stride) {
  __shared__ f32x4 red[8][32];
  const int il = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int64_t i4 = ((int64_t)blockIdx.x * 32 + il) * 4;
  const bool ok = i4 < per_slab;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (ok)
    for (int64_t c = sl; c < chunks; c += 8) s += gs_ld4(slabs + c * slab_stride + i4);
  red[sl][il] = s;
  __syncthreads();
  if (sl != 0 || !ok) return;
  for (int o = 1; o < 8; ++o) s += red[o][il];
  const int64_t r = i4 / cols, cc = i4 - r * cols;  // slab is dense [rows, cols]; out may be a column block
  const int64_t blk = r / so.rows_per_block;
  float *o = so.base[blk] + (r - blk * so.rows_per_block) * ld_out + cc;
  if (accumulate) s += gs_ld4(o);
  gs_st4(o, s);
}

// the same reduction for up to kMaxSlabJobs results in one launch (SlabQueue): blockIdx.y = job
struct SlabJobs {
  SlabJob j[kMaxSlabJobs];
};
__global__ __launch_bounds__(256) void k_sum_slabs_batched(SlabJobs jobs) {
  __shared__ f32x4 red[8][32];
  const SlabJob &jb = jobs.j[blockIdx.y];
  const int il = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int64_t i4 = ((int64_t)blockIdx.x * 32 + il) * 4;
  if ((int64_t)blockIdx.x * 128 >= jb.per_slab) return;  // block-uniform: the grid is sized for the largest job
  const bool ok = i4 < jb.per_slab;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (ok)
    for (int64_t c = sl; c < jb.chunks; c += 8) s += gs_ld4(jb.slabs + c * jb.per_slab + i4);
  red[sl][il] = s;
  __syncthreads();
  if (sl != 0 || !ok) return;
  for (int o = 1; o < 8; ++o) s += red[o][il];
  const int64_t r = i4 / jb.cols, cc = i4 - r * jb.cols;
  const int64_t blk = r / jb.so.rows_per_block;
  float *o = jb.so.base[blk] + (r - blk * jb.so.rows_per_block) * jb.ld_out + cc;
  if (jb.accumulate) s += gs_ld4(o);
  gs_st4(o, s);
}

// out[c][r] = in[r][c] for a batch of small matrices (weights): LDS-tiled 32x32
struct TransposeBatch {
  const float *in[kMaxTransposeBatch];
  float *out[kMaxTransposeBatch];
  int64_t ld_in[kMaxTransposeBatch];
  int64_t ld_out[kMaxTransposeBatch];
  int rows[kMaxTransposeBatch];
  int cols[kMaxTransposeBatch];
};
__global__ __launch_bounds__(256) void k_transpose(TransposeBatch tb) {
  __shared__ float t[32][33];
  const float *in = tb.in[blockIdx.z];
  float *out = tb.out[blockIdx.z];
  const int64_t ldi = tb.ld_in[blockIdx.z], ldo = tb.ld_out[blockIdx.z];
  const int rows = tb.rows[blockIdx.z], cols = tb.cols[blockIdx.z];
  if ((int)blockIdx.y * 32 >= rows || (int)blockIdx.x * 32 >= cols) return;  // grid is sized for the largest entry
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = r0 + ty + 8 * j, c = c0 + tx;
    t[ty + 8 * j][tx] = (r < rows && c < cols) ? in[r * ldi + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = c0 + ty + 8 * j, r = r0 + tx;
    if (c < cols && r < rows) out[c * ldo + r] = t[tx][ty + 8 * j];
  }
}

// column sums (bias gradients): partial[chunk][col] then fixed-order sum.  A workgroup owns 32 columns x
// kColChunk rows: 8 float4 column groups x 32 row lanes, every thread's 8 rows loaded back to back (independent
// loads in flight; the first version walked 128 rows per thread one dependent load at a time: 20 us per call).
constexpr int kColChunk = 256;
// `direct` (single chunk): write the sums to their destinations at once, no second launch
struct ColsumOut {
  float *base[4];
  int cols_per_block;
};
__global__ __launch_bounds__(256) void k_colsum_partial(const float *__restrict__ a, int64_t lda, int64_t m, int cols,
                                                        float *__restrict__ partial, ColsumOut co, int direct,
                                                        int accumulate) {
  __shared__ double red[32][33];
  const int cg = threadIdx.x & 7, rl = threadIdx.x >> 3;  // float4 column group, row lane
  const int c0 = blockIdx.x * 32 + cg * 4;
  const int64_t m_beg = (int64_t)blockIdx.y * kColChunk;
  const bool vec = c0 + 3 < cols;
  const int cc = c0 < cols ? c0 : 0;
  f32x4 v[kColChunk / 32];
#pragma unroll
  for (int j = 0; j < kColChunk / 32; ++j) {
    int64_t r = m_beg + rl + 32 * j;
    const bool ok = r < m;
    r = ok ? r : m - 1;
    const float *src = a + r * lda;
    if (vec) {
      v[j] = gs_ld4(src + cc);
    } else {  // ragged last column group (cols % 4 != 0: the [G, num_para] head)
      v[j].x = c0 + 0 < cols ? src[c0 + 0] : 0.f;
      v[j].y = c0 + 1 < cols ? src[c0 + 1] : 0.f;
      v[j].z = c0 + 2 < cols ? src[c0 + 2] : 0.f;
      v[j].w = 0.f;
    }
    if (!ok) v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // bias gradients in front of a train-mode BatchNorm are exactly zero: keep the sums exact (f64)
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
  for (int j = 0; j < kColChunk / 32; ++j) {
    s0 += (double)v[j].x;
    s1 += (double)v[j].y;
    s2 += (double)v[j].z;
    s3 += (double)v[j].w;
  }
  red[rl][cg * 4 + 0] = s0;
  red[rl][cg * 4 + 1] = s1;
  red[rl][cg * 4 + 2] = s2;
  red[rl][cg * 4 + 3] = s3;
  __syncthreads();
  if (threadIdx.x < 32) {
    const int c = blockIdx.x * 32 + threadIdx.x;
    double s = 0.0;
#pragma unroll
    for (int o = 0; o < 32; ++o) s += red[o][threadIdx.x];
    if (c < cols) {
      if (direct) {
        const int blk = c / co.cols_per_block;
        float *o = co.base[blk] + (c - blk * co.cols_per_block);
        *o = accumulate ? *o + (float)s : (float)s;
      } else {
        partial[(int64_t)blockIdx.y * cols + c] = (float)s;
      }
    }
  }
}
// 32 columns x 8 chunk lanes per workgroup
__global__ __launch_bounds__(256) void k_colsum_final(const float *__restrict__ partial, int64_t chunks, int cols,
                                                      ColsumOut co, int accumulate) {
  __shared__ double red[8][33];
  const int cl = threadIdx.x & 31, lane = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int cc = c < cols ? c : cols - 1;
  double s = 0.0;
  for (int64_t j0 = lane; j0 < chunks; j0 += 64) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t j = j0 + 8 * u;
      t[u] = j < chunks ? partial[j * cols + cc] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (double)t[u];
  }
  red[lane][cl] = s;
  __syncthreads();
  if (lane == 0 && c < cols) {
    for (int o = 1; o < 8; ++o) s += red[o][cl];
    const int blk = c / co.cols_per_block;
    float *o = co.base[blk] + (c - blk * co.cols_per_block);
    *o = accumulate ? *o + (float)s : (float)s;
  }
}

// wave grid for an [n_out, k] gradient over m rows: 4 x 4 waves of 64 x 64 for large problems (see above), else the
// 64 x 64 four-wave kernel
static inline int tn_waves(int64_t m, int n_out, int k) { return (m >= 32768 && n_out > 128 && k > 128) ? 4 : 1; }

template <int WN, int WK, class AProv, bool Y_CLASS>
static void launch_tn_wide(const float *dy, int64_t ldy, const AProv &ap, int64_t m, int n_out, int k, float *slabs,
                           int64_t rows_per_z, int64_t chunks, const TnDirect &direct, hipStream_t st) {
  constexpr int TN_ = 64 * WN, TK_ = 64 * WK;
  constexpr size_t lds_bytes = 2 * (size_t)kTnBK * (TN_ + TK_ + 8) * 4;
  static std::atomic<unsigned long long> lds_raised{0};   // > 64 KB of dynamic LDS: once per kernel and device
  (void)gs_raise_dynamic_lds(reinterpret_cast<const void *>(&k_gemm_tn_wide<WN, WK, AProv, Y_CLASS>), lds_bytes,
                             lds_raised);
  const dim3 grid((unsigned)gs_ceil_div(k, TK_), (unsigned)gs_ceil_div(n_out, TN_), (unsigned)chunks);
  hipLaunchKernelGGL((k_gemm_tn_wide<WN, WK, AProv, Y_CLASS>), grid, dim3(64 * WN * WK), lds_bytes, st, dy, ldy, ap, m,
                     n_out, k, slabs, rows_per_z, direct);
}

template <int WN, int WK, int TIN, int TIK, class AProv, bool Y_CLASS>
static void launch_tn_x6(const float *dy, int64_t ldy, const AProv &ap, int64_t m, int n_out, int k, float *slabs,
                         int64_t rows_per_z, int64_t chunks, const TnDirect &direct, hipStream_t st) {
  constexpr int TN_ = 32 * TIN * WN, TK_ = 32 * TIK * WK;
  constexpr size_t lds_bytes = 2 * 3 * (size_t)(TN_ + TK_) * kX6RowBytes;
  static_assert(lds_bytes <= 160 * 1024, "two stages of three planes");
  static std::atomic<unsigned long long> lds_raised{0};   // > 64 KB of dynamic LDS: once per kernel and device
  (void)gs_raise_dynamic_lds(reinterpret_cast<const void *>(&k_gemm_tn_x6<WN, WK, TIN, TIK, AProv, Y_CLASS>), lds_bytes,
                             lds_raised);
  const dim3 grid((unsigned)gs_ceil_div(k, TK_), (unsigned)gs_ceil_div(n_out, TN_), (unsigned)chunks);
  hipLaunchKernelGGL((k_gemm_tn_x6<WN, WK, TIN, TIK, AProv, Y_CLASS>), grid, dim3(64 * WN * WK), lds_bytes, st, dy, ldy, ap,
                     m, n_out, k, slabs, rows_per_z, direct);
}

// rows per slab and slab count: ~one resident set of workgroups, at least two 32-row stages each
static inline void tn_chunking(int64_t m, int64_t tiles, int64_t resident, int64_t &rows_per_z, int64_t &chunks) {
  chunks = gs_ceil_div(resident, tiles);
  const int64_t max_chunks = gs_ceil_div(m, 2 * kTnBK);
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  rows_per_z = gs_ceil_div(gs_ceil_div(m, chunks), kTnBK) * kTnBK;
  chunks = gs_ceil_div(m, rows_per_z);
}

template <class AProv, bool Y_CLASS = false>
static int launch_tn(const float *dy, int64_t ldy, const AProv &ap, int64_t m, int n_out, int k, float *out,
                     int64_t ld_out, int accumulate, float *slabs, size_t slab_bytes, hipStream_t st,
                     const SlabOut *scatter = nullptr, int force_wn = 0, int force_wk = 0, int64_t force_chunks = 0,
                     SlabQueue *defer = nullptr, int force_x6 = -1) {
  GS_REQUIRE(dy && (out || scatter) && (slabs || defer), GNNSAFT_ERR_NULL);
  GS_REQUIRE(m >= 1 && n_out >= 1 && k >= 4 && (k % 4) == 0 && (Y_CLASS || (ldy % 4) == 0) && (ld_out % 4) == 0,
             GNNSAFT_ERR_SHAPE);
  int wn = force_wn > 0 ? force_wn : tn_waves(m, n_out, k), wk = force_wk > 0 ? force_wk : tn_waves(m, n_out, k);
  GS_REQUIRE((wn == 1 && wk == 1) || (wn == 2 && wk == 2) || (wn == 4 && (wk == 2 || wk == 4)), GNNSAFT_ERR_UNSUPPORTED);
  const bool wide = wn * wk > 1;
  const int tn_ = wide ? 64 * wn : kTnTile, tk_ = wide ? 64 * wk : kTnTile;
  const int64_t tiles = gs_ceil_div(k, tk_) * gs_ceil_div(n_out, tn_);
  // workgroups one CU holds (LDS 160 KB; the 64 x 64 kernel: four)
  const int64_t per_cu = wide ? (160 * 1024) / (2 * kTnBK * (tn_ + tk_ + 8) * 4) : 4;
  int64_t rows_per_z, chunks;
  // (few output tiles: half a resident set of longer workgroups -- tools/tn_tune.py, C2 [128,128] 27 -> 23 us,
  //  C5 [256,64] 21 -> 15-18 us; with 16 tiles and more, or with the 60 k rows of the class sums, the full set is as
  //  good or better)
  const int64_t resident = 256 * (per_cu > 4 ? 4 : per_cu);
  tn_chunking(m, tiles, !wide && tiles <= 4 && m <= 32768 ? resident / 2 : resident, rows_per_z, chunks);
  if (force_chunks > 0) {
    rows_per_z = gs_ceil_div(gs_ceil_div(m, force_chunks), kTnBK) * kTnBK;
    chunks = gs_ceil_div(m, rows_per_z);
  }
  // per-class sums of many rows: the streaming one-hot kernel (three bf16 products, 64 classes x 256 columns per
  // workgroup, one workgroup per CU and column block)
  bool class_x3 = false;
  if constexpr (Y_CLASS && std::is_same_v<AProv, TnPlain>) {
    const bool x6_on = force_x6 < 0 ? gemm_x6_enabled() : force_x6 != 0;
    class_x3 = x6_on && force_wn == 0 && force_chunks == 0 && scatter == nullptr && n_out <= kClsX3Classes &&
               (k % kClsX3Cols) == 0 && ap.relu == 0 && (ap.lda % 4) == 0 && m >= (force_x6 > 0 ? 64 : 16384) &&
               (reinterpret_cast<uintptr_t>(dy) & 15) == 0 && (reinterpret_cast<uintptr_t>(ap.a) & 15) == 0;
    if (class_x3) {
      const int64_t col_blocks = k / kClsX3Cols;
      chunks = gs_ceil_div(512, col_blocks);   // two workgroups per CU ...
      const int64_t by_rows = gs_ceil_div(m, 256);   // ... of at least 16 stages (each chunk costs a slab: C2 36 us with 237 chunks, 46 with 512)
      if (chunks > by_rows) chunks = by_rows;
      rows_per_z = gs_ceil_div(gs_ceil_div(m, chunks), kClsX3Rows) * kClsX3Rows;
      if (rows_per_z < 64) rows_per_z = 64;   // (tn_slab_bytes: never finer than 64 rows)
      chunks = gs_ceil_div(m, rows_per_z);
      class_x3 = chunks > 1;
    }
  }
  if (defer != nullptr) {
    GS_REQUIRE(defer->count < kMaxSlabJobs, GNNSAFT_ERR_WORKSPACE);
    slabs = defer->take((size_t)chunks * n_out * k);
    GS_REQUIRE(slabs != nullptr, GNNSAFT_ERR_WORKSPACE);
  } else {
    GS_REQUIRE(slab_bytes >= (size_t)chunks * n_out * k * 4, GNNSAFT_ERR_WORKSPACE);
  }
  const int64_t per_slab = (int64_t)n_out * k;
  SlabOut so{{out, out, out, out}, (int64_t)1 << 40};
  if (scatter != nullptr) so = *scatter;
  TnDirect direct{so, ld_out, chunks == 1 ? 1 : 0, accumulate};
  // split-bf16 kernel for the 4 x 4 grid; the 64 x 64 kernel and the 2 x 2 / 4 x 2 grids stay f32
  const bool x6 = force_x6 < 0 ? gemm_x6_enabled() : force_x6 != 0;
  constexpr bool kHasX6 = std::is_same_v<AProv, TnPlain> && !Y_CLASS;   // (the one-hot providers spill in it)
  if constexpr (kHasX6) {
    if (x6 && wn == 4 && wk == 4) {
      launch_tn_x6<4, 4, 2, 2, AProv, Y_CLASS>(dy, ldy, ap, m, n_out, k, slabs, rows_per_z, chunks, direct, st);
      wn = wk = -1;   // done
    }
  }
  if constexpr (Y_CLASS && std::is_same_v<AProv, TnPlain>) {
    if (class_x3) {
      constexpr size_t lds_bytes = (size_t)2 * (kClsX3Classes + 3 * kClsX3Cols) * kX6RowBytes;
      static_assert(2 * lds_bytes <= 160 * 1024, "two workgroups of two stages per CU");
      static std::atomic<unsigned long long> lds_raised{0};
      GS_HIP(gs_raise_dynamic_lds(reinterpret_cast<const void *>(&k_class_sum_x3), lds_bytes, lds_raised));
      const dim3 grid((unsigned)(k / kClsX3Cols), 1u, (unsigned)chunks);
      hipLaunchKernelGGL(k_class_sum_x3, grid, dim3(256), lds_bytes, st, reinterpret_cast<const int32_t *>(dy), ap.a, ap.lda,
                         m, n_out, k, slabs, rows_per_z);
      wn = wk = -1;   // done
    }
  }
  if (wn > 0)
  switch (wn * 8 + wk) {
#define GS_TN_CASE(WN_, WK_)                                                                                  \
  case WN_ * 8 + WK_:                                                                                         \
    launch_tn_wide<WN_, WK_, AProv, Y_CLASS>(dy, ldy, ap, m, n_out, k, slabs, rows_per_z, chunks, direct, st); \
    break;
    GS_TN_CASE(2, 2)
    GS_TN_CASE(4, 2)
    GS_TN_CASE(4, 4)
#undef GS_TN_CASE
    default: {
      const dim3 grid((unsigned)gs_ceil_div(k, kTnTile), (unsigned)gs_ceil_div(n_out, kTnTile), (unsigned)chunks);
      hipLaunchKernelGGL((k_gemm_tn<AProv, Y_CLASS>), grid, dim3(256), 0, st, dy, ldy, ap, m, n_out, k, slabs,
                         rows_per_z, direct);
    }
  }
  if (chunks == 1) {   // one slab: the GEMM wrote the result itself
    GS_CHECK_LAUNCH();
    return GNNSAFT_OK;
  }
  if (defer != nullptr) {
    defer->jobs[defer->count++] = SlabJob{slabs, per_slab, chunks, ld_out, so, k, accumulate};
    GS_CHECK_LAUNCH();
    return GNNSAFT_OK;
  }
  hipLaunchKernelGGL(k_sum_slabs, dim3((unsigned)gs_ceil_div(per_slab / 4, 32)), dim3(256), 0, st, slabs, per_slab,
                     chunks, so, ld_out, k, accumulate, per_slab);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

int launch_slab_queue_flush(SlabQueue &q, hipStream_t st) {
  if (q.count > 0) {
    SlabJobs jobs;
    int64_t blocks = 1;
    for (int i = 0; i < kMaxSlabJobs; ++i) {
      jobs.j[i] = q.jobs[i < q.count ? i : 0];
      const int64_t b = gs_ceil_div(jobs.j[i].per_slab / 4, 32);
      blocks = b > blocks ? b : blocks;
    }
    hipLaunchKernelGGL(k_sum_slabs_batched, dim3((unsigned)blocks, (unsigned)q.count), dim3(256), 0, st, jobs);
    GS_CHECK_LAUNCH();
  }
  q.count = 0;
  q.off = 0;
  return GNNSAFT_OK;
}

size_t tn_slab_bytes(int64_t m, int n_out, int k) {
  // upper bound of launch_tn's slab count: at most 1024 workgroups' worth, never finer than 64 rows
  int64_t chunks = gs_ceil_div(m > 0 ? m : 1, 2 * kTnBK);
  if (chunks > 1024) chunks = 1024;
  return (size_t)chunks * (size_t)n_out * (size_t)k * 4;
}

int launch_wgrad_plain(const float *dy, int64_t ldy, const float *a, int64_t lda, int relu_a, int64_t m, int n_out,
                       int k, float *out, int64_t ld_out, int accumulate, float *slabs, size_t slab_bytes,
                       hipStream_t st, SlabQueue *defer) {
  GS_REQUIRE(a != nullptr && (lda % 4) == 0, GNNSAFT_ERR_SHAPE);
  TnPlain ap{a, lda, relu_a, k};
  return launch_tn(dy, ldy, ap, m, n_out, k, out, ld_out, accumulate, slabs, slab_bytes, st, nullptr, 0, 0, 0, defer);
}

int launch_wgrad_plain_blocks(const float *dy, int64_t ldy, const float *a, int64_t lda, int64_t m, int num_blocks,
                              int rows_per_block, int k, float *const *out_blocks, int64_t ld_out, float *slabs,
                              size_t slab_bytes, hipStream_t st, SlabQueue *defer) {
  GS_REQUIRE(a != nullptr && (lda % 4) == 0 && out_blocks != nullptr && num_blocks >= 1 && num_blocks <= 4,
             GNNSAFT_ERR_SHAPE);
  SlabOut so;
  for (int i = 0; i < 4; ++i) so.base[i] = out_blocks[i < num_blocks ? i : 0];
  so.rows_per_block = rows_per_block;
  TnPlain ap{a, lda, 0, k};
  return launch_tn(dy, ldy, ap, m, num_blocks * rows_per_block, k, nullptr, ld_out, 0, slabs, slab_bytes, st, &so, 0, 0,
                   0, defer);
}

int launch_wgrad_post_folded(const float *du, const float *x, const float *agg, const int32_t *perm,
                             const int32_t *tiles, const int32_t *num_tiles, int64_t tile_cap, int tile_rows,
                             const float *avg, int hidden, float *dw0, float *dw1 /* [F/2,13F] each */, float *slabs,
                             size_t slab_bytes, hipStream_t st, SlabQueue *defer, const int32_t *hist,
                             int64_t num_nodes) {
  GS_REQUIRE(du && x && agg && perm && tiles && num_tiles && avg && dw0 && dw1 && (slabs || defer), GNNSAFT_ERR_NULL);
  GS_REQUIRE((hidden % 64) == 0 && tile_rows >= kTnBK && (tile_rows % kTnBK) == 0 && tile_cap >= 1,
             GNNSAFT_ERR_UNSUPPORTED);
  // large batches, F a multiple of 128: the split-bf16 kernel over single-degree chunks (see above) -- 128 x 256 tiles
  // of 16 waves for F % 256 == 0, 64 x 128 tiles of 4 waves otherwise.  (At C2, 20 k rows, it is no faster than the
  // tile-table kernel -- 68 vs 75 us, both bound by load latency -- and its slab sum costs 16 us: not used there.)
  const char *pf_env = getenv("GNNSAFT_POSTFOLD_X6");   // tuning aid: 0 = the f32 tile-table kernel everywhere
  const bool pf_on = pf_env == nullptr || pf_env[0] != '0';
  if (hist != nullptr && gemm_x6_enabled() && pf_on && (hidden % 128) == 0 && num_nodes >= 32768) {
    const bool big = (hidden % 256) == 0;
    // ~64 chunks: with 10 (k tile, tower) workgroups per chunk that is 2-3 resident sets of workgroups
    int64_t chunk_rows = gs_ceil_div(gs_ceil_div(num_nodes, (int64_t)64), (int64_t)kTnX6BK) * kTnX6BK;
    const int64_t zcap = num_nodes / chunk_rows + 1 + kDegreeBuckets;
    const size_t need = (size_t)zcap * 2 * (hidden / 2) * 5 * hidden;   // floats
    float *sl = slabs;
    bool ok = zcap <= kFoldSumMaxChunks;
    if (defer != nullptr) {
      sl = ok ? defer->take(need) : nullptr;
      ok = sl != nullptr;
    } else {
      ok = ok && slab_bytes >= need * 4;
    }
    if (ok) {
      TnFoldWideArgs a{du, x, agg, perm, hist, avg, hidden, (int)chunk_rows};
      auto run = [&](auto wn_, auto wk_, auto tin_, auto tik_) {
        constexpr int WN = decltype(wn_)::value, WK = decltype(wk_)::value, TIN = decltype(tin_)::value,
                      TIK = decltype(tik_)::value;
        constexpr int TN_ = 32 * TIN * WN, TK_ = 32 * TIK * WK;
        constexpr size_t lds_bytes = 2 * 3 * (size_t)(TN_ + TK_) * kX6RowBytes;
        static std::atomic<unsigned long long> lds_raised{0};
        (void)gs_raise_dynamic_lds(reinterpret_cast<const void *>(&k_gemm_tn_postfold_x6<WN, WK, TIN, TIK>), lds_bytes,
                                   lds_raised);
        const dim3 grid((unsigned)(5 * hidden / TK_), (unsigned)(2 * (hidden / 2 / TN_)), (unsigned)zcap);
        hipLaunchKernelGGL((k_gemm_tn_postfold_x6<WN, WK, TIN, TIK>), grid, dim3(64 * WN * WK), lds_bytes, st, a, sl);
      };
      using std::integral_constant;
      if (big)
        run(integral_constant<int, 4>{}, integral_constant<int, 4>{}, integral_constant<int, 1>{},
            integral_constant<int, 2>{});   // 128 x 256, 16 waves of 32 x 64
      else
        run(integral_constant<int, 2>{}, integral_constant<int, 2>{}, integral_constant<int, 1>{},
            integral_constant<int, 2>{});   // 64 x 128, 4 waves of 32 x 64
      const int64_t out4 = 2 * (int64_t)(hidden / 2) * 5 * hidden / 4;
      hipLaunchKernelGGL(k_sum_slabs_fold, dim3((unsigned)gs_ceil_div(out4, 256)), dim3(256), 0, st, sl, hist, avg,
                         (int)chunk_rows, hidden, dw0, dw1);
      GS_CHECK_LAUNCH();
      return GNNSAFT_OK;
    }
  }
  // ~256 rows per workgroup along the contraction, as launch_tn does
  const int tiles_per_z = tile_rows >= 2 * kTnChunk ? 1 : 2 * kTnChunk / tile_rows;
  const int64_t chunks = gs_ceil_div(tile_cap, (int64_t)tiles_per_z);
  const int n_out = hidden / 2;
  const int64_t per_slab = (int64_t)n_out * 13 * hidden;
  if (defer != nullptr) {
    GS_REQUIRE(defer->count < kMaxSlabJobs, GNNSAFT_ERR_WORKSPACE);
    slabs = defer->take((size_t)chunks * 2 * per_slab);
    GS_REQUIRE(slabs != nullptr, GNNSAFT_ERR_WORKSPACE);
  } else {
    GS_REQUIRE(slab_bytes >= (size_t)chunks * 2 * per_slab * 4, GNNSAFT_ERR_WORKSPACE);
  }
  TnFoldArgs a{du, x, agg, perm, tiles, num_tiles, avg, hidden, tile_rows, tiles_per_z};
  const dim3 grid((unsigned)(5 * hidden / kTnTile), (unsigned)(2 * gs_ceil_div(n_out, kTnTile)), (unsigned)chunks);
  hipLaunchKernelGGL(k_gemm_tn_postfold, grid, dim3(256), 0, st, a, slabs);
  // both towers in one pass: the slab pair is a dense [2 * F/2, 13F] matrix whose row blocks go to dw0 / dw1
  SlabOut so{{dw0, dw1, dw1, dw1}, (int64_t)n_out};
  if (defer != nullptr) {
    defer->jobs[defer->count++] = SlabJob{slabs, 2 * per_slab, chunks, 13 * (int64_t)hidden, so, 13 * hidden, 0};
    GS_CHECK_LAUNCH();
    return GNNSAFT_OK;
  }
  hipLaunchKernelGGL(k_sum_slabs, dim3((unsigned)gs_ceil_div(2 * per_slab / 4, 32)), dim3(256), 0, st, slabs,
                     2 * per_slab, chunks, so, 13 * (int64_t)hidden, 13 * hidden, 0, 2 * per_slab);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

size_t wgrad_post_folded_slab_bytes(int64_t tile_cap, int tile_rows, int hidden) {
  const int tiles_per_z = tile_rows >= 2 * kTnChunk ? 1 : 2 * kTnChunk / tile_rows;
  return (size_t)gs_ceil_div(tile_cap, (int64_t)tiles_per_z) * 2 * (size_t)(hidden / 2) * 13 * hidden * 4;
}

int launch_wgrad_onehot(const float *dx, int64_t ldx, const int64_t *idx, int ncol, const int32_t *dims_host, int64_t n,
                        int hidden, float *dtab_t /* [H, total_rows] */, int total_rows_padded, float *slabs,
                        size_t slab_bytes, hipStream_t st) {
  GS_REQUIRE(idx != nullptr && ncol >= 1 && ncol <= GNNSAFT_MAX_TABLES, GNNSAFT_ERR_SHAPE);
  TnOneHot ap;
  ap.idx = idx;
  ap.ncol = ncol;
  int32_t off = 0;
  for (int t = 0; t <= GNNSAFT_MAX_TABLES; ++t) {
    ap.off[t] = off;
    if (t < ncol) off += dims_host[t];
  }
  GS_REQUIRE(total_rows_padded >= off && (total_rows_padded % 4) == 0, GNNSAFT_ERR_SHAPE);
  // output [n_out = H][k = total rows]: dE^T; the caller reads table t's row r at column off[t] + r
  return launch_tn(dx, ldx, ap, n, hidden, total_rows_padded, dtab_t, total_rows_padded, 0, slabs, slab_bytes, st);
}

int launch_sum_rows_by_class(const int32_t *cls, int num_classes, const float *a, int64_t lda, int64_t m, int k,
                             float *out /* [num_classes, k] */, int64_t ld_out, float *slabs, size_t slab_bytes,
                             hipStream_t st, SlabQueue *defer, int force_x6) {
  GS_REQUIRE(cls != nullptr && a != nullptr && (lda % 4) == 0 && num_classes >= 1, GNNSAFT_ERR_SHAPE);
  TnPlain ap{a, lda, 0, k};
  return launch_tn<TnPlain, true>(reinterpret_cast<const float *>(cls), 0, ap, m, num_classes, k, out, ld_out, 0, slabs,
                                  slab_bytes, st, nullptr, 0, 0, 0, defer, force_x6);
}

int launch_transpose_list(int count, const TransposeItem *items, hipStream_t st) {
  GS_REQUIRE(count >= 1 && count <= kMaxTransposeBatch && items != nullptr, GNNSAFT_ERR_SHAPE);
  TransposeBatch tb;
  int max_rows = 1, max_cols = 1;
  for (int i = 0; i < kMaxTransposeBatch; ++i) {
    const TransposeItem &it = items[i < count ? i : 0];
    tb.in[i] = it.in;
    tb.out[i] = it.out;
    tb.ld_in[i] = it.ld_in;
    tb.ld_out[i] = it.ld_out;
    tb.rows[i] = it.rows;
    tb.cols[i] = it.cols;
    GS_REQUIRE(it.in && it.out && it.rows >= 1 && it.cols >= 1, GNNSAFT_ERR_NULL);
    max_rows = it.rows > max_rows ? it.rows : max_rows;
    max_cols = it.cols > max_cols ? it.cols : max_cols;
  }
  hipLaunchKernelGGL(k_transpose, dim3((unsigned)gs_ceil_div(max_cols, 32), (unsigned)gs_ceil_div(max_rows, 32),
                                        (unsigned)count),
                     dim3(256), 0, st, tb);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

int launch_transpose(int count, const float *const *in, float *const *out, const int64_t *ld_in, const int64_t *ld_out,
                     int rows, int cols, hipStream_t st) {
  GS_REQUIRE(count >= 1 && count <= kMaxTransposeBatch, GNNSAFT_ERR_SHAPE);
  TransposeItem items[kMaxTransposeBatch];
  for (int i = 0; i < count; ++i) items[i] = TransposeItem{in[i], out[i], ld_in[i], ld_out[i], rows, cols};
  return launch_transpose_list(count, items, st);
}

static int colsum_impl(const float *a, int64_t lda, int64_t m, int cols, const ColsumOut &co, int accumulate,
                       float *partial, size_t partial_bytes, hipStream_t st, SlabQueue *defer = nullptr) {
  GS_REQUIRE(a && (partial || defer), GNNSAFT_ERR_NULL);
  const int64_t chunks = gs_ceil_div(m > 0 ? m : 1, kColChunk);
  const int direct = chunks == 1;
  if (defer != nullptr && !direct && (cols % 4) == 0 && (co.cols_per_block % 4) == 0) {
    // queued: the per-chunk partials [chunks][cols] are one more slab set (dense [num_blocks, cols_per_block] result,
    // row b -> bias tensor b)
    GS_REQUIRE(defer->count < kMaxSlabJobs, GNNSAFT_ERR_WORKSPACE);
    partial = defer->take((size_t)chunks * cols);
    GS_REQUIRE(partial != nullptr, GNNSAFT_ERR_WORKSPACE);
    hipLaunchKernelGGL(k_colsum_partial, dim3((unsigned)gs_ceil_div(cols, 32), (unsigned)chunks), dim3(256), 0, st, a,
                       lda, m, cols, partial, co, 0, accumulate);
    const bool one = co.cols_per_block >= cols;
    SlabOut so{{co.base[0], co.base[1], co.base[2], co.base[3]}, 1};
    defer->jobs[defer->count++] = SlabJob{partial, cols, chunks, one ? cols : co.cols_per_block, so,
                                          one ? cols : co.cols_per_block, accumulate};
    GS_CHECK_LAUNCH();
    return GNNSAFT_OK;
  }
  GS_REQUIRE(partial != nullptr && partial_bytes >= (size_t)chunks * cols * 4, GNNSAFT_ERR_WORKSPACE);
  hipLaunchKernelGGL(k_colsum_partial, dim3((unsigned)gs_ceil_div(cols, 32), (unsigned)chunks), dim3(256), 0, st, a,
                     lda, m, cols, partial, co, direct, accumulate);
  if (!direct)
    hipLaunchKernelGGL(k_colsum_final, dim3((unsigned)gs_ceil_div(cols, 32)), dim3(256), 0, st, partial, chunks, cols,
                       co, accumulate);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

int launch_colsum(const float *a, int64_t lda, int64_t m, int cols, float *out, int accumulate, float *partial,
                  size_t partial_bytes, hipStream_t st) {
  GS_REQUIRE(out != nullptr, GNNSAFT_ERR_NULL);
  ColsumOut co{{out, out, out, out}, 1 << 30};
  return colsum_impl(a, lda, m, cols, co, accumulate, partial, partial_bytes, st);
}

// column sums of [m, num_blocks * cols_per_block], block b written to outs[b] (several bias gradients in one pass)
int launch_colsum_blocks(const float *a, int64_t lda, int64_t m, int num_blocks, int cols_per_block, float *const *outs,
                         float *partial, size_t partial_bytes, hipStream_t st, SlabQueue *defer) {
  GS_REQUIRE(outs != nullptr && num_blocks >= 1 && num_blocks <= 4 && cols_per_block >= 1, GNNSAFT_ERR_SHAPE);
  ColsumOut co;
  for (int i = 0; i < 4; ++i) co.base[i] = outs[i < num_blocks ? i : 0];
  co.cols_per_block = cols_per_block;
  return colsum_impl(a, lda, m, num_blocks * cols_per_block, co, 0, partial, partial_bytes, st, defer);
}

}  // namespace gs

// tuning aid (tools/tn_tune.py): the weight-gradient GEMM with an explicit wave grid (wn x wk waves of 64 x 64 each;
// 1 x 1 = the 64 x 64 four-wave kernel) and slab count; 0 = the library's own choice; wn + 16 / wn + 32 force the
// split-bf16 / the f32 kernel (the 4 x 4 grid exists in both).  No global state.
extern "C" int gnnsaft_debug_linear_wgrad(const float *dy, int64_t ldy, const float *a, int64_t lda, int64_t m,
                                          int32_t n_out, int32_t k, float *dw, int64_t ld_dw, void *scratch,
                                          size_t scratch_bytes, int32_t wn, int32_t wk, int64_t chunks,
                                          gnnsaft_stream_t stream) {
  GS_REQUIRE(a != nullptr && (lda % 4) == 0, GNNSAFT_ERR_SHAPE);
  gs::TnPlain ap{a, lda, 0, k};
  int force_x6 = -1;   // wn + 16: the split-bf16 kernel, wn + 32: the f32 kernel, else the library's mode
  if (wn >= 32) {
    force_x6 = 0;
    wn -= 32;
  } else if (wn >= 16) {
    force_x6 = 1;
    wn -= 16;
  }
  return gs::launch_tn(dy, ldy, ap, m, n_out, k, dw, ld_dw, 0, static_cast<float *>(scratch), scratch_bytes,
                       static_cast<hipStream_t>(stream), nullptr, wn, wk, chunks, nullptr, force_x6);
}

// stage-test entry point (include/gnnsaft.h): out[c][:] = sum of the rows of `a` whose class id is c
extern "C" int gnnsaft_sum_rows_by_class(const int32_t *cls, int32_t num_classes, const float *a, int64_t lda, int64_t m,
                                         int32_t k, float *out, int64_t ld_out, void *scratch, size_t scratch_bytes,
                                         int32_t mode, gnnsaft_stream_t stream) {
  GS_REQUIRE(mode >= 0 && mode <= 2 && out != nullptr && scratch != nullptr, GNNSAFT_ERR_SHAPE);
  return gs::launch_sum_rows_by_class(cls, num_classes, a, lda, m, k, out, ld_out, static_cast<float *>(scratch),
                                      scratch_bytes, static_cast<hipStream_t>(stream), nullptr, mode == 0 ? -1 : mode - 1);
}

extern "C" size_t gnnsaft_wgrad_scratch_bytes(int64_t m, int32_t n_out, int32_t k) {
  return gs::tn_slab_bytes(m, n_out, k);
}

extern "C" int gnnsaft_linear_wgrad(const float *dy, int64_t ldy, const float *a, int64_t lda, int32_t relu_a,
                                    int64_t m, int32_t n_out, int32_t k, float *dw, int64_t ld_dw,
                                    int32_t accumulate, float *dbias /* or NULL */, void *scratch,
                                    size_t scratch_bytes, gnnsaft_stream_t stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc = gs::launch_wgrad_plain(dy, ldy, a, lda, relu_a, m, n_out, k, dw, ld_dw, accumulate,
                                  static_cast<float *>(scratch), scratch_bytes, st);
  if (rc != GNNSAFT_OK || dbias == nullptr) return rc;
  return gs::launch_colsum(dy, ldy, m, n_out, dbias, accumulate, static_cast<float *>(scratch), scratch_bytes, st);
}
