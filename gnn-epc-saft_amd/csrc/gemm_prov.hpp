// A-operand providers of the matrix-core GEMMs (gemm.hip: k_gemm_f32; gemm_w3.hip: k_gemm_w3).  A provider builds the
// virtual A operand while it is staged global -> registers -> LDS, so the tensors PyG materialises ([E',T,3F]
// message input, [N,T,13F] update input) never exist in HBM.
#pragma once
#include "common.hpp"

namespace gs {

constexpr int BK = 32;

// --------------------------------------------------------------------------
// A operand providers.  Every load is UNCONDITIONAL (hipcc branches around a
// guarded load and waits for it on the spot, serialising one L2 round trip per
// element): the kernel clamps row indices to the last valid row (rows past M
// are computed on duplicate data and never stored) and the K tail is zeroed by
// a select in finish().  `k0` is the wave-uniform base of the BK tile, `c` the
// lane's column offset inside it.
// --------------------------------------------------------------------------
// A block's rows: `count` valid rows starting at row0 (of the provider's row space); w_off is a
// per-tile offset into the weight matrix (degree-folded update).  count <= 0: nothing to do.
struct TileInfo {
  int64_t row0;
  int count;
  int64_t w_off;
};

__device__ __forceinline__ TileInfo gs_plain_tile(int bx, int bm, int64_t m) {
  const int64_t row0 = (int64_t)bx * bm;
  const int64_t left = m - row0;
  return TileInfo{row0, (int)(left < bm ? left : bm), 0};
}

template <bool RELU>
struct PlainAT {
  const float *a;
  int64_t lda;
  int64_t m;
  int k;
  struct Row {
    const float *p;
  };
  struct Raw {
    f32x4 v;
  };
  __device__ __forceinline__ TileInfo tile(int bx, int bm) const { return gs_plain_tile(bx, bm, m); }
  __device__ __forceinline__ int64_t out_row(const TileInfo &t, int lr) const { return t.row0 + lr; }
  __device__ __forceinline__ Row row(int64_t r, int64_t a_off) const { return Row{a + r * lda + a_off}; }
  __device__ __forceinline__ Raw load(const Row &r, int k0, int c) const {
    const int kk = k0 + c;
    return Raw{gs_ld4(r.p + (kk < k ? kk : 0))};
  }
  __device__ __forceinline__ f32x4 finish(const Raw &w, const Row &, int k0, int c) const {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 v = RELU ? gs_relu4(w.v) : w.v;
    return (k0 + c < k) ? v : zero;
  }
  // K % 16 == 0: no tail to zero
  __device__ __forceinline__ f32x4 finish_full(const Raw &w, const Row &, int, int) const {
    return RELU ? gs_relu4(w.v) : w.v;
  }
  // address of the float4 load() would fetch when k0 + c < k (gemm_ar.hip issues its A loads itself)
  __device__ __forceinline__ const float *ptr(const Row &r, int k0, int c) const { return r.p + k0 + c; }
};
using PlainA = PlainAT<false>;       // a row-major matrix
using PlainReluA = PlainAT<true>;    // ... with ReLU on load (the extra pre / post layers' inputs)

struct PostA {
  const float *x;        // [N,F]
  const float *agg;      // [N,2,4F]
  const float *log_amp;  // [N]
  const float *log_att;  // [N]
  const float *avg;      // device [1]
  int64_t n;
  int f;
  struct Row {
    const float *px;
    const float *pa;
    float amp, att;
  };
  struct Raw {
    f32x4 v;
  };
  __device__ __forceinline__ TileInfo tile(int bx, int bm) const { return gs_plain_tile(bx, bm, n); }
  __device__ __forceinline__ int64_t out_row(const TileInfo &t, int lr) const { return t.row0 + lr; }
  __device__ __forceinline__ Row row(int64_t r, int64_t a_off) const {
    const float avgv = avg[0];
    return Row{x + r * f, agg + r * (int64_t)(8 * f) + a_off, log_amp[r] / avgv, avgv / log_att[r]};
  }
  // K = 13F = [x | A | A*amp | A*att]; F is a multiple of BK, so a BK tile never straddles segments
  __device__ __forceinline__ Raw load(const Row &r, int k0, int c) const {
    const int j = k0 - f;  // wave-uniform
    const int seg = j < 4 * f ? 0 : (j < 8 * f ? 1 : 2);
    const float *p = j < 0 ? r.px + k0 : r.pa + (j - seg * 4 * f);
    return Raw{gs_ld4(p + c)};
  }
  __device__ __forceinline__ f32x4 finish(const Raw &w, const Row &r, int k0, int) const {
    const int j = k0 - f;
    const float s = j < 4 * f ? 1.f : (j < 8 * f ? r.amp : r.att);  // identity | amplification | attenuation
    return w.v * s;                                                 // x * 1.0f is exact
  }
  __device__ __forceinline__ f32x4 finish_full(const Raw &w, const Row &r, int k0, int c) const {
    return finish(w, r, k0, c);
  }
};

struct EdgeA {
  const int32_t *src;
  const int32_t *dst;
  const int32_t *combo;
  const float *pq;    // [N,4F]
  const float *rtab;  // [C,2F]
  int64_t rows;
  int f;
  struct Row {
    const float *p;
    const float *q;
    const float *r;
  };
  struct Raw {
    f32x4 a, b, t;
  };
  __device__ __forceinline__ TileInfo tile(int bx, int bm) const { return gs_plain_tile(bx, bm, rows); }
  __device__ __forceinline__ int64_t out_row(const TileInfo &t, int lr) const { return t.row0 + lr; }
  __device__ __forceinline__ Row row(int64_t r, int64_t a_off) const {
    return Row{pq + (int64_t)dst[r] * (4 * f) + a_off, pq + (int64_t)src[r] * (4 * f) + 2 * f + a_off,
               rtab + (int64_t)combo[r] * (2 * f) + a_off};
  }
  __device__ __forceinline__ Raw load(const Row &r, int k0, int c) const {
    const int kk = k0 + c;
    return Raw{gs_ld4(r.p + kk), gs_ld4(r.q + kk), gs_ld4(r.r + kk)};
  }
  __device__ __forceinline__ f32x4 finish(const Raw &w, const Row &, int, int) const {
    return gs_relu4((w.a + w.b) + w.t);
  }
  __device__ __forceinline__ f32x4 finish_full(const Raw &w, const Row &r, int k0, int c) const {
    return finish(w, r, k0, c);
  }
};

// The node state of the next layer formed while it is staged: A = relu(y * scale + shift) (+ x_prev), i.e. train-mode
// BatchNorm + ReLU + residual of models.py:128-131 applied to the previous layer's pre-activation y with the batch
// statistics k_bn_stats_close left in (scale, shift).  Replaces the k_bn_train_apply launch: the message
// GEMM of layer l+1 reads y_l and x_l instead of x_{l+1}, and its workgroups write x_{l+1} on their
// way (k-tile by k-tile, shared out among the column blocks) (the update GEMM and the backward's tape still want it in memory).  scale / shift sit in LDS for the whole kernel
// (K = H <= 256), so the staging registers hold only the two raw operands.
struct BnResA {
  const float *y;       // [M, K] pre-activation
  const float *xprev;   // [M, K] residual, or null
  const float *scale;   // [K]
  const float *shift;   // [K]
  float *xout;          // [M, K] side output, or null
  int64_t m;
  int k;                // multiple of BK, <= kMaxAffineK
  static constexpr bool kNeedsAffine = true;
  struct Row {
    const float *py;
    const float *px;
    float *po;
  };
  struct Raw {
    f32x4 y, x;
  };
  __device__ __forceinline__ TileInfo tile(int bx, int bm) const { return gs_plain_tile(bx, bm, m); }
  __device__ __forceinline__ int64_t out_row(const TileInfo &t, int lr) const { return t.row0 + lr; }
  __device__ __forceinline__ Row row(int64_t r, int64_t) const {
    // (rows past the end are clamped duplicates: they re-write the last row's values, harmless)
    return Row{y + r * k, (xprev != nullptr ? xprev : y) + r * k, xout != nullptr ? xout + r * k : nullptr};
  }
  __device__ __forceinline__ Raw load(const Row &r, int k0, int c) const {
    return Raw{gs_ld4(r.py + k0 + c), gs_ld4(r.px + k0 + c)};
  }
  // `aff`: LDS copy of [scale | shift]
  __device__ __forceinline__ f32x4 finish(const Raw &w, const Row &r, int k0, int c, const float *aff) const {
    const f32x4 sc = *reinterpret_cast<const f32x4 *>(aff + k0 + c);
    const f32x4 sh = *reinterpret_cast<const f32x4 *>(aff + k + k0 + c);
    f32x4 v = gs_relu4(w.y * sc + sh);      // the arithmetic of k_bn_train_apply: mul, add, max, add
    if (xprev != nullptr) v = v + w.x;
    // side output: the gridDim.y * gridDim.z workgroups of a row tile all stage the whole row tile; k-tile kt is
    // written by workgroup kt mod their number (k in steps of 32: a full 128-B line per row and writer; one writer per element, the stores spread over all)
    const unsigned writers = gridDim.y * gridDim.z, me = blockIdx.y + gridDim.y * blockIdx.z;
    if (r.po != nullptr && (unsigned)(k0 >> 5) % writers == me) gs_st4(r.po + k0 + c, v);
    return v;
  }
};
constexpr int kMaxAffineK = 256;

// Two row-major matrices side by side, A = [A0 | A1] (the backward's merged input-gradient GEMM
// dx = [du | dPQ] [W_x | W_pq]^T: one pass instead of two chained residual GEMMs).  k0 is a multiple of BK.
struct Concat2A {
  const float *a0;
  const float *a1;
  int64_t lda0, lda1, m;
  int k0, k;
  struct Row {
    const float *p0;
    const float *p1;
  };
  struct Raw {
    f32x4 v;
  };
  __device__ __forceinline__ TileInfo tile(int bx, int bm) const { return gs_plain_tile(bx, bm, m); }
  __device__ __forceinline__ int64_t out_row(const TileInfo &t, int lr) const { return t.row0 + lr; }
  __device__ __forceinline__ Row row(int64_t r, int64_t) const { return Row{a0 + r * lda0, a1 + r * lda1 - k0}; }
  __device__ __forceinline__ Raw load(const Row &r, int kt0, int c) const {
    const int kk = kt0 + c;
    const float *p = kt0 < k0 ? r.p0 : r.p1;  // wave-uniform
    return Raw{gs_ld4(p + (kk < k ? kk : kt0))};
  }
  __device__ __forceinline__ f32x4 finish(const Raw &w, const Row &, int kt0, int c) const {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    return (kt0 + c < k) ? w.v : zero;
  }
  __device__ __forceinline__ f32x4 finish_full(const Raw &w, const Row &, int, int) const { return w.v; }
  __device__ __forceinline__ const float *ptr(const Row &r, int kt0, int c) const { return (kt0 < k0 ? r.p0 : r.p1) + kt0 + c; }
};

// Degree-folded PNAConv update: rows are grouped by in-degree (tile table from csr.hip), so
// the three degree scalers are folded into per-degree weights and K shrinks from 13F to 5F:
//   u_t[i] = [W_x | W_id + amp(d) W_amp + att(d) W_att]_t(d_i)  .  cat[x_i, A_t[i]]
struct PostFoldA {
  const float *x;          // [N,F]
  const float *agg;        // [N,2,4F]
  const int32_t *perm;     // [N] node ids grouped by degree
  const int32_t *tiles;    // [max_tiles,4] = (degree, first slot, count, 0)
  const int32_t *num_tiles;
  int64_t w_stride;        // floats between two degrees' weight blocks
  int f;
  struct Row {
    const float *px;
    const float *pa;
  };
  struct Raw {
    f32x4 v;
  };
  __device__ __forceinline__ TileInfo tile(int bx, int) const {
    if (bx >= num_tiles[0]) return TileInfo{0, 0, 0};
    const int32_t *t = tiles + 4 * (int64_t)bx;
    return TileInfo{t[1], t[2], (int64_t)t[0] * w_stride};
  }
  __device__ __forceinline__ int64_t out_row(const TileInfo &t, int lr) const { return perm[t.row0 + lr]; }
  __device__ __forceinline__ Row row(int64_t slot, int64_t a_off) const {
    const int64_t node = perm[slot];
    return Row{x + node * f, agg + node * (int64_t)(8 * f) + a_off};
  }
  __device__ __forceinline__ Raw load(const Row &r, int k0, int c) const {
    const int j = k0 - f;  // wave-uniform; F is a multiple of BK
    const float *p = j < 0 ? r.px + k0 : r.pa + j;
    return Raw{gs_ld4(p + c)};
  }
  __device__ __forceinline__ f32x4 finish(const Raw &w, const Row &, int, int) const { return w.v; }
  __device__ __forceinline__ f32x4 finish_full(const Raw &w, const Row &, int, int) const { return w.v; }
  __device__ __forceinline__ const float *ptr(const Row &r, int k0, int c) const {
    const int j = k0 - f;  // wave-uniform
    return (j < 0 ? r.px + k0 : r.pa + j) + c;
  }
};

// Plain rows addressed through the degree permutation, weights selected per degree tile (backward of
// the folded update: dA_t[rows of degree d] = du_t[rows] W_eff(d,t)).
struct PermPlainA {
  const float *a;
  int64_t lda;
  const int32_t *perm;
  const int32_t *tiles;
  const int32_t *num_tiles;
  int64_t w_stride;
  int k;
  struct Row {
    const float *p;
  };
  struct Raw {
    f32x4 v;
  };
  __device__ __forceinline__ TileInfo tile(int bx, int) const {
    if (bx >= num_tiles[0]) return TileInfo{0, 0, 0};
    const int32_t *t = tiles + 4 * (int64_t)bx;
    return TileInfo{t[1], t[2], (int64_t)t[0] * w_stride};
  }
  __device__ __forceinline__ int64_t out_row(const TileInfo &t, int lr) const { return perm[t.row0 + lr]; }
  __device__ __forceinline__ Row row(int64_t slot, int64_t a_off) const {
    return Row{a + (int64_t)perm[slot] * lda + a_off};
  }
  __device__ __forceinline__ Raw load(const Row &r, int k0, int c) const {
    const int kk = k0 + c;
    return Raw{gs_ld4(r.p + (kk < k ? kk : 0))};
  }
  __device__ __forceinline__ f32x4 finish(const Raw &w, const Row &, int k0, int c) const {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    return (k0 + c < k) ? w.v : zero;
  }
  __device__ __forceinline__ f32x4 finish_full(const Raw &w, const Row &, int, int) const { return w.v; }
  __device__ __forceinline__ const float *ptr(const Row &r, int k0, int c) const { return r.p + k0 + c; }
};

}  // namespace gs
