// Shared by degree.hip (stand-alone launch) and elementwise.hip (fused forward prologue).
#pragma once
#include "common.hpp"

namespace gs {

// w_eff[d][t][o][0:F] = W_t[o][0:F];  w_eff[d][t][o][F+j] = W_t[o][F+j] + amp(d) W_t[o][5F+j] + att(d) W_t[o][9F+j]
// Every folded weight is formed in float64 and rounded to float32 ONCE (degree_scalers() below gives the scalers as
// the float32 values the reference multiplies the activations with).
// (degree_scalers: common.hpp)
struct FoldLayers {
  const float *w0[GNNSAFT_MAX_FOLD_LAYERS];
  const float *w1[GNNSAFT_MAX_FOLD_LAYERS];
  const float *avg[GNNSAFT_MAX_FOLD_LAYERS];
  const float *pre0[GNNSAFT_MAX_FOLD_LAYERS];  // pre_nns[t][0].weight [F,3F] or null: fold the destination term
  const float *pre1[GNNSAFT_MAX_FOLD_LAYERS];
};

// per-layer operands of the edge-class tables (edge_encoder, the edge columns of pre_nns[t][0]); elementwise.hip
struct EdgeTableLayers {
  const float *we[GNNSAFT_MAX_FOLD_LAYERS], *be[GNNSAFT_MAX_FOLD_LAYERS];          // [H,H], [H]
  const float *wpre0[GNNSAFT_MAX_FOLD_LAYERS], *wpre1[GNNSAFT_MAX_FOLD_LAYERS];    // [F,3F] each
  const float *bpre0[GNNSAFT_MAX_FOLD_LAYERS], *bpre1[GNNSAFT_MAX_FOLD_LAYERS];    // [F]
};

// Destination-term fold.  msg = P_i + m~ with P_i = W_dst x_i constant over a node's in-edges, so
// mean/min/max(msg) = P_i + mean/min/max(m~) and std(msg) = std(m~).  The update's aggregate block
// then contributes  sum_s scale_s(d) (W_s,mean + W_s,min + W_s,max) P_i, i.e. an extra x-block
//   G_s = (W_s,mean + W_s,min + W_s,max) W_dst      ([F/2, F] per layer, tower, scaler s)
// so the W_dst GEMM over all nodes disappears.  Tiled matmul, 32x32 outputs per workgroup, accumulated AND kept in
// float64: the product is a function of the weights only ([F/2, F] per layer / tower / scaler: tiny), and every
// rounding made here would sit on a WEIGHT -- a systematic error on all N rows of the update, which the reference's
// arithmetic (f32 roundings on activations only) does not have.  The one rounding to f32 happens when
// fold_post_weights_body forms the final folded weight.
// (bx, by, bz): the workgroup's coordinates in the (F/32, F/64, 6 L) grid of 256-thread workgroups
__device__ __forceinline__ void dst_fold_body(const FoldLayers &fl, int f, double *__restrict__ g_all, int bx, int by,
                                              int bz) {
  __shared__ double as[32][33], bs[32][33];
  const int layer = bz / 6, rem = bz % 6, t = rem / 3, sc = rem % 3;
  const float *wpost = t == 0 ? fl.w0[layer] : fl.w1[layer];
  const float *wpre = t == 0 ? fl.pre0[layer] : fl.pre1[layer];
  const int o0 = by * 32, j0 = bx * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < f; k0 += 32) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = o0 + ty + 8 * r;  // A[o][k] = sum over the mean, min, max column blocks of scaler sc
      const float *w = wpost + (int64_t)o * (13 * f) + f + sc * 4 * f + k0 + tx;
      as[ty + 8 * r][tx] = ((double)w[0] + (double)w[f]) + (double)w[2 * f];
      bs[ty + 8 * r][tx] = (double)wpre[(int64_t)(k0 + ty + 8 * r) * (3 * f) + j0 + tx];  // W_dst[k][j]
    }
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
      const double b = bs[k][tx];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = __builtin_fma(as[ty + 8 * r][k], b, acc[r]);
    }
    __syncthreads();
  }
  double *g = g_all + ((((int64_t)layer * 2 + t) * 3 + sc) * (f / 2)) * f;
#pragma unroll
  for (int r = 0; r < 4; ++r) g[(int64_t)(o0 + ty + 8 * r) * f + j0 + tx] = acc[r];
}

__global__ __launch_bounds__(256) void k_dst_fold(FoldLayers fl, int f, double *__restrict__ g_all);

}  // namespace gs
