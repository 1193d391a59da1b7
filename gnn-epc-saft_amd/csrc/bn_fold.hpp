// Train-mode BatchNorm statistics: combining the per-64-row (mean, M2) column partials a STATS GEMM epilogue leaves
// (torch BatchNorm1d semantics: SURVEY.md Appendix A.3; reference call sites models.py:82,128).  Shared by
// the kernels of bn_train.hip: combine + apply (readout blocks, C ABI) and k_bn_stats_close (node layers: the
// statistics are closed in ONE small launch, the normalisation is applied by the A-operand provider of the next GEMM
// instead of an apply pass over y).
//
// Division-free combine in f64:
//   S1 = sum n_g mean_g,  S2 = sum (M2_g + n_g mean_g^2);   mean = S1/N,  M2 = S2 - N mean^2
// (no pivot: in float64 the final subtraction loses 1e-16 (mean/std)^2 of M2 -- 1e-10 for a column whose mean is a
// thousand standard deviations)
// Two levels: segments of `per_seg` groups -> (S1, S2) per segment, then over the <= 64 segments.  Every sum has a
// fixed order (thread layout 32 columns x 8 partial-lanes, lanes folded 0..7), whoever executes it: both kernel
// families give the same bits.
#pragma once
#include "common.hpp"

namespace gs {

constexpr int kBnCols = 32;
constexpr int kBnGroupLanes = 8;    // 256 threads = 8 partial-lanes x 32 columns
constexpr int kBnFusedGroups = 64;  // up to this many partials: one segment
constexpr int kBnMaxSegments = 64;
constexpr int kBnSegGroups = 64;    // partials folded per segment (8 per thread, one round of loads)

// segmentation of `groups` partials: per_seg is a multiple of 4 (a STATS workgroup owns 1, 2 or 4 consecutive groups
// and must not straddle two segments), num_seg <= kBnMaxSegments
static inline void bn_segments(int64_t groups, int *num_seg, int64_t *per_seg) {
  if (groups <= kBnFusedGroups) {
    *num_seg = 1;
    *per_seg = groups > 0 ? (groups + 3) / 4 * 4 : 4;
    return;
  }
  int64_t ns = (groups + kBnSegGroups - 1) / kBnSegGroups;
  if (ns > kBnMaxSegments) ns = kBnMaxSegments;
  int64_t ps = (groups + ns - 1) / ns;
  ps = (ps + 3) / 4 * 4;
  *per_seg = ps;
  *num_seg = (int)((groups + ps - 1) / ps);
}

// (S1, S2) of one thread's share of the partials [g_beg, g_end) of column colc around `pivot`
__device__ __forceinline__ void bn_fold_partials(const float *__restrict__ stats, int64_t g_beg, int64_t g_end,
                                                 int64_t rows, int ch, int colc, int gl, double &s1, double &s2) {
  constexpr int kUnroll = 8;  // independent loads in flight per thread
  for (int64_t g0 = g_beg + gl; g0 < g_end; g0 += kBnGroupLanes * kUnroll) {
    float gm[kUnroll], g2[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      int64_t g = g0 + (int64_t)u * kBnGroupLanes;
      g = g < g_end ? g : g_end - 1;
      gm[u] = stats[(g * 2 + 0) * ch + colc];
      g2[u] = stats[(g * 2 + 1) * ch + colc];
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t g = g0 + (int64_t)u * kBnGroupLanes;
      if (g < g_end) {
        const int64_t left = rows - g * kBnRowsPerGroup;
        const double gn = (double)(left < kBnRowsPerGroup ? left : kBnRowsPerGroup);
        const double m = (double)gm[u];
        s1 += gn * m;
        s2 += (double)g2[u] + gn * m * m;
      }
    }
  }
}

// The same sums for NP column slabs of 32 at once (column of slab p: colc[p]): all slabs' loads of a round are in
// flight together.  Per slab the order of the additions is that of bn_fold_partials.
template <int NP>
__device__ __forceinline__ void bn_fold_partials_multi(const float *__restrict__ stats, int64_t g_beg, int64_t g_end,
                                                       int64_t rows, int ch, const int (&colc)[NP], int gl,
                                                       double (&s1)[NP], double (&s2)[NP]) {
  constexpr int kUnroll = 8;
  for (int64_t g0 = g_beg + gl; g0 < g_end; g0 += kBnGroupLanes * kUnroll) {
    float gm[NP][kUnroll], g2[NP][kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      int64_t g = g0 + (int64_t)u * kBnGroupLanes;
      g = g < g_end ? g : g_end - 1;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        gm[p][u] = __builtin_nontemporal_load(stats + (g * 2 + 0) * ch + colc[p]);
        g2[p][u] = __builtin_nontemporal_load(stats + (g * 2 + 1) * ch + colc[p]);
      }
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t g = g0 + (int64_t)u * kBnGroupLanes;
      if (g < g_end) {
        const int64_t left = rows - g * kBnRowsPerGroup;
        const double gn = (double)(left < kBnRowsPerGroup ? left : kBnRowsPerGroup);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const double m = (double)gm[p][u];
          s1[p] += gn * m;
          s2[p] += (double)g2[p][u] + gn * m * m;
        }
      }
    }
  }
}

// What the last step of the statistics produces for one column (both the apply launch and the in-GEMM tail)
struct BnColumn {
  float mean, rstd, scale, shift, unbiased;
};
__device__ __forceinline__ BnColumn bn_finish_column(double s1, double s2, int64_t rows, float gamma, float beta,
                                                     float eps) {
  const double n = (double)rows;
  const double mean = s1 / n;
  double m2 = s2 - n * mean * mean;
  m2 = m2 > 0.0 ? m2 : 0.0;
  BnColumn c;
  c.mean = (float)mean;
  const float var_f = (float)(m2 / n);  // biased: used for normalisation
  c.rstd = 1.f / sqrtf(var_f + eps);
  c.scale = c.rstd * gamma;
  c.shift = beta - c.mean * c.scale;
  c.unbiased = (float)(n > 1.0 ? m2 / (n - 1.0) : m2);
  return c;
}

}  // namespace gs
