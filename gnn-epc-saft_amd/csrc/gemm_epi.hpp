// Epilogue shared by the matrix-core GEMM kernels (gemm.hip: k_gemm_f32, gemm_w3.hip: k_gemm_w3): bias, BatchNorm
// column partials (STATS), eval-mode BatchNorm affine, ReLU, residual / ReLU-mask, stores.  `acc` holds the wave's
// TM x TN accumulator tiles of the 32x32 MFMA.
#pragma once
#include "common.hpp"
#include "gemm_prov.hpp"

namespace gs {

template <int TM, int TN, int WTM, int WTN, int BM, int BN, int WAVES_M, bool STATS, bool AFFINE, bool RESID, class AProv>
__device__ __forceinline__ void gemm_epilogue(const f32x16 (&acc)[TM][TN], const AProv &ap, const TileInfo &ti,
                                              const GemmBatchEntry &ent, const EpiArgs &epi, int n0, int n_out,
                                              int64_t ldo, int wm, int wn, int lane) {
  // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane & 31,
  //      row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const int half = lane >> 5;
  const int wrow0 = wm * WTM;  // block-local
  const bool full_tile = ti.count >= BM && n0 + BN <= n_out;  // block-uniform
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * WTN + j * 32 + (lane & 31);
    const bool col_ok = col < n_out;
    const int colc = col_ok ? col : n_out - 1;
    const float bias = ent.bias != nullptr ? ent.bias[colc] : 0.f;

    if (STATS) {
      // (mean, M2) of this wave's 64 rows for column `col`: two in-register
      // passes per lane half, then Chan's pairwise combine across the halves.
      float sum = 0.f;
      int cnt = 0;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int lr = wrow0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          const bool ok = lr < ti.count;
          sum += ok ? acc[i][j][r] + bias : 0.f;
          cnt += ok ? 1 : 0;
        }
      const float mean = cnt > 0 ? sum / (float)cnt : 0.f;
      float m2 = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int lr = wrow0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          const float d = (acc[i][j][r] + bias) - mean;
          m2 += lr < ti.count ? d * d : 0.f;
        }
      const float o_mean = __shfl_xor(mean, 32);
      const float o_m2 = __shfl_xor(m2, 32);
      const int o_cnt = __shfl_xor(cnt, 32);
      const int tot = cnt + o_cnt;
      if (half == 0 && col_ok && tot > 0) {
        const float delta = o_mean - mean;
        const float cmean = mean + delta * ((float)o_cnt / (float)tot);
        const float cm2 = m2 + o_m2 + delta * delta * ((float)cnt * (float)o_cnt / (float)tot);
        const int64_t group = (int64_t)blockIdx.x * WAVES_M + wm;
        epi.stats[(group * 2 + 0) * n_out + col] = cmean;
        epi.stats[(group * 2 + 1) * n_out + col] = cm2;
      }
    }

    float sc = 1.f, sh = 0.f;
    if (AFFINE) {
      sc = epi.scale[colc];
      sh = epi.shift[colc];
      if (epi.bn_var != nullptr) {   // the arithmetic of k_bn_finalize, eval branch (same rounding points)
        const float rstd = 1.f / sqrtf(epi.bn_var[colc] + epi.bn_eps);
        sc = rstd * sc;
        sh = sh - epi.bn_mean[colc] * sc;
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      // element indices in 32 bits (the launcher checks rows * ld < 2^31): a 64-bit multiply per stored element was
      // a third of this kernel's VALU instructions at K = 256, and VALU issue time adds to the matrix cores' here
      uint32_t grow[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lr = wrow0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        grow[r] = (uint32_t)ap.out_row(ti, lr < ti.count ? lr : ti.count - 1);
      }
      float res[16];
      if (RESID) {
#pragma unroll
        for (int r = 0; r < 16; ++r) res[r] = epi.residual[grow[r] * (uint32_t)epi.ldr + (uint32_t)colc];
      }
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        v[r] = acc[i][j][r] + bias;
        if (AFFINE) v[r] = v[r] * sc + sh;
        v[r] = epi.relu_out ? fmaxf(v[r], 0.f) : v[r];
        if (RESID) v[r] = epi.residual_is_mask ? (res[r] > 0.f ? v[r] : 0.f) : v[r] + res[r];
      }
      // A per-element guard makes hipcc branch around every store and put an s_waitcnt vmcnt(0) in front of it:
      // 16 serialised store round trips per 32x32 tile, more than the MFMA time of a short-K workgroup.  Full
      // tiles (all but the last row / column tile) take the unguarded path: 16 stores issued back to back.
      if (full_tile) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ent.out[grow[r] * (uint32_t)ldo + (uint32_t)col] = v[r];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int lr = wrow0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (lr < ti.count && col_ok) ent.out[grow[r] * (uint32_t)ldo + (uint32_t)col] = v[r];
        }
      }
    }
  }
}

}  // namespace gs
