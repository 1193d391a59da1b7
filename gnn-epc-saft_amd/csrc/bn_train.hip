// Train-mode BatchNorm in one launch after the producing GEMM
// (/root/reference/gnnepcsaft/train/models.py:82,87,94,98,128-131; torch BatchNorm1d
// semantics: SURVEY.md Appendix A.3).  The GEMM epilogue left per-64-row (mean, M2) column
// partials; every workgroup here owns a 32-column slab x a chunk of rows, first folds the
// partials of its 32 columns (Chan's combine in f64 -- redundant across the row chunks, but a
// few tens of KB of L2 reads per workgroup, cheaper than a separate finalize launch), then
// streams its rows:  out = relu(y * scale + shift) (+ residual).  Row chunk 0 also updates
// running_mean / running_var (unbiased) / num_batches_tracked.
// With more than kBnFusedGroups partials that redundant fold costs more L2 traffic than the pass over y itself
// (every workgroup re-reads all partials of its slab), so k_bn_combine first folds them into <= 64 segment sums
// (f64, same pivot) in its own small launch and the apply workgroups fold only those.
#include "bn_fold.hpp"

namespace gs {

// segment sums: seg[(s*2 + {0,1}) * ch + col] = (S1, S2) of partials [s*kBnSegGroups, (s+1)*kBnSegGroups)
__global__ __launch_bounds__(256) void k_bn_combine(const float *__restrict__ stats, int64_t rows, int ch,
                                                    double *__restrict__ seg, int64_t per_seg) {
  __shared__ double s_a[kBnGroupLanes][kBnCols], s_b[kBnGroupLanes][kBnCols];
  const int cl = threadIdx.x & (kBnCols - 1), gl = threadIdx.x / kBnCols;
  const int col = blockIdx.x * kBnCols + cl;
  const int colc = col < ch ? col : ch - 1;
  const int64_t groups = (rows + kBnRowsPerGroup - 1) / kBnRowsPerGroup;
  const int64_t g_beg = (int64_t)blockIdx.y * per_seg;
  int64_t g_end = g_beg + per_seg;
  if (g_end > groups) g_end = groups;
  double s1 = 0.0, s2 = 0.0;
  if (g_beg < g_end) bn_fold_partials(stats, g_beg, g_end, rows, ch, colc, gl, s1, s2);
  s_a[gl][cl] = s1;
  s_b[gl][cl] = s2;
  __syncthreads();
  if (gl == 0 && col < ch) {
    for (int o = 1; o < kBnGroupLanes; ++o) {
      s1 += s_a[o][cl];
      s2 += s_b[o][cl];
    }
    seg[((int64_t)blockIdx.y * 2 + 0) * ch + col] = s1;
    seg[((int64_t)blockIdx.y * 2 + 1) * ch + col] = s2;
  }
}

// combine + finalize in one launch: workgroup (slab, segment) folds its segment like k_bn_combine, then takes a
// ticket of its slab; the LAST of the slab's workgroups folds the segment sums and writes what k_bn_train_apply's
// first phase computes -- scale, shift, saved (mean, rstd), running statistics.  (The partials were written by the
// previous kernel; only the few KB of segment sums cross workgroups inside this one, behind an agent-scope release /
// acquire on the ticket.)  Same sums, same order as combine + apply.
__global__ __launch_bounds__(256) void k_bn_stats_close(const float *__restrict__ stats, int64_t rows, int ch,
                                                        double *__restrict__ seg, int64_t per_seg,
                                                        int32_t *__restrict__ tickets,
                                                        const float *__restrict__ gamma,
                                                        const float *__restrict__ beta,
                                                        float *__restrict__ running_mean,
                                                        float *__restrict__ running_var, int64_t *nbt,
                                                        float momentum, float eps, float *__restrict__ scale,
                                                        float *__restrict__ shift, float *__restrict__ save_stat) {
  __shared__ double s_a[kBnGroupLanes][kBnCols], s_b[kBnGroupLanes][kBnCols];
  __shared__ int s_last;
  const int cl = threadIdx.x & (kBnCols - 1), gl = threadIdx.x / kBnCols;
  const int col = blockIdx.x * kBnCols + cl;
  const bool col_ok = col < ch;
  const int colc = col_ok ? col : ch - 1;
  const int num_seg = gridDim.y;
  const int64_t groups = (rows + kBnRowsPerGroup - 1) / kBnRowsPerGroup;
  const int64_t g_beg = (int64_t)blockIdx.y * per_seg;
  int64_t g_end = g_beg + per_seg;
  if (g_end > groups) g_end = groups;
  double s1 = 0.0, s2 = 0.0;
  if (g_beg < g_end) bn_fold_partials(stats, g_beg, g_end, rows, ch, colc, gl, s1, s2);
  s_a[gl][cl] = s1;
  s_b[gl][cl] = s2;
  __syncthreads();
  if (gl == 0) {
    for (int o = 1; o < kBnGroupLanes; ++o) {
      s1 += s_a[o][cl];
      s2 += s_b[o][cl];
    }
    if (col_ok) {
      seg[((int64_t)blockIdx.y * 2 + 0) * ch + col] = s1;
      seg[((int64_t)blockIdx.y * 2 + 1) * ch + col] = s2;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0)
    s_last = __hip_atomic_fetch_add(tickets + blockIdx.x, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == num_seg - 1;
  __syncthreads();
  if (!s_last) return;   // block-uniform
  s1 = s2 = 0.0;
  for (int sg = gl; sg < num_seg; sg += kBnGroupLanes) {
    s1 += __builtin_nontemporal_load(seg + ((int64_t)sg * 2 + 0) * ch + colc);
    s2 += __builtin_nontemporal_load(seg + ((int64_t)sg * 2 + 1) * ch + colc);
  }
  s_a[gl][cl] = s1;
  s_b[gl][cl] = s2;
  __syncthreads();
  if (gl == 0 && col_ok) {
    for (int o = 1; o < kBnGroupLanes; ++o) {
      s1 += s_a[o][cl];
      s2 += s_b[o][cl];
    }
    const BnColumn bc = bn_finish_column(s1, s2, rows, gamma != nullptr ? gamma[col] : 1.f,
                                         beta != nullptr ? beta[col] : 0.f, eps);
    scale[col] = bc.scale;
    shift[col] = bc.shift;
    if (save_stat != nullptr) {
      save_stat[col] = bc.mean;
      save_stat[ch + col] = bc.rstd;
    }
    if (running_mean != nullptr) {
      running_mean[col] = (1.f - momentum) * running_mean[col] + momentum * bc.mean;
      running_var[col] = (1.f - momentum) * running_var[col] + momentum * bc.unbiased;
    }
    if (nbt != nullptr && col == 0) nbt[0] += 1;
  }
  if (threadIdx.x == 0) tickets[blockIdx.x] = 0;   // last to touch it: zero again for the next launch on this stream
}

__global__ __launch_bounds__(256) void k_bn_train_apply(const float *__restrict__ stats, const float *__restrict__ y,
                                                        int64_t rows, int ch, const float *__restrict__ gamma,
                                                        const float *__restrict__ beta,
                                                        float *__restrict__ running_mean,
                                                        float *__restrict__ running_var, int64_t *nbt,
                                                        float momentum, float eps,
                                                        const float *__restrict__ residual, float *__restrict__ out,
                                                        int64_t rows_per_chunk, float *__restrict__ save_stat,
                                                        const double *__restrict__ seg, int num_seg) {
  __shared__ double s_mean[kBnGroupLanes][kBnCols], s_m2[kBnGroupLanes][kBnCols];
  __shared__ float s_scale[kBnCols], s_shift[kBnCols];
  const int c0 = blockIdx.x * kBnCols;
  const int cl = threadIdx.x & (kBnCols - 1);
  const int gl = threadIdx.x / kBnCols;
  const int col = c0 + cl;
  const bool col_ok = col < ch;
  const int colc = col_ok ? col : ch - 1;

  // ---- phase 1: batch statistics of this slab's columns
  const int64_t groups = (rows + kBnRowsPerGroup - 1) / kBnRowsPerGroup;
  // (S1, S2) in f64: bn_fold.hpp
  double s1 = 0.0, s2 = 0.0;
  if (seg != nullptr) {
    for (int sg = gl; sg < num_seg; sg += kBnGroupLanes) {
      s1 += seg[((int64_t)sg * 2 + 0) * ch + colc];
      s2 += seg[((int64_t)sg * 2 + 1) * ch + colc];
    }
  } else {
    bn_fold_partials(stats, 0, groups, rows, ch, colc, gl, s1, s2);
  }
  s_mean[gl][cl] = s1;
  s_m2[gl][cl] = s2;
  __syncthreads();
  if (gl == 0) {
    for (int o = 1; o < kBnGroupLanes; ++o) {
      s1 += s_mean[o][cl];
      s2 += s_m2[o][cl];
    }
    const BnColumn bc = bn_finish_column(s1, s2, rows, gamma != nullptr ? gamma[colc] : 1.f,
                                         beta != nullptr ? beta[colc] : 0.f, eps);
    s_scale[cl] = bc.scale;
    s_shift[cl] = bc.shift;
    if (blockIdx.y == 0 && col_ok && save_stat != nullptr) {
      save_stat[col] = bc.mean;
      save_stat[ch + col] = bc.rstd;
    }
    if (blockIdx.y == 0 && col_ok) {
      if (running_mean != nullptr) {
        running_mean[col] = (1.f - momentum) * running_mean[col] + momentum * bc.mean;
        running_var[col] = (1.f - momentum) * running_var[col] + momentum * bc.unbiased;
      }
      if (nbt != nullptr && col == 0) nbt[0] += 1;
    }
  }
  __syncthreads();

  // ---- phase 2: normalise + ReLU (+ residual) over this workgroup's rows; 8 float4 per 32-column row
  const int c4 = (threadIdx.x & 7) * 4;
  const int rl = threadIdx.x >> 3;  // 32 rows per pass
  if (c0 + c4 >= ch) return;
  const f32x4 sc = {s_scale[c4], s_scale[c4 + 1], s_scale[c4 + 2], s_scale[c4 + 3]};
  const f32x4 sh = {s_shift[c4], s_shift[c4 + 1], s_shift[c4 + 2], s_shift[c4 + 3]};
  const int64_t r_beg = (int64_t)blockIdx.y * rows_per_chunk;
  int64_t r_end = r_beg + rows_per_chunk;
  if (r_end > rows) r_end = rows;
  for (int64_t r = r_beg + rl; r < r_end; r += 64) {
    const int64_t r2 = r + 32;
    const bool two = r2 < r_end;
    const int64_t o1 = r * ch + c0 + c4;
    const int64_t o2 = (two ? r2 : r) * ch + c0 + c4;
    const f32x4 v1 = gs_ld4(y + o1), v2 = gs_ld4(y + o2);
    f32x4 q1 = {0.f, 0.f, 0.f, 0.f}, q2 = {0.f, 0.f, 0.f, 0.f};
    if (residual != nullptr) {
      q1 = gs_ld4(residual + o1);
      q2 = gs_ld4(residual + o2);
    }
    f32x4 a = v1 * sc + sh, b = v2 * sc + sh;
    a.x = fmaxf(a.x, 0.f);
    a.y = fmaxf(a.y, 0.f);
    a.z = fmaxf(a.z, 0.f);
    a.w = fmaxf(a.w, 0.f);
    b.x = fmaxf(b.x, 0.f);
    b.y = fmaxf(b.y, 0.f);
    b.z = fmaxf(b.z, 0.f);
    b.w = fmaxf(b.w, 0.f);
    gs_st4(out + o1, a + q1);
    if (two) gs_st4(out + o2, b + q2);
  }
}

// eval-mode BatchNorm (+ ReLU, + residual) of a kept pre-activation tensor: out = relu(y scale + shift) (+ residual)
// with scale = gamma / sqrt(running_var + eps), shift = beta - running_mean scale formed per thread; (running_mean, rstd)
// go to `save_stat` in the layout k_bn_train_apply keeps its batch statistics in, so that the backward reads either.
__global__ __launch_bounds__(256) void k_bn_eval_apply(const float *__restrict__ y, int64_t rows, int ch,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta,
                                                       const float *__restrict__ rmean, const float *__restrict__ rvar,
                                                       float eps, const float *__restrict__ residual,
                                                       float *__restrict__ out, float *__restrict__ save_stat) {
  const int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= rows * ch) return;
  const int c = (int)(i4 % ch);
  const f32x4 yv = gs_ld4(y + i4), gm = gs_ld4(gamma + c), bt = gs_ld4(beta + c), mn = gs_ld4(rmean + c),
              vr = gs_ld4(rvar + c);
  f32x4 o, rs;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float rstd = 1.f / sqrtf(vr[j] + eps);
    const float sc = rstd * gm[j];
    rs[j] = rstd;
    o[j] = fmaxf(yv[j] * sc + (bt[j] - mn[j] * sc), 0.f);
  }
  if (residual != nullptr) o += gs_ld4(residual + i4);
  gs_st4(out + i4, o);
  if (save_stat != nullptr && i4 < ch) {   // the first row's threads cover every channel once
    gs_st4(save_stat + c, mn);
    gs_st4(save_stat + ch + c, rs);
  }
}

int launch_bn_stats_close(const float *stats, int64_t rows, int ch, const float *gamma, const float *beta,
                          float *running_mean, float *running_var, int64_t *nbt, float momentum, float eps,
                          float *scale, float *shift, float *save_stat, double *seg, int32_t *tickets, hipStream_t st) {
  GS_REQUIRE(stats && scale && shift && seg && tickets, GNNSAFT_ERR_NULL);
  GS_REQUIRE(rows >= 2 && ch >= 4 && (ch % 4) == 0 && (reinterpret_cast<uintptr_t>(seg) & 7) == 0, GNNSAFT_ERR_SHAPE);
  const int slabs = (ch + kBnCols - 1) / kBnCols;
  GS_REQUIRE(slabs <= kBnTailCounterInts, GNNSAFT_ERR_UNSUPPORTED);
  int num_seg = 0;
  int64_t per_seg = 0;
  bn_segments(gs_ceil_div(rows, (int64_t)kBnRowsPerGroup), &num_seg, &per_seg);
  hipLaunchKernelGGL(k_bn_stats_close, dim3((unsigned)slabs, (unsigned)num_seg), dim3(256), 0, st, stats, rows, ch, seg,
                     per_seg, tickets, gamma, beta, running_mean, running_var, nbt, momentum, eps, scale, shift,
                     save_stat);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

int launch_bn_eval_apply(const float *y, int64_t rows, int ch, const float *gamma, const float *beta,
                         const float *rmean, const float *rvar, float eps, const float *residual, float *out,
                         float *save_stat, hipStream_t st) {
  GS_REQUIRE(y && out && gamma && beta && rmean && rvar, GNNSAFT_ERR_NULL);
  GS_REQUIRE(rows >= 1 && ch >= 4 && (ch % 4) == 0, GNNSAFT_ERR_SHAPE);
  hipLaunchKernelGGL(k_bn_eval_apply, dim3((unsigned)gs_ceil_div(rows * ch / 4, 256)), dim3(256), 0, st, y, rows, ch,
                     gamma, beta, rmean, rvar, eps, residual, out, save_stat);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

}  // namespace gs

extern "C" size_t gnnsaft_bn_train_scratch_bytes(int64_t num_rows, int32_t channels) {
  if (num_rows <= 0 || channels <= 0) return 0;
  const int64_t groups = gs_ceil_div(num_rows, (int64_t)gs::kBnRowsPerGroup);
  if (groups <= gs::kBnFusedGroups) return 0;
  return (size_t)gs::kBnMaxSegments * 2 * (size_t)channels * sizeof(double);
}

extern "C" int gnnsaft_bn_train_apply(const float *stats, const float *y, int64_t num_rows, int32_t channels,
                                      const float *gamma, const float *beta, float *running_mean,
                                      float *running_var, int64_t *num_batches_tracked, float momentum, float eps,
                                      const float *residual, float *out, float *save_mean_rstd, void *scratch,
                                      size_t scratch_bytes, gnnsaft_stream_t stream) {
  GS_REQUIRE(stats && y && out, GNNSAFT_ERR_NULL);
  GS_REQUIRE(num_rows >= 2, GNNSAFT_ERR_SHAPE);  // torch: "Expected more than 1 value per channel"
  GS_REQUIRE(channels >= 4 && (channels % 4) == 0, GNNSAFT_ERR_SHAPE);
  const int slabs = (channels + gs::kBnCols - 1) / gs::kBnCols;
  const int64_t groups = gs_ceil_div(num_rows, (int64_t)gs::kBnRowsPerGroup);
  double *seg = nullptr;
  int num_seg = 0;
  if (groups > gs::kBnFusedGroups) {
    GS_REQUIRE(scratch != nullptr && scratch_bytes >= gnnsaft_bn_train_scratch_bytes(num_rows, channels) &&
                   (reinterpret_cast<uintptr_t>(scratch) & 7) == 0,
               GNNSAFT_ERR_WORKSPACE);
    seg = static_cast<double *>(scratch);
    int64_t per_seg = 0;
    gs::bn_segments(groups, &num_seg, &per_seg);
    hipLaunchKernelGGL(gs::k_bn_combine, dim3((unsigned)slabs, (unsigned)num_seg), dim3(256), 0,
                       static_cast<hipStream_t>(stream), stats, num_rows, channels, seg, per_seg);
    GS_CHECK_LAUNCH();
  }
  // ~1024 workgroups in total, at least 64 rows each
  int64_t chunks = 1024 / slabs;
  const int64_t max_chunks = gs_ceil_div(num_rows, 64);
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  int64_t rows_per_chunk = gs_ceil_div(num_rows, chunks);
  rows_per_chunk = gs_ceil_div(rows_per_chunk, 64) * 64;
  chunks = gs_ceil_div(num_rows, rows_per_chunk);
  hipLaunchKernelGGL(gs::k_bn_train_apply, dim3((unsigned)slabs, (unsigned)chunks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), stats, y, num_rows, channels, gamma, beta, running_mean,
                     running_var, num_batches_tracked, momentum, eps, residual, out, rows_per_chunk, save_mean_rstd, seg,
                     num_seg);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}
