// Degree-folded PNAConv update, structure side.
//
// PyG's DegreeScalerAggregation (reference call site
// /root/reference/gnnepcsaft/train/models.py:59-80) multiplies the [mean|min|max|std]
// block of every node by three scalars that depend only on the node's in-degree d:
//   1, amp(d) = log(d+1)/avg_deg_log, att(d) = avg_deg_log/log(max(d,1)+1)
// and feeds cat[x, A, A*amp, A*att] (13F wide) to post_nns.  Molecular in-degrees take a
// handful of values, so nodes are grouped by degree (tiles of `tile_rows` rows with one
// degree each) and the scalers are folded into per-degree weights
//   W_eff(d) = [W_x | W_id + amp(d) W_amp + att(d) W_att]        ([F/2, 5F] per tower)
// which cuts the dominant GEMM's K from 13F to 5F.  Degrees >= kDegreeBuckets are flagged
// (GNNSAFT_FLAG_BAD_DEGREE) and clamped; callers with such graphs disable folding.
#include "common.hpp"

namespace gs {

__device__ __forceinline__ int clamp_degree(int d, int32_t *err) {
  if (d >= kDegreeBuckets) {
    if (err) atomicOr(err, GNNSAFT_FLAG_BAD_DEGREE);
    d = kDegreeBuckets - 1;
  }
  return d;
}

// wave-aggregated histogram: one atomic per (wave, distinct degree)
__global__ __launch_bounds__(256) void k_degree_hist(const int32_t *__restrict__ rowptr, int64_t n,
                                                     int32_t *__restrict__ hist, int32_t *err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;
  const int d = live ? clamp_degree(rowptr[i + 1] - rowptr[i], err) : -1;
  unsigned long long todo = __ballot(live);
  while (todo != 0ull) {
    const int leader = __ffsll((long long)todo) - 1;
    const int dl = __shfl(d, leader);
    const unsigned long long same = __ballot(d == dl);
    if ((threadIdx.x & 63) == leader) atomicAdd(&hist[dl], __popcll(same));
    todo &= ~same;
  }
}

// single thread: bucket starts (slots in `perm`) and the tile table
__global__ void k_degree_tiles(const int32_t *__restrict__ hist, int tile_rows, int32_t *__restrict__ start,
                               int32_t *__restrict__ cursor, int32_t *__restrict__ tiles,
                               int32_t *__restrict__ num_tiles) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int slot = 0, nt = 0;
  for (int d = 0; d < kDegreeBuckets; ++d) {
    const int cnt = hist[d];
    start[d] = slot;
    cursor[d] = 0;
    for (int o = 0; o < cnt; o += tile_rows) {
      tiles[4 * nt + 0] = d;
      tiles[4 * nt + 1] = slot + o;
      tiles[4 * nt + 2] = cnt - o < tile_rows ? cnt - o : tile_rows;
      tiles[4 * nt + 3] = 0;
      ++nt;
    }
    slot += cnt;
  }
  num_tiles[0] = nt;
}

__global__ __launch_bounds__(256) void k_degree_fill(const int32_t *__restrict__ rowptr, int64_t n,
                                                     const int32_t *__restrict__ start,
                                                     int32_t *__restrict__ cursor, int32_t *__restrict__ perm) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;
  int d = live ? rowptr[i + 1] - rowptr[i] : -1;
  if (d >= kDegreeBuckets) d = kDegreeBuckets - 1;
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(live);
  while (todo != 0ull) {
    const int leader = __ffsll((long long)todo) - 1;
    const int dl = __shfl(d, leader);
    const unsigned long long same = __ballot(d == dl);
    int base = 0;
    if (lane == leader) base = atomicAdd(&cursor[dl], __popcll(same));
    base = __shfl(base, leader);
    if (d == dl) perm[start[dl] + base + __popcll(same & ((1ull << lane) - 1ull))] = (int32_t)i;
    todo &= ~same;
  }
}

// w_eff[d][t][o][0:F] = W_t[o][0:F];  w_eff[d][t][o][F+j] = W_t[o][F+j] + amp(d) W_t[o][5F+j] + att(d) W_t[o][9F+j]
__global__ __launch_bounds__(256) void k_fold_post_weights(const float *__restrict__ w0, const float *__restrict__ w1,
                                                           const float *__restrict__ avg,
                                                           const int32_t *__restrict__ hist, int f,
                                                           float *__restrict__ w_eff) {
  const int d = blockIdx.z;
  if (hist[d] == 0) return;  // degree absent from this batch
  const int t = blockIdx.y;
  const int per_row4 = 5 * f / 4;
  const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t o = slot / per_row4;
  if (o >= f / 2) return;
  const int c = (int)(slot - o * per_row4) * 4;
  const float *w = (t == 0 ? w0 : w1) + o * (int64_t)(13 * f);
  f32x4 v;
  if (c < f) {
    v = gs_ld4(w + c);
  } else {
    const float avgv = avg[0];
    const float amp = logf((float)d + 1.f) / avgv;
    const float att = avgv / logf(fmaxf((float)d, 1.f) + 1.f);
    v = gs_ld4(w + c) + gs_ld4(w + 4 * f + c) * amp + gs_ld4(w + 8 * f + c) * att;
  }
  gs_st4(w_eff + (((int64_t)d * 2 + t) * (f / 2) + o) * (int64_t)(5 * f) + c, v);
}

}  // namespace gs

extern "C" int64_t gnnsaft_degree_tiles_capacity(int64_t num_nodes, int32_t hidden) {
  const int tr = gs::pna_fold_tile_rows(hidden);
  return gs_ceil_div(num_nodes > 0 ? num_nodes : 1, tr) + gs::kDegreeBuckets;
}

extern "C" int32_t gnnsaft_degree_buckets(void) { return gs::kDegreeBuckets; }

extern "C" int gnnsaft_degree_tiles(const int32_t *rowptr, int64_t num_nodes, int32_t hidden, int32_t *perm,
                                    int32_t *tiles, int32_t *num_tiles, int32_t *hist3, int32_t *err_flag,
                                    gnnsaft_stream_t stream) {
  GS_REQUIRE(rowptr && perm && tiles && num_tiles && hist3, GNNSAFT_ERR_NULL);
  GS_REQUIRE(num_nodes >= 1 && hidden >= 32 && (hidden % 32) == 0, GNNSAFT_ERR_SHAPE);
  hipStream_t st = static_cast<hipStream_t>(stream);
  int32_t *hist = hist3, *start = hist3 + gs::kDegreeBuckets, *cursor = hist3 + 2 * gs::kDegreeBuckets;
  hipError_t e = hipMemsetAsync(hist, 0, sizeof(int32_t) * gs::kDegreeBuckets, st);
  if (e != hipSuccess) return (int)e;
  const dim3 grid((unsigned)gs_ceil_div(num_nodes, 256)), block(256);
  hipLaunchKernelGGL(gs::k_degree_hist, grid, block, 0, st, rowptr, num_nodes, hist, err_flag);
  hipLaunchKernelGGL(gs::k_degree_tiles, dim3(1), dim3(64), 0, st, hist, gs::pna_fold_tile_rows(hidden), start, cursor,
                     tiles, num_tiles);
  hipLaunchKernelGGL(gs::k_degree_fill, grid, block, 0, st, rowptr, num_nodes, start, cursor, perm);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

extern "C" int gnnsaft_pna_fold_post_weights(const float *w_post0, const float *w_post1, const float *avg_deg_log,
                                             const int32_t *hist, int32_t hidden, float *w_eff,
                                             gnnsaft_stream_t stream) {
  GS_REQUIRE(w_post0 && w_post1 && avg_deg_log && hist && w_eff, GNNSAFT_ERR_NULL);
  GS_REQUIRE(hidden >= 32 && (hidden % 32) == 0, GNNSAFT_ERR_SHAPE);
  const int64_t threads = (int64_t)(hidden / 2) * (5 * hidden / 4);
  const dim3 grid((unsigned)gs_ceil_div(threads, 256), 2, gs::kDegreeBuckets);
  hipLaunchKernelGGL(gs::k_fold_post_weights, grid, dim3(256), 0, static_cast<hipStream_t>(stream), w_post0, w_post1,
                     avg_deg_log, hist, hidden, w_eff);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

extern "C" int gnnsaft_pna_update_folded(const float *x, const float *agg, const int32_t *perm, const int32_t *tiles,
                                         const int32_t *num_tiles, int64_t num_nodes, int32_t hidden,
                                         const float *w_eff, const float *b_post0, const float *b_post1, float *u,
                                         gnnsaft_stream_t stream) {
  return gs::launch_pna_update_folded(x, agg, perm, tiles, num_tiles,
                                      gnnsaft_degree_tiles_capacity(num_nodes, hidden), num_nodes, hidden, w_eff,
                                      b_post0, b_post1, u, static_cast<hipStream_t>(stream));
}

extern "C" void gnnsaft_debug_set_gemm_config(int32_t cfg) { gs::debug_set_gemm_config(cfg); }
