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
#include "fold.hpp"
#include "k0_chain.hpp"
#include "w3.hpp"

namespace gs {

// pass 1: block_hist[b][d] = number of nodes with (clamped) in-degree d in block b
__global__ __launch_bounds__(kDegBlock) void k_degree_block_hist(const int32_t *__restrict__ rowptr, int64_t n,
                                                                 int32_t *__restrict__ block_hist, int32_t *err) {
  const int64_t i = (int64_t)blockIdx.x * kDegBlock + threadIdx.x;
  const bool live = i < n;
  const int d = live ? clamp_degree(rowptr[i + 1] - rowptr[i], err) : 0;
  block_degree_hist(d, live, block_hist);
}

// pass 2 (one workgroup): per-bucket exclusive scan over blocks (in place), bucket starts, tile table (k0_chain.hpp)
__global__ __launch_bounds__(256) void k_degree_plan(int32_t *__restrict__ block_hist, int64_t num_blocks,
                                                     int tile_rows, int32_t *__restrict__ hist,
                                                     int32_t *__restrict__ start, int32_t *__restrict__ tiles,
                                                     int32_t *__restrict__ num_tiles) {
  __shared__ DegreePlanLds s;
  degree_plan_body(block_hist, num_blocks, tile_rows, hist, start, tiles, num_tiles, s);
}

// pass 3: perm[start[d] + (nodes of degree d in earlier blocks / waves / lanes)] = node
__device__ __forceinline__ void degree_fill_body(const int32_t *__restrict__ rowptr, int64_t n,
                                                 const int32_t *__restrict__ block_base,
                                                 const int32_t *__restrict__ start, int32_t *__restrict__ perm,
                                                 int base_stride = 1 /* histogram entries per 1024-node block */) {
  __shared__ int32_t wcount[kDegBlock / 64][kDegreeBuckets];
  for (int t = threadIdx.x; t < (kDegBlock / 64) * kDegreeBuckets; t += kDegBlock) (&wcount[0][0])[t] = 0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * kDegBlock + threadIdx.x;
  const bool live = i < n;
  int d = live ? rowptr[i + 1] - rowptr[i] : 0;
  if (d >= kDegreeBuckets) d = kDegreeBuckets - 1;
  const int rank = wave_degree_ranks(d, live, wcount);
  __syncthreads();
  if (!live) return;
  const int wave = threadIdx.x >> 6;
  int before = 0;
  for (int w = 0; w < wave; ++w) before += wcount[w][d];
  perm[start[d] + block_base[(int64_t)blockIdx.x * base_stride * kDegreeBuckets + d] + before + rank] = (int32_t)i;
}

__global__ __launch_bounds__(kDegBlock) void k_degree_fill(const int32_t *__restrict__ rowptr, int64_t n,
                                                           const int32_t *__restrict__ block_base,
                                                           const int32_t *__restrict__ start,
                                                           int32_t *__restrict__ perm) {
  degree_fill_body(rowptr, n, block_base, start, perm);
}

__global__ __launch_bounds__(256) void k_dst_fold(FoldLayers fl, int f, double *__restrict__ g_all) {
  dst_fold_body(fl, f, g_all, blockIdx.x, blockIdx.y, blockIdx.z);
}

// slot: float4 index inside one [F/2, 5F] block; t: tower; dz: degree + kDegreeBuckets * layer
__device__ __forceinline__ void fold_post_weights_body(const FoldLayers &fl, const int32_t *__restrict__ hist, int f,
                                                       float *__restrict__ w_eff_all, int64_t layer_stride,
                                                       const double *__restrict__ g_all, int64_t slot, int t, int dz,
                                                       bool only_degree0 = false,
                                                       char *__restrict__ w_eff3_all = nullptr) {
  const int d = dz % kDegreeBuckets;
  const int layer = dz / kDegreeBuckets;
  if (only_degree0 ? d != 0 : hist[d] == 0) return;  // degree absent from this batch
  const float *w0 = fl.w0[layer], *w1 = fl.w1[layer], *avg = fl.avg[layer];
  float *w_eff = w_eff_all + layer * layer_stride;
  const int per_row4 = 5 * f / 4;
  const int64_t o = slot / per_row4;
  if (o >= f / 2) return;
  const int c = (int)(slot - o * per_row4) * 4;
  const float *w = (t == 0 ? w0 : w1) + o * (int64_t)(13 * f);
  float amp_f, att_f;
  degree_scalers(d, avg[0], amp_f, att_f);
  const double amp = (double)amp_f, att = (double)att_f;
  // float64 accumulate, ONE rounding per folded weight (fold.hpp)
  f32x4 v;
  if (c < f) {
    v = gs_ld4(w + c);
    if (g_all != nullptr && d > 0) {  // d == 0: no in-edge, the aggregates (and the P term) are zero
      const double *g = g_all + ((((int64_t)layer * 2 + t) * 3) * (f / 2) + o) * f + c;
      const int64_t gs = (int64_t)(f / 2) * f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        v[j] = (float)((double)v[j] + (g[j] + g[gs + j] * amp + g[2 * gs + j] * att));
    }
  } else {
    const f32x4 w_id = gs_ld4(w + c), w_amp = gs_ld4(w + 4 * f + c), w_att = gs_ld4(w + 8 * f + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (float)((double)w_id[j] + (double)w_amp[j] * amp + (double)w_att[j] * att);
  }
  gs_st4(w_eff + (((int64_t)d * 2 + t) * (f / 2) + o) * (int64_t)(5 * f) + c, v);
  // the same folded weight as bf16 planes for the GEMM that copies its B tiles straight into LDS (w3.hpp): one image of
  // [F/2, 5F] per (layer, degree, tower), in the block order of w_eff
  if (w_eff3_all != nullptr)
    w3_store4(w_eff3_all + (layer * layer_stride + ((int64_t)d * 2 + t) * (f / 2) * (int64_t)(5 * f)) * 6, f / 2, o, c, v);
}

__global__ __launch_bounds__(256) void k_fold_post_weights(FoldLayers fl, const int32_t *__restrict__ hist, int f,
                                                           float *__restrict__ w_eff_all, int64_t layer_stride,
                                                           const double *__restrict__ g_all,
                                                           char *__restrict__ w_eff3_all) {
  fold_post_weights_body(fl, hist, f, w_eff_all, layer_stride, g_all, (int64_t)blockIdx.x * blockDim.x + threadIdx.x,
                         blockIdx.y, blockIdx.z, false, w_eff3_all);
}

// pass 3 of the degree bucketing and the weight fold both need only the plan (bucket starts / histogram): one
// launch, workgroups [0, fill_blocks) fill the permutation, the rest fold the weights.
struct FoldJob {
  FoldLayers fl;
  const int32_t *hist;
  int f;
  float *w_eff_all;
  int64_t layer_stride;
  const double *g_all;
  int x_blocks;      // kDegBlock-thread workgroups per [F/2, 5F] block
  int num_layers;
  char *w_eff3_all;  // W3 images of the folded weights (6 bytes per weight, same block order), or null
};

// What the fill launch installs when the cooperative K0 chain of the prologue launch lost a grid barrier (the chain's
// "lost" word of THIS call says so; the sticky flag word tells the host): an EMPTY structure -- no rows, every node of
// in-degree 0, identity permutation -- so that nothing downstream indexes with half-built tables; the chain's
// persistent words (fill cursors, barrier, ticket) are restored to zero here, so the next call starts clean without
// the host's help; `lost_out` tells the end of the forward to write NaN (the outputs are garbage).
struct K0Sanitize {
  const int32_t *err = nullptr;   // null: never
  int32_t *sync = nullptr, *cursor = nullptr;   // persistent words of the chain (sync[kK0LostWord]: this call lost)
  int32_t *lost_out = nullptr;
  int32_t *rowptr = nullptr, *src = nullptr, *dst = nullptr, *combo = nullptr;
  int64_t ep = 0;
  float *log_amp = nullptr, *log_att = nullptr;
  int32_t *hist = nullptr, *start = nullptr, *tiles = nullptr, *num_tiles = nullptr;
  int tile_rows = 64;
};

__device__ __forceinline__ void k0_install_empty(const K0Sanitize &z, int64_t n, int32_t *__restrict__ perm,
                                                 unsigned fill_blocks) {
  const int64_t i = (int64_t)blockIdx.x * kDegBlock + threadIdx.x;
  if (i < n) {
    z.cursor[i] = 0;     // (the chain's phase that resets them did not run)
    z.rowptr[i] = 0;
    perm[i] = (int32_t)i;
    z.log_amp[i] = degree_log_amp(0);
    z.log_att[i] = degree_log_att(0);
  }
  for (int64_t r = i; r < z.ep; r += (int64_t)fill_blocks * kDegBlock) z.src[r] = z.dst[r] = z.combo[r] = 0;
  if (blockIdx.x != 0) return;
  if (threadIdx.x == 0) {
    z.rowptr[n] = 0;
    z.num_tiles[0] = (int32_t)((n + z.tile_rows - 1) / z.tile_rows);
    z.sync[0] = 0;       // barrier and ticket (workgroups that gave up at different times may have left them anywhere);
    z.sync[1] = 0;       // the lost word itself is still being read by this launch: the pooling launch zeroes it
  }
  if (threadIdx.x < kDegreeBuckets) {
    z.hist[threadIdx.x] = threadIdx.x == 0 ? (int32_t)n : 0;
    z.start[threadIdx.x] = threadIdx.x == 0 ? 0 : (int32_t)n;
  }
  const int64_t nt = (n + z.tile_rows - 1) / z.tile_rows;
  for (int64_t t = threadIdx.x; t < nt; t += kDegBlock) {
    const int64_t o = t * z.tile_rows;
    z.tiles[4 * t + 0] = 0;
    z.tiles[4 * t + 1] = (int32_t)o;
    z.tiles[4 * t + 2] = (int32_t)(n - o < z.tile_rows ? n - o : z.tile_rows);
    z.tiles[4 * t + 3] = 0;
  }
}

__global__ __launch_bounds__(kDegBlock) void k_degree_fill_and_fold(const int32_t *__restrict__ rowptr, int64_t n,
                                                                    const int32_t *__restrict__ block_base,
                                                                    const int32_t *__restrict__ start,
                                                                    int32_t *__restrict__ perm, unsigned fill_blocks,
                                                                    FoldJob job, K0Sanitize z) {
  const bool lost = z.err != nullptr && __hip_atomic_load(z.sync + kK0LostWord, __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT) != 0;   // grid-uniform, THIS call
  if (blockIdx.x == 0 && threadIdx.x == 0 && z.lost_out != nullptr) z.lost_out[0] = lost ? 1 : 0;
  if (blockIdx.x < fill_blocks) {  // block-uniform
    if (lost)
      k0_install_empty(z, n, perm, fill_blocks);
    else
      degree_fill_body(rowptr, n, block_base, start, perm, z.err != nullptr ? kDegBlock / kK0Group : 1);
    return;
  }
  const unsigned b = blockIdx.x - fill_blocks;
  const int xb = b % job.x_blocks, rest = b / job.x_blocks;
  fold_post_weights_body(job.fl, job.hist, job.f, job.w_eff_all, job.layer_stride, job.g_all,
                         (int64_t)xb * kDegBlock + threadIdx.x, rest & 1, rest >> 1, lost, job.w_eff3_all);
}

}  // namespace gs

extern "C" int64_t gnnsaft_degree_tiles_capacity(int64_t num_nodes, int32_t hidden) {
  const int tr = gs::pna_fold_tile_rows(hidden);
  return gs_ceil_div(num_nodes > 0 ? num_nodes : 1, tr) + gs::kDegreeBuckets;
}

extern "C" int32_t gnnsaft_degree_buckets(void) { return gs::kDegreeBuckets; }

extern "C" size_t gnnsaft_degree_scratch_ints(int64_t num_nodes) {
  // hist[B] | start[B] | block_hist[ceil(N/1024)][B] (the cooperative chain of the forward: per group of 256 nodes)
  return (size_t)(2 + gs_ceil_div(num_nodes > 0 ? num_nodes : 1, (int64_t)gs::kK0Group)) * gs::kDegreeBuckets;
}

int gs::launch_degree_tiles(const int32_t *rowptr, int64_t num_nodes, int32_t hidden, int32_t *perm, int32_t *tiles,
                            int32_t *num_tiles, int32_t *scratch, int32_t *err_flag, bool have_block_hist,
                            hipStream_t st, const DegreeFoldRequest *fold, const K0Installed *installed) {
  GS_REQUIRE(rowptr && perm && tiles && num_tiles && scratch, GNNSAFT_ERR_NULL);
  GS_REQUIRE(num_nodes >= 1 && hidden >= 32 && (hidden % 32) == 0, GNNSAFT_ERR_SHAPE);
  int32_t *hist = scratch, *start = scratch + gs::kDegreeBuckets, *block_hist = scratch + 2 * gs::kDegreeBuckets;
  const int64_t nb = gs_ceil_div(num_nodes, gs::kDegBlock);
  // `installed`: the cooperative chain of the prologue launch already left the histogram, the plan and the tile table
  GS_REQUIRE(installed == nullptr || fold != nullptr, GNNSAFT_ERR_SHAPE);
  if (installed == nullptr) {
    if (!have_block_hist)  // otherwise launch_csr_build's last kernel already left it there
      hipLaunchKernelGGL(gs::k_degree_block_hist, dim3((unsigned)nb), dim3(gs::kDegBlock), 0, st, rowptr, num_nodes,
                         block_hist, err_flag);
    hipLaunchKernelGGL(gs::k_degree_plan, dim3(1), dim3(256), 0, st, block_hist, nb, gs::pna_fold_tile_rows(hidden),
                       hist, start, tiles, num_tiles);
  }
  if (fold == nullptr) {
    hipLaunchKernelGGL(gs::k_degree_fill, dim3((unsigned)nb), dim3(gs::kDegBlock), 0, st, rowptr, num_nodes,
                       block_hist, start, perm);
  } else {  // the degree-folded update weights of all layers ride along (they need only `hist`)
    GS_REQUIRE(fold->num_layers >= 1 && fold->num_layers <= GNNSAFT_MAX_FOLD_LAYERS && fold->w_eff != nullptr,
               GNNSAFT_ERR_SHAPE);
    gs::FoldJob job;
    const bool fold_dst = fold->g_all != nullptr;
    for (int i = 0; i < GNNSAFT_MAX_FOLD_LAYERS; ++i) {
      const int j = i < fold->num_layers ? i : 0;
      job.fl.w0[i] = fold->w_post0[j];
      job.fl.w1[i] = fold->w_post1[j];
      job.fl.avg[i] = fold->avg[j];
      job.fl.pre0[i] = nullptr;
      job.fl.pre1[i] = nullptr;
      GS_REQUIRE(job.fl.w0[i] && job.fl.w1[i] && job.fl.avg[i], GNNSAFT_ERR_NULL);
    }
    job.hist = hist;
    job.f = hidden;
    job.w_eff_all = fold->w_eff;
    job.layer_stride = fold->layer_stride;
    job.g_all = fold_dst ? fold->g_all : nullptr;
    const int64_t threads = (int64_t)(hidden / 2) * (5 * hidden / 4);
    job.x_blocks = (int)gs_ceil_div(threads, (int64_t)gs::kDegBlock);
    job.num_layers = fold->num_layers;
    job.w_eff3_all = fold->w_eff3;
    const int64_t fold_blocks = (int64_t)job.x_blocks * 2 * gs::kDegreeBuckets * fold->num_layers;
    gs::K0Sanitize z;
    if (installed != nullptr) {
      z.err = err_flag;
      z.rowptr = installed->rowptr;
      z.src = installed->src;
      z.dst = installed->dst;
      z.combo = installed->combo;
      z.ep = installed->ep;
      z.log_amp = installed->log_amp;
      z.log_att = installed->log_att;
      z.sync = installed->sync;
      z.cursor = installed->cursor;
      z.lost_out = installed->lost_out;
      GS_REQUIRE(z.sync != nullptr && z.cursor != nullptr, GNNSAFT_ERR_NULL);
      z.hist = hist;
      z.start = start;
      z.tiles = tiles;
      z.num_tiles = num_tiles;
      z.tile_rows = gs::pna_fold_tile_rows(hidden);
    }
    hipLaunchKernelGGL(gs::k_degree_fill_and_fold, dim3((unsigned)(nb + fold_blocks)), dim3(gs::kDegBlock), 0, st, rowptr,
                       num_nodes, block_hist, start, perm, (unsigned)nb, job, z);
  }
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

extern "C" int gnnsaft_degree_tiles(const int32_t *rowptr, int64_t num_nodes, int32_t hidden, int32_t *perm,
                                    int32_t *tiles, int32_t *num_tiles, int32_t *scratch, int32_t *err_flag,
                                    gnnsaft_stream_t stream) {
  return gs::launch_degree_tiles(rowptr, num_nodes, hidden, perm, tiles, num_tiles, scratch, err_flag, false,
                                 static_cast<hipStream_t>(stream), nullptr);
}

extern "C" int gnnsaft_pna_fold_post_weights_multi(int32_t num_layers, const float *const *w_post0_host,
                                                   const float *const *w_post1_host,
                                                   const float *const *avg_deg_log_host,
                                                   const float *const *w_pre0_host, const float *const *w_pre1_host,
                                                   void *g_scratch, const int32_t *hist, int32_t hidden,
                                                   float *w_eff, int64_t layer_stride, gnnsaft_stream_t stream) {
  GS_REQUIRE((reinterpret_cast<uintptr_t>(g_scratch) & 7) == 0, GNNSAFT_ERR_SHAPE);
  return gs::launch_fold_post_weights(num_layers, w_post0_host, w_post1_host, avg_deg_log_host, w_pre0_host,
                                      w_pre1_host, static_cast<double *>(g_scratch), hist, hidden, w_eff, layer_stride, 3,
                                      static_cast<hipStream_t>(stream), nullptr);
}

int gs::launch_fold_post_weights(int32_t num_layers, const float *const *w_post0_host,
                                 const float *const *w_post1_host, const float *const *avg_deg_log_host,
                                 const float *const *w_pre0_host, const float *const *w_pre1_host, double *g_scratch,
                                 const int32_t *hist, int32_t hidden, float *w_eff, int64_t layer_stride, int phases,
                                 hipStream_t st, char *w_eff3) {
  GS_REQUIRE(w_post0_host && w_post1_host && avg_deg_log_host && hist && w_eff, GNNSAFT_ERR_NULL);
  GS_REQUIRE(hidden >= 32 && (hidden % 32) == 0, GNNSAFT_ERR_SHAPE);
  GS_REQUIRE(num_layers >= 1 && num_layers <= GNNSAFT_MAX_FOLD_LAYERS, GNNSAFT_ERR_SHAPE);
  const bool fold_dst = w_pre0_host != nullptr;
  GS_REQUIRE(!fold_dst || (hidden % 64) == 0, GNNSAFT_ERR_SHAPE);  // k_dst_fold works on 32x32 tiles of [F/2, F]
  GS_REQUIRE(!fold_dst || (w_pre1_host != nullptr && g_scratch != nullptr), GNNSAFT_ERR_NULL);
  gs::FoldLayers fl;
  for (int i = 0; i < GNNSAFT_MAX_FOLD_LAYERS; ++i) {
    const int j = i < num_layers ? i : 0;
    fl.w0[i] = w_post0_host[j];
    fl.w1[i] = w_post1_host[j];
    fl.avg[i] = avg_deg_log_host[j];
    fl.pre0[i] = fold_dst ? w_pre0_host[j] : nullptr;
    fl.pre1[i] = fold_dst ? w_pre1_host[j] : nullptr;
    GS_REQUIRE(fl.w0[i] && fl.w1[i] && fl.avg[i], GNNSAFT_ERR_NULL);
    GS_REQUIRE(!fold_dst || (fl.pre0[i] && fl.pre1[i]), GNNSAFT_ERR_NULL);
  }
  if (fold_dst && (phases & 1)) {
    const dim3 gridg((unsigned)(hidden / 32), (unsigned)(hidden / 2 / 32), (unsigned)(6 * num_layers));
    hipLaunchKernelGGL(gs::k_dst_fold, gridg, dim3(256), 0, st, fl, hidden, g_scratch);
  }
  if (phases & 2) {
    const int64_t threads = (int64_t)(hidden / 2) * (5 * hidden / 4);
    const dim3 grid((unsigned)gs_ceil_div(threads, 256), 2, (unsigned)(gs::kDegreeBuckets * num_layers));
    GS_REQUIRE(w_eff3 == nullptr || (hidden % 64) == 0, GNNSAFT_ERR_SHAPE);
    hipLaunchKernelGGL(gs::k_fold_post_weights, grid, dim3(256), 0, st, fl, hist, hidden, w_eff, layer_stride,
                       fold_dst ? g_scratch : nullptr, w_eff3);
  }
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

extern "C" int gnnsaft_pna_fold_post_weights(const float *w_post0, const float *w_post1, const float *avg_deg_log,
                                             const int32_t *hist, int32_t hidden, float *w_eff,
                                             gnnsaft_stream_t stream) {
  return gnnsaft_pna_fold_post_weights_multi(1, &w_post0, &w_post1, &avg_deg_log, nullptr, nullptr, nullptr, hist,
                                             hidden, w_eff, 0, stream);
}

extern "C" int gnnsaft_pna_update_folded(const float *x, const float *agg, const int32_t *perm, const int32_t *tiles,
                                         const int32_t *num_tiles, int64_t num_nodes, int32_t hidden,
                                         const float *w_eff, const float *b_post0, const float *b_post1, float *u,
                                         gnnsaft_stream_t stream) {
  return gs::launch_pna_update_folded(x, agg, perm, tiles, num_tiles,
                                      gnnsaft_degree_tiles_capacity(num_nodes, hidden), num_nodes, hidden, w_eff,
                                      b_post0, b_post1, u, static_cast<hipStream_t>(stream));
}
