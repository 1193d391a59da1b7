// K0 -- destination-sorted CSR of a PyG edge list, built once per batch and
// reused by every layer.  Replaces add_self_loops
// (/root/reference/gnnepcsaft/train/models.py:118-121) and the index plumbing
// PyG re-derives per layer from the unsorted `edge_index` (index_select for
// x_i / x_j, scatter index, degree()).  Self-loops are implicit: row
// rowptr[i+1]-1 of every node is (src=i, combo=0), i.e. the loop PyG appends
// after all real edges with attributes [0,0,0].
//
// histogram (int atomics) -> 3-phase exclusive scan -> atomic fill -> per-node
// ascending sort of the segment's edge ids (segments are molecular in-degrees,
// 1..5) so that the row order, hence every floating-point sum downstream, is
// independent of atomic arrival order.
#include "common.hpp"
#include "k0_chain.hpp"

namespace gs {

__global__ void k_zero_i32(int32_t *p, int64_t n, int32_t *also = nullptr) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0;
  if (i == 0 && also != nullptr) also[0] = 0;
}


// in-degree histogram; the same launch optionally converts the PyG `batch` vector to graph offsets (an
// independent job with the same parallel shape: one thread per node)
__global__ void k_count_in_degree(const int64_t *__restrict__ edge_index, int64_t n, int64_t e,
                                  int32_t *__restrict__ counts, int32_t *err,
                                  const int64_t *__restrict__ batch = nullptr, int64_t g = 0,
                                  int32_t *__restrict__ graph_ptr = nullptr) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (graph_ptr != nullptr) {
    if (batch != nullptr) {
      batch_to_ptr_slot(batch, n, g, graph_ptr, err, i);
    } else if (i == 0) {  // un-batched Data: one graph spanning all nodes
      graph_ptr[0] = 0;
      graph_ptr[1] = (int32_t)n;
    }
  }
  if (i >= e) return;
  const int64_t s = edge_index[i];
  const int64_t d = edge_index[e + i];
  if (s < 0 || s >= n || d < 0 || d >= n) {
    if (err) atomicOr(err, GNNSAFT_FLAG_BAD_EDGE);
    return;  // dropped
  }
  atomicAdd(&counts[d], 1);
}

__global__ __launch_bounds__(kScanBlock) void k_scan_tiles(const int32_t *__restrict__ counts, int64_t n, int extra,
                                                           int32_t *__restrict__ rowptr,
                                                           int32_t *__restrict__ tile_sums,
                                                           int clamp = 0x7fffffff) {
  __shared__ int lds[kScanBlock / 64];
  scan_tile_body(counts, n, extra, rowptr, tile_sums, clamp, blockIdx.x, lds);
}

// single workgroup: exclusive scan of the tile sums, in place; writes the grand total to rowptr[n]
__global__ __launch_bounds__(kScanBlock) void k_scan_tile_sums(int32_t *__restrict__ tile_sums, int64_t num_tiles,
                                                               int32_t *__restrict__ rowptr, int64_t n) {
  __shared__ int lds[kScanBlock / 64];
  int carry = 0;
  for (int64_t b = 0; b < num_tiles; b += kScanBlock) {
    const int64_t i = b + threadIdx.x;
    const int v = i < num_tiles ? tile_sums[i] : 0;
    int total;
    const int off = block_exclusive_scan(v, lds, total);
    if (i < num_tiles) tile_sums[i] = carry + off;
    carry += total;
  }
  if (threadIdx.x == 0) rowptr[n] = carry;
}

__global__ void k_scan_add(int32_t *__restrict__ rowptr, const int32_t *__restrict__ tile_sums, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) rowptr[i] += tile_sums[i / kScanTile];
}

// phases 2 + 3 in one launch when there are few tiles: every workgroup sums the tile totals in front of its own
// tile (its 256 entries lie in one tile); the last workgroup also writes the grand total to rowptr[n]
constexpr int kScanFusedTiles = 4096;
__global__ __launch_bounds__(256) void k_scan_add_fused(int32_t *__restrict__ rowptr,
                                                        const int32_t *__restrict__ tile_sums, int64_t num_tiles,
                                                        int64_t n) {
  __shared__ int lds[4];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t tile = ((int64_t)blockIdx.x * blockDim.x) / kScanTile;
  const bool last = blockIdx.x == gridDim.x - 1;
  const int64_t upto = last ? num_tiles : tile;
  int before = 0, all = 0;
  for (int64_t t = threadIdx.x; t < upto; t += blockDim.x) {
    const int v = tile_sums[t];
    all += v;
    if (t < tile) before += v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    before += __shfl_xor(before, o);
    all += __shfl_xor(all, o);
  }
  __shared__ int lds_all[4];
  if ((threadIdx.x & 63) == 0) {
    lds[threadIdx.x >> 6] = before;
    lds_all[threadIdx.x >> 6] = all;
  }
  __syncthreads();
  before = lds[0] + lds[1] + lds[2] + lds[3];
  if (i < n) rowptr[i] += before;
  if (last && threadIdx.x == 0) rowptr[n] = lds_all[0] + lds_all[1] + lds_all[2] + lds_all[3];
}

// exclusive scan of (counts + extra) into rowptr[0..n], rowptr[n] = total
static void launch_exclusive_scan(const int32_t *counts, int64_t n, int extra, int32_t *rowptr, int32_t *tile_sums,
                                  hipStream_t st) {
  const int64_t tiles = gs_ceil_div(n > 0 ? n : 1, kScanTile);
  hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)tiles), dim3(kScanBlock), 0, st, counts, n, extra, rowptr,
                     tile_sums);
  if (tiles <= kScanFusedTiles) {
    hipLaunchKernelGGL(k_scan_add_fused, dim3((unsigned)gs_ceil_div(n > 0 ? n : 1, 256)), dim3(256), 0, st, rowptr,
                       tile_sums, tiles, n);
  } else {
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(1), dim3(kScanBlock), 0, st, tile_sums, tiles, rowptr, n);
    if (n > 0)
      hipLaunchKernelGGL(k_scan_add, dim3((unsigned)gs_ceil_div(n, 256)), dim3(256), 0, st, rowptr, tile_sums, n);
  }
}

__global__ void k_fill_edge_ids(const int64_t *__restrict__ edge_index, int64_t n, int64_t e,
                                const int32_t *__restrict__ rowptr, int32_t *__restrict__ cursor,
                                int32_t *__restrict__ eid) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= e) return;
  const int64_t s = edge_index[i];
  const int64_t d = edge_index[e + i];
  if (s < 0 || s >= n || d < 0 || d >= n) return;
  const int pos = rowptr[d] + atomicAdd(&cursor[d], 1);
  eid[pos] = (int32_t)i;
}

// one thread per node: order the segment, emit src / dst / combo rows and the degree scaler logs
__device__ __forceinline__ int finish_row(const int64_t *__restrict__ edge_index,
                                          const int64_t *__restrict__ edge_attr, int64_t i, const BondDims &bd,
                                          int self_loops, const int32_t *__restrict__ rowptr,
                                          const int32_t *__restrict__ counts, int32_t *__restrict__ eid,
                                          int32_t *__restrict__ src, int32_t *__restrict__ dst,
                                          int32_t *__restrict__ combo, float *__restrict__ log_amp,
                                          float *__restrict__ log_att, int32_t *err) {
  const int beg = rowptr[i];
  const int cnt = counts[i];
  // insertion sort, ascending edge id (stable wrt. the edge list)
  for (int a = 1; a < cnt; ++a) {
    const int key = eid[beg + a];
    int b = a - 1;
    while (b >= 0 && eid[beg + b] > key) {
      eid[beg + b + 1] = eid[beg + b];
      --b;
    }
    eid[beg + b + 1] = key;
  }
  for (int a = 0; a < cnt; ++a) {
    const int64_t id = eid[beg + a];
    int cid = 0;
    for (int k = 0; k < bd.n; ++k) {
      int64_t v = edge_attr[id * bd.n + k];
      if (v < 0 || v >= bd.dims[k]) {
        if (err) atomicOr(err, GNNSAFT_FLAG_BAD_ATTR);
        v = 0;
      }
      cid = cid * bd.dims[k] + (int)v;
    }
    src[beg + a] = (int32_t)edge_index[id];
    dst[beg + a] = (int32_t)i;
    combo[beg + a] = cid;
  }
  int deg = cnt;
  if (self_loops) {
    src[beg + cnt] = (int32_t)i;
    dst[beg + cnt] = (int32_t)i;
    combo[beg + cnt] = 0;
    deg += 1;
  }
  log_amp[i] = degree_log_amp(deg);
  log_att[i] = degree_log_att(deg);
  return deg;
}

__global__ void k_finish_rows(const int64_t *__restrict__ edge_index, const int64_t *__restrict__ edge_attr,
                              int64_t n, int64_t e, BondDims bd, int self_loops,
                              const int32_t *__restrict__ rowptr, const int32_t *__restrict__ counts,
                              int32_t *__restrict__ eid, int32_t *__restrict__ src, int32_t *__restrict__ dst,
                              int32_t *__restrict__ combo, float *__restrict__ log_amp,
                              float *__restrict__ log_att, int32_t *err, int32_t *__restrict__ block_hist) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;
  int deg = 0;
  if (live)
    deg = finish_row(edge_index, edge_attr, i, bd, self_loops, rowptr, counts, eid, src, dst, combo, log_amp, log_att,
                     err);
  // first pass of the degree bucketing (degree.hip) for free: launched with kDegBlock threads in that case
  if (block_hist != nullptr) block_degree_hist(live ? clamp_degree(deg, err) : 0, live, block_hist);
}

// ---- bounded in-degree (the folded update: every in-degree < kDegreeBuckets, anything above is an input error that
// the degree bucketing flags): the edge ids go to a fixed row of kCsrSlots slots per destination node in ONE pass
// over the edge list -- no histogram pass before the fill -- and the row offsets are completed by the kernel that
// orders and emits the rows.  prologue (zero) -> k_fill_slots -> k_scan_tiles -> k_finish_rows_slots: two launches
// fewer than the general chain, and the per-node sort runs in registers instead of through memory.  (Bodies in
// k0_chain.hpp: the forward runs the same chain as workgroups of its prologue launch where it can.)
__global__ void k_fill_slots(const int64_t *__restrict__ edge_index, int64_t n, int64_t e,
                             int32_t *__restrict__ cursor, int32_t *__restrict__ slots, int32_t *err,
                             const int64_t *__restrict__ batch, int64_t g, int32_t *__restrict__ graph_ptr) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (graph_ptr != nullptr) {
    if (batch != nullptr) {
      batch_to_ptr_slot(batch, n, g, graph_ptr, err, i);
    } else if (i == 0) {  // un-batched Data: one graph spanning all nodes
      graph_ptr[0] = 0;
      graph_ptr[1] = (int32_t)n;
    }
  }
  fill_slot_body(edge_index, n, e, cursor, slots, err, i);
}

// blockDim.x divides kScanTile; block_hist != nullptr: launched with kDegBlock threads (first pass of the degree
// bucketing for free)
__global__ void k_finish_rows_slots(const int64_t *__restrict__ edge_index, const int64_t *__restrict__ edge_attr,
                                    int64_t n, int64_t e, BondDims bd, int self_loops,
                                    int32_t *__restrict__ rowptr, const int32_t *__restrict__ tile_sums,
                                    int64_t num_tiles, const int32_t *__restrict__ cursor,
                                    int32_t *__restrict__ slots, int32_t *__restrict__ src,
                                    int32_t *__restrict__ dst, int32_t *__restrict__ combo,
                                    float *__restrict__ log_amp, float *__restrict__ log_att, int32_t *err,
                                    int32_t *__restrict__ block_hist) {
  __shared__ int s_before[16], s_all[16];
  const int deg = finish_rows_slots_body(edge_index, edge_attr, n, bd, self_loops, rowptr, tile_sums, num_tiles, cursor,
                                         slots, src, dst, combo, log_amp, log_att, err, blockIdx.x, (int)blockDim.x,
                                         blockIdx.x == gridDim.x - 1, s_before, s_all);
  const bool live = (int64_t)blockIdx.x * blockDim.x + threadIdx.x < n;
  if (block_hist != nullptr) block_degree_hist(live ? clamp_degree(deg, err) : 0, live, block_hist);
}

__global__ void k_batch_to_ptr(const int64_t *__restrict__ batch, int64_t n, int64_t g, int32_t *__restrict__ ptr,
                               int32_t *err) {
  batch_to_ptr_slot(batch, n, g, ptr, err, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// ---- group an int32 key array (e.g. the CSR rows' source node) by key: the transposed CSR that
//      the backward pass uses to turn the scatter-add of message gradients into a gather
__global__ void k_count_keys(const int32_t *__restrict__ keys, int64_t count, int64_t num_keys,
                             int32_t *__restrict__ counts) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int32_t k = i < count ? keys[i] : -1;
  const bool live = k >= 0 && k < num_keys;
  // one atomic per (wave, distinct key): a few hot keys (edge classes) would otherwise serialise
  unsigned long long todo = __ballot(live);
  while (todo != 0ull) {
    const int leader = __ffsll((long long)todo) - 1;
    const int kl = __shfl(k, leader);
    const unsigned long long same = __ballot(live && k == kl);
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&counts[kl], __popcll(same));
    todo &= ~same;
  }
}

__global__ void k_fill_by_key(const int32_t *__restrict__ keys, int64_t count, int64_t num_keys,
                              const int32_t *__restrict__ rowptr, int32_t *__restrict__ cursor,
                              int32_t *__restrict__ rows) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int32_t k = i < count ? keys[i] : -1;
  const bool live = k >= 0 && k < num_keys;
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(live);
  while (todo != 0ull) {
    const int leader = __ffsll((long long)todo) - 1;
    const int kl = __shfl(k, leader);
    const unsigned long long same = __ballot(live && k == kl);
    int base = 0;
    if (lane == leader) base = atomicAdd(&cursor[kl], __popcll(same));
    base = __shfl(base, leader);
    if (live && k == kl) rows[rowptr[kl] + base + __popcll(same & ((1ull << lane) - 1ull))] = (int32_t)i;
    todo &= ~same;
  }
}

__global__ void k_sort_segments(const int32_t *__restrict__ rowptr, int64_t num_keys, int32_t *__restrict__ rows) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= num_keys) return;
  const int beg = rowptr[k], end = rowptr[k + 1];
  for (int a = beg + 1; a < end; ++a) {  // ascending row id: fixed summation order downstream
    const int key = rows[a];
    int b = a - 1;
    while (b >= beg && rows[b] > key) {
      rows[b + 1] = rows[b];
      --b;
    }
    rows[b + 1] = key;
  }
}

size_t group_by_key_workspace_bytes(int64_t num_keys) {
  const size_t tiles = (size_t)gs_ceil_div(num_keys > 0 ? num_keys : 1, kScanTile);
  return gs_align_up((size_t)num_keys * 4, 256) * 2 + gs_align_up(tiles * 4, 256);
}

int launch_group_by_key(const int32_t *keys, int64_t count, int64_t num_keys, int32_t *rowptr, int32_t *rows,
                        void *workspace, size_t workspace_bytes, int sort_segments, hipStream_t st) {
  GS_REQUIRE(keys && rowptr && rows && workspace, GNNSAFT_ERR_NULL);
  GS_REQUIRE(num_keys >= 1 && count >= 0, GNNSAFT_ERR_SHAPE);
  GS_REQUIRE(workspace_bytes >= group_by_key_workspace_bytes(num_keys), GNNSAFT_ERR_WORKSPACE);
  char *ws = static_cast<char *>(workspace);
  int32_t *counts = reinterpret_cast<int32_t *>(ws);
  int32_t *cursor = reinterpret_cast<int32_t *>(ws + gs_align_up((size_t)num_keys * 4, 256));
  int32_t *tile_sums = reinterpret_cast<int32_t *>(ws + 2 * gs_align_up((size_t)num_keys * 4, 256));
  const int64_t zero_ints = 2 * (int64_t)gs_align_up((size_t)num_keys * 4, 256) / 4;
  hipLaunchKernelGGL(k_zero_i32, dim3((unsigned)gs_ceil_div(zero_ints, 256)), dim3(256), 0, st, counts, zero_ints);
  if (count > 0)
    hipLaunchKernelGGL(k_count_keys, dim3((unsigned)gs_ceil_div(count, 256)), dim3(256), 0, st, keys, count, num_keys,
                       counts);
  launch_exclusive_scan(counts, num_keys, 0, rowptr, tile_sums, st);
  if (count > 0) {
    hipLaunchKernelGGL(k_fill_by_key, dim3((unsigned)gs_ceil_div(count, 256)), dim3(256), 0, st, keys, count, num_keys,
                       rowptr, cursor, rows);
    // insertion sort per segment: only for short segments (node degrees), never for the few huge edge classes
    if (sort_segments)
      hipLaunchKernelGGL(k_sort_segments, dim3((unsigned)gs_ceil_div(num_keys, 256)), dim3(256), 0, st, rowptr,
                         num_keys, rows);
  }
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

__global__ void k_single_graph_ptr(int32_t *ptr, int32_t n) {
  ptr[0] = 0;
  ptr[1] = n;
}

}  // namespace gs

extern "C" size_t gnnsaft_csr_workspace_bytes(int64_t num_nodes, int64_t num_edges) {
  // counts[N] + cursor[N] + tile_sums[tiles] + eid[E + N] (general chain) or slots[N][kCsrSlots] (bounded in-degree)
  const size_t tiles = (size_t)gs_ceil_div(num_nodes > 0 ? num_nodes : 1, gs::kScanTile);
  const size_t eid = (size_t)(num_edges + num_nodes + 1), slots = (size_t)num_nodes * gs::kCsrSlots;
  return gs_align_up((size_t)num_nodes * 4, 256) * 2 + gs_align_up(tiles * 4, 256) +
         gs_align_up((eid > slots ? eid : slots) * 4, 256);
}

// the int32 region launch_csr_build expects zeroed (in-degree counts + fill cursors) when told `counts_zeroed`
void gs::csr_zero_region(void *workspace, int64_t num_nodes, int32_t **ptr, int64_t *count) {
  *ptr = static_cast<int32_t *>(workspace);
  *count = 2 * (int64_t)gs_align_up((size_t)num_nodes * 4, 256) / 4;
}

void gs::csr_workspace_parts(void *workspace, int64_t num_nodes, int32_t **cursor, int32_t **tile_sums,
                             int32_t **slots) {
  const int64_t tiles = gs_ceil_div(num_nodes > 0 ? num_nodes : 1, gs::kScanTile);
  char *ws = static_cast<char *>(workspace) + gs_align_up((size_t)num_nodes * 4, 256);   // (counts first)
  *cursor = reinterpret_cast<int32_t *>(ws);
  ws += gs_align_up((size_t)num_nodes * 4, 256);
  *tile_sums = reinterpret_cast<int32_t *>(ws);
  ws += gs_align_up((size_t)tiles * 4, 256);
  *slots = reinterpret_cast<int32_t *>(ws);
}

int gs::launch_csr_build(const int64_t *edge_index, const int64_t *edge_attr, int64_t num_nodes, int64_t num_edges,
                         int32_t num_bond_cols, const int32_t *bond_dims_host, int32_t self_loops, int32_t *rowptr,
                         int32_t *src, int32_t *dst, int32_t *combo, float *log_amp, float *log_att,
                         int32_t *err_flag, void *workspace, size_t workspace_bytes, const int64_t *batch,
                         int64_t num_graphs, int32_t *graph_ptr, int32_t *degree_block_hist, bool counts_zeroed,
                         hipStream_t st, bool bounded_degree) {
  GS_REQUIRE(rowptr && src && dst && combo && log_amp && log_att && workspace, GNNSAFT_ERR_NULL);
  GS_REQUIRE(num_edges == 0 || (edge_index != nullptr && edge_attr != nullptr), GNNSAFT_ERR_NULL);
  GS_REQUIRE(num_nodes >= 0 && num_edges >= 0, GNNSAFT_ERR_SHAPE);
  GS_REQUIRE(num_nodes + num_edges < ((int64_t)1 << 31) - 1, GNNSAFT_ERR_SHAPE);
  GS_REQUIRE(num_bond_cols >= 0 && num_bond_cols <= GNNSAFT_MAX_TABLES, GNNSAFT_ERR_SHAPE);
  GS_REQUIRE(workspace_bytes >= gnnsaft_csr_workspace_bytes(num_nodes, num_edges), GNNSAFT_ERR_WORKSPACE);
  GS_REQUIRE(graph_ptr == nullptr || batch != nullptr || num_graphs == 1, GNNSAFT_ERR_SHAPE);
  const int64_t n = num_nodes, e = num_edges;
  const int64_t tiles = gs_ceil_div(n > 0 ? n : 1, gs::kScanTile);
  char *ws = static_cast<char *>(workspace);
  int32_t *counts = reinterpret_cast<int32_t *>(ws);
  ws += gs_align_up((size_t)n * 4, 256);
  int32_t *cursor = reinterpret_cast<int32_t *>(ws);
  ws += gs_align_up((size_t)n * 4, 256);
  int32_t *tile_sums = reinterpret_cast<int32_t *>(ws);
  ws += gs_align_up((size_t)tiles * 4, 256);
  int32_t *eid = reinterpret_cast<int32_t *>(ws);

  gs::BondDims bd;
  bd.n = num_bond_cols;
  int64_t combos = 1;
  for (int k = 0; k < GNNSAFT_MAX_TABLES; ++k) {
    bd.dims[k] = k < num_bond_cols ? bond_dims_host[k] : 1;
    GS_REQUIRE(bd.dims[k] >= 1, GNNSAFT_ERR_SHAPE);
    combos *= bd.dims[k];
    GS_REQUIRE(combos <= (1 << 20), GNNSAFT_ERR_UNSUPPORTED);
  }

  const int tb = 256;
  if (n > 0 && !counts_zeroed) {
    // counts and cursor are adjacent up to alignment: zero both
    hipLaunchKernelGGL(gs::k_zero_i32, dim3((unsigned)gs_ceil_div(2 * (int64_t)gs_align_up((size_t)n * 4, 256) / 4, tb)),
                       dim3(tb), 0, st, counts, 2 * (int64_t)gs_align_up((size_t)n * 4, 256) / 4, nullptr);
  }
  const int64_t count_threads = graph_ptr != nullptr ? (e > n + 1 ? e : n + 1) : e;
  if (bounded_degree && n > 0) {
    // in-degree < kDegreeBuckets promised (violations are flagged, the surplus edges dropped): slotted chain
    if (count_threads > 0)
      hipLaunchKernelGGL(gs::k_fill_slots, dim3((unsigned)gs_ceil_div(count_threads, tb)), dim3(tb), 0, st, edge_index,
                         n, e, cursor, eid, err_flag, batch, num_graphs, graph_ptr);
    hipLaunchKernelGGL(gs::k_scan_tiles, dim3((unsigned)tiles), dim3(gs::kScanBlock), 0, st, cursor, n,
                       self_loops ? 1 : 0, rowptr, tile_sums, gs::kCsrSlots);
    const int fb = degree_block_hist != nullptr ? gs::kDegBlock : tb;
    hipLaunchKernelGGL(gs::k_finish_rows_slots, dim3((unsigned)gs_ceil_div(n, fb)), dim3(fb), 0, st, edge_index,
                       edge_attr, n, e, bd, self_loops ? 1 : 0, rowptr, tile_sums, tiles, cursor, eid, src, dst, combo,
                       log_amp, log_att, err_flag, degree_block_hist);
    GS_CHECK_LAUNCH();
    return GNNSAFT_OK;
  }
  // in-degree histogram (+ graph offsets from `batch` in the same launch when asked for)
  if (n > 0 && count_threads > 0)
    hipLaunchKernelGGL(gs::k_count_in_degree, dim3((unsigned)gs_ceil_div(count_threads, tb)), dim3(tb), 0, st,
                       edge_index, n, e, counts, err_flag, batch, num_graphs, graph_ptr);
  gs::launch_exclusive_scan(counts, n, self_loops ? 1 : 0, rowptr, tile_sums, st);
  if (n > 0) {
    if (e > 0)
      hipLaunchKernelGGL(gs::k_fill_edge_ids, dim3((unsigned)gs_ceil_div(e, tb)), dim3(tb), 0, st, edge_index, n, e,
                         rowptr, cursor, eid);
    const int fb = degree_block_hist != nullptr ? gs::kDegBlock : tb;
    hipLaunchKernelGGL(gs::k_finish_rows, dim3((unsigned)gs_ceil_div(n, fb)), dim3(fb), 0, st, edge_index, edge_attr,
                       n, e, bd, self_loops ? 1 : 0, rowptr, counts, eid, src, dst, combo, log_amp, log_att,
                       err_flag, degree_block_hist);
  }
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

extern "C" int gnnsaft_csr_build(const int64_t *edge_index, const int64_t *edge_attr, int64_t num_nodes,
                                 int64_t num_edges, int32_t num_bond_cols, const int32_t *bond_dims_host,
                                 int32_t self_loops, int32_t *rowptr, int32_t *src, int32_t *dst, int32_t *combo,
                                 float *log_amp, float *log_att, int32_t *err_flag, void *workspace,
                                 size_t workspace_bytes, gnnsaft_stream_t stream) {
  return gs::launch_csr_build(edge_index, edge_attr, num_nodes, num_edges, num_bond_cols, bond_dims_host, self_loops,
                              rowptr, src, dst, combo, log_amp, log_att, err_flag, workspace, workspace_bytes, nullptr,
                              0, nullptr, nullptr, false, static_cast<hipStream_t>(stream));
}

extern "C" int gnnsaft_batch_to_ptr(const int64_t *batch, int64_t num_nodes, int64_t num_graphs, int32_t *graph_ptr,
                                    int32_t *err_flag, gnnsaft_stream_t stream) {
  GS_REQUIRE(graph_ptr != nullptr, GNNSAFT_ERR_NULL);
  GS_REQUIRE(num_graphs >= 0 && num_nodes >= 0 && num_nodes < ((int64_t)1 << 31), GNNSAFT_ERR_SHAPE);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (batch == nullptr) {  // un-batched Data: a single graph spanning every node
    GS_REQUIRE(num_graphs == 1, GNNSAFT_ERR_SHAPE);
    hipLaunchKernelGGL(gs::k_single_graph_ptr, dim3(1), dim3(1), 0, st, graph_ptr, (int32_t)num_nodes);
    GS_CHECK_LAUNCH();
    return GNNSAFT_OK;
  }
  hipLaunchKernelGGL(gs::k_batch_to_ptr, dim3((unsigned)gs_ceil_div(num_nodes + 1, 256)), dim3(256), 0, st, batch,
                     num_nodes, num_graphs, graph_ptr, err_flag);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}
