// Device bodies of the K0 structure chain (destination-sorted CSR with slotted fill, degree plan) shared by the
// stand-alone kernels of csr.hip / degree.hip and by the cooperative chain that runs INSIDE the forward's prologue
// launch (elementwise.hip: k0_chain_body).  Reference semantics: add_self_loops + the index plumbing PyG derives per
// layer from the unsorted edge_index (/root/reference/gnnepcsaft/train/models.py:118-121).
#pragma once
#include "common.hpp"

namespace gs {

constexpr int kScanBlock = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanBlock * kScanItems;  // 2048 counts per workgroup
constexpr int kCsrSlots = kDegreeBuckets;           // slotted fill: edge ids per destination node

// thread i in [0,n) closes the gap between batch[i-1] and batch[i]; thread n closes the tail
__device__ __forceinline__ void batch_to_ptr_slot(const int64_t *__restrict__ batch, int64_t n, int64_t g,
                                                  int32_t *__restrict__ ptr, int32_t *err, int64_t i) {
  if (i > n) return;
  int64_t prev = i == 0 ? -1 : batch[i - 1];
  int64_t cur = i == n ? g : batch[i];
  if (i < n && (cur < 0 || cur >= g || cur < prev)) {
    if (err) atomicOr(err, GNNSAFT_FLAG_BAD_BATCH);
    return;
  }
  if (prev < -1) prev = -1;
  if (prev >= g) return;
  for (int64_t q = prev + 1; q <= cur && q <= g; ++q) ptr[q] = (int32_t)i;
}

// block-level exclusive scan of one value per thread (kScanBlock threads); `lds`: kScanBlock / 64 ints
__device__ __forceinline__ int block_exclusive_scan(int v, int *lds, int &total) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  int wave_off = 0;
  int tot = 0;
#pragma unroll
  for (int w = 0; w < kScanBlock / 64; ++w) {
    const int t = lds[w];
    if (w < wave) wave_off += t;
    tot += t;
  }
  __syncthreads();
  total = tot;
  return wave_off + inc - v;
}

// tile-local exclusive scan of min(counts[i], clamp) + extra over tile `tile` (kScanTile entries, kScanBlock threads):
// rowptr[i] = offset inside the tile, tile_sums[tile] = the tile's total
__device__ __forceinline__ void scan_tile_body(const int32_t *__restrict__ counts, int64_t n, int extra,
                                               int32_t *__restrict__ rowptr, int32_t *__restrict__ tile_sums,
                                               int clamp, int64_t tile, int *lds) {
  const int64_t base = tile * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int v[kScanItems];
  int local = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    const int64_t i = base + j;
    v[j] = i < n ? (counts[i] < clamp ? counts[i] : clamp) + extra : 0;
    local += v[j];
  }
  int total;
  int off = block_exclusive_scan(local, lds, total);
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    const int64_t i = base + j;
    if (i < n) rowptr[i] = off;  // tile-local; the tile offset is added by whoever completes the rows
    off += v[j];
  }
  if (threadIdx.x == 0) tile_sums[tile] = total;
}

// edge i -> the next free slot of its destination's row (cursor: zero before the first edge); an in-degree beyond the
// slots is flagged, the surplus edge dropped
__device__ __forceinline__ void fill_slot_body(const int64_t *__restrict__ edge_index, int64_t n, int64_t e,
                                               int32_t *__restrict__ cursor, int32_t *__restrict__ slots,
                                               int32_t *err, int64_t i) {
  if (i >= e) return;
  const int64_t s = edge_index[i];
  const int64_t d = edge_index[e + i];
  if (s < 0 || s >= n || d < 0 || d >= n) {
    if (err) atomicOr(err, GNNSAFT_FLAG_BAD_EDGE);
    return;  // dropped
  }
  const int pos = atomicAdd(&cursor[d], 1);
  if ((unsigned)pos < (unsigned)kCsrSlots) {
    slots[d * kCsrSlots + pos] = (int32_t)i;
  } else if (err) {
    atomicOr(err, GNNSAFT_FLAG_BAD_DEGREE);  // (the edge is dropped; the row keeps kCsrSlots edges)
  }
}

__device__ __forceinline__ void sort_exchange(int &a, int &b) {
  const int lo = a < b ? a : b, hi = a < b ? b : a;
  a = lo;
  b = hi;
}

// the (<= 8) edge ids of a slot row into registers, unused entries = INT_MAX (cnt > 8: sorted in memory by emit_node_rows)
__device__ __forceinline__ void load_slot_keys(const int32_t *__restrict__ row, int cnt, int (&key)[8]) {
  if (cnt <= 8) {
    const int4 lo = cnt > 0 ? *reinterpret_cast<const int4 *>(row) : int4{0, 0, 0, 0};
    const int4 hi = cnt > 4 ? *reinterpret_cast<const int4 *>(row + 4) : int4{0, 0, 0, 0};
    const int raw[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int a = 0; a < 8; ++a) key[a] = a < cnt ? raw[a] : 0x7fffffff;
  }
}

// node i with `cnt` in-edges (ids in key[] / in its slot row) whose rows start at `beg`: ids ascending (8 or fewer: a
// sorting network in registers), src / dst / combo rows, the loop row, degree scaler logs.  Returns the in-degree
// with the loop.
__device__ __forceinline__ int emit_node_rows(const int64_t *__restrict__ edge_index,
                                              const int64_t *__restrict__ edge_attr, const BondDims &bd, int self_loops,
                                              int64_t i, int cnt, int beg, int (&key)[8], int32_t *__restrict__ row,
                                              int32_t *__restrict__ src, int32_t *__restrict__ dst,
                                              int32_t *__restrict__ combo, float *__restrict__ log_amp,
                                              float *__restrict__ log_att, int32_t *err) {
  if (cnt <= 8) {
    // Batcher odd-even merge sort, 8 keys, 19 exchanges
    sort_exchange(key[0], key[1]); sort_exchange(key[2], key[3]); sort_exchange(key[4], key[5]); sort_exchange(key[6], key[7]);
    sort_exchange(key[0], key[2]); sort_exchange(key[1], key[3]); sort_exchange(key[4], key[6]); sort_exchange(key[5], key[7]);
    sort_exchange(key[1], key[2]); sort_exchange(key[5], key[6]);
    sort_exchange(key[0], key[4]); sort_exchange(key[1], key[5]); sort_exchange(key[2], key[6]); sort_exchange(key[3], key[7]);
    sort_exchange(key[2], key[4]); sort_exchange(key[3], key[5]);
    sort_exchange(key[1], key[2]); sort_exchange(key[3], key[4]); sort_exchange(key[5], key[6]);
    int64_t sv[8];
    int cid[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) {   // every row's loads are independent: all in flight together
      const int64_t id = a < cnt ? key[a] : 0;
      sv[a] = a < cnt ? edge_index[id] : 0;
      int c = 0;
      for (int k = 0; k < bd.n; ++k) {
        int64_t v = a < cnt ? edge_attr[id * bd.n + k] : 0;
        if (v < 0 || v >= bd.dims[k]) {
          if (err) atomicOr(err, GNNSAFT_FLAG_BAD_ATTR);
          v = 0;
        }
        c = c * bd.dims[k] + (int)v;
      }
      cid[a] = c;
    }
#pragma unroll
    for (int a = 0; a < 8; ++a)
      if (a < cnt) {
        src[beg + a] = (int32_t)sv[a];
        dst[beg + a] = (int32_t)i;
        combo[beg + a] = cid[a];
      }
  } else {
    for (int a = 1; a < cnt; ++a) {   // insertion sort in the slot row, ascending edge id
      const int k2 = row[a];
      int b = a - 1;
      while (b >= 0 && row[b] > k2) {
        row[b + 1] = row[b];
        --b;
      }
      row[b + 1] = k2;
    }
    for (int a = 0; a < cnt; ++a) {
      const int64_t id = row[a];
      int c = 0;
      for (int k = 0; k < bd.n; ++k) {
        int64_t v = edge_attr[id * bd.n + k];
        if (v < 0 || v >= bd.dims[k]) {
          if (err) atomicOr(err, GNNSAFT_FLAG_BAD_ATTR);
          v = 0;
        }
        c = c * bd.dims[k] + (int)v;
      }
      src[beg + a] = (int32_t)edge_index[id];
      dst[beg + a] = (int32_t)i;
      combo[beg + a] = c;
    }
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

// One thread per node of the group of `bsize` nodes number `group` (bsize divides kScanTile; every thread of the
// workgroup calls): final row offset (tile-local scan value + the totals of the tiles in front), then the node's rows.
// Returns the node's in-degree (with the loop), 0 where there is no node.  `last`: this group also writes rowptr[n].
__device__ __forceinline__ int finish_rows_slots_body(
    const int64_t *__restrict__ edge_index, const int64_t *__restrict__ edge_attr, int64_t n, const BondDims &bd,
    int self_loops, int32_t *__restrict__ rowptr, const int32_t *__restrict__ tile_sums, int64_t num_tiles,
    const int32_t *__restrict__ cursor, int32_t *__restrict__ slots, int32_t *__restrict__ src,
    int32_t *__restrict__ dst, int32_t *__restrict__ combo, float *__restrict__ log_amp, float *__restrict__ log_att,
    int32_t *err, int64_t group, int bsize, bool last, int *s_before, int *s_all /* 16 ints each */) {
  const int64_t i = group * bsize + threadIdx.x;
  const bool live = i < n;
  // loads that do not depend on the row offset first
  int cnt = live ? cursor[i] : 0;
  cnt = cnt < 0 ? 0 : (cnt < kCsrSlots ? cnt : kCsrSlots);
  int32_t *row = slots + (live ? i : 0) * kCsrSlots;
  int key[8];
  load_slot_keys(row, cnt, key);
  // tile totals in front of this group's tile (the last group also sums all of them: rowptr[n])
  const int64_t tile = (group * bsize) / kScanTile;
  const int64_t upto = last ? num_tiles : tile;
  int before = 0, all = 0;
  for (int64_t t = threadIdx.x; t < upto; t += bsize) {
    const int v = tile_sums[t];
    all += v;
    if (t < tile) before += v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    before += __shfl_xor(before, o);
    all += __shfl_xor(all, o);
  }
  __syncthreads();   // (s_before / s_all may still be read from the previous group of this workgroup)
  if ((threadIdx.x & 63) == 0) {
    s_before[threadIdx.x >> 6] = before;
    s_all[threadIdx.x >> 6] = all;
  }
  __syncthreads();
  before = 0;
  all = 0;
  for (int w = 0; w < (bsize >> 6); ++w) {
    before += s_before[w];
    all += s_all[w];
  }
  if (last && threadIdx.x == 0) rowptr[n] = all;
  int deg = 0;
  if (live) {
    const int beg = rowptr[i] + before;
    rowptr[i] = beg;
    deg = emit_node_rows(edge_index, edge_attr, bd, self_loops, i, cnt, beg, key, row, src, dst, combo, log_amp, log_att,
                         err);
  }
  return deg;
}

// One 256-thread workgroup: per-bucket exclusive scan over the blocks of block_hist (in place), bucket totals, bucket
// starts, tile table.  8 lanes x 32 buckets: every lane scans a contiguous run of blocks, chained through LDS.
struct DegreePlanLds {
  int32_t lane[256 / kDegreeBuckets][kDegreeBuckets];
  int32_t cnt[kDegreeBuckets], start[kDegreeBuckets], tile0[kDegreeBuckets + 1];
};
__device__ __forceinline__ void degree_plan_body(int32_t *__restrict__ block_hist, int64_t num_blocks, int tile_rows,
                                                 int32_t *__restrict__ hist, int32_t *__restrict__ start,
                                                 int32_t *__restrict__ tiles, int32_t *__restrict__ num_tiles,
                                                 DegreePlanLds &s) {
  constexpr int kLanes = 256 / kDegreeBuckets;
  const int bucket = threadIdx.x % kDegreeBuckets, lane = threadIdx.x / kDegreeBuckets;
  const int64_t per_lane = (num_blocks + kLanes - 1) / kLanes;
  const int64_t b_beg = lane * per_lane;
  int64_t b_end = b_beg + per_lane;
  if (b_end > num_blocks) b_end = num_blocks;
  int sum = 0;
  for (int64_t b0 = b_beg; b0 < b_end; b0 += 8) {
    int c[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u] = b0 + u < b_end ? block_hist[(b0 + u) * kDegreeBuckets + bucket] : 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) sum += c[u];
  }
  s.lane[lane][bucket] = sum;
  __syncthreads();
  int run = 0, total = 0;
  for (int l = 0; l < kLanes; ++l) {
    const int v = s.lane[l][bucket];
    if (l < lane) run += v;
    total += v;
  }
  for (int64_t b0 = b_beg; b0 < b_end; b0 += 8) {
    int c[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u] = b0 + u < b_end ? block_hist[(b0 + u) * kDegreeBuckets + bucket] : 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (b0 + u < b_end) block_hist[(b0 + u) * kDegreeBuckets + bucket] = run;
      run += c[u];
    }
  }
  if (lane == 0) {
    s.cnt[bucket] = total;
    hist[bucket] = total;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int slot = 0, nt = 0;
    for (int d = 0; d < kDegreeBuckets; ++d) {
      s.start[d] = slot;
      s.tile0[d] = nt;
      slot += s.cnt[d];
      nt += (s.cnt[d] + tile_rows - 1) / tile_rows;
    }
    s.tile0[kDegreeBuckets] = nt;
    num_tiles[0] = nt;
  }
  __syncthreads();
  if (threadIdx.x < kDegreeBuckets) start[threadIdx.x] = s.start[threadIdx.x];
  const int nt = s.tile0[kDegreeBuckets];
  for (int t = threadIdx.x; t < nt; t += blockDim.x) {
    int d = 0;
    while (t >= s.tile0[d + 1]) ++d;
    const int o = (t - s.tile0[d]) * tile_rows;
    tiles[4 * t + 0] = d;
    tiles[4 * t + 1] = s.start[d] + o;
    tiles[4 * t + 2] = s.cnt[d] - o < tile_rows ? s.cnt[d] - o : tile_rows;
    tiles[4 * t + 3] = 0;
  }
}

}  // namespace gs
