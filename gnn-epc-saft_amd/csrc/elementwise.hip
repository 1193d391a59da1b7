// HBM-bound row kernels of the PNAPCSAFT forward: categorical embedding sums
// (ogb AtomEncoder / BondEncoder, /root/reference/gnnepcsaft/train/models.py:65-66,
// 122-123), BatchNorm finalize / apply (models.py:82,128-131), global_add_pool
// (models.py:133) and the MAPE loss (models.py:194).  All float4 per lane.
#include "common.hpp"
#include "fold.hpp"
#include "k0_chain.hpp"

namespace gs {

struct TableSet {
  int32_t n;
  int32_t dims[GNNSAFT_MAX_TABLES];
  const float *tab[GNNSAFT_MAX_TABLES];
};

// out[i, :] = sum_k tab_k[idx[i,k], :]   (left-to-right, as ogb's encoder loop)
template <int MAXT>
__device__ __forceinline__ void embed_sum_body(int64_t slot, const int64_t *__restrict__ idx, int64_t rows,
                                               const TableSet &ts, int h, float *__restrict__ out, int32_t *err,
                                               const RowSplit &rs) {
  int64_t i;
  int lane_in_row;
  gs_split(rs, slot, i, lane_in_row);
  if (i >= rows) return;
  const int c = lane_in_row * 4;
  // all index loads first, then all table-row loads (no branch between them: a guarded load would be
  // waited for on the spot), then the left-to-right sum
  int64_t v[MAXT];
#pragma unroll
  for (int k = 0; k < MAXT; ++k) v[k] = idx[i * ts.n + (k < ts.n ? k : 0)];
  bool bad = false;
  f32x4 row[MAXT];
#pragma unroll
  for (int k = 0; k < MAXT; ++k) {
    const int kk = k < ts.n ? k : 0;
    const bool oob = v[k] < 0 || v[k] >= ts.dims[kk];
    bad |= oob && k < ts.n;
    row[k] = gs_ld4(ts.tab[kk] + (oob ? 0 : v[k]) * h + c);
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < MAXT; ++k)
    if (k < ts.n) acc += row[k];
  if (bad && err) atomicOr(err, GNNSAFT_FLAG_BAD_ATTR);
  gs_st4(out + i * h + c, acc);
}

template <int MAXT>
__global__ __launch_bounds__(256) void k_embed_sum(const int64_t *__restrict__ idx, int64_t rows, TableSet ts, int h,
                                                   float *__restrict__ out, int32_t *err, RowSplit rs) {
  embed_sum_body<MAXT>((int64_t)blockIdx.x * blockDim.x + threadIdx.x, idx, rows, ts, h, out, err, rs);
}

// out[c, :] = sum_k tab_k[digit_k(c), :] for every attribute combination c
__device__ __forceinline__ void combo_embed_body(int64_t slot, const TableSet &ts, int64_t combos, int h,
                                                 float *__restrict__ out) {
  const int per_row = h / 4;
  const int64_t cid = slot / per_row;
  if (cid >= combos) return;
  const int c = (int)(slot - cid * per_row) * 4;
  int digit[GNNSAFT_MAX_TABLES];
  int64_t rem = cid;
  for (int k = ts.n - 1; k >= 0; --k) {
    digit[k] = (int)(rem % ts.dims[k]);
    rem /= ts.dims[k];
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < ts.n; ++k) acc += gs_ld4(ts.tab[k] + (int64_t)digit[k] * h + c);
  gs_st4(out + cid * h + c, acc);
}

__global__ __launch_bounds__(256) void k_combo_embed(TableSet ts, int64_t combos, int h, float *__restrict__ out) {
  combo_embed_body((int64_t)blockIdx.x * blockDim.x + threadIdx.x, ts, combos, h, out);
}

// The four independent first kernels of a forward in ONE launch (workgroup ranges dispatch to the job bodies):
// atom embedding sum | bond-class embedding table | zeroing of the CSR histogram / cursor | destination-term
// weight fold.  None reads what another writes, and each is otherwise a 3-10 us launch at the head of a
// dependent chain.
struct PrologueArgs {
  // embedding sum
  const int64_t *x_idx;
  int64_t rows;
  TableSet atoms;
  float *x_out;
  int32_t *err;
  RowSplit rs;
  // bond-class table
  TableSet bonds;
  int64_t combos;
  float *cemb;
  // zero fill (CSR histogram / cursors) + a few counters of later kernels (fused readout barriers)
  int32_t *zero_ptr;
  int64_t zero_count;
  int32_t *zero2_ptr;
  int zero2_count;
  // destination-term fold (dst_blocks == 0: off)
  FoldLayers fl;
  double *g_all;
  int dst_gx, dst_gy;
  int h;
  unsigned dst_blocks;             // workgroups [0, dst_blocks): destination fold
  unsigned end_embed, end_combo;   // then embedding sum [0, end_embed), class table [.., end_combo), zero fill
  // edge-class tables of every layer (tab_layers == 0: off): workgroup (layer, class)
  EdgeTableLayers et;
  float *cenc, *rtab;              // [L][C][H], [L][C][2H]
  unsigned tab_blocks;             // tab_layers * combos workgroups at the very front of the grid
  int tab_layers;
  // the batch structure (K0 chain) by the FIRST k0.wgs workgroups of the grid, grid barriers among them (0 = off)
  K0ChainArgs k0;
};

// Edge-class tables of one (layer, class): cenc[c] = W_e emb_c + b_e, rtab[c, tF + f] = W_pre,t[f, 2F:3F] cenc[c] + b_pre,t[f]
// (models.py:65-66 + the edge part of PNAConv's pre_nns; the class embedding emb_c = sum of its bond-table rows is
// formed here, so the job depends on weights only).  60 classes x L layers: two [60, H] x [H, H | 2H] GEMM launches at
// the head of every forward (6.5 us each at C2) become workgroups of the prologue launch.  One thread per output,
// 16 weight loads in flight, operand vector in LDS.
__device__ __forceinline__ void edge_table_body(const PrologueArgs &a, unsigned wg) {
  __shared__ __attribute__((aligned(16))) double s_in[256], s_cenc[256];
  const int h = a.h;
  const int layer = (int)(wg / (unsigned)a.combos);
  const int64_t cid = wg - (unsigned)layer * (unsigned)a.combos;
  const int tid = threadIdx.x;
  if (tid < h) {
    int64_t rem = cid;
    int digit[GNNSAFT_MAX_TABLES];
    for (int k = a.bonds.n - 1; k >= 0; --k) {
      digit[k] = (int)(rem % a.bonds.dims[k]);
      rem /= a.bonds.dims[k];
    }
    float acc = 0.f;
    for (int k = 0; k < a.bonds.n; ++k) acc += a.bonds.tab[k][(int64_t)digit[k] * h + tid];   // left to right, as ogb
    s_in[tid] = (double)acc;     // the class embedding IS a float32 tensor of the reference (BondEncoder output)
  }
  __syncthreads();
  // wrow: h contiguous floats (16-B aligned).  The tables are functions of the weights alone, 60 rows per layer: they
  // are accumulated in float64 and rounded to float32 once, so that their value does not depend on a summation order
  // (in float32 the 4-row BatchNorm fixtures of tests/golden moved at the 1e-5 level with the order chosen here) and
  // every node's message carries a correctly rounded edge term.
  auto dot = [&](const float *__restrict__ wrow, const double *vec) {
    double s = 0.0;
    for (int i = 0; i < h; i += 16) {
      f32x4 w[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) w[u] = gs_ld4(wrow + i + 4 * u);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double *v = vec + i + 4 * u;
        s = __builtin_fma((double)w[u].x, v[0], s);
        s = __builtin_fma((double)w[u].y, v[1], s);
        s = __builtin_fma((double)w[u].z, v[2], s);
        s = __builtin_fma((double)w[u].w, v[3], s);
      }
    }
    return s;
  };
  if (tid < h) {
    const double v = dot(a.et.we[layer] + (int64_t)tid * h, s_in) + (double)a.et.be[layer][tid];
    s_cenc[tid] = v;             // kept in float64 for the second product: one rounding on the way to rtab
    a.cenc[((int64_t)layer * a.combos + cid) * h + tid] = (float)v;
  }
  __syncthreads();
  for (int o = tid; o < 2 * h; o += 256) {
    const int t = o >= h ? 1 : 0, f = o - t * h;
    const float *w = (t == 0 ? a.et.wpre0[layer] : a.et.wpre1[layer]) + (int64_t)f * (3 * h) + 2 * h;
    const float *b = t == 0 ? a.et.bpre0[layer] : a.et.bpre1[layer];
    a.rtab[((int64_t)layer * a.combos + cid) * (2 * h) + o] = (float)(dot(w, s_cenc) + (double)b[f]);
  }
}

// ---- the K0 chain as `wgs` cooperating workgroups at the FRONT of the prologue grid (dispatched first, hence
// resident together from the start; the embedding / table / fold workgroups behind them do not depend on them):
//   slotted fill (+ graph offsets) | ONE grid barrier | per group of 256 nodes: row offsets by a decoupled look-back
//   scan, rows, degree histogram of the group | ticket: the last workgroup out makes the degree plan
// A release at agent scope writes back an L2 that the embedding workgroups keep full of dirty lines (~7 us per grid
// barrier here against 1-1.5 us in the readout), hence ONE barrier: the fill cursors live in caller memory that is zero
// between calls (gnnsaft_model_desc.persistent_sync_words; every cursor is reset by the thread that reads it), the
// look-back words are relaxed agent-scope atomics (no fence needed: they carry their own data), the histogram is
// per group and not atomic.  A lost barrier / a look-back that never resolves raises GNNSAFT_FLAG_BARRIER_TIMEOUT (sticky,
// for the host) AND sets the chain's "lost" word for THIS call: every workgroup that sees it skips the rest of the chain,
// the degree-fill launch that follows installs an empty structure and restores the persistent words to zero itself
// (degree.hip), and the end of the forward writes NaN to the outputs in every mode -- a caller that never reads the
// flag word gets NaN for that call and a correct forward on the next one.
__device__ __forceinline__ bool k0_lost(const int32_t *lost) {
  return __hip_atomic_load(lost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}
__device__ __forceinline__ void k0_give_up(int32_t *lost, int32_t *err) {
  if (err) atomicOr(err, GNNSAFT_FLAG_BARRIER_TIMEOUT);
  __hip_atomic_store(lost, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool k0_barrier(int32_t *counter, int expected, int32_t *err, int32_t *lost) {
  __shared__ int s_k0_ok;
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    int ok = 1;
    // poll relaxed (an acquire load invalidates the caches on EVERY poll), one acquire fence at the end
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expected) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1L << 22)) {
        k0_give_up(lost, err);
        ok = 0;
        break;
      }
      if ((spins & 1023) == 0 && k0_lost(lost)) {   // somebody else gave up: do not wait out the whole bound too
        ok = 0;
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    s_k0_ok = ok && !k0_lost(lost);
  }
  __syncthreads();
  return s_k0_ok != 0;
}

constexpr unsigned long long kLbAggregate = 1ull << 32, kLbInclusive = 2ull << 32;

// exclusive prefix of group gi (> 0) from the look-back words of its predecessors; called by wave 0, all 64 lanes
__device__ __forceinline__ long long k0_look_back(unsigned long long *lb, int64_t gi, int32_t *err, int32_t *lost) {
  const int lane = threadIdx.x & 63;
  long long prefix = 0;
  int64_t j = gi - 1;
  for (;;) {
    const int64_t jj = j - lane;   // lane 0: the nearest predecessor
    unsigned long long v = kLbInclusive;   // in front of group 0: an inclusive prefix of 0
    if (jj >= 0) {
      long spins = 0;
      for (;;) {
        v = __hip_atomic_load(lb + jj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((v >> 32) != 0ull) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1L << 22)) k0_give_up(lost, err);
        if ((spins & 1023) == 0 && k0_lost(lost)) {   // (a predecessor that skipped the chain never publishes)
          v = kLbInclusive;
          break;
        }
      }
    }
    const unsigned long long incl = __ballot((v >> 32) == 2ull);
    const int first = incl != 0ull ? __ffsll((long long)incl) - 1 : 64;
    long long c = lane <= first ? (long long)(v & 0xffffffffull) : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    prefix += c;
    if (first < 64) return prefix;
    j -= 64;
  }
}

__device__ __forceinline__ void k0_chain_body(const K0ChainArgs &k, int w) {
  __shared__ int s_scan[kScanBlock / 64], s_hist[kDegreeBuckets], s_last;
  __shared__ long long s_prefix;
  __shared__ DegreePlanLds s_plan;
  const int W = k.wgs, tid = threadIdx.x;
  const int64_t n = k.n, e = k.e;
  const int64_t stride = (int64_t)W * kK0Group, gtid = (int64_t)w * kK0Group + tid;
  const int64_t groups = (n + kK0Group - 1) / kK0Group;
  // phase 1: edge ids into the slot rows of their destinations (cursors: zero at entry); look-back words to "empty"
  for (int64_t i = gtid; i < e; i += stride) fill_slot_body(k.edge_index, n, e, k.cursor, k.slots, k.err, i);
  for (int64_t i = gtid; i < groups; i += stride)
    __hip_atomic_store(k.lookback + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (k.graph_ptr != nullptr) {
    if (k.batch != nullptr) {
      for (int64_t i = gtid; i <= n; i += stride) batch_to_ptr_slot(k.batch, n, k.g, k.graph_ptr, k.err, i);
    } else if (gtid == 0) {  // un-batched Data: one graph spanning all nodes
      k.graph_ptr[0] = 0;
      k.graph_ptr[1] = (int32_t)n;
    }
  }
  int32_t *lost = k.sync + kK0LostWord;
  const bool met = k0_barrier(k.sync + 0, W + k.barrier_extra, k.err, lost);
  // phase 2: groups of 256 nodes, ascending per workgroup (a group only ever waits for groups in front of it).  Not
  // after a lost barrier: the slot rows and cursors are incomplete, the next launch installs an empty structure
  for (int64_t gi = w; met && gi < groups; gi += W) {
    const int64_t i = gi * kK0Group + tid;
    const bool live = i < n;
    int cnt = 0;
    if (live) {
      cnt = k.cursor[i];
      k.cursor[i] = 0;     // left zero for the next call
      cnt = cnt < 0 ? 0 : (cnt < kCsrSlots ? cnt : kCsrSlots);
    }
    int32_t *row = k.slots + (live ? i : 0) * kCsrSlots;
    int key[8];
    load_slot_keys(row, cnt, key);
    if (tid < kDegreeBuckets) s_hist[tid] = 0;
    int total;
    const int off = block_exclusive_scan(live ? cnt + k.self_loops : 0, s_scan, total);   // (syncs inside)
    if (tid < 64) {   // wave 0: publish the group's rows, resolve its exclusive prefix, publish the inclusive one
      long long prefix = 0;
      if (gi > 0) {
        if (tid == 0)
          __hip_atomic_store(k.lookback + gi, kLbAggregate | (unsigned)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        prefix = k0_look_back(k.lookback, gi, k.err, lost);
      }
      if (tid == 0) {
        __hip_atomic_store(k.lookback + gi, kLbInclusive | (unsigned long long)(unsigned)(prefix + total),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_prefix = prefix;
        if (gi == groups - 1) k.rowptr[n] = (int32_t)(prefix + total);
      }
    }
    __syncthreads();
    int d = 0;
    if (live) {
      const int beg = (int)s_prefix + off;
      k.rowptr[i] = beg;
      const int deg = emit_node_rows(k.edge_index, k.edge_attr, k.bd, k.self_loops, i, cnt, beg, key, row, k.src, k.dst,
                                     k.combo, k.log_amp, k.log_att, k.err);
      d = clamp_degree(deg, k.err);
    }
    unsigned long long todo = __ballot(live);
    while (todo != 0ull) {   // the group's degree histogram: one LDS add per (wave, degree)
      const int leader = __ffsll((long long)todo) - 1;
      const int dl = __shfl(d, leader);
      const unsigned long long same = __ballot(live && d == dl);
      if ((tid & 63) == leader) atomicAdd(&s_hist[dl], __popcll(same));
      todo &= ~same;
    }
    __syncthreads();
    if (tid < kDegreeBuckets) k.group_hist[gi * kDegreeBuckets + tid] = s_hist[tid];
    __syncthreads();   // (s_hist, s_prefix are reused by the next group)
  }
  // ticket: the last workgroup out plans (bucket starts, tile table; the permutation fill is the next launch) and
  // leaves the persistent words zero
  __syncthreads();
  if (tid == 0) {
    const int before = __hip_atomic_fetch_add(k.sync + 1, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    s_last = before == W - 1;
  }
  __syncthreads();
  if (s_last != 0) {   // (workgroup-uniform)
    if (!k0_lost(lost)) degree_plan_body(k.group_hist, groups, k.tile_rows, k.hist, k.start, k.tiles, k.num_tiles, s_plan);
    if (tid == 0) {   // (the lost word stays for the launches behind this one; the pooling launch zeroes it)
      __hip_atomic_store(k.sync + 0, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(k.sync + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int MAXT>
__global__ __launch_bounds__(256) void k_forward_prologue(PrologueArgs a) {
  // the destination fold goes FIRST in the grid: its workgroups are the long-latency ones (a k-loop with
  // barriers), so they should be resident from the start while the streaming jobs fill the remaining CUs
  // ... and in front of them the edge-table workgroups (two dependent rounds of L2 latency each)
  if (blockIdx.x < (unsigned)a.k0.wgs) {
    k0_chain_body(a.k0, (int)blockIdx.x);
    return;
  }
  const unsigned bx = blockIdx.x - (unsigned)a.k0.wgs;
  if (bx < a.tab_blocks) {
    edge_table_body(a, bx);
    return;
  }
  if (bx < a.tab_blocks + a.dst_blocks) {
    const unsigned d = bx - a.tab_blocks;
    const int per_z = a.dst_gx * a.dst_gy;
    const int bz = d / per_z, r = d - bz * per_z;
    dst_fold_body(a.fl, a.h, a.g_all, r % a.dst_gx, r / a.dst_gx, bz);
    return;
  }
  const unsigned b = bx - a.tab_blocks - a.dst_blocks;
  if (b < a.end_embed) {
    embed_sum_body<MAXT>((int64_t)b * 256 + threadIdx.x, a.x_idx, a.rows, a.atoms, a.h, a.x_out, a.err, a.rs);
  } else if (b < a.end_combo) {
    combo_embed_body((int64_t)(b - a.end_embed) * 256 + threadIdx.x, a.bonds, a.combos, a.h, a.cemb);
  } else {
    const int64_t i = (int64_t)(b - a.end_combo) * 256 + threadIdx.x;
    if (i < a.zero_count) a.zero_ptr[i] = 0;
    if (b == a.end_combo)
      for (int t = threadIdx.x; t < a.zero2_count; t += 256) a.zero2_ptr[t] = 0;
  }
}

// ---- BatchNorm: one workgroup per channel, Chan combine of (mean, M2) partials in f64
__global__ __launch_bounds__(256) void k_bn_finalize(const float *__restrict__ stats, int64_t rows, int ch,
                                                     const float *__restrict__ gamma, const float *__restrict__ beta,
                                                     float *__restrict__ running_mean,
                                                     float *__restrict__ running_var, int64_t *nbt, float momentum,
                                                     float eps, int training, float *__restrict__ scale,
                                                     float *__restrict__ shift) {
  const int c = blockIdx.x;
  __shared__ double s_n[256], s_mean[256], s_m2[256];
  float mean_f, var_f;
  if (training) {
    const int64_t groups = (rows + kBnRowsPerGroup - 1) / kBnRowsPerGroup;
    double n = 0.0, mean = 0.0, m2 = 0.0;
    for (int64_t g = threadIdx.x; g < groups; g += blockDim.x) {
      const int64_t left = rows - g * kBnRowsPerGroup;
      const double gn = (double)(left < kBnRowsPerGroup ? left : kBnRowsPerGroup);
      const double gm = (double)stats[(g * 2 + 0) * ch + c];
      const double g2 = (double)stats[(g * 2 + 1) * ch + c];
      const double tot = n + gn;
      const double delta = gm - mean;
      mean += delta * (gn / tot);
      m2 += g2 + delta * delta * (n * gn / tot);
      n = tot;
    }
    s_n[threadIdx.x] = n;
    s_mean[threadIdx.x] = mean;
    s_m2[threadIdx.x] = m2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) {
        const double na = s_n[threadIdx.x], nb = s_n[threadIdx.x + o];
        const double tot = na + nb;
        if (tot > 0.0) {
          const double delta = s_mean[threadIdx.x + o] - s_mean[threadIdx.x];
          s_mean[threadIdx.x] += delta * (nb / tot);
          s_m2[threadIdx.x] += s_m2[threadIdx.x + o] + delta * delta * (na * nb / tot);
          s_n[threadIdx.x] = tot;
        }
      }
      __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const double tot = s_n[0];
    mean_f = (float)s_mean[0];
    var_f = (float)(s_m2[0] / tot);  // biased, used for normalisation
    if (running_mean != nullptr) {
      const float unbiased = (float)(tot > 1.0 ? s_m2[0] / (tot - 1.0) : s_m2[0]);
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean_f;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
    if (nbt != nullptr && c == 0) nbt[0] += 1;
  } else {
    if (threadIdx.x != 0) return;
    mean_f = running_mean[c];
    var_f = running_var[c];
  }
  const float rstd = 1.f / sqrtf(var_f + eps);
  const float sc = rstd * (gamma != nullptr ? gamma[c] : 1.f);
  scale[c] = sc;
  shift[c] = (beta != nullptr ? beta[c] : 0.f) - mean_f * sc;
}

__global__ __launch_bounds__(256) void k_bn_relu_residual(const float *__restrict__ y, const float *__restrict__ scale,
                                                          const float *__restrict__ shift,
                                                          const float *__restrict__ residual, float *__restrict__ out,
                                                          int64_t total4, int ch) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
    const int c = (int)((i * 4) % ch);
    const f32x4 v = gs_ld4(y + i * 4);
    const f32x4 sc = gs_ld4(scale + c), sh = gs_ld4(shift + c);
    f32x4 o = v * sc + sh;
    o.x = fmaxf(o.x, 0.f);
    o.y = fmaxf(o.y, 0.f);
    o.z = fmaxf(o.z, 0.f);
    o.w = fmaxf(o.w, 0.f);
    if (residual != nullptr) o += gs_ld4(residual + i * 4);
    gs_st4(out + i * 4, o);
  }
}

// out[g, :] = sum over rows ptr[g] .. ptr[g+1]-1, sequential (graph-contiguous rows)
__global__ __launch_bounds__(256) void k_add_pool(const float *__restrict__ x, const int32_t *__restrict__ ptr,
                                                  int64_t graphs, int64_t nodes, int h, float *__restrict__ out,
                                                  RowSplit rs, int32_t *clear_word) {
  const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (slot == 0 && clear_word != nullptr) __hip_atomic_store(clear_word, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int64_t g;
  int lane_in_row;
  gs_split(rs, slot, g, lane_in_row);
  if (g >= graphs) return;
  const int c = lane_in_row * 4;
  int64_t beg = ptr[g], end = ptr[g + 1];
  beg = beg < 0 ? 0 : (beg > nodes ? nodes : beg);
  end = end < beg ? beg : (end > nodes ? nodes : end);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int64_t r = beg;
  for (; r + 4 <= end; r += 4) {
    const f32x4 a = gs_ld4(x + (r + 0) * h + c), b = gs_ld4(x + (r + 1) * h + c);
    const f32x4 d = gs_ld4(x + (r + 2) * h + c), e = gs_ld4(x + (r + 3) * h + c);
    acc += a;
    acc += b;
    acc += d;
    acc += e;
  }
  for (; r < end; ++r) acc += gs_ld4(x + r * h + c);
  gs_st4(out + g * h + c, acc);
}

// global_add_pool of x = relu(y * scale + shift) (+ xprev): the last layer's train-mode BatchNorm + ReLU + residual
// (models.py:128-131) applied while pooling (models.py:133) -- same arithmetic and summation order as
// k_bn_train_apply followed by k_add_pool, one pass over y instead of a write and a re-read of x_L.  `xout` (tape)
// receives x_L.
__global__ __launch_bounds__(256) void k_add_pool_bn(const float *__restrict__ y, const float *__restrict__ xprev,
                                                     const float *__restrict__ scale, const float *__restrict__ shift,
                                                     float *__restrict__ xout, const int32_t *__restrict__ ptr,
                                                     int64_t graphs, int64_t nodes, int h, float *__restrict__ out,
                                                     RowSplit rs, int32_t *clear_word) {
  const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (slot == 0 && clear_word != nullptr) __hip_atomic_store(clear_word, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int64_t g;
  int lane_in_row;
  gs_split(rs, slot, g, lane_in_row);
  if (g >= graphs) return;
  const int c = lane_in_row * 4;
  int64_t beg = ptr[g], end = ptr[g + 1];
  beg = beg < 0 ? 0 : (beg > nodes ? nodes : beg);
  end = end < beg ? beg : (end > nodes ? nodes : end);
  const f32x4 sc = gs_ld4(scale + c), sh = gs_ld4(shift + c);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc = zero;
  int64_t r = beg;
  for (; r + 4 <= end; r += 4) {
    f32x4 v[4], q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[u] = gs_ld4(y + (r + u) * h + c);
      q[u] = xprev != nullptr ? gs_ld4(xprev + (r + u) * h + c) : zero;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const f32x4 o = gs_relu4(v[u] * sc + sh) + q[u];
      if (xout != nullptr) gs_st4(xout + (r + u) * h + c, o);
      acc += o;
    }
  }
  for (; r < end; ++r) {
    const f32x4 o = gs_relu4(gs_ld4(y + r * h + c) * sc + sh) + (xprev != nullptr ? gs_ld4(xprev + r * h + c) : zero);
    if (xout != nullptr) gs_st4(xout + r * h + c, o);
    acc += o;
  }
  gs_st4(out + g * h + c, acc);
}

__global__ __launch_bounds__(1024) void k_mape(const float *__restrict__ pred, const float *__restrict__ target,
                                               int64_t n, float *__restrict__ out3) {
  __shared__ float part[1024 / 64];
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
    const float t = target[i];
    acc += fabsf(pred[i] - t) / fmaxf(fabsf(t), 1.17e-06f);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int w = 0; w < 1024 / 64; ++w) tot += part[w];
    out3[0] = tot / (float)n;
    out3[1] = tot;
    out3[2] = (float)n;
  }
}

static int make_tables(int32_t num_cols, const float *const *tables_host, const int32_t *dims_host, TableSet &ts) {
  GS_REQUIRE(num_cols >= 1 && num_cols <= GNNSAFT_MAX_TABLES, GNNSAFT_ERR_SHAPE);
  GS_REQUIRE(tables_host != nullptr && dims_host != nullptr, GNNSAFT_ERR_NULL);
  ts.n = num_cols;
  for (int k = 0; k < GNNSAFT_MAX_TABLES; ++k) {
    ts.dims[k] = k < num_cols ? dims_host[k] : 1;
    ts.tab[k] = k < num_cols ? tables_host[k] : nullptr;
    if (k < num_cols) GS_REQUIRE(ts.tab[k] != nullptr && ts.dims[k] >= 1, GNNSAFT_ERR_NULL);
  }
  return GNNSAFT_OK;
}

int launch_forward_prologue(const int64_t *x_idx, int64_t num_rows, int32_t num_atom_cols,
                            const float *const *atom_tables_host, const int32_t *atom_dims_host,
                            int32_t num_bond_cols, const float *const *bond_tables_host,
                            const int32_t *bond_dims_host, int32_t hidden, float *x_out, float *cemb,
                            int32_t *zero_ptr, int64_t zero_count, int32_t fold_layers,
                            const float *const *w_post0_host, const float *const *w_post1_host,
                            const float *const *w_pre0_host, const float *const *w_pre1_host, double *g_all,
                            int32_t *err_flag, hipStream_t st, int32_t *zero2_ptr, int zero2_count,
                            const EdgeTableLayers *tables, int32_t table_layers, float *cenc, float *rtab,
                            const K0ChainArgs *k0) {
  GS_REQUIRE(x_idx && x_out && cemb && num_rows >= 1, GNNSAFT_ERR_NULL);
  GS_REQUIRE(zero2_count >= 0 && zero2_count <= 65536, GNNSAFT_ERR_SHAPE);
  GS_REQUIRE(hidden >= 4 && (hidden % 4) == 0, GNNSAFT_ERR_SHAPE);
  GS_REQUIRE(fold_layers >= 0 && fold_layers <= GNNSAFT_MAX_FOLD_LAYERS, GNNSAFT_ERR_SHAPE);
  PrologueArgs a;
  int rc = make_tables(num_atom_cols, atom_tables_host, atom_dims_host, a.atoms);
  if (rc != GNNSAFT_OK) return rc;
  rc = make_tables(num_bond_cols, bond_tables_host, bond_dims_host, a.bonds);
  if (rc != GNNSAFT_OK) return rc;
  a.x_idx = x_idx;
  a.rows = num_rows;
  a.x_out = x_out;
  a.err = err_flag;
  a.rs = gs_row_split(hidden / 4);
  a.combos = 1;
  for (int k = 0; k < num_bond_cols; ++k) a.combos *= a.bonds.dims[k];
  GS_REQUIRE(a.combos <= (1 << 20), GNNSAFT_ERR_UNSUPPORTED);
  a.cemb = cemb;
  a.zero_ptr = zero_ptr;
  a.zero_count = zero_ptr != nullptr ? zero_count : 0;
  a.zero2_ptr = zero2_ptr;
  a.zero2_count = zero2_ptr != nullptr ? zero2_count : 0;
  a.g_all = g_all;
  a.h = hidden;
  for (int i = 0; i < GNNSAFT_MAX_FOLD_LAYERS; ++i) {
    const int j = i < fold_layers ? i : 0;
    a.fl.w0[i] = fold_layers ? w_post0_host[j] : nullptr;
    a.fl.w1[i] = fold_layers ? w_post1_host[j] : nullptr;
    a.fl.avg[i] = nullptr;
    a.fl.pre0[i] = fold_layers ? w_pre0_host[j] : nullptr;
    a.fl.pre1[i] = fold_layers ? w_pre1_host[j] : nullptr;
  }
  GS_REQUIRE(fold_layers == 0 || ((hidden % 64) == 0 && g_all != nullptr), GNNSAFT_ERR_SHAPE);
  a.dst_gx = hidden / 32;
  a.dst_gy = hidden / 64;
  const int64_t be = gs_ceil_div(num_rows * (hidden / 4), 256), bc = gs_ceil_div(a.combos * (hidden / 4), 256);
  int64_t bz = gs_ceil_div(a.zero_count, 256);
  if (bz == 0 && a.zero2_count > 0) bz = 1;   // the counters ride on the first zero-fill workgroup
  const int64_t bd = (int64_t)a.dst_gx * a.dst_gy * 6 * fold_layers;
  GS_REQUIRE(be + bc + bz + bd < ((int64_t)1 << 31), GNNSAFT_ERR_SHAPE);
  a.dst_blocks = (unsigned)bd;
  a.end_embed = (unsigned)be;
  a.end_combo = (unsigned)(be + bc);
  a.tab_blocks = 0;
  a.tab_layers = 0;
  a.cenc = cenc;
  a.rtab = rtab;
  int64_t bt = 0;
  if (tables != nullptr && table_layers > 0) {
    GS_REQUIRE(table_layers <= GNNSAFT_MAX_FOLD_LAYERS && cenc != nullptr && rtab != nullptr && hidden <= 256 &&
                   (hidden % 16) == 0,
               GNNSAFT_ERR_SHAPE);
    a.et = *tables;
    a.tab_layers = table_layers;
    bt = (int64_t)table_layers * a.combos;
    a.tab_blocks = (unsigned)bt;
  } else {
    for (int i = 0; i < GNNSAFT_MAX_FOLD_LAYERS; ++i)
      a.et.we[i] = a.et.be[i] = a.et.wpre0[i] = a.et.wpre1[i] = a.et.bpre0[i] = a.et.bpre1[i] = nullptr;
  }
  int64_t bk = 0;
  a.k0.wgs = 0;
  if (k0 != nullptr && k0->wgs > 0) {
    GS_REQUIRE(k0->wgs <= kK0MaxWgs && k0->sync != nullptr && k0->cursor != nullptr && k0->lookback != nullptr,
               GNNSAFT_ERR_SHAPE);
    a.k0 = *k0;
    bk = k0->wgs;
  }
  GS_REQUIRE(bk + be + bc + bz + bd + bt < ((int64_t)1 << 31), GNNSAFT_ERR_SHAPE);
  const dim3 grid((unsigned)(bk + be + bc + bz + bd + bt));
  if (num_atom_cols <= 9)
    hipLaunchKernelGGL(k_forward_prologue<9>, grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(k_forward_prologue<GNNSAFT_MAX_TABLES>, grid, dim3(256), 0, st, a);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

int launch_add_pool_bn(const float *y, const float *xprev, const float *scale, const float *shift, float *xout,
                       const int32_t *graph_ptr, int64_t num_graphs, int64_t num_nodes, int hidden, float *out,
                       hipStream_t st, int32_t *clear_word) {
  GS_REQUIRE(y && scale && shift && graph_ptr && out, GNNSAFT_ERR_NULL);
  GS_REQUIRE(hidden >= 4 && (hidden % 4) == 0 && num_graphs >= 0, GNNSAFT_ERR_SHAPE);
  if (num_graphs == 0) return GNNSAFT_OK;
  const int64_t threads = num_graphs * (hidden / 4);
  hipLaunchKernelGGL(k_add_pool_bn, dim3((unsigned)gs_ceil_div(threads, 256)), dim3(256), 0, st, y, xprev, scale, shift,
                     xout, graph_ptr, num_graphs, num_nodes, hidden, out, gs_row_split(hidden / 4), clear_word);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

int launch_add_pool(const float *x, const int32_t *graph_ptr, int64_t num_graphs, int64_t num_nodes, int hidden,
                    float *out, hipStream_t st, int32_t *clear_word) {
  GS_REQUIRE(graph_ptr && out && (x != nullptr || num_nodes == 0), GNNSAFT_ERR_NULL);
  GS_REQUIRE(hidden >= 4 && (hidden % 4) == 0 && num_graphs >= 0, GNNSAFT_ERR_SHAPE);
  if (num_graphs == 0) return GNNSAFT_OK;
  const int64_t threads = num_graphs * (hidden / 4);
  hipLaunchKernelGGL(k_add_pool, dim3((unsigned)gs_ceil_div(threads, 256)), dim3(256), 0, st, x, graph_ptr, num_graphs,
                     num_nodes, hidden, out, gs_row_split(hidden / 4), clear_word);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

__global__ __launch_bounds__(256) void k_poison_if(const int32_t *__restrict__ lost, float *__restrict__ out,
                                                   int64_t count, float *__restrict__ loss3) {
  if (lost[0] == 0) return;
  const float nanv = __builtin_nanf("");
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) out[i] = nanv;
  if (loss3 != nullptr && blockIdx.x == 0 && threadIdx.x < 3) loss3[threadIdx.x] = nanv;
}

int launch_poison_if(const int32_t *lost, float *out, int64_t count, float *loss3, hipStream_t st) {
  GS_REQUIRE(lost != nullptr && out != nullptr && count >= 0, GNNSAFT_ERR_NULL);
  const int64_t blocks = gs_ceil_div(count > 0 ? count : 1, 256);
  hipLaunchKernelGGL(k_poison_if, dim3((unsigned)(blocks < 256 ? blocks : 256)), dim3(256), 0, st, lost, out, count, loss3);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

}  // namespace gs

extern "C" int gnnsaft_embed_sum(const int64_t *idx, int64_t num_rows, int32_t num_cols,
                                 const float *const *tables_host, const int32_t *dims_host, int32_t hidden,
                                 float *out, int32_t *err_flag, gnnsaft_stream_t stream) {
  GS_REQUIRE(out != nullptr && (idx != nullptr || num_rows == 0), GNNSAFT_ERR_NULL);
  GS_REQUIRE(hidden >= 4 && (hidden % 4) == 0 && num_rows >= 0, GNNSAFT_ERR_SHAPE);
  gs::TableSet ts;
  const int rc = gs::make_tables(num_cols, tables_host, dims_host, ts);
  if (rc != GNNSAFT_OK) return rc;
  if (num_rows == 0) return GNNSAFT_OK;
  const int64_t threads = num_rows * (hidden / 4);
  const dim3 grid((unsigned)gs_ceil_div(threads, 256)), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const RowSplit rs = gs_row_split(hidden / 4);
  if (num_cols <= 3)
    hipLaunchKernelGGL(gs::k_embed_sum<3>, grid, block, 0, st, idx, num_rows, ts, hidden, out, err_flag, rs);
  else if (num_cols <= 9)
    hipLaunchKernelGGL(gs::k_embed_sum<9>, grid, block, 0, st, idx, num_rows, ts, hidden, out, err_flag, rs);
  else
    hipLaunchKernelGGL(gs::k_embed_sum<GNNSAFT_MAX_TABLES>, grid, block, 0, st, idx, num_rows, ts, hidden, out,
                       err_flag, rs);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

extern "C" int gnnsaft_bond_combo_embed(int32_t num_cols, const float *const *tables_host,
                                        const int32_t *dims_host, int32_t hidden, float *out,
                                        gnnsaft_stream_t stream) {
  GS_REQUIRE(out != nullptr, GNNSAFT_ERR_NULL);
  GS_REQUIRE(hidden >= 4 && (hidden % 4) == 0, GNNSAFT_ERR_SHAPE);
  gs::TableSet ts;
  const int rc = gs::make_tables(num_cols, tables_host, dims_host, ts);
  if (rc != GNNSAFT_OK) return rc;
  int64_t combos = 1;
  for (int k = 0; k < num_cols; ++k) combos *= ts.dims[k];
  GS_REQUIRE(combos <= (1 << 20), GNNSAFT_ERR_UNSUPPORTED);
  const int64_t threads = combos * (hidden / 4);
  hipLaunchKernelGGL(gs::k_combo_embed, dim3((unsigned)gs_ceil_div(threads, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), ts, combos, hidden, out);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

extern "C" int gnnsaft_bn_finalize(const float *stats, int64_t num_rows, int32_t channels, const float *gamma,
                                   const float *beta, float *running_mean, float *running_var,
                                   int64_t *num_batches_tracked, float momentum, float eps, int32_t training,
                                   float *scale, float *shift, gnnsaft_stream_t stream) {
  GS_REQUIRE(scale != nullptr && shift != nullptr && channels >= 1, GNNSAFT_ERR_NULL);
  if (training) {
    GS_REQUIRE(stats != nullptr, GNNSAFT_ERR_NULL);
    GS_REQUIRE(num_rows >= 2, GNNSAFT_ERR_SHAPE);  // torch: "Expected more than 1 value per channel"
  } else {
    GS_REQUIRE(running_mean != nullptr && running_var != nullptr, GNNSAFT_ERR_NULL);
  }
  hipLaunchKernelGGL(gs::k_bn_finalize, dim3((unsigned)channels), dim3(256), 0, static_cast<hipStream_t>(stream),
                     stats, num_rows, channels, gamma, beta, running_mean, running_var, num_batches_tracked, momentum,
                     eps, training, scale, shift);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

extern "C" int gnnsaft_bn_relu_residual(const float *y, const float *scale, const float *shift,
                                        const float *residual, float *out, int64_t num_rows, int32_t channels,
                                        gnnsaft_stream_t stream) {
  GS_REQUIRE(y && scale && shift && out, GNNSAFT_ERR_NULL);
  GS_REQUIRE(channels >= 4 && (channels % 4) == 0 && num_rows >= 0, GNNSAFT_ERR_SHAPE);
  if (num_rows == 0) return GNNSAFT_OK;
  const int64_t total4 = num_rows * channels / 4;
  int64_t blocks = gs_ceil_div(total4, 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(gs::k_bn_relu_residual, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     y, scale, shift, residual, out, total4, channels);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

extern "C" int gnnsaft_add_pool(const float *x, const int32_t *graph_ptr, int64_t num_graphs, int64_t num_nodes,
                                int32_t hidden, float *out, gnnsaft_stream_t stream) {
  return gs::launch_add_pool(x, graph_ptr, num_graphs, num_nodes, hidden, out, static_cast<hipStream_t>(stream), nullptr);
}

extern "C" int gnnsaft_mape(const float *pred, const float *target, int64_t numel, float *out3,
                            gnnsaft_stream_t stream) {
  GS_REQUIRE(pred && target && out3, GNNSAFT_ERR_NULL);
  GS_REQUIRE(numel >= 1, GNNSAFT_ERR_SHAPE);
  hipLaunchKernelGGL(gs::k_mape, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), pred, target, numel, out3);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}
