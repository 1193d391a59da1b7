// Shared helpers for the gfx950 kernels of libgnnsaft.  CDNA4 only: wave = 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gnnsaft.h"

#define GS_WAVE 64

#define GS_HIP(expr)                          \
  do {                                        \
    const hipError_t gs_e_ = (expr);          \
    if (gs_e_ != hipSuccess) return (int)gs_e_; \
  } while (0)
#define GS_CHECK_LAUNCH()                         \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return (int)e__;       \
  } while (0)

#define GS_REQUIRE(cond, code) \
  do {                         \
    if (!(cond)) return (code); \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int64_t gs_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t gs_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// slot -> (row, lane-in-row) without 64-bit integer division (AMDGPU has no hardware divide):
// shift/mask when `per_row` is a power of two (always for H in {64,128,256}), 32-bit division otherwise.
struct RowSplit {
  int per_row;
  int shift;  // log2(per_row) or -1
};
static inline RowSplit gs_row_split(int per_row) {
  int sh = -1;
  if (per_row > 0 && (per_row & (per_row - 1)) == 0) {
    sh = 0;
    while ((1 << sh) < per_row) ++sh;
  }
  return RowSplit{per_row, sh};
}
__device__ __forceinline__ void gs_split(const RowSplit rs, int64_t slot, int64_t &row, int &lane) {
  if (rs.shift >= 0) {
    row = slot >> rs.shift;
    lane = (int)(slot & (rs.per_row - 1));
  } else if (slot < 0x7fffffffll) {
    const unsigned s = (unsigned)slot, q = s / (unsigned)rs.per_row;
    row = q;
    lane = (int)(s - q * (unsigned)rs.per_row);
  } else {
    row = slot / rs.per_row;
    lane = (int)(slot - row * rs.per_row);
  }
}

__device__ __forceinline__ f32x4 gs_ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ void gs_st4(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }
// streaming store (written once, read by a later kernel: should not evict what this kernel re-reads)
__device__ __forceinline__ void gs_st4_stream(float *p, f32x4 v) {
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p));
}

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share an L2: MI355X_MICROARCH.md, "Workgroup
// dispatch").  Bijective renumbering that gives every XCD ONE contiguous range of the logical block ids, so that
// neighbouring rows (a molecule's nodes and their neighbours) are gathered through one L2 instead of eight.  A
// speed choice only.
__device__ __forceinline__ unsigned gs_xcd_block(unsigned orig, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7u, xcd = orig & 7u;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}
// The same inside groups of 8 * span consecutive blocks: every XCD gets `span` consecutive logical blocks of each
// group, and all XCDs stay inside one stretch of memory (DRAM page locality of the output stream).  The tail that does
// not fill a group keeps its numbering.
__device__ __forceinline__ unsigned gs_xcd_block_grouped(unsigned orig, unsigned nwg, unsigned span) {
  const unsigned group = 8u * span;
  if (orig >= (nwg / group) * group) return orig;
  const unsigned o = orig % group;
  return orig - o + (o & 7u) * span + (o >> 3);
}

// span for the row kernels with 2F / 4 lanes per node (256-thread workgroups hold 512 / F nodes): ~32 nodes, i.e. one or
// two molecules, per XCD and group -- measured best at C2 and C3 (spans of 16 / 64 / 256 nodes and one range per XCD:
// DESIGN.md section 9)
__device__ __forceinline__ unsigned gs_xcd_span(int f) { return f >= 16 ? (unsigned)(f >> 4) : 1u; }

// The dynamic-LDS limit above 64 KB is a property of (kernel, DEVICE): raise it once per device this process drives
// (`devices`: one static bitmask per kernel instantiation; bit d = done on device d; devices >= 64: every launch).
// Safe from several host threads.
#include <atomic>
static inline hipError_t gs_raise_dynamic_lds(const void *kernel, size_t bytes, std::atomic<unsigned long long> &devices) {
  int dev = 0;
  hipError_t rc = hipGetDevice(&dev);
  if (rc != hipSuccess) return rc;
  const unsigned long long bit = dev >= 0 && dev < 64 ? 1ull << dev : 0ull;
  if ((devices.load(std::memory_order_acquire) & bit) != 0ull) return hipSuccess;
  rc = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (rc == hipSuccess) devices.fetch_or(bit, std::memory_order_release);
  return rc;
}

namespace gs {

// ---- internal launchers shared between the C ABI wrappers and gnnsaft_forward

struct GemmBatchEntry {
  const float *w;     // [n_out, ldw]
  const float *bias;  // [n_out] or null
  float *out;         // out + column offset already applied
  int64_t a_off;      // provider-specific offset (e.g. tower offset into agg)
  const char *w3 = nullptr;   // W3 image of the weights (w3.hpp) for the kernels of gemm_w3.hip, or null
};

constexpr int kMaxGemmBatch = 8;
constexpr int kBnRowsPerGroup = 64;  // rows covered by one wave's accumulator tile

struct LinearEpilogue {
  const float *scale = nullptr;   // eval-mode BN folded
  const float *shift = nullptr;
  int relu_out = 0;
  const float *residual = nullptr;
  int64_t ldr = 0;
  float *stats = nullptr;         // [groups, 2, n_out] (mean, M2) partials
  int residual_is_mask = 0;       // out = residual > 0 ? v : 0 (ReLU backward) instead of v + residual
  // eval-mode BatchNorm straight from its parameters: with bn_var set, `scale` / `shift` hold gamma / beta and the
  // epilogue forms scale = gamma / sqrt(var + eps), shift = beta - mean scale itself (no finalize launch)
  const float *bn_mean = nullptr, *bn_var = nullptr;
  float bn_eps = 0.f;
};

struct GemmBatch {
  GemmBatchEntry e[kMaxGemmBatch];
};

struct EpiArgs {
  const float *scale;
  const float *shift;
  int relu_out;
  const float *residual;
  int64_t ldr;
  float *stats;
  int residual_is_mask;  // 1: out = residual > 0 ? v : 0 (ReLU backward) instead of v + residual
  const float *bn_mean, *bn_var;   // eval-mode BatchNorm parameters (scale / shift then hold gamma / beta), or null
  float bn_eps;
};

// x / c for a small positive integer count c (inv = 1.0f / c): quotient estimate, exact residual, one correction --
// the correctly rounded quotient (what `/` gives, at a third of its instructions: hipcc expands an f32 division into
// v_div_scale x 2, v_rcp, four fmas, v_div_fmas, v_div_fixup; operands are far from the overflow / underflow ranges
// where the residual would lose bits).  Used by the segment means / variances of the PNA aggregation.
__device__ __forceinline__ float gs_div_count(float x, float c, float inv) {
  const float qe = x * inv;
  const float r = __builtin_fmaf(-qe, c, x);
  return __builtin_fmaf(r, inv, qe);
}

// sqrt(x), correctly rounded, for x in the normal range (callers pass variances above PyG's 1e-5 mask): the 1-ulp
// v_sqrt_f32 plus the one-ulp-down / one-ulp-up residual test of hipcc's own sqrtf() expansion -- without that
// expansion's rescaling of tiny arguments and its special-value selects (9 instructions instead of ~13; same bits).
// (A bare v_sqrt_f32 was measured first: 5 % off the fused aggregation + update, and the frozen population bar of
// tests/helpers.py failed on one train-mode case -- an ulp of std is visible through train-mode BatchNorm.)
__device__ __forceinline__ float gs_sqrt_rn(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float down = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
  const float vp = __builtin_fmaf(-down, s, x);   // x - down s
  const float vs = __builtin_fmaf(-up, s, x);     // x - up s
  float r = vp <= 0.f ? down : s;
  r = vs > 0.f ? up : r;
  return r;
}

__device__ __forceinline__ f32x4 gs_relu4(f32x4 v) {
  v.x = fmaxf(v.x, 0.f);
  v.y = fmaxf(v.y, 0.f);
  v.z = fmaxf(v.z, 0.f);
  v.w = fmaxf(v.w, 0.f);
  return v;
}

int launch_linear(const float *a, int64_t lda, int relu_in, int nbatch, const GemmBatchEntry *entries,
                  int64_t ldw, int64_t ldo, int64_t m, int n_out, int k, const LinearEpilogue &epi,
                  hipStream_t stream, int cfg = -1 /* tile configuration, -1 = measured heuristic */);
// out = [a0 | a1] W^T (+ residual): a0 [m,k0], a1 [m,k1], W [n_out, k0 + k1]; k0 a multiple of 32
int launch_linear_concat2(const float *a0, int64_t lda0, int k0, const float *a1, int64_t lda1, int k1,
                          const GemmBatchEntry &entry, int64_t ldw, int64_t ldo, int64_t m, int n_out,
                          const LinearEpilogue &epi, hipStream_t stream);

// out = relu(y * scale + shift) (+ xprev) as A operand, W^T as B; workgroups of column block 0 also write that A
// to `xout` (or null): train-mode BatchNorm + ReLU + residual applied while the next GEMM stages its operand
constexpr int kBnTailCounterInts = 64;     // one ticket per 32-column slab of k_bn_stats_close (H <= 2048)
int launch_linear_bnres(const float *y, const float *xprev, const float *scale, const float *shift, float *xout,
                        int nbatch, const GemmBatchEntry *entries, int64_t ldw, int64_t ldo, int64_t m, int n_out, int k,
                        hipStream_t stream);

int launch_pna_update(const float *x, const float *agg, const float *log_amp, const float *log_att,
                      const float *avg_deg_log, int64_t n, int hidden, const GemmBatchEntry *entries /*2*/,
                      int64_t ldo, hipStream_t stream);

int launch_pna_edge_mlp(const int32_t *src, const int32_t *dst, const int32_t *combo, int64_t rows, int hidden,
                        const float *pq, const float *rtab, const GemmBatchEntry *entries /*2*/, int64_t ldo,
                        hipStream_t stream);

int pna_fold_tile_rows(int hidden);

// ---- split-bf16 GEMMs on pre-split weight images (gemm_w3.hip, w3.hpp); cfg from w3_pick_cfg / w3_cfg_for_update
int launch_linear_w3(const float *a, int64_t lda, int nbatch, const GemmBatchEntry *entries /* .w3 set */, int n_pad,
                     int64_t ldo, int64_t m, int n_out, int k, const LinearEpilogue &epi, hipStream_t stream, int cfg);
// A operand in registers (gemm_ar.hip): cfg = ArCfg (w3.hpp)
int launch_linear_ar(const float *a, int64_t lda, int nbatch, const GemmBatchEntry *entries /* .w3 set */, int n_pad,
                     int64_t ldo, int64_t m, int n_out, int k, const LinearEpilogue &epi, hipStream_t stream, int cfg);
// the degree-folded update on k_gemm_ar, both towers per workgroup (small batches; hidden 128 / 256)
bool ar_update_supported(int hidden);
int launch_pna_update_folded_ar(const float *x, const float *agg, const int32_t *perm, const int32_t *tiles,
                                const int32_t *num_tiles, int64_t max_tiles, int64_t n, int hidden, const char *w_eff3,
                                const float *b_post0, const float *b_post1, float *u, hipStream_t stream);
// wave-specialised form (gemm_w3s.hip): cfg 0 = 128 x 128, 1 = 64 x 128
int launch_linear_w3s(const float *a, int64_t lda, int nbatch, const GemmBatchEntry *entries /* .w3 set */, int n_pad,
                      int64_t ldo, int64_t m, int n_out, int k, const LinearEpilogue &epi, hipStream_t stream, int cfg);
// fused aggregation + folded update of the no-tape forward (update_agg.hip): the aggregates never reach HBM
bool update_agg_supported(int hidden, int classes);
int launch_pna_update_agg(const float *x, const float *q, const float *rtab, int classes, const int32_t *rowptr,
                          const int32_t *src, const int32_t *combo, const int32_t *perm, const int32_t *tiles,
                          const int32_t *num_tiles, int64_t max_tiles, int64_t n, int hidden, const char *w_eff3,
                          const float *b_post0, const float *b_post1, float *u, hipStream_t stream);
int w3_cfg_for_update(int hidden, int64_t n);   // -1: the folded update stays on k_gemm_f32
int launch_pna_update_folded_w3(const float *x, const float *agg, const int32_t *perm, const int32_t *tiles,
                                const int32_t *num_tiles, int64_t max_tiles, int64_t n, int hidden, const char *w_eff3,
                                const float *b_post0, const float *b_post1, float *u, hipStream_t stream);

int launch_pna_update_folded(const float *x, const float *agg, const int32_t *perm, const int32_t *tiles,
                             const int32_t *num_tiles, int64_t max_tiles, int64_t n, int hidden, const float *w_eff,
                             const float *b_post0, const float *b_post1, float *u, hipStream_t stream);

// degree-tiled dgrad of the folded update: out[perm[slot], :] = a[perm[slot], a_off:a_off+k] x W(d)^T for the
// rows of every degree tile; entries[t].w points at the (transposed, folded) weights of degree 0, the weights
// of degree d sit w_stride floats further per degree
int launch_linear_degree_tiled(const float *a, int64_t lda, const int32_t *perm, const int32_t *tiles,
                               const int32_t *num_tiles, int64_t max_tiles, int64_t w_stride, int nbatch,
                               const GemmBatchEntry *entries, int64_t ldw, int64_t ldo, int64_t n, int n_out, int k,
                               int hidden /* the tile table was built for */, hipStream_t stream);

// Train-mode BatchNorm statistics without the apply pass (bn_train.hip: k_bn_stats_close): folds the (mean, M2)
// partials a STATS GEMM left, writes scale / shift [ch] (y * scale + shift normalises), the saved (mean, rstd) and the
// running statistics.  `seg`: [kBnMaxSegments][2][ch] f64 scratch; `tickets`: >= ceil(ch / 32) ints, zero at launch
// (left zero).  Whoever consumes y applies relu(y scale + shift) (+ residual) on load.
int launch_bn_stats_close(const float *stats, int64_t rows, int ch, const float *gamma, const float *beta,
                          float *running_mean, float *running_var, int64_t *nbt, float momentum, float eps,
                          float *scale, float *shift, float *save_stat, double *seg, int32_t *tickets, hipStream_t st);

// eval-mode BatchNorm + ReLU (+ residual) of a kept pre-activation tensor (bn_train.hip); `save_stat` [2][ch] receives
// (running_mean, rstd) for the backward
int launch_bn_eval_apply(const float *y, int64_t rows, int ch, const float *gamma, const float *beta,
                         const float *rmean, const float *rvar, float eps, const float *residual, float *out,
                         float *save_stat, hipStream_t st);

// ---- backward building blocks (gemm_tn.hip, csr.hip)
// Row blocks of a dense [rows, cols] result go to separate matrices: block b = row / rows_per_block -> base[b]
struct SlabOut {
  float *base[4];
  int64_t rows_per_block;
};
// Deferred slab reductions.  Every weight-gradient GEMM / column-sum pass leaves per-chunk partial results ("slabs")
// that a second, tiny launch sums in chunk order.  A backward layer issues five of them: with a queue they write to
// disjoint regions of one arena and ONE launch (launch_slab_queue_flush) reduces them all.
constexpr int kMaxSlabJobs = 8;
struct SlabJob {
  const float *slabs;
  int64_t per_slab, chunks, ld_out;
  SlabOut so;
  int cols, accumulate;
};
struct SlabQueue {
  float *base = nullptr;
  size_t cap = 0, off = 0;   // floats
  int count = 0;
  SlabJob jobs[kMaxSlabJobs];
  float *take(size_t floats) {
    const size_t o = off;
    const size_t next = o + ((floats + 63) / 64) * 64;
    if (next > cap) return nullptr;
    off = next;
    return base + o;
  }
  size_t left_bytes() const { return (cap - off) * 4; }
};
int launch_slab_queue_flush(SlabQueue &q, hipStream_t st);
size_t tn_slab_bytes(int64_t m, int n_out, int k);
int launch_wgrad_plain(const float *dy, int64_t ldy, const float *a, int64_t lda, int relu_a, int64_t m, int n_out,
                       int k, float *out, int64_t ld_out, int accumulate, float *slabs, size_t slab_bytes,
                       hipStream_t st, SlabQueue *defer = nullptr /* slabs from its arena, reduction queued */);
int launch_wgrad_onehot(const float *dx, int64_t ldx, const int64_t *idx, int ncol, const int32_t *dims_host, int64_t n,
                        int hidden, float *dtab_t, int total_rows_padded, float *slabs, size_t slab_bytes,
                        hipStream_t st);
// dW blocks [rows_per_block, k] of one TN GEMM dy[:, b*rows_per_block ..]^T a, each written to its own matrix
int launch_wgrad_plain_blocks(const float *dy, int64_t ldy, const float *a, int64_t lda, int64_t m, int num_blocks,
                              int rows_per_block, int k, float *const *out_blocks, int64_t ld_out, float *slabs,
                              size_t slab_bytes, hipStream_t st, SlabQueue *defer = nullptr);
// post_nns weight gradients of both towers through the degree tiles (K = 5F contraction, scalers folded per tile)
int launch_wgrad_post_folded(const float *du, const float *x, const float *agg, const int32_t *perm,
                             const int32_t *tiles, const int32_t *num_tiles, int64_t tile_cap, int tile_rows,
                             const float *avg, int hidden, float *dw0, float *dw1, float *slabs, size_t slab_bytes,
                             hipStream_t st, SlabQueue *defer = nullptr,
                             const int32_t *hist = nullptr /* nodes per in-degree: enables the wide kernel */,
                             int64_t num_nodes = 0);
size_t wgrad_post_folded_slab_bytes(int64_t tile_cap, int tile_rows, int hidden);
// out[c, :] = sum of the rows of `a` whose class id is c (one-hot TN GEMM: deterministic, no atomics)
int launch_sum_rows_by_class(const int32_t *cls, int num_classes, const float *a, int64_t lda, int64_t m, int k,
                             float *out, int64_t ld_out, float *slabs, size_t slab_bytes, hipStream_t st,
                             SlabQueue *defer = nullptr, int force_x6 = -1 /* 0: the f32 one-hot GEMM */);
constexpr int kMaxTransposeBatch = 64;   // 48 B of kernel arguments per entry
struct TransposeItem {
  const float *in;
  float *out;
  int64_t ld_in, ld_out;
  int rows, cols;  // of `in`
};
// out[c][r] = in[r][c] for up to kMaxTransposeBatch small matrices of different shapes in one launch
int launch_transpose_list(int count, const TransposeItem *items, hipStream_t st);
int launch_transpose(int count, const float *const *in, float *const *out, const int64_t *ld_in, const int64_t *ld_out,
                     int rows, int cols, hipStream_t st);
int launch_colsum(const float *a, int64_t lda, int64_t m, int cols, float *out, int accumulate, float *partial,
                  size_t partial_bytes, hipStream_t st);
int launch_colsum_blocks(const float *a, int64_t lda, int64_t m, int num_blocks, int cols_per_block, float *const *outs,
                         float *partial, size_t partial_bytes, hipStream_t st, SlabQueue *defer = nullptr);
size_t group_by_key_workspace_bytes(int64_t num_keys);
int launch_group_by_key(const int32_t *keys, int64_t count, int64_t num_keys, int32_t *rowptr, int32_t *rows,
                        void *workspace, size_t workspace_bytes, int sort_segments, hipStream_t st);

constexpr int kDegreeBuckets = 32;  // folded update: exact in-degree buckets 0..31
constexpr int kDegBlock = 1024;     // nodes per workgroup in the degree bucketing passes (16 waves)

__device__ __forceinline__ int clamp_degree(int d, int32_t *err) {
  if (d >= kDegreeBuckets) {
    if (err) atomicOr(err, GNNSAFT_FLAG_BAD_DEGREE);
    d = kDegreeBuckets - 1;
  }
  return d;
}

// The degree scalers of PyG's DegreeScalerAggregation, as float32 values the way the
// reference forms them: log of a float32 integer (here the correctly rounded float32 logarithm -- one definition for
// every kernel that needs it), float32 division by / of the float32 buffer avg_deg_log.
__device__ __forceinline__ float degree_log_amp(int d) { return (float)log((double)d + 1.0); }                 // log(d + 1)
__device__ __forceinline__ float degree_log_att(int d) { return (float)log((double)(d > 1 ? d : 1) + 1.0); }   // log(max(d,1) + 1)
__device__ __forceinline__ void degree_scalers(int d, float avg, float &amp, float &att) {
  amp = degree_log_amp(d) / avg;
  att = avg / degree_log_att(d);
}

// Per-wave counts of every degree among this wave's nodes -> wcount[wave][bucket] (LDS), and
// the lane's rank among the wave's nodes of the same degree.  No atomics: deterministic.
__device__ __forceinline__ int wave_degree_ranks(int d, bool live, int32_t (*wcount)[kDegreeBuckets]) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  int rank = 0;
  unsigned long long todo = __ballot(live);
  while (todo != 0ull) {
    const int leader = __ffsll((long long)todo) - 1;
    const int dl = __shfl(d, leader);
    const unsigned long long same = __ballot(live && d == dl);
    if (lane == leader) wcount[wave][dl] = __popcll(same);
    if (live && d == dl) rank = __popcll(same & ((1ull << lane) - 1ull));
    todo &= ~same;
  }
  return rank;
}

// block_hist[block][d] = number of nodes with (clamped) in-degree d among the kDegBlock nodes of this workgroup;
// every thread of a kDegBlock-wide workgroup must call it (d ignored where !live)
__device__ __forceinline__ void block_degree_hist(int d, bool live, int32_t *__restrict__ block_hist) {
  __shared__ int32_t wcount[kDegBlock / 64][kDegreeBuckets];
  for (int t = threadIdx.x; t < (kDegBlock / 64) * kDegreeBuckets; t += kDegBlock) (&wcount[0][0])[t] = 0;
  __syncthreads();
  wave_degree_ranks(d, live, wcount);
  __syncthreads();
  if (threadIdx.x < kDegreeBuckets) {
    int tot = 0;
    for (int w = 0; w < kDegBlock / 64; ++w) tot += wcount[w][threadIdx.x];
    block_hist[(int64_t)blockIdx.x * kDegreeBuckets + threadIdx.x] = tot;
  }
}

// The K0 chain run by cooperating workgroups of the prologue launch (elementwise.hip: k0_chain_body)
constexpr int kK0MaxWgs = 128;        // co-resident by construction: the first workgroups of the launch
constexpr int kK0SyncInts = 3;        // grid barrier, ticket, "this call lost a barrier" (<= GNNSAFT_K0_SYNC_WORDS)
constexpr int kK0LostWord = 2;        // index of the third among the sync words
constexpr int kK0Group = 256;         // nodes per group of the chain (rows, look-back scan, degree histogram)
struct BondDims {                     // vocabulary sizes of the bond attribute columns (edge class = mixed radix)
  int32_t n;
  int32_t dims[GNNSAFT_MAX_TABLES];
};
struct K0ChainArgs {
  int wgs = 0;                        // 0 = off
  int self_loops = 0, tile_rows = 0, barrier_extra = 0;
  BondDims bd;
  const int64_t *edge_index = nullptr, *edge_attr = nullptr, *batch = nullptr;
  int64_t n = 0, e = 0, g = 0;
  int32_t *graph_ptr = nullptr, *slots = nullptr, *rowptr = nullptr;
  int32_t *src = nullptr, *dst = nullptr, *combo = nullptr;
  float *log_amp = nullptr, *log_att = nullptr;
  unsigned long long *lookback = nullptr;   // [groups] (status << 32 | rows): decoupled look-back of the row offsets
  int32_t *group_hist = nullptr;      // [groups][kDegreeBuckets]; the plan turns it into prefixes in place
  int32_t *hist = nullptr, *start = nullptr, *tiles = nullptr, *num_tiles = nullptr;
  int32_t *sync = nullptr;            // PERSISTENT: kK0SyncInts ints, zero at entry; barrier and ticket zero again at exit of
                                      // the launch, the lost word at the end of the forward (the pooling launch)
  int32_t *cursor = nullptr;          // PERSISTENT: n fill cursors, zero at entry, zero again at exit
  int32_t *err = nullptr;
};

// internal fused launchers used by gnnsaft_forward (the C entry points keep the one-job-per-call form)
int launch_forward_prologue(const int64_t *x_idx, int64_t num_rows, int32_t num_atom_cols,
                            const float *const *atom_tables_host, const int32_t *atom_dims_host,
                            int32_t num_bond_cols, const float *const *bond_tables_host,
                            const int32_t *bond_dims_host, int32_t hidden, float *x_out, float *cemb,
                            int32_t *zero_ptr, int64_t zero_count, int32_t fold_layers,
                            const float *const *w_post0_host, const float *const *w_post1_host,
                            const float *const *w_pre0_host, const float *const *w_pre1_host, double *g_all,
                            int32_t *err_flag, hipStream_t st, int32_t *zero2_ptr = nullptr, int zero2_count = 0,
                            const struct EdgeTableLayers *tables = nullptr /* fold.hpp */, int32_t table_layers = 0,
                            float *cenc = nullptr, float *rtab = nullptr, const K0ChainArgs *k0 = nullptr);
// `clear_word` (both pooling launchers; or null): a word this launch leaves zero -- the structure chain's "lost" word,
// read by the launches in front of the pooling, zero again for the next call
int launch_add_pool_bn(const float *y, const float *xprev, const float *scale, const float *shift, float *xout,
                       const int32_t *graph_ptr, int64_t num_graphs, int64_t num_nodes, int hidden, float *out,
                       hipStream_t st, int32_t *clear_word = nullptr);
int launch_add_pool(const float *x, const int32_t *graph_ptr, int64_t num_graphs, int64_t num_nodes, int hidden,
                    float *out, hipStream_t st, int32_t *clear_word = nullptr);
// out[0 .. count) and loss3[0 .. 3) become NaN when *lost != 0 (the per-op readout after a lost structure barrier)
int launch_poison_if(const int32_t *lost, float *out, int64_t count, float *loss3, hipStream_t st);
void csr_zero_region(void *workspace, int64_t num_nodes, int32_t **ptr, int64_t *count);
int launch_csr_build(const int64_t *edge_index, const int64_t *edge_attr, int64_t num_nodes, int64_t num_edges,
                     int32_t num_bond_cols, const int32_t *bond_dims_host, int32_t self_loops, int32_t *rowptr,
                     int32_t *src, int32_t *dst, int32_t *combo, float *log_amp, float *log_att, int32_t *err_flag,
                     void *workspace, size_t workspace_bytes, const int64_t *batch, int64_t num_graphs,
                     int32_t *graph_ptr, int32_t *degree_block_hist, bool counts_zeroed, hipStream_t st,
                     bool bounded_degree = false /* in-degree < kDegreeBuckets: the slotted chain (csr.hip) */);
// optional rider of launch_degree_tiles: fold the update weights of all layers in the launch that fills the
// permutation (both need only the degree plan)
struct DegreeFoldRequest {
  int num_layers;
  const float *const *w_post0, *const *w_post1, *const *avg;
  const double *g_all;  // destination-term products (k_dst_fold, float64) or null
  float *w_eff;
  int64_t layer_stride;
  char *w_eff3 = nullptr;   // also as W3 images (w3.hpp; 6 bytes per weight, block order of w_eff), or null
};
// the arrays the cooperative K0 chain wrote (for the empty structure installed after a lost barrier: degree.hip)
struct K0Installed {
  int32_t *rowptr, *src, *dst, *combo;
  int64_t ep;
  float *log_amp, *log_att;
  int32_t *sync, *cursor;   // the chain's persistent words: restored to zero by the launch that installs the empty structure
  int32_t *lost_out;        // workspace word written by EVERY call: 1 = this call's structure chain lost a barrier
};
int launch_degree_tiles(const int32_t *rowptr, int64_t num_nodes, int32_t hidden, int32_t *perm, int32_t *tiles,
                        int32_t *num_tiles, int32_t *scratch, int32_t *err_flag, bool have_block_hist, hipStream_t st,
                        const DegreeFoldRequest *fold = nullptr, const K0Installed *installed = nullptr);
// where launch_csr_build keeps the fill cursors, the scan's tile sums and the slot rows inside its workspace
void csr_workspace_parts(void *workspace, int64_t num_nodes, int32_t **cursor, int32_t **tile_sums, int32_t **slots);
int launch_fold_post_weights(int32_t num_layers, const float *const *w_post0_host, const float *const *w_post1_host,
                             const float *const *avg_deg_log_host, const float *const *w_pre0_host,
                             const float *const *w_pre1_host, double *g_scratch, const int32_t *hist, int32_t hidden,
                             float *w_eff, int64_t layer_stride, int phases /* 1: dst fold, 2: degree fold */,
                             hipStream_t st, char *w_eff3 = nullptr /* W3 images of w_eff (w3.hpp) */);

}  // namespace gs
