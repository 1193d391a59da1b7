// K4 -- PNA multi-aggregator segmented reduction (the "scatter-add" kernel of
// BASELINE.json): mean | min | max | std over the in-edges of every node, PyG
// MultiAggregation([mean,min,max,std]) as configured at
// /root/reference/gnnepcsaft/train/models.py:59,69-80 (semantics: SURVEY.md
// Appendix A.2 step 3).
//
// PyG runs 8 scatter passes with atomics over the unsorted edge list; here the
// rows are destination-sorted once per batch (csr.hip), so one thread owns a
// float4 column slice of one node and walks the node's in-edges with the
// running sum / sum of squares / min / max in registers: every message row is
// read exactly once, every output written exactly once, no atomics, edge order
// fixed => bitwise reproducible.  A message row is [T*F] = 2H floats: 64 lanes
// x float4 = one full 1 KiB wave load at H = 128 (two waves per node at 256,
// two nodes per wave at 64).
//
// Three sources for the message row r of node i:
//   kMsgs    (pre_layers >= 2): m = msgs[r, c]                      (materialised edge-MLP output)
//   kFusedPQ (pre_layers == 1): m = pq[i, c] + pq[src_r, 2F + c] + rtab[combo_r, c]
//   kFusedQ  (pre_layers == 1, destination term folded into the update weights):
//                               m~ = q[src_r, c] + rtab[combo_r, c];  the outputs are the
//            aggregates of m~ = m - P_i: mean/min/max shifted by the per-node constant P_i (which
//            gnnsaft_pna_dst_fold moves into W_eff), std identical.
//
// HBM-bound.  Algorithmic bytes per launch:
//   4*2F*E' (each message row once) + 8*E' (ids) + 4*2F*4*N (four aggregates).
#include "common.hpp"

namespace gs {

constexpr int kEdgeBatch = 4;  // gathers kept in flight per thread
enum AggSource { kMsgs = 0, kFusedPQ = 1, kFusedQ = 2 };

// UNIFORM: a node's 2F / 4 lanes are a whole number of waves (F a multiple of 128), so the node, its CSR range and the
// edge loop's trip count are wave-uniform: read as scalars (readfirstlane), the loop control and the edge-guard run on the
// scalar unit (no per-lane exec masks), the neighbour / class indices come through scalar loads.
template <int MODE, bool STREAM_OUT = false, bool UNIFORM = false>
__global__ __launch_bounds__(256) void k_pna_aggregate(const int32_t *__restrict__ rowptr,
                                                       const int32_t *__restrict__ src,
                                                       const int32_t *__restrict__ combo,
                                                       const float *__restrict__ pq, const float *__restrict__ rtab,
                                                       const float *__restrict__ msgs, float *__restrict__ agg,
                                                       int64_t num_nodes, int f, RowSplit rs) {
  // rs.per_row = f / 2 lanes per node (2F floats / 4 per thread)
  // (XCD-contiguous block numbering: the q rows of a molecule are gathered by ~3 nodes each -- through one L2)
  const int64_t slot = (int64_t)gs_xcd_block_grouped(blockIdx.x, gridDim.x, gs_xcd_span(f)) * blockDim.x + threadIdx.x;
  int64_t node;
  int lane_in_node;
  gs_split(rs, slot, node, lane_in_node);
  if (node >= num_nodes) return;
  if constexpr (UNIFORM) {
    const int nlo = __builtin_amdgcn_readfirstlane((int)node), nhi = __builtin_amdgcn_readfirstlane((int)(node >> 32));
    node = ((int64_t)nhi << 32) | (uint32_t)nlo;
  }
  const int c = lane_in_node * 4;  // column in [0, 2F)
  const int tower = c >= f ? 1 : 0;
  const int col = c - tower * f;
  // source-term row layout: [N,4F] with Q at column 2F (kFusedPQ) or [N,2F] (kFusedQ)
  const int q_stride = MODE == kFusedPQ ? 4 * f : 2 * f;
  const int q_off = MODE == kFusedPQ ? 2 * f : 0;

  int beg = rowptr[node];
  int end = rowptr[node + 1];
  if constexpr (UNIFORM) {
    beg = __builtin_amdgcn_readfirstlane(beg);
    end = __builtin_amdgcn_readfirstlane(end);
  }

  f32x4 p = {0.f, 0.f, 0.f, 0.f};
  if (MODE == kFusedPQ) p = gs_ld4(pq + node * (int64_t)(4 * f) + c);

  // Sums are taken of d = m - m_first (the segment's first row): var = E[d^2] - E[d]^2 then has no cancellation.
  // The textbook E[m^2] - E[m]^2 in f32 carries an absolute error of ~1e-7 m^2, i.e. ~1 % of PyG's 1e-5 masking
  // threshold: std within 0.8 % only near the threshold, mask flips, and -- through d std/dm = (m - mean)/(n std) --
  // gradient errors of 2e-3 on the message weights (profiles/r02_std_variance_gradient_analysis.txt).
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f}, v0 = {0.f, 0.f, 0.f, 0.f};
  const float inf = __builtin_huge_valf();
  f32x4 mn = {inf, inf, inf, inf}, mx = {-inf, -inf, -inf, -inf};

  for (int r0 = beg; r0 < end; r0 += kEdgeBatch) {
    f32x4 mrow[kEdgeBatch];
    if (MODE != kMsgs) {
      int sidx[kEdgeBatch], cidx[kEdgeBatch];
#pragma unroll
      for (int j = 0; j < kEdgeBatch; ++j) {
        const int r = r0 + j < end ? r0 + j : end - 1;
        sidx[j] = src[r];
        cidx[j] = combo[r];
      }
      f32x4 q[kEdgeBatch], t[kEdgeBatch];
#pragma unroll
      for (int j = 0; j < kEdgeBatch; ++j) {
        q[j] = gs_ld4(pq + (int64_t)sidx[j] * q_stride + q_off + c);
        t[j] = gs_ld4(rtab + (int64_t)cidx[j] * (2 * f) + c);
      }
#pragma unroll
      for (int j = 0; j < kEdgeBatch; ++j) mrow[j] = MODE == kFusedPQ ? (p + q[j]) + t[j] : q[j] + t[j];
    } else {
#pragma unroll
      for (int j = 0; j < kEdgeBatch; ++j) {
        const int r = r0 + j < end ? r0 + j : end - 1;
        mrow[j] = gs_ld4(msgs + (int64_t)r * (2 * f) + c);
      }
    }
    if (r0 == beg) v0 = mrow[0];
#pragma unroll
    for (int j = 0; j < kEdgeBatch; ++j) {
      if (r0 + j < end) {
        const f32x4 v = mrow[j];
        const f32x4 d = v - v0;
        s += d;
        s2 += d * d;
        mn.x = fminf(mn.x, v.x);
        mn.y = fminf(mn.y, v.y);
        mn.z = fminf(mn.z, v.z);
        mn.w = fminf(mn.w, v.w);
        mx.x = fmaxf(mx.x, v.x);
        mx.y = fmaxf(mx.y, v.y);
        mx.z = fmaxf(mx.z, v.z);
        mx.w = fmaxf(mx.w, v.w);
      }
    }
  }

  const int cnt = end - beg;
  f32x4 mean = {0.f, 0.f, 0.f, 0.f}, sd = {0.f, 0.f, 0.f, 0.f};
  if (cnt > 0) {
    // (round 4: hipcc expands every f32 `/` into ~10 and every sqrtf into ~13 instructions -- 140 of this thread's ~225
    //  for a node of three in-edges.  The quotients by the edge count keep their correct rounding at a third of the
    //  instructions (gs_div_count); PyG's mask -- std = var.clamp(min=1e-5).sqrt(), 0 where std <= sqrt(1e-5) -- is
    //  exactly `var <= 1e-5f` with a correctly rounded root (sqrt(1e-5f) rounds to the threshold, the next float above
    //  it does not), so it is taken on the variance, the clamp has nothing left to do (what it would clamp is masked;
    //  a NaN variance stays NaN, as in torch) and the root is gs_sqrt_rn: v_sqrt_f32 + the one-ulp residual test,
    //  correctly rounded without the expansion's rescaling of tiny arguments.  Same bits as before.)
    const float fc = (float)cnt, inv = 1.f / fc;
    float sv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float dm = gs_div_count(s[j], fc, inv);
      mean[j] = v0[j] + dm;
      const float var = gs_div_count(s2[j], fc, inv) - dm * dm;
      sv[j] = var <= 1e-5f ? 0.f : gs_sqrt_rn(var);
    }
    sd = f32x4{sv[0], sv[1], sv[2], sv[3]};
  } else {
    mn = f32x4{0.f, 0.f, 0.f, 0.f};
    mx = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float *o = agg + node * (int64_t)(8 * f) + tower * (4 * f) + col;
  if constexpr (STREAM_OUT) {   // beyond the Infinity Cache: the aggregates must not evict the q rows being re-read
    // (a template parameter: under a run-time condition the compiler merges the two arms into plain stores)
    gs_st4_stream(o, mean);
    gs_st4_stream(o + f, mn);
    gs_st4_stream(o + 2 * f, mx);
    gs_st4_stream(o + 3 * f, sd);
  } else {
    gs_st4(o, mean);
    gs_st4(o + f, mn);
    gs_st4(o + 2 * f, mx);
    gs_st4(o + 3 * f, sd);
  }
}

// h1pre[r, c] = pq[dst_r, c] + pq[src_r, 2F + c] + rtab[combo_r, c]: pre-activation of the first pre layer per
// CSR row, materialised only on the tape path of pre_layers >= 2 (the backward needs it as ReLU mask / wgrad operand)
__global__ __launch_bounds__(256) void k_edge_preact(const int32_t *__restrict__ src, const int32_t *__restrict__ dst,
                                                     const int32_t *__restrict__ combo, const float *__restrict__ pq,
                                                     const float *__restrict__ rtab, float *__restrict__ out,
                                                     int64_t rows, int f, RowSplit rs) {
  const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t r;
  int lane;
  gs_split(rs, slot, r, lane);
  if (r >= rows) return;
  const int c = lane * 4;
  const f32x4 p = gs_ld4(pq + (int64_t)dst[r] * (4 * f) + c), q = gs_ld4(pq + (int64_t)src[r] * (4 * f) + 2 * f + c),
              t = gs_ld4(rtab + (int64_t)combo[r] * (2 * f) + c);
  gs_st4(out + r * (int64_t)(2 * f) + c, (p + q) + t);
}

static int launch_aggregate(int mode, const int32_t *rowptr, const int32_t *src, const int32_t *combo,
                            int64_t num_nodes, int32_t hidden, const float *pq, const float *rtab, const float *msgs,
                            float *agg, hipStream_t st) {
  GS_REQUIRE(rowptr != nullptr && agg != nullptr, GNNSAFT_ERR_NULL);
  GS_REQUIRE(hidden >= 32 && (hidden % 32) == 0, GNNSAFT_ERR_SHAPE);
  GS_REQUIRE(num_nodes >= 0 && num_nodes * (int64_t)(hidden / 2) < ((int64_t)1 << 40), GNNSAFT_ERR_SHAPE);
  if (num_nodes == 0) return GNNSAFT_OK;
  const int64_t threads = num_nodes * (hidden / 2);
  const dim3 grid((unsigned)gs_ceil_div(threads, 256)), block(256);
  // streaming stores once the aggregates (N x 8F floats) exceed the 256 MiB Infinity Cache: measured C3 (1.34 GB) K4
  // 432 -> 400 us; at C2 (84 MB) they make the update GEMM, which finds the aggregates in cache, 3 us slower
  const int stream_out = num_nodes * (int64_t)(8 * hidden) * 4 > ((int64_t)256 << 20) ? 1 : 0;
  const RowSplit rs = gs_row_split(hidden / 2);
  const bool uni = (hidden / 2) % 64 == 0;   // a node = whole waves
  GS_REQUIRE(mode == kMsgs ? msgs != nullptr : (src && combo && pq && rtab), GNNSAFT_ERR_NULL);
#define GS_K4(MODE_, STREAM_, UNI_)                                                                                  \
  hipLaunchKernelGGL((k_pna_aggregate<MODE_, STREAM_, UNI_>), grid, block, 0, st, rowptr, src, combo, pq, rtab, msgs, agg, \
                     num_nodes, hidden, rs)
#define GS_K4_MODE(MODE_)                  \
  do {                                     \
    if (stream_out && uni)                 \
      GS_K4(MODE_, true, true);            \
    else if (stream_out)                   \
      GS_K4(MODE_, true, false);           \
    else if (uni)                          \
      GS_K4(MODE_, false, true);           \
    else                                   \
      GS_K4(MODE_, false, false);          \
  } while (0)
  if (mode == kMsgs) {
    if (uni)
      GS_K4(kMsgs, false, true);
    else
      GS_K4(kMsgs, false, false);
  } else if (mode == kFusedPQ) {
    GS_K4_MODE(kFusedPQ);
  } else {
    GS_K4_MODE(kFusedQ);
  }
#undef GS_K4_MODE
#undef GS_K4
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

}  // namespace gs

extern "C" int gnnsaft_pna_aggregate(const int32_t *rowptr, const int32_t *src, const int32_t *combo,
                                     int64_t num_nodes, int32_t hidden, const float *pq, const float *rtab,
                                     const float *msgs, float *agg, gnnsaft_stream_t stream) {
  return gs::launch_aggregate(msgs != nullptr ? gs::kMsgs : gs::kFusedPQ, rowptr, src, combo, num_nodes, hidden, pq,
                              rtab, msgs, agg, static_cast<hipStream_t>(stream));
}

extern "C" int gnnsaft_pna_aggregate_src(const int32_t *rowptr, const int32_t *src, const int32_t *combo,
                                         int64_t num_nodes, int32_t hidden, const float *q, const float *rtab,
                                         float *agg, gnnsaft_stream_t stream) {
  return gs::launch_aggregate(gs::kFusedQ, rowptr, src, combo, num_nodes, hidden, q, rtab, nullptr, agg,
                              static_cast<hipStream_t>(stream));
}

extern "C" int gnnsaft_pna_edge_preact(const int32_t *src, const int32_t *dst, const int32_t *combo, int64_t num_rows,
                                       int32_t hidden, const float *pq, const float *rtab, float *out,
                                       gnnsaft_stream_t stream) {
  GS_REQUIRE(src && dst && combo && pq && rtab && out, GNNSAFT_ERR_NULL);
  GS_REQUIRE(hidden >= 32 && (hidden % 32) == 0 && num_rows >= 0, GNNSAFT_ERR_SHAPE);
  if (num_rows == 0) return GNNSAFT_OK;
  hipLaunchKernelGGL(gs::k_edge_preact, dim3((unsigned)gs_ceil_div(num_rows * (hidden / 2), 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), src, dst, combo, pq, rtab, out, num_rows, hidden,
                     gs_row_split(hidden / 2));
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}
