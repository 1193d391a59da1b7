// Backward of the PNAPCSAFT forward + MAPE loss: what autograd does through
// /root/reference/gnnepcsaft/train/models.py:105-135,191-194 when Lightning calls loss.backward()
// (SURVEY.md section 8(f) rank 1).  Everything is recomputed from the forward's tape (x_l, pq_l, agg_l,
// u_l, y_l, BatchNorm batch statistics) that gnnsaft_forward keeps in its workspace when
// desc->save_tape is set; nothing here runs on the CPU and nothing synchronises.
//
// Per layer, in reverse:  BatchNorm+ReLU backward -> lin (dgrad NT / wgrad TN) -> update (dgrad through
// the degree-folded weights, wgrad against the virtual cat[x, A, A*amp, A*att]) -> aggregation backward
// (mean / tie-aware min, max / std, per CSR row) -> gather of the row gradients by SOURCE node through a
// transposed CSR (no atomics) + reduction by edge class -> message GEMMs (dgrad / wgrad) -> edge-class
// table (edge_encoder, bond embeddings).  Supported: pre_layers == post_layers == 1 (the shipped
// default, configs/default.py:39-40), hidden % 64 == 0.
#include "plan.hpp"
#include "readout.hpp"

namespace gs {

// ------------------------------------------------------------------ small elementwise pieces
__global__ void k_mape_bwd(const float *__restrict__ pred, const float *__restrict__ target, int64_t g, int p, int ldp,
                           const float *__restrict__ dloss, float *__restrict__ dout) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= g * ldp) return;
  const int64_t r = i / ldp;
  const int c = (int)(i - r * ldp);
  float v = 0.f;
  if (c < p) {
    const float t = target[r * p + c], d = pred[r * p + c] - t;
    const float s = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    v = s / fmaxf(fabsf(t), 1.17e-06f) / (float)(g * p) * (dloss != nullptr ? dloss[0] : 1.f);
  }
  dout[i] = v;
}

__global__ void k_pad_cols(const float *__restrict__ in, int64_t rows, int cols, int ld_out, float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * ld_out) return;
  const int64_t r = i / ld_out;
  const int c = (int)(i - r * ld_out);
  out[i] = c < cols ? in[r * cols + c] : 0.f;
}

// gradients that are exactly zero by construction (biases in front of a train-mode BatchNorm: the normalisation
// removes any per-column constant): written as zeros in one launch instead of a column-sum pass each
constexpr int kMaxZeroList = 24;
struct ZeroList {
  float *p[kMaxZeroList];
  int n[kMaxZeroList];
};
__global__ void k_fill_zero_list(ZeroList z) {
  float *p = z.p[blockIdx.y];
  const int n = z.n[blockIdx.y];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0.f;
}

__global__ void k_fill_zero(float *__restrict__ p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

// dx[i, :] = dg[graph(i), :]  (backward of global_add_pool)
__global__ __launch_bounds__(256) void k_pool_bwd(const float *__restrict__ dg, const int32_t *__restrict__ ptr,
                                                  int64_t graphs, int64_t nodes, int h, float *__restrict__ dx,
                                                  RowSplit rs) {
  const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t g;
  int lane;
  gs_split(rs, slot, g, lane);
  if (g >= graphs) return;
  const int c = lane * 4;
  int64_t beg = ptr[g], end = ptr[g + 1];
  beg = beg < 0 ? 0 : (beg > nodes ? nodes : beg);
  end = end < beg ? beg : (end > nodes ? nodes : end);
  const f32x4 v = gs_ld4(dg + g * h + c);
  for (int64_t r = beg; r < end; ++r) gs_st4(dx + r * h + c, v);
}

// ------------------------------------------------------------------ BatchNorm (+ReLU) backward, train mode
constexpr int kBwdCols = 32;
// partial[chunk][0][col] = sum dz, partial[chunk][1][col] = sum dz yhat over the chunk's rows (dz = dout masked by the
// ReLU).  A workgroup owns 32 columns x one row chunk: 8 float4 column lanes x 32 row lanes, four rows per thread in
// flight (the first version read one float per thread per row: 1.7 TB/s on a 2 x [N,H] stream).  ch % 4 == 0.
__global__ __launch_bounds__(256) void k_bn_bwd_partial(const float *__restrict__ y, const float *__restrict__ dout,
                                                        const float *__restrict__ stat,
                                                        const float *__restrict__ gamma,
                                                        const float *__restrict__ beta, int64_t rows, int ch,
                                                        int64_t rows_per_chunk, float *__restrict__ partial) {
  __shared__ double s1s[32][kBwdCols + 1], s2s[32][kBwdCols + 1];
  const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int col = blockIdx.x * kBwdCols + cl * 4;
  const bool col_ok = col < ch;
  const int colc = col_ok ? col : 0;
  const f32x4 mean = gs_ld4(stat + colc), rstd = gs_ld4(stat + ch + colc), gm = gs_ld4(gamma + colc),
              bt = gs_ld4(beta + colc);
  const int64_t r_beg = (int64_t)blockIdx.y * rows_per_chunk;
  int64_t r_end = r_beg + rows_per_chunk;
  if (r_end > rows) r_end = rows;
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t r0 = r_beg + rl; r0 < r_end; r0 += 4 * 32) {
    f32x4 yv[4], dv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int64_t r = r0 + 32 * u;
      r = r < r_end ? r : r_end - 1;
      yv[u] = gs_ld4(y + r * ch + colc);
      dv[u] = gs_ld4(dout + r * ch + colc);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool ok = r0 + 32 * u < r_end;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float yh = (yv[u][j] - mean[j]) * rstd[j];
        // the ReLU gate exactly as the forward took it: relu(y * scale + shift) with scale = rstd * gamma,
        // shift = beta - mean * scale (bn_fold.hpp) -- "yhat * gamma + beta > 0" rounds differently near zero and
        // then differentiates another branch than the one the forward evaluated
        const float sc = rstd[j] * gm[j], sh = bt[j] - mean[j] * sc;
        const float dz = (ok && (yv[u][j] * sc + sh) > 0.f) ? dv[u][j] : 0.f;
        s1[j] += (double)dz;
        s2[j] += (double)dz * (double)yh;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    s1s[rl][cl * 4 + j] = s1[j];
    s2s[rl][cl * 4 + j] = s2[j];
  }
  __syncthreads();
  if (threadIdx.x < 2 * kBwdCols) {   // 32 columns x (sum dz | sum dz yhat), row lanes combined in a fixed order
    const int which = threadIdx.x / kBwdCols, c = threadIdx.x % kBwdCols;
    double t = 0.0;
    for (int o = 0; o < 32; ++o) t += which == 0 ? s1s[o][c] : s2s[o][c];
    const int cc = blockIdx.x * kBwdCols + c;
    if (cc < ch) partial[((int64_t)blockIdx.y * 2 + which) * ch + cc] = (float)t;
  }
}

__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float *__restrict__ y, const float *__restrict__ dout,
                                                      const float *__restrict__ stat, const float *__restrict__ gamma,
                                                      const float *__restrict__ beta, int64_t rows, int ch,
                                                      int64_t rows_per_chunk, const float *__restrict__ partial,
                                                      int64_t chunks, float *__restrict__ dgamma,
                                                      float *__restrict__ dbeta, float *__restrict__ dy, int eval_bn) {
  __shared__ double s1s[8][kBwdCols], s2s[8][kBwdCols];
  __shared__ float s_a[kBwdCols], s_b[kBwdCols], s_c[kBwdCols], s_mean[kBwdCols], s_rstd[kBwdCols], s_g[kBwdCols],
      s_bt[kBwdCols];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c0 = blockIdx.x * kBwdCols;
  const int col = c0 + cl;
  const int colc = col < ch ? col : ch - 1;
  double s1 = 0.0, s2 = 0.0;
  for (int64_t j = rl; j < chunks; j += 8) {
    s1 += (double)partial[(j * 2 + 0) * ch + colc];
    s2 += (double)partial[(j * 2 + 1) * ch + colc];
  }
  s1s[rl][cl] = s1;
  s2s[rl][cl] = s2;
  __syncthreads();
  if (rl == 0) {
    for (int o = 1; o < 8; ++o) {
      s1 += s1s[o][cl];
      s2 += s2s[o][cl];
    }
    const float gm = gamma[colc], rstd = stat[ch + colc];
    // dy = gamma*rstd*(dz - s1/N - yhat*s2/N) = a*dz - b - c*yhat   (running statistics: a constant affine map,
    // the two batch-mean terms vanish)
    s_a[cl] = gm * rstd;
    s_b[cl] = eval_bn ? 0.f : gm * rstd * (float)(s1 / (double)rows);
    s_c[cl] = eval_bn ? 0.f : gm * rstd * (float)(s2 / (double)rows);
    s_mean[cl] = stat[colc];
    s_rstd[cl] = rstd;
    s_g[cl] = gm;
    s_bt[cl] = beta[colc];
    if (blockIdx.y == 0 && col < ch) {
      if (dgamma != nullptr) dgamma[col] = (float)s2;
      if (dbeta != nullptr) dbeta[col] = (float)s1;
    }
  }
  __syncthreads();
  const int c4 = (threadIdx.x & 7) * 4;
  const int rr = threadIdx.x >> 3;
  if (c0 + c4 >= ch) return;
  const int64_t r_beg = (int64_t)blockIdx.y * rows_per_chunk;
  int64_t r_end = r_beg + rows_per_chunk;
  if (r_end > rows) r_end = rows;
  for (int64_t r = r_beg + rr; r < r_end; r += 32) {
    const int64_t o = r * ch + c0 + c4;
    const f32x4 yv = gs_ld4(y + o), dv = gs_ld4(dout + o);
    f32x4 res;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float yh = (yv[j] - s_mean[c4 + j]) * s_rstd[c4 + j];
      const float sc = s_rstd[c4 + j] * s_g[c4 + j], sh = s_bt[c4 + j] - s_mean[c4 + j] * sc;   // the forward's gate
      const float dz = (yv[j] * sc + sh) > 0.f ? dv[j] : 0.f;
      res[j] = s_a[c4 + j] * dz - s_b[c4 + j] - s_c[c4 + j] * yh;
    }
    gs_st4(dy + o, res);
  }
}

static int bn_relu_backward(const float *y, const float *dout, const float *stat, const float *gamma,
                            const float *beta, int64_t rows, int ch, float *dgamma, float *dbeta, float *dy,
                            float *partial, hipStream_t st, int eval_bn = 0) {
  GS_REQUIRE(ch >= 4 && (ch % 4) == 0, GNNSAFT_ERR_UNSUPPORTED);
  const int slabs = (ch + kBwdCols - 1) / kBwdCols;
  int64_t chunks = 512 / slabs;
  const int64_t max_chunks = gs_ceil_div(rows, 64);
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  int64_t rpc = gs_ceil_div(rows, chunks);
  rpc = gs_ceil_div(rpc, 32) * 32;
  chunks = gs_ceil_div(rows, rpc);
  hipLaunchKernelGGL(k_bn_bwd_partial, dim3((unsigned)slabs, (unsigned)chunks), dim3(256), 0, st, y, dout, stat, gamma,
                     beta, rows, ch, rpc, partial);
  hipLaunchKernelGGL(k_bn_bwd_apply, dim3((unsigned)slabs, (unsigned)chunks), dim3(256), 0, st, y, dout, stat, gamma,
                     beta, rows, ch, rpc, partial, chunks, dgamma, dbeta, dy, eval_bn);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

// ------------------------------------------------------------------ aggregation backward
// For node i, column slice c: recompute its in-edge messages exactly as the forward did, then
//   dm_e = dmean/cnt + [m_e == min] dmin/#ties + [m_e == max] dmax/#ties + dstd (m_e - mean)/(cnt std)
// (std term only where the forward left std unmasked).  dm rows are written in CSR order, dP_i = sum_e dm_e.
template <bool MSGS>  // MSGS: the messages are a materialised [E',2F] tensor (pre_layers >= 2), dP comes later
__global__ __launch_bounds__(256) void k_agg_bwd(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ src,
                                                 const int32_t *__restrict__ combo, const float *__restrict__ pq,
                                                 const float *__restrict__ rtab, const float *__restrict__ msgs,
                                                 const float *__restrict__ agg, const float *__restrict__ dagg,
                                                 float *__restrict__ dm, float *__restrict__ dp, int64_t num_nodes,
                                                 int f, RowSplit rs) {
  const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t node;
  int lane;
  gs_split(rs, slot, node, lane);
  if (node >= num_nodes) return;
  const int c = lane * 4;
  const int tower = c >= f ? 1 : 0;
  const int col = c - tower * f;
  const int beg = rowptr[node], end = rowptr[node + 1];
  const int cnt = end - beg;
  f32x4 dpacc = {0.f, 0.f, 0.f, 0.f};
  if (cnt > 0) {
    const int64_t ao = node * (int64_t)(8 * f) + tower * (4 * f) + col;
    const f32x4 mean = gs_ld4(agg + ao), mn = gs_ld4(agg + ao + f), mx = gs_ld4(agg + ao + 2 * f),
                sd = gs_ld4(agg + ao + 3 * f);
    const f32x4 dmean = gs_ld4(dagg + ao), dmn = gs_ld4(dagg + ao + f), dmx = gs_ld4(dagg + ao + 2 * f),
                dsd = gs_ld4(dagg + ao + 3 * f);
    f32x4 p = {0.f, 0.f, 0.f, 0.f};
    if (!MSGS) p = gs_ld4(pq + node * (int64_t)(4 * f) + c);
    auto message = [&](int r) -> f32x4 {
      if (MSGS) return gs_ld4(msgs + (int64_t)r * (2 * f) + c);
      return (p + gs_ld4(pq + (int64_t)src[r] * (4 * f) + 2 * f + c)) + gs_ld4(rtab + (int64_t)combo[r] * (2 * f) + c);
    };
    f32x4 nmin = {0.f, 0.f, 0.f, 0.f}, nmax = {0.f, 0.f, 0.f, 0.f};
    for (int r = beg; r < end; ++r) {
      const f32x4 m = message(r);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        nmin[j] += m[j] == mn[j] ? 1.f : 0.f;
        nmax[j] += m[j] == mx[j] ? 1.f : 0.f;
      }
    }
    const float fc = (float)cnt;
    f32x4 cs;
#pragma unroll
    for (int j = 0; j < 4; ++j) cs[j] = sd[j] > 0.f ? dsd[j] / (fc * sd[j]) : 0.f;
    for (int r = beg; r < end; ++r) {
      const f32x4 m = message(r);
      f32x4 d;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = dmean[j] / fc + cs[j] * (m[j] - mean[j]);
        if (m[j] == mn[j]) v += dmn[j] / nmin[j];
        if (m[j] == mx[j]) v += dmx[j] / nmax[j];
        d[j] = v;
      }
      gs_st4(dm + (int64_t)r * (2 * f) + c, d);
      dpacc += d;
    }
  }
  if (!MSGS) gs_st4(dp + node * (int64_t)(4 * f) + c, dpacc);  // dPQ[:, 0:2F]
}

// dP_i = sum of the row gradients of node i's in-edges (CSR rows are contiguous per destination)
__global__ __launch_bounds__(256) void k_segment_sum(const int32_t *__restrict__ rowptr, const float *__restrict__ dm,
                                                     float *__restrict__ dpq, int64_t num_nodes, int f, RowSplit rs) {
  const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t node;
  int lane;
  gs_split(rs, slot, node, lane);
  if (node >= num_nodes) return;
  const int c = lane * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int r = rowptr[node]; r < rowptr[node + 1]; ++r) acc += gs_ld4(dm + (int64_t)r * (2 * f) + c);
  gs_st4(dpq + node * (int64_t)(4 * f) + c, acc);
}

// dq[j, :] = sum over the CSR rows whose source is j (ascending row id) -> dPQ[:, 2F:4F]
__global__ __launch_bounds__(256) void k_gather_rows_sum(const int32_t *__restrict__ rowptr_s,
                                                         const int32_t *__restrict__ rows_s,
                                                         const float *__restrict__ dm, float *__restrict__ dpq,
                                                         int64_t num_nodes, int f, RowSplit rs) {
  const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t node;
  int lane;
  gs_split(rs, slot, node, lane);
  if (node >= num_nodes) return;
  const int c = lane * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s = rowptr_s[node]; s < rowptr_s[node + 1]; ++s) acc += gs_ld4(dm + (int64_t)rows_s[s] * (2 * f) + c);
  gs_st4(dpq + node * (int64_t)(4 * f) + 2 * f + c, acc);
}

constexpr int64_t kClassGemmMax = 256;  // up to this many edge classes the class sums run as a one-hot TN GEMM

// Fallback for many classes: dr[class, :] = sum of dm[row, :] over the rows of that edge class.  `rows_c` lists the CSR rows grouped by
// class (built once per backward); a thread owns a float4 column slice of a 32-row chunk of that list, keeps a
// running sum while the class stays the same and flushes with one atomic add per class change (chunks hold
// one or two classes), so the 60-row table is not hammered.
__global__ __launch_bounds__(256) void k_class_reduce(const int32_t *__restrict__ rows_c,
                                                      const int32_t *__restrict__ combo, const float *__restrict__ dm,
                                                      int64_t rows, int classes, int f, float *__restrict__ dr,
                                                      RowSplit rs) {
  const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t chunk;
  int lane;
  gs_split(rs, slot, chunk, lane);
  const int64_t r_beg = chunk * 32;
  if (r_beg >= rows) return;
  int64_t r_end = r_beg + 32;
  if (r_end > rows) r_end = rows;
  const int c = lane * 4, width = 2 * f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int cur = -1;
  for (int64_t i = r_beg; i < r_end; ++i) {
    const int r = rows_c[i];
    const int cls = combo[r];
    if (cls != cur) {
      if (cur >= 0 && cur < classes) {
        float *o = dr + (int64_t)cur * width + c;
        atomicAdd(o + 0, acc.x);
        atomicAdd(o + 1, acc.y);
        atomicAdd(o + 2, acc.z);
        atomicAdd(o + 3, acc.w);
      }
      acc = f32x4{0.f, 0.f, 0.f, 0.f};
      cur = cls;
    }
    acc += gs_ld4(dm + (int64_t)r * width + c);
  }
  if (cur >= 0 && cur < classes) {
    float *o = dr + (int64_t)cur * width + c;
    atomicAdd(o + 0, acc.x);
    atomicAdd(o + 1, acc.y);
    atomicAdd(o + 2, acc.z);
    atomicAdd(o + 3, acc.w);
  }
}

// WTA[d][t][c][o] = W_t[o][F+c] + amp(d) W_t[o][5F+c] + att(d) W_t[o][9F+c]   (c < 4F, o < F/2)
// for every layer in one launch: blockIdx.z = (layer, degree, tower)
constexpr int kMaxFoldLayers = 16;
struct PostPair {
  const float *w0, *w1;
};
struct FoldLayers {
  PostPair pp[kMaxFoldLayers];
  const float *avg[kMaxFoldLayers];
};
__global__ __launch_bounds__(256) void k_fold_post_weights_t(FoldLayers fl, const int32_t *__restrict__ hist, int f,
                                                             float *__restrict__ wta_all, int64_t wta_per_layer) {
  __shared__ float tl[32][33];
  const int layer = blockIdx.z / (kDegreeBuckets * 2);
  const int dz = blockIdx.z - layer * (kDegreeBuckets * 2);
  const int d = dz / 2, t = dz % 2;
  if (hist[d] == 0) return;
  const float *w = t == 0 ? fl.pp[layer].w0 : fl.pp[layer].w1;
  float amp_f, att_f;
  degree_scalers(d, fl.avg[layer][0], amp_f, att_f);
  const double amp = (double)amp_f, att = (double)att_f;   // float64 accumulate, one rounding (fold.hpp)
  const int o0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int o = o0 + ty + 8 * j;
    const float *wr = w + (int64_t)o * (13 * f) + f + c0 + tx;
    tl[ty + 8 * j][tx] = (float)((double)wr[0] + (double)wr[4 * f] * amp + (double)wr[8 * f] * att);
  }
  __syncthreads();
  float *out = wta_all + layer * wta_per_layer + (((int64_t)d * 2 + t) * (4 * f)) * (f / 2);
#pragma unroll
  for (int j = 0; j < 4; ++j) out[(int64_t)(c0 + ty + 8 * j) * (f / 2) + o0 + tx] = tl[tx][ty + 8 * j];
}

// ---- edge-class table chain for few classes (the reference has 5*6*2 = 60 bond-feature combinations):
//   rtab[c, tF + f] = sum_j W_t[f][2F + j] cenc[c][j] + b_t[f]       cenc[c][j] = sum_i W_e[j][i] cemb[c][i] + b_e[j]
// Given dr = d rtab [C, 2F] of one layer, two launches replace six (two TN GEMMs, two column sums, two dgrads whose
// contraction is only C rows long): every output element is one thread's loop over the classes / channels, in a fixed
// order.
struct EdgeTableBwd {
  const float *dr;     // [C, 2F]
  const float *cenc;   // [C, F]
  const float *cemb;   // [C, F]
  const float *wpre0, *wpre1;   // [F, 3F]
  const float *we;     // [F, F]
  float *dwpre0, *dwpre1;       // [F, 3F]: columns 2F.. written
  float *dbpre0, *dbpre1;       // [F]
  float *dwe, *dbe;    // [F, F], [F]
  float *dcenc;        // [C, F] scratch
  float *dcemb;        // [C, F] accumulated over the layers
  int classes, f;
};
// phase A: dW_t[:, 2F:3F] = dr_t^T cenc, db_t = column sums of dr_t, dcenc = sum_t dr_t W_t[:, 2F:3F]
// (the loops carry nothing but the sum: unrolled so that a batch of independent L2 loads is in flight -- the rolled
// version waited for one load pair per term, 92 us for 60 classes)
constexpr int kEdgeUnroll = 16;
__global__ __launch_bounds__(256) void k_edge_table_bwd_a(EdgeTableBwd a) {
  const int f = a.f, C = a.classes;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n_w = 2 * (int64_t)f * f, n_b = 2 * (int64_t)f, n_c = (int64_t)C * f;
  if (i < n_w) {
    const int t = (int)(i / ((int64_t)f * f));
    const int rem = (int)(i - (int64_t)t * f * f);
    const int row = rem / f, j = rem - row * f;
    const float *d = a.dr + t * f + row;
    const float *e = a.cenc + j;
    float s = 0.f;
    int c = 0;
    for (; c + kEdgeUnroll <= C; c += kEdgeUnroll) {
      float dv[kEdgeUnroll], ev[kEdgeUnroll];
#pragma unroll
      for (int u = 0; u < kEdgeUnroll; ++u) {
        dv[u] = d[(int64_t)(c + u) * (2 * f)];
        ev[u] = e[(int64_t)(c + u) * f];
      }
#pragma unroll
      for (int u = 0; u < kEdgeUnroll; ++u) s += dv[u] * ev[u];
    }
    for (; c < C; ++c) s += d[(int64_t)c * (2 * f)] * e[(int64_t)c * f];
    (t == 0 ? a.dwpre0 : a.dwpre1)[(int64_t)row * (3 * f) + 2 * f + j] = s;
  } else if (i < n_w + n_b) {
    const int k = (int)(i - n_w);
    float s = 0.f;
#pragma unroll 8
    for (int c = 0; c < C; ++c) s += a.dr[(int64_t)c * (2 * f) + k];
    (k < f ? a.dbpre0 : a.dbpre1)[k < f ? k : k - f] = s;
  } else if (i < n_w + n_b + n_c) {
    const int64_t r = i - n_w - n_b;
    const int c = (int)(r / f), j = (int)(r - (int64_t)c * f);
    const float *d = a.dr + (int64_t)c * (2 * f);
    float s = 0.f;
    for (int t = 0; t < 2; ++t) {
      const float *w = (t == 0 ? a.wpre0 : a.wpre1) + 2 * f + j;
      const float *dt = d + t * f;
      for (int k = 0; k < f; k += kEdgeUnroll) {   // f % 16 == 0 (hidden % 64 == 0)
        float wv[kEdgeUnroll];
#pragma unroll
        for (int u = 0; u < kEdgeUnroll; ++u) wv[u] = w[(int64_t)(k + u) * (3 * f)];
#pragma unroll
        for (int u = 0; u < kEdgeUnroll; ++u) s += dt[k + u] * wv[u];
      }
    }
    a.dcenc[r] = s;
  }
}
// phase B: dW_e = dcenc^T cemb, db_e = column sums of dcenc, dcemb += dcenc W_e
__global__ __launch_bounds__(256) void k_edge_table_bwd_b(EdgeTableBwd a) {
  const int f = a.f, C = a.classes;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n_w = (int64_t)f * f, n_b = f, n_c = (int64_t)C * f;
  if (i < n_w) {
    const int j = (int)(i / f), col = (int)(i - (int64_t)j * f);
    const float *d = a.dcenc + j;
    const float *e = a.cemb + col;
    float s = 0.f;
    int c = 0;
    for (; c + kEdgeUnroll <= C; c += kEdgeUnroll) {
      float dv[kEdgeUnroll], ev[kEdgeUnroll];
#pragma unroll
      for (int u = 0; u < kEdgeUnroll; ++u) {
        dv[u] = d[(int64_t)(c + u) * f];
        ev[u] = e[(int64_t)(c + u) * f];
      }
#pragma unroll
      for (int u = 0; u < kEdgeUnroll; ++u) s += dv[u] * ev[u];
    }
    for (; c < C; ++c) s += d[(int64_t)c * f] * e[(int64_t)c * f];
    a.dwe[i] = s;
  } else if (i < n_w + n_b) {
    const int j = (int)(i - n_w);
    float s = 0.f;
#pragma unroll 8
    for (int c = 0; c < C; ++c) s += a.dcenc[(int64_t)c * f + j];
    a.dbe[j] = s;
  } else if (i < n_w + n_b + n_c) {
    const int64_t r = i - n_w - n_b;
    const int c = (int)(r / f), col = (int)(r - (int64_t)c * f);
    const float *d = a.dcenc + (int64_t)c * f;
    const float *w = a.we + col;
    float s = 0.f;
    for (int j = 0; j < f; j += kEdgeUnroll) {
      float wv[kEdgeUnroll];
#pragma unroll
      for (int u = 0; u < kEdgeUnroll; ++u) wv[u] = w[(int64_t)(j + u) * f];
#pragma unroll
      for (int u = 0; u < kEdgeUnroll; ++u) s += d[j + u] * wv[u];
    }
    a.dcemb[r] += s;
  }
}

// bond tables: dtab_k[v, :] (+)= sum over combinations c with digit_k(c) == v of dcemb[c, :]
struct TableGrads {
  int32_t n;
  int32_t dims[GNNSAFT_MAX_TABLES];
  float *grad[GNNSAFT_MAX_TABLES];
};
__global__ __launch_bounds__(256) void k_combo_embed_bwd(const float *__restrict__ dcemb, int64_t combos, int h,
                                                         TableGrads tg) {
  const int k = blockIdx.y;
  if (tg.grad[k] == nullptr) return;
  const int per_row = h / 4;
  const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t v = slot / per_row;
  if (v >= tg.dims[k]) return;
  const int c = (int)(slot - v * per_row) * 4;
  int64_t stride = 1;
  for (int j = tg.n - 1; j > k; --j) stride *= tg.dims[j];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int64_t cid = 0; cid < combos; ++cid)
    if ((cid / stride) % tg.dims[k] == v) acc += gs_ld4(dcemb + cid * h + c);
  gs_st4(tg.grad[k] + v * h + c, acc);
}

// atom tables: dtab_k[v, hh] = det[hh][off_k + v]   (det = one-hot^T dx0, stored transposed)
// all tables in one launch: thread i owns element (vocabulary row i / h of the concatenated tables, channel i % h)
__global__ __launch_bounds__(256) void k_unpack_embed_grad(const float *__restrict__ det, int ld, int h, TableGrads tg,
                                                           int total_rows) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)total_rows * h) return;
  int v = (int)(i / h);
  const int hh = (int)(i - (int64_t)v * h);
  const int row = v;
  int table = 0;
  while (table + 1 < tg.n && v >= tg.dims[table]) v -= tg.dims[table++];
  tg.grad[table][(int64_t)v * h + hh] = det[(int64_t)hh * ld + row];
}

struct Scratch {
  char *base;
  size_t off = 0, cap;
  template <class T>
  T *take(size_t count) {
    const size_t o = off;
    off += gs_align_up(count * sizeof(T), 256);
    return off <= cap ? reinterpret_cast<T *>(base + o) : nullptr;
  }
};

struct BwdSizes {
  size_t slab, total;
};

static BwdSizes backward_sizes(const gnnsaft_model_desc *d, const Plan &p) {
  const size_t h = d->hidden, nn = p.n > 0 ? p.n : 1, ee = p.ep > 0 ? p.ep : 1, gg = p.g > 0 ? p.g : 1;
  const size_t rows = nn > gg ? nn : gg;
  // the slab arena holds the partial results of ONE layer's five reductions side by side (SlabQueue): post_nns
  // weights of both towers, lin weights, the message weights' four [F,F] blocks, per-class sums, bias column sums
  const size_t s1 = wgrad_post_folded_slab_bytes(p.tile_cap, pna_fold_tile_rows((int)h), (int)h);
  const size_t s2 = tn_slab_bytes(p.n, (int)h, 176 + 16);                 // one-hot (atom vocabulary rows)
  size_t s3 = tn_slab_bytes(rows, (int)(4 * h), (int)h);                 // message weights: four [F,F] blocks at once
  const size_t s3e = d->pre_layers > 1 ? tn_slab_bytes(ee, (int)h, (int)h) : 0;  // edge-row wgrads of extra pre layers
  s3 = s3 > s3e ? s3 : s3e;
  const size_t s4 = (size_t)p.combos <= (size_t)kClassGemmMax ? tn_slab_bytes(p.ep, (int)p.combos, (int)(2 * h)) : 0;
  const size_t s5 = tn_slab_bytes(rows, (int)h, (int)h);                 // lin
  const size_t s6 = (rows / 256 + 2) * h * 4;                            // column-sum partials
  size_t slab = s1 + s3 + s4 + s5 + 2 * s6 + 8 * 256;   // (two column-sum sets with an eval-mode BatchNorm)
  slab = slab > s2 ? slab : s2;
  const bool rb_fused = !d->unfused_readout && d->training && readout_bwd_fused_supported(p.g, (int)h, d->num_para, p.nb);
  const size_t s7 = rb_fused ? readout_bwd_slab_floats(p.g, (int)h, d->num_para, p.nb) * 4 : 0;
  slab = slab > s7 ? slab : s7;
  size_t tot = 0;
  auto add = [&](size_t b) { tot += gs_align_up(b, 256); };
  add(slab);
  const size_t nlay_b = (size_t)(d->num_layers > 0 ? d->num_layers : 1);
  for (int i = 0; i < 2; ++i) add(nn * h * 4);          // dxa, dxb
  // every layer keeps its own dy / du / dm / dpq / dr: the side stream reads them (weight gradients, class sums)
  // while the main stream is already in the next layer
  add(nlay_b * nn * h * 4);                              // dy
  add(nlay_b * nn * h * 4);                              // du
  add(nn * 8 * h * 4);                                   // dagg
  add(nlay_b * ee * 2 * h * 4);                          // dm
  if (d->pre_layers > 1) add(ee * 2 * h * 4);            // dm2
  add(nlay_b * nn * 4 * h * 4);                          // dpq
  add(nlay_b * (size_t)p.combos * 2 * h * 4);            // dr
  add((size_t)p.combos * h * 4 * 2);                     // dcenc, dcemb
  add(1024 * 2 * h * 4);                                 // bn partials
  add(h * h * 4);                                        // wlinT (extra pre / post layers)
  {
    const size_t nlay = (size_t)(d->num_layers > 0 ? d->num_layers : 1);
    add(nlay * 9 * h * h * 4);                           // per layer: wlinT | [wxT | wpqT] | wcT | weT
    add((size_t)(d->num_mlp_layers + 2) * h * h * 4);    // readout: W^T of every BatchNorm block
  }
  add(nlay_b * (size_t)kDegreeBuckets * 2 * 4 * h * (h / 2) * 4); // wta, per layer
  add((nn + 1) * 4);                                     // rowptr_s
  add(ee * 4);                                           // rows_s
  add((size_t)(p.combos + 1) * 4);                        // rowptr_c
  add(ee * 4);                                           // rows_c
  add(group_by_key_workspace_bytes(p.n > p.combos ? p.n : p.combos));
  add(gg * h * 4 * 3);                                   // readout dcur, dnext, dyr
  add(gg * 8 * 4);                                       // padded dout
  add(h * 8 * 4);                                        // padded W3^T
  add(h * (size_t)(176 + 16) * 4);                       // one-hot result
  add(readout_bwd_scratch_floats(p.g, (int)h, p.nb) * 4); // fused readout backward: BatchNorm / bias partials
  add(kRdSyncInts * 4);                                   //   ... and its barrier counters
  return BwdSizes{slab, tot + 65536};
}

}  // namespace gs

using namespace gs;

extern "C" size_t gnnsaft_backward_scratch_bytes(const gnnsaft_model_desc *desc, int64_t num_nodes, int64_t num_edges,
                                                 int64_t num_graphs) {
  Plan p;
  if (make_plan(desc, num_nodes, num_edges, num_graphs, p) != GNNSAFT_OK) return 0;
  return backward_sizes(desc, p).total;
}

extern "C" int gnnsaft_mape_backward(const float *pred, const float *target, int64_t num_graphs, int32_t num_para,
                                     const float *dloss, float *dpred, gnnsaft_stream_t stream) {
  GS_REQUIRE(pred && target && dpred, GNNSAFT_ERR_NULL);
  const int64_t tot = num_graphs * num_para;
  hipLaunchKernelGGL(k_mape_bwd, dim3((unsigned)gs_ceil_div(tot, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     pred, target, num_graphs, num_para, num_para, dloss, dpred);
  GS_CHECK_LAUNCH();
  return GNNSAFT_OK;
}

extern "C" int gnnsaft_backward(const gnnsaft_model_desc *d, const void *const *weights_host,
                                void *const *grads_host, int32_t num_weights, const int64_t *x_idx,
                                const int64_t *batch, int64_t num_nodes, int64_t num_edges, int64_t num_graphs,
                                const float *grad_out /* [G,P] */, void *tape, size_t tape_bytes, void *scratch,
                                size_t scratch_bytes, void *const *segment_events, int32_t *err_flag,
                                gnnsaft_aux *aux, gnnsaft_stream_t stream) {
  (void)batch;
  GS_REQUIRE(d && weights_host && grads_host && grad_out && tape && scratch && x_idx, GNNSAFT_ERR_NULL);
  GS_REQUIRE(d->save_tape, GNNSAFT_ERR_UNSUPPORTED);
  // eval-mode BatchNorm (running statistics; fine-tuning with frozen statistics): BatchNorm is a constant affine map,
  // and the biases in front of it have real gradients again
  const int eval_bn = d->training ? 0 : 1;
  GS_REQUIRE(d->pre_layers >= 1 && d->pre_layers <= 8 && d->post_layers >= 1 && d->post_layers <= 8 &&
                 (d->hidden % 64) == 0 && d->fold_degree_scalers && !d->fold_dst_term,
             GNNSAFT_ERR_UNSUPPORTED);
  Plan p;
  GS_TRY(make_plan(d, num_nodes, num_edges, num_graphs, p));
  GS_REQUIRE(tape_bytes >= p.total, GNNSAFT_ERR_WORKSPACE);
  const BwdSizes bs = backward_sizes(d, p);
  GS_REQUIRE(scratch_bytes >= bs.total && (reinterpret_cast<uintptr_t>(scratch) & 255) == 0, GNNSAFT_ERR_WORKSPACE);
  ParsedWeights pw;
  GS_TRY(parse_weights(d, weights_host, num_weights, pw));
  hipStream_t st = static_cast<hipStream_t>(stream);
  char *tp = static_cast<char *>(tape);
  auto F = [&](size_t off) { return reinterpret_cast<float *>(tp + off); };
  auto I = [&](size_t off) { return reinterpret_cast<int32_t *>(tp + off); };
  auto G = [&](int idx) { return static_cast<float *>(grads_host[idx]); };
  const int h = d->hidden, P = d->num_para, L = d->num_layers;
  const int64_t n = num_nodes, g = num_graphs;
  const int64_t C = p.combos;
  const int q = d->post_layers, pl = d->pre_layers;

  // Two branches.  The MAIN stream carries the chain every layer's input gradient depends on (BatchNorm backward ->
  // lin dgrad -> update dgrads -> aggregation backward -> gather by source -> message dgrad); the SIDE stream
  // (aux->stream) everything that only consumes that chain: weight and bias gradients, the per-edge-class sums and
  // the edge-table chain, plus the one-off preparation (transposed CSR, folded transposed update weights).  Each of
  // these kernels alone leaves most of the 256 CUs idle (a few hundred short workgroups), so the two streams
  // overlap almost completely.  Layers keep their own dy / du / dm / dpq / dr so that the side stream may lag.
  // Extra pre / post layers reuse buffers within a layer: they run single-stream.
  const bool two = aux != nullptr && aux->stream != nullptr && pl == 1 && q == 1;
  hipStream_t sa = two ? aux->stream : st;
  size_t next_event = 0;
  // `to` waits for everything enqueued on `from` so far
  auto order = [&](hipStream_t from, hipStream_t to) -> int {
    if (from == to) return GNNSAFT_OK;
    hipEvent_t e = aux_event(aux, next_event++);
    GS_REQUIRE(e != nullptr, GNNSAFT_ERR_WORKSPACE);
    GS_HIP(hipEventRecord(e, from));
    GS_HIP(hipStreamWaitEvent(to, e, 0));
    return GNNSAFT_OK;
  };
  auto mark = [&](hipStream_t from, hipEvent_t &e) -> int {   // ... or wait later (`await`)
    e = nullptr;
    if (!two) return GNNSAFT_OK;
    e = aux_event(aux, next_event++);
    GS_REQUIRE(e != nullptr, GNNSAFT_ERR_WORKSPACE);
    GS_HIP(hipEventRecord(e, from));
    return GNNSAFT_OK;
  };
  auto await = [&](hipStream_t to, hipEvent_t e) -> int {
    if (e != nullptr) GS_HIP(hipStreamWaitEvent(to, e, 0));
    return GNNSAFT_OK;
  };

  const int64_t Ln = L > 0 ? L : 1;
  const int64_t epn = p.ep > 0 ? p.ep : 1;
  Scratch sc{static_cast<char *>(scratch), 0, scratch_bytes};
  float *slabs = sc.take<float>(bs.slab / 4);
  float *dxa = sc.take<float>(n * h), *dxb = sc.take<float>(n * h);
  float *dy_all = sc.take<float>(Ln * n * h), *du_all = sc.take<float>(Ln * n * h);
  float *dagg = sc.take<float>(n * 8 * h);
  float *dm_all = sc.take<float>(Ln * epn * 2 * h);
  float *dm2 = pl > 1 ? sc.take<float>(epn * 2 * h) : nullptr;
  float *dpq_all = sc.take<float>(Ln * n * 4 * h);
  float *dr_all = sc.take<float>(Ln * C * 2 * h);
  float *dcenc = sc.take<float>(C * h), *dcemb = sc.take<float>(C * h);
  float *bnpart = sc.take<float>(1024 * 2 * h);
  float *wlinT = sc.take<float>((size_t)h * h);
  const size_t wta_per_layer = (size_t)kDegreeBuckets * 2 * 4 * h * (h / 2);
  float *wta_all = sc.take<float>((size_t)Ln * wta_per_layer);
  int32_t *rowptr_s = sc.take<int32_t>(n + 1);
  int32_t *rows_s = sc.take<int32_t>(epn);
  int32_t *rowptr_c = sc.take<int32_t>(C + 1);
  int32_t *rows_c = sc.take<int32_t>(epn);
  char *grp_ws = sc.take<char>(group_by_key_workspace_bytes(n > C ? n : C));
  float *dcur = sc.take<float>(g * h), *dnext = sc.take<float>(g * h), *dyr = sc.take<float>(g * h);
  float *dout_pad = sc.take<float>(g * 8);
  float *w3T = sc.take<float>((size_t)h * 8);
  const int vocab_pad = 192;
  float *det = sc.take<float>((size_t)h * vocab_pad);
  const size_t per_layer_t = 9 * (size_t)h * h;      // wlinT [H][H] | wxpqT [H][5H] = [wxT | wpqT] | wcT [H][2H] | weT [H][H]
  float *wt_layers = sc.take<float>((size_t)Ln * per_layer_t);
  float *wt_readout = sc.take<float>((size_t)p.nb * h * h);
  float *rb_scratch = sc.take<float>(readout_bwd_scratch_floats(g, h, p.nb));
  int32_t *rb_sync = sc.take<int32_t>(kRdSyncInts);
  GS_REQUIRE(det != nullptr && wt_layers != nullptr && wt_readout != nullptr && rb_scratch != nullptr &&
                 rb_sync != nullptr,
             GNNSAFT_ERR_WORKSPACE);
  const bool rb_fused = !d->unfused_readout && d->training && readout_fused_launchable(g, h, P, p.nb, true);
  const size_t slab_bytes = bs.slab;
  SlabQueue sq;   // one layer's reductions, summed by one launch (side stream)
  sq.base = slabs;
  sq.cap = slab_bytes / 4;
  // extra pre / post layers reduce on the spot through the head of the same buffer: no queue for them
  SlabQueue *dq = (pl == 1 && q == 1) ? &sq : nullptr;

  auto transpose1 = [&](const float *in, int64_t ld_in, float *out, int64_t ld_out, int rows, int cols) {
    const float *i1[1] = {in};
    float *o1[1] = {out};
    return launch_transpose(1, i1, o1, &ld_in, &ld_out, rows, cols, st);
  };
  auto dgrad = [&](hipStream_t s_, const float *a, int64_t lda, const float *wT, int64_t ldw, float *out, int64_t ldo,
                   int64_t rows, int n_out, int k, const float *residual) {
    GemmBatchEntry e{wT, nullptr, out, 0};
    LinearEpilogue epi;
    epi.residual = residual;
    epi.ldr = ldo;
    return launch_linear(a, lda, 0, 1, &e, ldw, ldo, rows, n_out, k, epi, s_);
  };

  // =========================== side branch: one-off preparation ===========================
  GS_TRY(order(st, sa));  // fork: the tape and the scratch buffer belong to the caller's stream
  hipEvent_t ev_wta = nullptr, ev_csr = nullptr;
  for (int l0 = 0; l0 < L; l0 += kMaxFoldLayers) {  // transposed, degree-folded update weights (weights only)
    FoldLayers fl;
    const int cnt = L - l0 < kMaxFoldLayers ? L - l0 : kMaxFoldLayers;
    for (int i = 0; i < kMaxFoldLayers; ++i) {
      const LayerW &w = pw.layers[l0 + (i < cnt ? i : 0)];
      fl.pp[i] = PostPair{w.wpost[0][0], w.wpost[1][0]};
      fl.avg[i] = w.avg;
    }
    hipLaunchKernelGGL(k_fold_post_weights_t,
                       dim3((unsigned)(4 * h / 32), (unsigned)(h / 2 / 32), (unsigned)(cnt * kDegreeBuckets * 2)),
                       dim3(256), 0, sa, fl, I(p.hist3), h, wta_all + (size_t)l0 * wta_per_layer,
                       (int64_t)wta_per_layer);
  }
  GS_TRY(mark(sa, ev_wta));
  // transposed CSR (rows grouped by source) ... and, with many edge classes, rows grouped by class.  Few classes
  // (the reference has 5*6*2 = 60): the per-class sums of dm are a one-hot TN GEMM, no grouping needed.
  const bool class_gemm = C <= kClassGemmMax;
  if (L > 0) {
    GS_TRY(launch_group_by_key(I(p.src), p.ep, n, rowptr_s, rows_s, grp_ws, group_by_key_workspace_bytes(n), 1, sa));
    GS_TRY(mark(sa, ev_csr));
    if (!class_gemm)
      GS_TRY(launch_group_by_key(I(p.combo), p.ep, C, rowptr_c, rows_c, grp_ws,
                                 group_by_key_workspace_bytes(n > C ? n : C), 0, sa));
  }
  hipLaunchKernelGGL(k_fill_zero, dim3((unsigned)gs_ceil_div(C * h, 256)), dim3(256), 0, sa, dcemb, C * h);

  // =========================== readout backward (main) ===========================
  const int nb = p.nb;
  const int64_t rs = g * (int64_t)h;
  GS_REQUIRE(P <= 8, GNNSAFT_ERR_UNSUPPORTED);
  {  // biases in front of a train-mode BatchNorm (lin of every layer, every readout block): gradient exactly zero
    ZeroList z;
    int cnt = 0;
    auto flush = [&]() {
      if (cnt == 0) return;
      for (int i = cnt; i < kMaxZeroList; ++i) {
        z.p[i] = z.p[0];
        z.n[i] = 0;
      }
      hipLaunchKernelGGL(k_fill_zero_list, dim3(2, (unsigned)cnt), dim3(256), 0, st, z);
      cnt = 0;
    };
    auto push = [&](float *ptr, int count) {
      if (ptr == nullptr) return;
      z.p[cnt] = ptr;
      z.n[cnt] = count;
      if (++cnt == kMaxZeroList) flush();
    };
    if (!eval_bn) {
      for (int l = 0; l < L; ++l) {
        const int i_lin = pw.layer_base[l] + 3 + 4 * d->pre_layers + 4 * d->post_layers;
        push(G(i_lin + 1), h);
      }
      for (int bi = 0; bi < nb; ++bi) push(G(pw.readout_base[bi] + 1), pw.readout[bi].n_out);
    }
    push(w3T, h * 8);   // zero padding of the final Linear's transposed weight (filled by the batched transpose)
    push(reinterpret_cast<float *>(rb_sync), kRdSyncInts);   // barrier counters of the fused readout backward
    flush();
  }
  {  // every weight transpose the dgrads of this backward need, in one launch (64 matrices per launch)
    TransposeItem items[kMaxTransposeBatch];
    int cnt = 0;
    auto flush = [&]() -> int {
      if (cnt == 0) return GNNSAFT_OK;
      const int rc = launch_transpose_list(cnt, items, st);
      cnt = 0;
      return rc;
    };
    auto push = [&](const TransposeItem &it) -> int {
      items[cnt++] = it;
      return cnt == kMaxTransposeBatch ? flush() : GNNSAFT_OK;
    };
    const int64_t h3 = 3 * (int64_t)h, h13 = 13 * (int64_t)h, h5 = 5 * (int64_t)h, h2 = 2 * (int64_t)h;
    for (int l = 0; l < L; ++l) {
      const LayerW &w = pw.layers[l];
      float *wl = wt_layers + (size_t)l * per_layer_t;
      float *wlinT_l = wl, *wxpqT_l = wl + (size_t)h * h, *wcT_l = wl + 6 * (size_t)h * h, *weT_l = wl + 8 * (size_t)h * h;
      // wxpqT[j][0:H] = [W_x,0 ; W_x,1]^T,  wxpqT[j][H + blk F + f] = W_pre,t[f][part F + j]  (blk = 2 part + t)
      GS_TRY(push({w.wlin, wlinT_l, h, h, h, h}));
      GS_TRY(push({w.wpost[0][0], wxpqT_l, h13, h5, h / 2, h}));
      GS_TRY(push({w.wpost[1][0], wxpqT_l + h / 2, h13, h5, h / 2, h}));
      GS_TRY(push({w.wpre[0][0], wxpqT_l + h, h3, h5, h, h}));
      GS_TRY(push({w.wpre[1][0], wxpqT_l + 2 * h, h3, h5, h, h}));
      GS_TRY(push({w.wpre[0][0] + h, wxpqT_l + 3 * h, h3, h5, h, h}));
      GS_TRY(push({w.wpre[1][0] + h, wxpqT_l + 4 * h, h3, h5, h, h}));
      GS_TRY(push({w.wpre[0][0] + 2 * h, wcT_l, h3, h2, h, h}));
      GS_TRY(push({w.wpre[1][0] + 2 * h, wcT_l + h, h3, h2, h, h}));
      GS_TRY(push({w.we, weT_l, h, h, h, h}));
    }
    for (int bi = 0; bi < nb; ++bi) {
      const ReadoutW &rw = pw.readout[bi];
      GS_TRY(push({rw.w, wt_readout + (size_t)bi * h * h, rw.n_in, rw.n_out, rw.n_out, rw.n_in}));  // [n_in][n_out]
    }
    GS_TRY(push({pw.readout[nb].w, w3T, pw.readout[nb].n_in, 8, P, pw.readout[nb].n_in}));  // W3^T, 8 columns
    GS_TRY(flush());
  }
  GS_REQUIRE(rb_fused || !(d->training && d->readout_dropout > 0.f), GNNSAFT_ERR_UNSUPPORTED);   // dropout: fused only
  if (rb_fused) {
    // one launch for the whole readout (readout.hip: k_readout_bwd_fused) + one reduction of its per-workgroup partial
    // weight gradients
    ReadoutBwdParams rp;
    rp.g = g;
    rp.h = h;
    rp.num_para = P;
    rp.nblocks = nb;
    rp.grad_out = grad_out;
    for (int i = 0; i <= kRdMaxBlocks; ++i) {
      const bool live = i <= nb;
      rp.w[i] = live ? pw.readout[i].w : nullptr;
      rp.dw[i] = live ? G(pw.readout_base[i]) : nullptr;
      if (i < kRdMaxBlocks) {
        const bool bl = i < nb;
        rp.wt[i] = bl ? wt_readout + (size_t)i * h * h : nullptr;
        rp.gamma[i] = bl ? pw.readout[i].bn.gamma : nullptr;
        rp.beta[i] = bl ? pw.readout[i].bn.beta : nullptr;
        rp.dgamma[i] = bl ? G(pw.readout_base[i] + 2) : nullptr;
        rp.dbeta[i] = bl ? G(pw.readout_base[i] + 3) : nullptr;
      }
    }
    rp.db_final = G(pw.readout_base[nb] + 1);
    rp.pooled = F(p.pooled);
    rp.ry = F(p.ry);
    rp.ro = F(p.ro);
    rp.rstat = F(p.rstat);
    rp.dpooled = dcur;
    rp.scratch = rb_scratch;
    rp.sync = rb_sync;
    rp.err = err_flag;   // a lost barrier raises GNNSAFT_FLAG_BARRIER_TIMEOUT and poisons the gradients with NaN
    rp.barrier_extra = d->debug_barrier_extra;
    rp.dropout_p = d->training ? d->readout_dropout : 0.f;
    rp.dropout_seed = d->dropout_seed;
    rp.dropout_step = d->dropout_step;
    GS_TRY(launch_readout_bwd_fused(rp, sq, st));
    GS_TRY(launch_slab_queue_flush(sq, st));
  } else {
    {
      // final Linear(H/4 -> P): pad dOut to 8 columns so that the GEMMs can use 16-byte loads
      hipLaunchKernelGGL(k_pad_cols, dim3((unsigned)gs_ceil_div(g * 8, 256)), dim3(256), 0, st, grad_out, g, P, 8, dout_pad);
      const ReadoutW &fin = pw.readout[nb];
      const int ib = pw.readout_base[nb];
      const float *in = F(p.ro) + (nb - 1) * rs;  // output of the last BN block, width H/4
      GS_TRY(launch_wgrad_plain(dout_pad, 8, in, fin.n_in, 0, g, P, fin.n_in, G(ib), fin.n_in, 0, slabs, slab_bytes, st));
      GS_TRY(launch_colsum(dout_pad, 8, g, P, G(ib + 1), 0, slabs, slab_bytes, st));
      // dIn = dOut W3: W'[n_out = H/4][k = 8] = W3^T zero-padded (zero list + batched transpose above)
      GS_TRY(dgrad(st, dout_pad, 8, w3T, 8, dcur, fin.n_in, g, fin.n_in, 8, nullptr));
    }
    for (int bi = nb - 1; bi >= 0; --bi) {
      const ReadoutW &rw = pw.readout[bi];
      const int ib = pw.readout_base[bi];
      const float *yb = F(p.ry) + bi * rs;
      const float *in = bi == 0 ? F(p.pooled) : F(p.ro) + (bi - 1) * rs;
      const float *stat = F(p.rstat) + (int64_t)bi * 2 * h;
      GS_TRY(bn_relu_backward(yb, dcur, stat, rw.bn.gamma, rw.bn.beta, g, rw.n_out, G(ib + 2), G(ib + 3), dyr, bnpart, st,
                              eval_bn));
      if (eval_bn) GS_TRY(launch_colsum(dyr, rw.n_out, g, rw.n_out, G(ib + 1), 0, slabs, slab_bytes, st));
      GS_TRY(launch_wgrad_plain(dyr, rw.n_out, in, rw.n_in, 0, g, rw.n_out, rw.n_in, G(ib), rw.n_in, 0, slabs, slab_bytes,
                                st));
      // (bias gradient: exactly zero, written above)
      GS_TRY(dgrad(st, dyr, rw.n_out, wt_readout + (size_t)bi * h * h, rw.n_out, dnext, rw.n_in, g, rw.n_in, rw.n_out,
                   nullptr));
      float *t = dcur;
      dcur = dnext;
      dnext = t;
    }
  }
  // gradient segments complete in the order readout, layer L-1 .. 0, embeddings: an event per segment lets the
  // data-parallel exchange of a finished segment run under the rest of the backward (parallel.py).  A layer's
  // segment is finished by the side stream (its weight gradients come last).
  auto segment_done = [&](int i, hipStream_t s_) -> int {
    if (segment_events != nullptr && segment_events[i] != nullptr)
      GS_HIP(hipEventRecord(static_cast<hipEvent_t>(segment_events[i]), s_));
    return GNNSAFT_OK;
  };
  GS_TRY(segment_done(0, st));
  // pool backward: dx_L
  float *dx = dxa, *dx_other = dxb;
  hipLaunchKernelGGL(k_pool_bwd, dim3((unsigned)gs_ceil_div(g * (h / 4), 256)), dim3(256), 0, st, dcur,
                     I(p.graph_ptr), g, n, h, dx, gs_row_split(h / 4));
  // the readout's weight gradients used `slabs` on the main stream; from here on it belongs to the side stream
  // (... which also orders the weight transposes above before the side stream's edge-table dgrads)
  GS_TRY(order(st, sa));

  // =========================== layers, in reverse ===========================
  for (int l = L - 1; l >= 0; --l) {
    const LayerW &w = pw.layers[l];
    const int base = pw.layer_base[l];
    // table indices inside the layer (q = post_layers): avg 0 | we 1 be 2 | pre0 w3 b4 | pre1 w5 b6 |
    //   post0 (w,b) x q from 7 | post1 (w,b) x q from 7+2q | lin w,b | bn gamma, beta (running stats, counter)
    const int i_pre0 = base + 3, i_pre1 = base + 3 + 2 * pl;
    const int i_post0 = base + 3 + 4 * pl, i_post1 = i_post0 + 2 * q, i_lin = i_post1 + 2 * q, i_bn = i_lin + 2;
    const float *x_l = F(p.x0) + l * p.sx, *pq_l = F(p.pq) + l * p.spq, *agg_l = F(p.agg) + l * p.sagg;
    const float *u_first = F(p.u0) + l * p.su;                        // output of post layer 0
    const float *u_l = u_first + (int64_t)(q - 1) * n * h;            // output of the last post layer = lin's input
    const float *y_l = F(p.y) + l * p.sy;
    const float *stat = F(p.bnstat) + (int64_t)l * 2 * h;
    const float *rtab = F(p.rtab) + l * C * (int64_t)(2 * h);
    const float *cenc = F(p.cenc) + l * C * (int64_t)h;
    float *dy = dy_all + (int64_t)l * n * h, *du = du_all + (int64_t)l * n * h;
    float *dm = dm_all + (int64_t)l * epn * 2 * h, *dpq = dpq_all + (int64_t)l * n * 4 * h;
    float *dr = dr_all + (int64_t)l * C * 2 * h;
    float *wta = wta_all + (size_t)l * wta_per_layer;

    // transposed weights of this layer (built by the one batched launch at the top):
    //   wlinT | wxpqT[j][0:H] = W_post,t[o][j] (x block), wxpqT[j][H + blk F + f] = W_pre,t[f][part F + j] | wcT | weT
    float *wl = wt_layers + (size_t)l * per_layer_t;
    float *wlinT_l = wl, *wxpqT_l = wl + (size_t)h * h, *wcT_l = wl + 6 * (size_t)h * h, *weT_l = wl + 8 * (size_t)h * h;
    const int64_t h5 = 5 * (int64_t)h;
    // x_{l+1} = relu(bn(y)) + x_l : dy through BN+ReLU; the skip gradient stays in dx
    GS_TRY(bn_relu_backward(y_l, dx, stat, w.bn.gamma, w.bn.beta, n, h, G(i_bn), G(i_bn + 1), dy, bnpart, st, eval_bn));
    // lin: input gradient (main), then the fork for this layer's first side batch: weight gradient of lin (needs
    // dy), of the update and its biases (need du)
    // (lin.bias sits in front of the BatchNorm: gradient exactly zero, written by the zero list)
    GS_TRY(dgrad(st, dy, h, wlinT_l, h, du, h, n, h, h, nullptr));
    GS_TRY(order(st, sa));
    GS_TRY(launch_wgrad_plain(dy, h, u_l, h, 0, n, h, h, G(i_lin), h, 0, slabs, slab_bytes, sa, dq));
    if (eval_bn) {   // lin.bias: a real gradient behind a frozen BatchNorm
      float *outs[1] = {G(i_lin + 1)};
      GS_TRY(launch_colsum_blocks(dy, h, n, 1, h, outs, slabs, slab_bytes, sa, dq));
    }
    // extra post layers (Linear(F/2,F/2) after a ReLU, per tower), last to first: du_j -> du_{j-1}  (single stream)
    for (int j = q - 1; j >= 1; --j) {
      const float *u_prev = u_first + (int64_t)(j - 1) * n * h;  // pre-ReLU input of post layer j
      float *du_prev = dy;                                        // dy is free after the lin dgrad
      for (int t = 0; t < 2; ++t) {
        const float *wj = w.wpost[t][j];
        const int iw = (t == 0 ? i_post0 : i_post1) + 2 * j;
        GS_TRY(launch_wgrad_plain(du + t * (h / 2), h, u_prev + t * (h / 2), h, 1, n, h / 2, h / 2, G(iw), h / 2, 0,
                                  slabs, slab_bytes, st));
        GS_TRY(launch_colsum(du + t * (h / 2), h, n, h / 2, G(iw + 1), 0, slabs, slab_bytes, st));
        GS_TRY(transpose1(wj, h / 2, wlinT, h / 2, h / 2, h / 2));
        // d(relu input) = (du_t W_j) masked by u_prev > 0
        GemmBatchEntry e{wlinT, nullptr, du_prev + t * (h / 2), t * (int64_t)(h / 2)};
        LinearEpilogue epi;
        epi.residual = u_prev + t * (h / 2);
        epi.ldr = h;
        epi.residual_is_mask = 1;
        GS_TRY(launch_linear(du, h, 0, 1, &e, h / 2, h, n, h / 2, h / 2, epi, st));
      }
      float *tsw = du;
      du = du_prev;
      dy = tsw;
    }
    // update weight / bias gradients (side; du is final: with extra post layers everything is on one stream)
    GS_TRY(launch_wgrad_post_folded(du, x_l, agg_l, I(p.perm), I(p.tiles), I(p.num_tiles), p.tile_cap,
                                    pna_fold_tile_rows(h), w.avg, h, G(i_post0), G(i_post1), slabs, slab_bytes, sa,
                                    dq, I(p.hist3), n));
    {  // both towers' bias gradients: column sums of du, halves to two tensors
      float *outs[2] = {G(i_post0 + 1), G(i_post1 + 1)};
      GS_TRY(launch_colsum_blocks(du, h, n, 2, h / 2, outs, slabs, slab_bytes, sa, dq));
    }
    // (update dgrad, x part: merged with the message dgrad below -- one GEMM over [du | dPQ])
    // update dgrad, aggregate part (degree-tiled, scalers folded): dagg[i,t,:] = du_t[i] W_A,eff(d_i, t)
    {
      GS_TRY(await(st, ev_wta));
      ev_wta = nullptr;
      const int64_t per_t = (int64_t)(4 * h) * (h / 2);
      GemmBatchEntry e[2] = {{wta, nullptr, dagg, 0}, {wta + per_t, nullptr, dagg + 4 * h, h / 2}};
      GS_TRY(launch_linear_degree_tiled(du, h, I(p.perm), I(p.tiles), I(p.num_tiles), p.tile_cap, 2 * per_t, 2, e,
                                        h / 2, 8 * (int64_t)h, n, 4 * h, h / 2, h, st));
    }
    // aggregation backward: dm rows, dP; then dQ by source, dR by class
    if (pl == 1) {
      hipLaunchKernelGGL(k_agg_bwd<false>, dim3((unsigned)gs_ceil_div(n * (h / 2), 256)), dim3(256), 0, st,
                         I(p.rowptr), I(p.src), I(p.combo), pq_l, rtab, nullptr, agg_l, dagg, dm, dpq, n, h,
                         gs_row_split(h / 2));
    } else {
      // extra pre layers: messages = output of the last one (materialised per CSR row on the tape)  (single stream)
      const int64_t ms = p.ep * (int64_t)(2 * h);
      const float *m_l = F(p.msg0) + l * p.smsg;
      hipLaunchKernelGGL(k_agg_bwd<true>, dim3((unsigned)gs_ceil_div(n * (h / 2), 256)), dim3(256), 0, st,
                         I(p.rowptr), I(p.src), I(p.combo), pq_l, rtab, m_l + (pl - 1) * ms, agg_l, dagg, dm, dpq, n,
                         h, gs_row_split(h / 2));
      float *dcur_m = dm, *dnext_m = dm2;
      for (int j = pl - 1; j >= 1; --j) {
        const float *a_prev = m_l + (j - 1) * ms;  // pre-ReLU input of pre layer j
        for (int t = 0; t < 2; ++t) {
          const int iw = (t == 0 ? i_pre0 : i_pre1) + 2 * j;
          GS_TRY(launch_wgrad_plain(dcur_m + t * h, 2 * (int64_t)h, a_prev + t * h, 2 * (int64_t)h, 1, p.ep, h, h, G(iw), h,
                                    0, slabs, slab_bytes, st));
          GS_TRY(launch_colsum(dcur_m + t * h, 2 * (int64_t)h, p.ep, h, G(iw + 1), 0, slabs, slab_bytes, st));
          GS_TRY(transpose1(w.wpre[t][j], h, wlinT, h, h, h));
          GemmBatchEntry e{wlinT, nullptr, dnext_m + t * h, t * (int64_t)h};
          LinearEpilogue epi;
          epi.residual = a_prev + t * h;
          epi.ldr = 2 * (int64_t)h;
          epi.residual_is_mask = 1;
          GS_TRY(launch_linear(dcur_m, 2 * (int64_t)h, 0, 1, &e, h, 2 * (int64_t)h, p.ep, h, h, epi, st));
        }
        float *tsw = dcur_m;
        dcur_m = dnext_m;
        dnext_m = tsw;
      }
      if (dcur_m != dm) {  // the downstream kernels read `dm`
        float *tsw = dm;
        dm = dm2;
        dm2 = tsw;
      }
      hipLaunchKernelGGL(k_segment_sum, dim3((unsigned)gs_ceil_div(n * (h / 2), 256)), dim3(256), 0, st, I(p.rowptr), dm,
                         dpq, n, h, gs_row_split(h / 2));
    }
    // dQ by source (main; the transposed CSR comes from the side stream's preparation)
    GS_TRY(await(st, ev_csr));
    ev_csr = nullptr;
    hipLaunchKernelGGL(k_gather_rows_sum, dim3((unsigned)gs_ceil_div(n * (h / 2), 256)), dim3(256), 0, st, rowptr_s,
                       rows_s, dm, dpq, n, h, gs_row_split(h / 2));
    // side: per-class sums of dm, message weight gradients dW_dst,t / dW_src,t = dP_t^T x / dQ_t^T x, then ONE launch
    // that reduces this layer's five slab sets (lin, post_nns, bias column sums, class sums, message weights), then
    // the edge-class table chain  rtab[c, tF:(t+1)F] = W_t[:,2F:3F] cenc[c] + b_t ; cenc = cemb W_e^T + b_e
    GS_TRY(order(st, sa));
    if (class_gemm) {
      GS_TRY(launch_sum_rows_by_class(I(p.combo), (int)C, dm, 2 * (int64_t)h, p.ep, 2 * h, dr, 2 * (int64_t)h, slabs,
                                      slab_bytes, sa, dq));
    } else {
      hipLaunchKernelGGL(k_fill_zero, dim3((unsigned)gs_ceil_div(C * 2 * h, 256)), dim3(256), 0, sa, dr, C * 2 * h);
      hipLaunchKernelGGL(k_class_reduce, dim3((unsigned)gs_ceil_div(gs_ceil_div(p.ep, 32) * (h / 2), 256)),
                         dim3(256), 0, sa, rows_c, I(p.combo), dm, p.ep, (int)C, h, dr, gs_row_split(h / 2));
    }
    {  // dW_dst,t0 | dW_dst,t1 | dW_src,t0 | dW_src,t1 = dPQ^T x: one TN GEMM, four [F,F] blocks of two matrices
      float *blocks[4] = {G(i_pre0), G(i_pre1), G(i_pre0) + h, G(i_pre1) + h};
      GS_TRY(launch_wgrad_plain_blocks(dpq, 4 * (int64_t)h, x_l, h, n, 4, h, h, blocks, 3 * (int64_t)h, slabs,
                                       slab_bytes, sa, dq));
    }
    if (dq != nullptr) GS_TRY(launch_slab_queue_flush(sq, sa));
    if (class_gemm) {  // few classes: the six GEMM / column-sum launches below as two elementwise launches
      EdgeTableBwd eb{dr, cenc, F(p.cemb), w.wpre[0][0], w.wpre[1][0], w.we, G(i_pre0), G(i_pre1), G(i_pre0 + 1),
                      G(i_pre1 + 1), G(base + 1), G(base + 2), dcenc, dcemb, (int)C, h};
      const int64_t na = 2 * (int64_t)h * h + 2 * h + C * h, nb2 = (int64_t)h * h + h + C * h;
      hipLaunchKernelGGL(k_edge_table_bwd_a, dim3((unsigned)gs_ceil_div(na, 256)), dim3(256), 0, sa, eb);
      hipLaunchKernelGGL(k_edge_table_bwd_b, dim3((unsigned)gs_ceil_div(nb2, 256)), dim3(256), 0, sa, eb);
    } else {
      {
        float *blocks[2] = {G(i_pre0) + 2 * h, G(i_pre1) + 2 * h};
        GS_TRY(launch_wgrad_plain_blocks(dr, 2 * (int64_t)h, cenc, h, C, 2, h, h, blocks, 3 * (int64_t)h, slabs,
                                         slab_bytes, sa));
      }
      {
        float *outs[2] = {G(i_pre0 + 1), G(i_pre1 + 1)};
        GS_TRY(launch_colsum_blocks(dr, 2 * (int64_t)h, C, 2, h, outs, slabs, slab_bytes, sa));
      }
      GemmBatchEntry e1{wcT_l, nullptr, dcenc, 0};
      LinearEpilogue epi1;
      GS_TRY(launch_linear(dr, 2 * (int64_t)h, 0, 1, &e1, 2 * (int64_t)h, h, C, h, 2 * h, epi1, sa));
      GS_TRY(launch_wgrad_plain(dcenc, h, F(p.cemb), h, 0, C, h, h, G(base + 1), h, 0, slabs, slab_bytes, sa));
      GS_TRY(launch_colsum(dcenc, h, C, h, G(base + 2), 0, slabs, slab_bytes, sa));
      GS_TRY(dgrad(sa, dcenc, h, weT_l, h, dcemb, h, C, h, h, dcemb));  // accumulate over layers (in place)
    }
    // dx_in = (skip ? dx : 0) + [du | dP | dQ] [W_x,0 ; W_x,1 | W_pq]^T : K = 5F in one pass
    {
      GemmBatchEntry e{wxpqT_l, nullptr, dx_other, 0};
      LinearEpilogue epi;
      epi.residual = d->skip_connections ? dx : nullptr;
      epi.ldr = h;
      GS_TRY(launch_linear_concat2(du, h, h, dpq, 4 * (int64_t)h, 4 * h, e, h5, h, n, h, epi, st));
      float *tsw = dx;
      dx = dx_other;
      dx_other = tsw;
    }
    // dx now holds dL/dx_l; dx_other is free again.  The layer's last gradient is written by the side stream (which
    // has waited for this layer's BatchNorm backward, the only main-stream writer of the segment).
    GS_TRY(segment_done(1 + (L - 1 - l), sa));
  }

  // =========================== embeddings ===========================
  GS_TRY(order(sa, st));  // join: dcemb is complete, `slabs` is the main stream's again
  {
    TableGrads tg;
    tg.n = d->num_bond_cols;
    for (int k = 0; k < GNNSAFT_MAX_TABLES; ++k) {
      tg.dims[k] = k < d->num_bond_cols ? d->bond_dims[k] : 1;
      tg.grad[k] = k < d->num_bond_cols ? G(pw.bond0 + k) : nullptr;
    }
    int maxdim = 1;
    for (int k = 0; k < d->num_bond_cols; ++k) maxdim = d->bond_dims[k] > maxdim ? d->bond_dims[k] : maxdim;
    hipLaunchKernelGGL(k_combo_embed_bwd, dim3((unsigned)gs_ceil_div((int64_t)maxdim * (h / 4), 256),
                                                 (unsigned)d->num_bond_cols),
                       dim3(256), 0, st, dcemb, C, h, tg);
  }
  {
    int total = 0;
    for (int k = 0; k < d->num_atom_cols; ++k) total += d->atom_dims[k];
    GS_REQUIRE(total <= vocab_pad, GNNSAFT_ERR_UNSUPPORTED);
    GS_TRY(launch_wgrad_onehot(dx, h, x_idx, d->num_atom_cols, d->atom_dims, n, h, det, vocab_pad, slabs, slab_bytes,
                               st));
    TableGrads tg;
    tg.n = d->num_atom_cols;
    for (int k = 0; k < GNNSAFT_MAX_TABLES; ++k) {
      tg.dims[k] = k < d->num_atom_cols ? d->atom_dims[k] : 1;
      tg.grad[k] = k < d->num_atom_cols ? G(pw.atom0 + k) : nullptr;
    }
    hipLaunchKernelGGL(k_unpack_embed_grad, dim3((unsigned)gs_ceil_div((int64_t)total * h, 256)), dim3(256), 0, st, det,
                       vocab_pad, h, tg, total);
  }
  GS_CHECK_LAUNCH();
  GS_TRY(segment_done(L + 1, st));
  return GNNSAFT_OK;
}
