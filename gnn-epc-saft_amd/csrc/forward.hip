// Whole-network orchestration: PNAPCSAFT.forward
// (/root/reference/gnnepcsaft/train/models.py:105-135) + MAPE loss
// (models.py:191-194) as one C call that enqueues every kernel on the caller's
// stream.  No allocation, no synchronisation: capturable into a hipGraph.
//
// Canonical weight table (HOST array of device pointers), in order:
//   atom tables [num_atom_cols], bond tables [num_bond_cols]
//   per conv layer l:
//     aggr_module.avg_deg_log
//     edge_encoder.weight, edge_encoder.bias
//     pre_nns[0]: (weight, bias) x pre_layers ; pre_nns[1]: (weight, bias) x pre_layers
//     post_nns[0]: (weight, bias) x post_layers ; post_nns[1]: (weight, bias) x post_layers
//     lin.weight, lin.bias
//     batch_norms[l].module: weight, bias, running_mean, running_var, num_batches_tracked
//   readout: num_mlp_layers x (Linear weight, bias, BN x5)
//            Linear(H,H/2) w,b, BN x5, Linear(H/2,H/4) w,b, BN x5, Linear(H/4,P) w,b
#include <cstdlib>
#include <vector>

#include "bn_fold.hpp"
#include "common.hpp"
#include "fold.hpp"
#include "plan.hpp"
#include "readout.hpp"
#include "w3.hpp"

// Event pairs recorded on the launch stream around selected launches (bench.py roofline).
struct gnnsaft_profile {
  uint32_t mask = 0;
  int used = 0;
  std::vector<hipEvent_t> start, stop;
  std::vector<uint32_t> kind;
};

// gnnsaft_aux (plan.hpp): side stream for the structure chain (K0: CSR, graph ptr, degree tiles, folded weights).
// It depends only on edge_index / batch / weights, the embedding + edge-table + first message GEMM chain only on
// x / weights, so the two run concurrently between a fork and a join event (captured as parallel branches under
// hipGraph capture).
extern "C" int gnnsaft_aux_create(gnnsaft_aux **out) {
  if (out == nullptr) return GNNSAFT_ERR_NULL;
  gnnsaft_aux *a = new (std::nothrow) gnnsaft_aux();
  if (a == nullptr) return GNNSAFT_ERR_WORKSPACE;
  hipError_t e = hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&a->fork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&a->join, hipEventDisableTiming);
  if (e != hipSuccess) {
    gnnsaft_aux_destroy(a);
    return (int)e;
  }
  *out = a;
  return GNNSAFT_OK;
}

extern "C" void gnnsaft_aux_destroy(gnnsaft_aux *a) {
  if (a == nullptr) return;
  if (a->fork) (void)hipEventDestroy(a->fork);
  if (a->join) (void)hipEventDestroy(a->join);
  for (hipEvent_t e : a->pool) (void)hipEventDestroy(e);
  if (a->stream) (void)hipStreamDestroy(a->stream);
  delete a;
}

namespace gs {

struct ProfScope {
  gnnsaft_profile *p;
  int slot = -1;
  hipStream_t st;
  ProfScope(gnnsaft_profile *prof, uint32_t bit, hipStream_t stream) : p(prof), st(stream) {
    if (p != nullptr && (p->mask & bit) != 0 && p->used < (int)p->start.size()) {
      slot = p->used++;
      p->kind[slot] = bit;
      (void)hipEventRecord(p->start[slot], st);
    }
  }
  ~ProfScope() {
    if (slot >= 0) (void)hipEventRecord(p->stop[slot], st);
  }
};

// Linear -> BatchNorm -> ReLU (-> + residual): eval folds BN into the GEMM epilogue,
// train writes y with (mean, M2) partials, finalises the statistics and applies them.
// `defer_apply` (train mode): the batch statistics are closed by ONE small launch (k_bn_stats_close: scale / shift
// land in ws + p.scale / p.shift, running statistics updated) and NOTHING is applied here -- the consumer of the layer's
// output applies relu(y scale + shift) (+ residual) while it loads y (launch_linear_bnres, launch_add_pool_bn).
static int linear_bn_relu(const float *a, int64_t lda, const float *w, const float *b, int64_t rows, int n_out, int k,
                          const BnPtrs &bn, const gnnsaft_model_desc *d, char *ws, const Plan &p, float *y_tmp,
                          const float *residual, float *out, hipStream_t st, gnnsaft_profile *prof = nullptr,
                          float *save_stat = nullptr, bool defer_apply = false,
                          const char *w3 = nullptr /* W3 image of w (w3.hpp): the GEMM reads it instead of w */) {
  float *stats = reinterpret_cast<float *>(ws + p.stats);
  GemmBatchEntry ent{w, b, nullptr, 0, w3};
  const int w3cfg = w3 != nullptr ? w3_pick_cfg(rows, n_out, k, d->training != 0) : -1;
  auto launch_linear = [&](const float *a_, int64_t lda_, int relu_in, int nbatch, const GemmBatchEntry *e, int64_t ldw,
                           int64_t ldo, int64_t m, int n_out_, int k_, const LinearEpilogue &epi, hipStream_t s_) -> int {
    if (w3cfg >= 0 && relu_in == 0) return launch_linear_w3(a_, lda_, nbatch, e, n_out_, ldo, m, n_out_, k_, epi, s_, w3cfg);
    return gs::launch_linear(a_, lda_, relu_in, nbatch, e, ldw, ldo, m, n_out_, k_, epi, s_);
  };
  if (d->training && defer_apply) {
    GS_REQUIRE(rows >= 2, GNNSAFT_ERR_SHAPE);
    ent.out = y_tmp;
    LinearEpilogue epi;
    epi.stats = stats;
    {
      ProfScope ps(prof, GNNSAFT_PROF_LIN, st);
      GS_TRY(launch_linear(a, lda, 0, 1, &ent, k, n_out, rows, n_out, k, epi, st));
    }
    return launch_bn_stats_close(stats, rows, n_out, bn.gamma, bn.beta, bn.rmean, bn.rvar, bn.nbt, d->bn_momentum,
                                 d->bn_eps, reinterpret_cast<float *>(ws + p.scale),
                                 reinterpret_cast<float *>(ws + p.shift), save_stat,
                                 reinterpret_cast<double *>(ws + p.bn_tail_seg),
                                 reinterpret_cast<int32_t *>(ws + p.rd_sync) + kRdSyncInts, st);
  }
  if (d->training) {
    GS_REQUIRE(rows >= 2, GNNSAFT_ERR_SHAPE);
    ent.out = y_tmp;
    LinearEpilogue epi;
    epi.stats = stats;
    {
      ProfScope ps(prof, GNNSAFT_PROF_LIN, st);
      GS_TRY(launch_linear(a, lda, 0, 1, &ent, k, n_out, rows, n_out, k, epi, st));
    }
    GS_TRY(gnnsaft_bn_train_apply(stats, y_tmp, rows, n_out, bn.gamma, bn.beta, bn.rmean, bn.rvar, bn.nbt,
                                  d->bn_momentum, d->bn_eps, residual, out, save_stat, ws + p.bnseg, p.bnseg_bytes, st));
  } else if (d->save_tape) {
    // eval mode WITH a tape (fine-tuning with frozen statistics): the pre-activation is kept, as in training
    ent.out = y_tmp;
    LinearEpilogue epi;
    {
      ProfScope ps(prof, GNNSAFT_PROF_LIN, st);
      GS_TRY(launch_linear(a, lda, 0, 1, &ent, k, n_out, rows, n_out, k, epi, st));
    }
    GS_TRY(launch_bn_eval_apply(y_tmp, rows, n_out, bn.gamma, bn.beta, bn.rmean, bn.rvar, d->bn_eps, residual, out,
                                save_stat, st));
  } else {
    // running statistics: the GEMM epilogue forms scale / shift from (gamma, beta, mean, var) itself
    ent.out = out;
    LinearEpilogue epi;
    epi.scale = bn.gamma;
    epi.shift = bn.beta;
    epi.bn_mean = bn.rmean;
    epi.bn_var = bn.rvar;
    epi.bn_eps = d->bn_eps;
    epi.relu_out = 1;
    epi.residual = residual;
    epi.ldr = n_out;
    ProfScope ps(prof, GNNSAFT_PROF_LIN, st);
    GS_TRY(launch_linear(a, lda, 0, 1, &ent, k, n_out, rows, n_out, k, epi, st));
  }
  return GNNSAFT_OK;
}

// The node state a layer reads: materialised (`x`), or still pending as the previous layer's pre-activation `y`
// with its BatchNorm (scale, shift) and residual -- then the message GEMM applies it on load and writes it to `xdst`.
struct NodeState {
  const float *x = nullptr;
  const float *y = nullptr, *xres = nullptr, *scale = nullptr, *shift = nullptr;
  float *xdst = nullptr;
};

static int node_terms(const NodeState &ns, int64_t n, int h, const float *w0, const float *w1, float *pq,
                      hipStream_t st) {
  GemmBatchEntry e[4] = {{w0, nullptr, pq, 0}, {w1, nullptr, pq + h, 0}, {w0 + h, nullptr, pq + 2 * h, 0},
                         {w1 + h, nullptr, pq + 3 * h, 0}};
  if (ns.y != nullptr)
    return launch_linear_bnres(ns.y, ns.xres, ns.scale, ns.shift, ns.xdst, 4, e, 3 * (int64_t)h, 4 * (int64_t)h, n, h, h,
                               st);
  LinearEpilogue epi;
  return launch_linear(ns.x, h, 0, 4, e, 3 * (int64_t)h, 4 * (int64_t)h, n, h, h, epi, st);
}

static int src_terms(const NodeState &ns, int64_t n, int h, const float *w0, const float *w1, float *q,
                     hipStream_t st, const char *w3 = nullptr /* image of [W0_src ; W1_src] = [2H, H] */) {
  const int w3cfg = w3 != nullptr && ns.y == nullptr ? w3_pick_cfg(n, 2 * h, h, false) : -1;
  if (w3cfg >= 0) {   // both towers as ONE GEMM over the stacked image: A is split once per 256 output columns
    GemmBatchEntry e1{nullptr, nullptr, q, 0, w3};
    LinearEpilogue epi;
    return launch_linear_w3(ns.x, h, 1, &e1, 2 * h, 2 * (int64_t)h, n, 2 * h, h, epi, st, w3cfg);
  }
  GemmBatchEntry e[2] = {{w0 + h, nullptr, q, 0}, {w1 + h, nullptr, q + h, 0}};
  if (ns.y != nullptr)
    return launch_linear_bnres(ns.y, ns.xres, ns.scale, ns.shift, ns.xdst, 2, e, 3 * (int64_t)h, 2 * (int64_t)h, n, h, h,
                               st);
  LinearEpilogue epi;
  return launch_linear(ns.x, h, 0, 2, e, 3 * (int64_t)h, 2 * (int64_t)h, n, h, h, epi, st);
}

static int edge_table(const float *cemb, int64_t combos, int h, const float *we, const float *be, const float *w0,
                      const float *b0, const float *w1, const float *b1, float *cenc, float *rtab, hipStream_t st) {
  LinearEpilogue epi;
  GemmBatchEntry e1{we, be, cenc, 0};
  GS_TRY(launch_linear(cemb, h, 0, 1, &e1, h, h, combos, h, h, epi, st));
  GemmBatchEntry e2[2] = {{w0 + 2 * h, b0, rtab, 0}, {w1 + 2 * h, b1, rtab + h, 0}};
  return launch_linear(cenc, h, 0, 2, e2, 3 * (int64_t)h, 2 * (int64_t)h, combos, h, h, epi, st);
}

}  // namespace gs

using namespace gs;

extern "C" int gnnsaft_abi_version(void) { return GNNSAFT_ABI_VERSION; }

extern "C" const char *gnnsaft_error_string(int code) {
  if (code == GNNSAFT_OK) return "ok";
  if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
  switch (code) {
    case GNNSAFT_ERR_SHAPE: return "unsupported or inconsistent sizes";
    case GNNSAFT_ERR_WORKSPACE: return "workspace too small";
    case GNNSAFT_ERR_NULL: return "required pointer is NULL";
    case GNNSAFT_ERR_UNSUPPORTED: return "configuration outside the supported shape envelope";
    default: return "unknown gnnsaft error";
  }
}

extern "C" int32_t gnnsaft_bn_rows_per_group(void) { return kBnRowsPerGroup; }

extern "C" int gnnsaft_linear(const float *a, int64_t lda, int32_t relu_in, const float *w, int64_t ldw,
                              const float *bias, float *out, int64_t ldo, int64_t m, int32_t n_out, int32_t k,
                              const float *scale, const float *shift, int32_t relu_out, const float *residual,
                              int64_t ldr, float *stats, gnnsaft_stream_t stream) {
  GemmBatchEntry ent{w, bias, out, 0};
  LinearEpilogue epi;
  epi.scale = scale;
  epi.shift = shift;
  epi.relu_out = relu_out;
  epi.residual = residual;
  epi.ldr = ldr;
  epi.stats = stats;
  GS_REQUIRE(stats == nullptr || ldo == n_out, GNNSAFT_ERR_SHAPE);
  return launch_linear(a, lda, relu_in, 1, &ent, ldw, ldo, m, n_out, k, epi, static_cast<hipStream_t>(stream));
}

extern "C" int gnnsaft_debug_linear_tile(const float *a, int64_t lda, const float *w, int64_t ldw, const float *bias,
                                         float *out, int64_t ldo, int64_t m, int32_t n_out, int32_t k, float *stats,
                                         int32_t tile_config, gnnsaft_stream_t stream) {
  GemmBatchEntry ent{w, bias, out, 0};
  LinearEpilogue epi;
  epi.stats = stats;
  GS_REQUIRE(stats == nullptr || ldo == n_out, GNNSAFT_ERR_SHAPE);
  GS_REQUIRE(tile_config >= 0, GNNSAFT_ERR_SHAPE);
  return launch_linear(a, lda, 0, 1, &ent, ldw, ldo, m, n_out, k, epi, static_cast<hipStream_t>(stream), tile_config);
}

extern "C" int gnnsaft_pna_node_terms(const float *x, int64_t num_nodes, int32_t hidden, const float *w_pre0,
                                      const float *w_pre1, float *pq, gnnsaft_stream_t stream) {
  GS_REQUIRE(x && w_pre0 && w_pre1 && pq, GNNSAFT_ERR_NULL);
  GS_REQUIRE(hidden >= 32 && (hidden % 32) == 0, GNNSAFT_ERR_SHAPE);
  NodeState ns;
  ns.x = x;
  return node_terms(ns, num_nodes, hidden, w_pre0, w_pre1, pq, static_cast<hipStream_t>(stream));
}

extern "C" int gnnsaft_pna_src_terms(const float *x, int64_t num_nodes, int32_t hidden, const float *w_pre0,
                                     const float *w_pre1, float *q, gnnsaft_stream_t stream) {
  GS_REQUIRE(x && w_pre0 && w_pre1 && q, GNNSAFT_ERR_NULL);
  GS_REQUIRE(hidden >= 32 && (hidden % 32) == 0, GNNSAFT_ERR_SHAPE);
  NodeState ns;
  ns.x = x;
  return src_terms(ns, num_nodes, hidden, w_pre0, w_pre1, q, static_cast<hipStream_t>(stream));
}

extern "C" int gnnsaft_pna_edge_table(const float *combo_emb, int32_t num_combos, int32_t hidden,
                                      const float *w_edge, const float *b_edge, const float *w_pre0,
                                      const float *b_pre0, const float *w_pre1, const float *b_pre1, float *enc_tmp,
                                      float *rtab, gnnsaft_stream_t stream) {
  GS_REQUIRE(combo_emb && w_edge && b_edge && w_pre0 && b_pre0 && w_pre1 && b_pre1 && enc_tmp && rtab,
             GNNSAFT_ERR_NULL);
  GS_REQUIRE(hidden >= 32 && (hidden % 32) == 0 && num_combos >= 1, GNNSAFT_ERR_SHAPE);
  return edge_table(combo_emb, num_combos, hidden, w_edge, b_edge, w_pre0, b_pre0, w_pre1, b_pre1, enc_tmp, rtab,
                    static_cast<hipStream_t>(stream));
}

extern "C" int gnnsaft_pna_edge_mlp(const int32_t *src, const int32_t *dst, const int32_t *combo, int64_t num_rows,
                                    int32_t hidden, const float *pq, const float *rtab, const float *w2_t0,
                                    const float *b2_t0, const float *w2_t1, const float *b2_t1, float *msgs,
                                    gnnsaft_stream_t stream) {
  GS_REQUIRE(w2_t0 && b2_t0 && w2_t1 && b2_t1 && msgs, GNNSAFT_ERR_NULL);
  GemmBatchEntry e[2] = {{w2_t0, b2_t0, msgs, 0}, {w2_t1, b2_t1, msgs + hidden, 0}};
  return launch_pna_edge_mlp(src, dst, combo, num_rows, hidden, pq, rtab, e, 2 * (int64_t)hidden,
                             static_cast<hipStream_t>(stream));
}

extern "C" int gnnsaft_pna_update(const float *x, const float *agg, const float *log_amp, const float *log_att,
                                  const float *avg_deg_log, int64_t num_nodes, int32_t hidden, const float *w_post0,
                                  const float *b_post0, const float *w_post1, const float *b_post1, float *u,
                                  gnnsaft_stream_t stream) {
  GS_REQUIRE(w_post0 && b_post0 && w_post1 && b_post1 && u, GNNSAFT_ERR_NULL);
  GemmBatchEntry e[2] = {{w_post0, b_post0, u, 0}, {w_post1, b_post1, u + hidden / 2, 0}};
  return launch_pna_update(x, agg, log_amp, log_att, avg_deg_log, num_nodes, hidden, e, hidden,
                           static_cast<hipStream_t>(stream));
}

extern "C" int gnnsaft_profile_create(int32_t capacity, uint32_t mask, gnnsaft_profile **out) {
  GS_REQUIRE(out != nullptr, GNNSAFT_ERR_NULL);
  GS_REQUIRE(capacity >= 1 && capacity <= (1 << 20), GNNSAFT_ERR_SHAPE);
  gnnsaft_profile *p = new gnnsaft_profile();
  p->mask = mask;
  p->start.resize(capacity);
  p->stop.resize(capacity);
  p->kind.assign(capacity, 0);
  for (int i = 0; i < capacity; ++i) {
    hipError_t e1 = hipEventCreate(&p->start[i]);
    hipError_t e2 = hipEventCreate(&p->stop[i]);
    if (e1 != hipSuccess || e2 != hipSuccess) {
      delete p;
      return (int)(e1 != hipSuccess ? e1 : e2);
    }
  }
  *out = p;
  return GNNSAFT_OK;
}

extern "C" void gnnsaft_profile_destroy(gnnsaft_profile *p) {
  if (p == nullptr) return;
  for (size_t i = 0; i < p->start.size(); ++i) {
    (void)hipEventDestroy(p->start[i]);
    (void)hipEventDestroy(p->stop[i]);
  }
  delete p;
}

extern "C" int gnnsaft_profile_reset(gnnsaft_profile *p) {
  GS_REQUIRE(p != nullptr, GNNSAFT_ERR_NULL);
  p->used = 0;
  return GNNSAFT_OK;
}

extern "C" int gnnsaft_profile_summary(gnnsaft_profile *p, uint32_t kernel_bit, int32_t *count, float *total_ms) {
  GS_REQUIRE(p && count && total_ms, GNNSAFT_ERR_NULL);
  int c = 0;
  double tot = 0.0;
  for (int i = 0; i < p->used; ++i) {
    if (p->kind[i] != kernel_bit) continue;
    hipError_t e = hipEventSynchronize(p->stop[i]);
    if (e != hipSuccess) return (int)e;
    float ms = 0.f;
    e = hipEventElapsedTime(&ms, p->start[i], p->stop[i]);
    if (e != hipSuccess) return (int)e;
    tot += ms;
    ++c;
  }
  *count = c;
  *total_ms = (float)tot;
  return GNNSAFT_OK;
}

extern "C" int32_t gnnsaft_num_weights(const gnnsaft_model_desc *d) {
  if (d == nullptr) return -1;
  const int per_layer = 1 + 2 + 4 * d->pre_layers + 4 * d->post_layers + 2 + 5;
  return d->num_atom_cols + d->num_bond_cols + d->num_layers * per_layer + d->num_mlp_layers * 7 + (2 + 5 + 2 + 5 + 2);
}

extern "C" size_t gnnsaft_forward_workspace_bytes(const gnnsaft_model_desc *desc, int64_t num_nodes,
                                                  int64_t num_edges, int64_t num_graphs) {
  Plan p;
  if (make_plan(desc, num_nodes, num_edges, num_graphs, p) != GNNSAFT_OK) return 0;
  return p.total;
}

extern "C" int gnnsaft_forward_workspace_map(const gnnsaft_model_desc *desc, int64_t num_nodes, int64_t num_edges,
                                             int64_t num_graphs, gnnsaft_workspace_map *map) {
  GS_REQUIRE(map != nullptr, GNNSAFT_ERR_NULL);
  Plan p;
  GS_TRY(make_plan(desc, num_nodes, num_edges, num_graphs, p));
  map->rowptr = p.rowptr;
  map->src = p.src;
  map->dst = p.dst;
  map->combo = p.combo;
  map->log_amp = p.log_amp;
  map->log_att = p.log_att;
  map->graph_ptr = p.graph_ptr;
  map->x_embed = p.x0;
  map->x_final = desc->save_tape ? p.x0 + (size_t)desc->num_layers * (size_t)p.sx * 4
                                 : ((desc->num_layers % 2) ? p.x1 : p.x0);
  map->pq = p.pq;
  map->agg = p.agg;
  map->u = p.u0;
  map->y = p.y;
  map->rtab = p.rtab;
  map->pooled = p.pooled;
  map->total = p.total;
  map->ro = p.ro;
  map->x_stride = (size_t)p.sx * 4;
  map->bnstat = p.bnstat;
  map->y_stride = (size_t)p.sy * 4;
  map->ry = p.ry;
  map->rstat = p.rstat;
  return GNNSAFT_OK;
}

static int gs_forward_impl(const gnnsaft_model_desc *d, const void *const *weights_host, int32_t num_weights,
                           const int64_t *x_idx, const int64_t *edge_index, const int64_t *edge_attr,
                           const int64_t *batch, int64_t num_nodes, int64_t num_edges, int64_t num_graphs,
                           const float *target, float *out, float *loss3, int32_t *err_flag, void *workspace,
                           size_t workspace_bytes, const void *structure_in, void *structure_out,
                           gnnsaft_profile *prof, gnnsaft_aux *aux, gnnsaft_stream_t stream);

extern "C" int gnnsaft_forward(const gnnsaft_model_desc *d, const void *const *weights_host, int32_t num_weights,
                               const int64_t *x_idx, const int64_t *edge_index, const int64_t *edge_attr,
                               const int64_t *batch, int64_t num_nodes, int64_t num_edges, int64_t num_graphs,
                               const float *target, float *out, float *loss3, int32_t *err_flag, void *workspace,
                               size_t workspace_bytes, const void *structure, gnnsaft_profile *prof,
                               gnnsaft_aux *aux, gnnsaft_stream_t stream) {
  return gs_forward_impl(d, weights_host, num_weights, x_idx, edge_index, edge_attr, batch, num_nodes, num_edges,
                         num_graphs, target, out, loss3, err_flag, workspace, workspace_bytes, structure, nullptr, prof,
                         aux, stream);
}

extern "C" size_t gnnsaft_structure_bytes(const gnnsaft_model_desc *desc, int64_t num_nodes, int64_t num_edges,
                                          int64_t num_graphs) {
  Plan p;
  if (make_plan(desc, num_nodes, num_edges, num_graphs, p) != GNNSAFT_OK) return 0;
  return p.struct_bytes;
}

extern "C" int gnnsaft_structure_build(const gnnsaft_model_desc *d, const int64_t *edge_index, const int64_t *edge_attr,
                                       const int64_t *batch, int64_t num_nodes, int64_t num_edges, int64_t num_graphs,
                                       void *structure_out, int32_t *err_flag, void *workspace, size_t workspace_bytes,
                                       gnnsaft_stream_t stream) {
  GS_REQUIRE(d && structure_out && workspace, GNNSAFT_ERR_NULL);
  return gs_forward_impl(d, nullptr, 0, nullptr, edge_index, edge_attr, batch, num_nodes, num_edges, num_graphs,
                         nullptr, nullptr, nullptr, err_flag, workspace, workspace_bytes, nullptr, structure_out,
                         nullptr, nullptr, stream);
}

// structure_in: skip the K0 chain, copy the cached segment into the workspace instead.
// structure_out (gnnsaft_structure_build): run ONLY the K0 chain and copy the segment out.
static int gs_forward_impl(const gnnsaft_model_desc *d, const void *const *weights_host, int32_t num_weights,
                           const int64_t *x_idx, const int64_t *edge_index, const int64_t *edge_attr,
                           const int64_t *batch, int64_t num_nodes, int64_t num_edges, int64_t num_graphs,
                           const float *target, float *out, float *loss3, int32_t *err_flag, void *workspace,
                           size_t workspace_bytes, const void *structure_in, void *structure_out,
                           gnnsaft_profile *prof, gnnsaft_aux *aux, gnnsaft_stream_t stream) {
  const bool structure_only = structure_out != nullptr;
  GS_REQUIRE(d && workspace && (structure_only || (weights_host && out)), GNNSAFT_ERR_NULL);
  if (structure_only) {
    GS_REQUIRE(num_nodes >= 1 && num_graphs >= 1, GNNSAFT_ERR_SHAPE);
    GS_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, GNNSAFT_ERR_WORKSPACE);
    Plan p;
    GS_TRY(make_plan(d, num_nodes, num_edges, num_graphs, p));
    GS_REQUIRE(workspace_bytes >= p.total, GNNSAFT_ERR_WORKSPACE);
    GS_REQUIRE(batch != nullptr || num_graphs == 1, GNNSAFT_ERR_SHAPE);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace);
    auto I = [&](size_t off) { return reinterpret_cast<int32_t *>(ws + off); };
    auto F = [&](size_t off) { return reinterpret_cast<float *>(ws + off); };
    GS_TRY(launch_csr_build(edge_index, edge_attr, num_nodes, num_edges, d->num_bond_cols, d->bond_dims, d->self_loops,
                            I(p.rowptr), I(p.src), I(p.dst), I(p.combo), F(p.log_amp), F(p.log_att), err_flag,
                            ws + p.csr_ws, gnnsaft_csr_workspace_bytes(num_nodes, num_edges), batch, num_graphs,
                            I(p.graph_ptr), I(p.hist3) + 2 * kDegreeBuckets, false, st,
                            d->fold_degree_scalers != 0));
    GS_TRY(launch_degree_tiles(I(p.rowptr), num_nodes, d->hidden, I(p.perm), I(p.tiles), I(p.num_tiles), I(p.hist3),
                               err_flag, true, st));
    GS_HIP(hipMemcpyAsync(structure_out, ws + p.struct_begin, p.struct_bytes, hipMemcpyDeviceToDevice, st));
    return GNNSAFT_OK;
  }
  GS_REQUIRE(num_weights == gnnsaft_num_weights(d), GNNSAFT_ERR_SHAPE);
  GS_REQUIRE(num_nodes >= 1 && num_graphs >= 1 && x_idx != nullptr, GNNSAFT_ERR_SHAPE);
  GS_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, GNNSAFT_ERR_WORKSPACE);
  Plan p;
  GS_TRY(make_plan(d, num_nodes, num_edges, num_graphs, p));
  GS_REQUIRE(workspace_bytes >= p.total, GNNSAFT_ERR_WORKSPACE);
  if (d->training) GS_REQUIRE(num_nodes >= 2 && num_graphs >= 2, GNNSAFT_ERR_SHAPE);
  hipStream_t st = static_cast<hipStream_t>(stream);
  char *ws = static_cast<char *>(workspace);
  const int h = d->hidden;
  const int64_t n = num_nodes, g = num_graphs;
  auto F = [&](size_t off) { return reinterpret_cast<float *>(ws + off); };
  auto I = [&](size_t off) { return reinterpret_cast<int32_t *>(ws + off); };
  auto D = [&](size_t off) { return reinterpret_cast<double *>(ws + off); };

  WeightCursor wc{weights_host, num_weights};
  const float *atom_tab[GNNSAFT_MAX_TABLES], *bond_tab[GNNSAFT_MAX_TABLES];
  for (int k = 0; k < d->num_atom_cols; ++k) atom_tab[k] = wc.f();
  for (int k = 0; k < d->num_bond_cols; ++k) bond_tab[k] = wc.f();
  GS_REQUIRE(wc.ok, GNNSAFT_ERR_NULL);

  // ---- parse every layer's weights, then do the x-independent prologue work of ALL layers in a
  //      few batched launches: edge-class tables (edge_encoder + pre_nns[t][0] edge block) and the
  //      degree-folded update weights
  struct LayerW {
    const float *avg, *we, *be, *wlin, *blin;
    const float *wpre[2][8], *bpre[2][8], *wpost[2][8], *bpost[2][8];
    BnPtrs bn;
  };
  GS_REQUIRE(d->pre_layers <= 8 && d->post_layers <= 8, GNNSAFT_ERR_UNSUPPORTED);
  std::vector<LayerW> lw(d->num_layers);
  for (int l = 0; l < d->num_layers; ++l) {
    LayerW &w = lw[l];
    w.avg = wc.f();
    w.we = wc.f();
    w.be = wc.f();
    for (int t = 0; t < 2; ++t)
      for (int j = 0; j < d->pre_layers; ++j) {
        w.wpre[t][j] = wc.f();
        w.bpre[t][j] = wc.f();
      }
    for (int t = 0; t < 2; ++t)
      for (int j = 0; j < d->post_layers; ++j) {
        w.wpost[t][j] = wc.f();
        w.bpost[t][j] = wc.f();
      }
    w.wlin = wc.f();
    w.blin = wc.f();
    w.bn = wc.bn();
    GS_REQUIRE(wc.ok, GNNSAFT_ERR_NULL);
  }
  // destination-term fold: only with the degree-folded update and a purely linear message (pre_layers == 1)
  const bool fold_dst = d->fold_dst_term && d->fold_degree_scalers && d->pre_layers == 1 && (h % 64) == 0;
  const int64_t cstride = p.combos * (int64_t)h;          // floats per layer in cenc
  const int64_t rstride = p.combos * (int64_t)(2 * h);    // floats per layer in rtab
  const int64_t wstride = (int64_t)kDegreeBuckets * 5 * h * h;  // floats per layer in weff
  // ---- K0 structure chain, on the side stream when the caller lends one: destination-term fold (weights only),
  //      CSR, graph offsets, degree tiles, degree-folded update weights
  // The four independent first jobs (atom embedding sum, bond-class embedding table, zeroing of the CSR
  // histogram, destination-term weight fold) share ONE launch on the caller's stream; then the chains fork.
  const bool dst_in_prologue = fold_dst && d->num_layers <= GNNSAFT_MAX_FOLD_LAYERS;
  // K0 as cooperating workgroups of the prologue launch: needs the caller's persistent barrier words behind err_flag
  // (gnnsaft_model_desc.persistent_sync_words), the bounded in-degree of the folded update and one stream
  // (at most kK0MaxWgs cooperating workgroups build the structure: beside 20 k nodes' worth of embedding work that
  // hides four launch boundaries -- C2: -14 us; at C3's 164 k nodes and 492 k edges the chain becomes the launch's
  // critical path: prologue 165 -> 255 us against 61 us of chain launches (profiles/r02_ / r04_c3_kernel_stats.csv),
  // so from 64 k nodes up the structure is built by the launches)
  const bool k0_fused = structure_in == nullptr && aux == nullptr && d->fold_degree_scalers != 0 && n < 65536 &&
                        d->num_layers >= 1 && d->num_layers <= GNNSAFT_MAX_FOLD_LAYERS && err_flag != nullptr &&
                        d->persistent_sync_words >= GNNSAFT_K0_SYNC_WORDS + n;
  // the edge-class tables (cenc, rtab: weights only) as workgroups of the same launch instead of two small GEMMs
  const bool tables_in_prologue = d->num_layers >= 1 && d->num_layers <= GNNSAFT_MAX_FOLD_LAYERS && h <= 256 &&
                                  (h % 16) == 0 && p.combos <= 4096;
  {
    const float *w0[GNNSAFT_MAX_FOLD_LAYERS], *w1[GNNSAFT_MAX_FOLD_LAYERS];
    const float *p0[GNNSAFT_MAX_FOLD_LAYERS], *p1[GNNSAFT_MAX_FOLD_LAYERS];
    for (int i = 0; i < GNNSAFT_MAX_FOLD_LAYERS && i < d->num_layers; ++i) {
      w0[i] = lw[i].wpost[0][0];
      w1[i] = lw[i].wpost[1][0];
      p0[i] = lw[i].wpre[0][0];
      p1[i] = lw[i].wpre[1][0];
    }
    int32_t *zero_ptr = nullptr;
    int64_t zero_count = 0;
    if (structure_in == nullptr && !k0_fused) csr_zero_region(ws + p.csr_ws, n, &zero_ptr, &zero_count);
    K0ChainArgs k0;
    if (k0_fused) {
      // the batch structure by cooperating workgroups at the front of this launch (elementwise.hip: k0_chain_body)
      GS_REQUIRE(batch != nullptr || g == 1, GNNSAFT_ERR_SHAPE);
      GS_REQUIRE(n + num_edges < ((int64_t)1 << 31) - 1, GNNSAFT_ERR_SHAPE);
      const int64_t want = gs_ceil_div(n, (int64_t)kK0Group);
      k0.wgs = (int)(want < kK0MaxWgs ? want : kK0MaxWgs);
      k0.self_loops = d->self_loops ? 1 : 0;
      k0.tile_rows = pna_fold_tile_rows(h);
      k0.barrier_extra = d->debug_barrier_extra;
      k0.bd.n = d->num_bond_cols;
      for (int k = 0; k < GNNSAFT_MAX_TABLES; ++k) k0.bd.dims[k] = k < d->num_bond_cols ? d->bond_dims[k] : 1;
      k0.edge_index = edge_index;
      k0.edge_attr = edge_attr;
      k0.batch = batch;
      k0.n = n;
      k0.e = num_edges;
      k0.g = g;
      k0.graph_ptr = I(p.graph_ptr);
      int32_t *unused_cursor, *unused_sums;
      csr_workspace_parts(ws + p.csr_ws, n, &unused_cursor, &unused_sums, &k0.slots);
      k0.lookback = reinterpret_cast<unsigned long long *>(ws + p.csr_ws);   // the general chain's in-degree counts
      k0.rowptr = I(p.rowptr);
      k0.src = I(p.src);
      k0.dst = I(p.dst);
      k0.combo = I(p.combo);
      k0.log_amp = F(p.log_amp);
      k0.log_att = F(p.log_att);
      k0.hist = I(p.hist3);
      k0.start = I(p.hist3) + kDegreeBuckets;
      k0.group_hist = I(p.hist3) + 2 * kDegreeBuckets;
      k0.tiles = I(p.tiles);
      k0.num_tiles = I(p.num_tiles);
      k0.sync = err_flag + 1;
      k0.cursor = err_flag + 1 + GNNSAFT_K0_SYNC_WORDS;
      k0.err = err_flag;
    }
    EdgeTableLayers et;
    for (int i = 0; i < GNNSAFT_MAX_FOLD_LAYERS; ++i) {
      et.we[i] = et.be[i] = et.wpre0[i] = et.wpre1[i] = et.bpre0[i] = et.bpre1[i] = nullptr;
      if (!tables_in_prologue) continue;
      const LayerW &lwi = lw[i < d->num_layers ? i : 0];
      et.we[i] = lwi.we;
      et.be[i] = lwi.be;
      et.wpre0[i] = lwi.wpre[0][0];
      et.wpre1[i] = lwi.wpre[1][0];
      et.bpre0[i] = lwi.bpre[0][0];
      et.bpre1[i] = lwi.bpre[1][0];
    }
    GS_TRY(launch_forward_prologue(x_idx, n, d->num_atom_cols, atom_tab, d->atom_dims, d->num_bond_cols, bond_tab,
                                   d->bond_dims, h, F(p.x0), F(p.cemb), zero_ptr, zero_count,
                                   dst_in_prologue ? d->num_layers : 0, w0, w1, p0, p1, D(p.gfold), err_flag, st,
                                   I(p.rd_sync), kRdSyncInts + kBnTailCounterInts, &et, tables_in_prologue ? d->num_layers : 0, F(p.cenc),
                                   F(p.rtab), k0_fused ? &k0 : nullptr));
  }
  // ---- W3 images (w3.hpp) of the weights the big GEMMs read: the source block of pre_nns[t][0] of both towers
  //      stacked [2H, H] and lin [H, H], every layer, one launch; the folded update weights get theirs from the fold
  const bool w3_src = p.w3src_stride != 0 && fold_dst && w3_pick_cfg(n, 2 * h, h, false) >= 0;
  const bool w3_lin = p.w3lin_stride != 0 && w3_pick_cfg(n, h, h, d->training != 0) >= 0;
  const bool w3_upd_gemm = p.w3eff_stride != 0 && d->fold_degree_scalers && w3_cfg_for_update(h, n) >= 0;
  // the no-tape forward's fused aggregation + update (update_agg.hip): the aggregates never reach HBM.  Not while the
  // aggregation kernel itself is being timed (bench.py's roofline events around k_pna_aggregate), not with a tape (the
  // backward reads the aggregates), not for the explicit destination term.  GNNSAFT_FUSED_AGG = 0 / 1 forces the two
  // launches / the fused launch; unset: fused from 64 k nodes up (measured: C3 -6 % per step; at C2's 20 k nodes the
  // two launches are 10 us per layer faster -- profiles/r04_*)
  static const int fused_agg_env = [] {
    const char *e = getenv("GNNSAFT_FUSED_AGG");
    return e == nullptr ? -1 : (e[0] != '0' ? 1 : 0);
  }();
  const bool fused_agg_on = fused_agg_env < 0 ? n >= 65536 : fused_agg_env == 1;
  const bool fuse_agg = fused_agg_on && p.w3eff_stride != 0 && d->fold_degree_scalers && fold_dst && !d->save_tape &&
                        update_agg_supported(h, (int)p.combos) && !(prof != nullptr && (prof->mask & GNNSAFT_PROF_AGGREGATE) != 0);
  // below 64 k nodes, opt-in (GNNSAFT_AR_UPDATE = 1): the folded update on k_gemm_ar with both towers in one workgroup
  // per degree tile (gemm_ar.hip).  Measured at C2 in the replayed step: 39.1 us against 36.1 us for
  // k_gemm_f32<PostFoldA, X6> (step 0.383 against 0.375 ms) although one tower alone ran 19.0 against 21.4 us in the
  // microbenchmark -- 320 workgroups of 20 dependent stages are a longer latency chain than 640 of 40 shorter ones
  static const int ar_upd_env = [] {
    const char *e = getenv("GNNSAFT_AR_UPDATE");
    return e != nullptr && e[0] == '1' ? 1 : 0;
  }();
  const bool ar_upd = ar_upd_env == 1 && !w3_upd_gemm && !fuse_agg && p.w3eff_stride != 0 && d->fold_degree_scalers &&
                      n < 65536 && ar_update_supported(h);
  const bool w3_upd = w3_upd_gemm || fuse_agg || ar_upd;   // the fold leaves W3 images of the folded weights
  if (w3_src || w3_lin) {
    std::vector<W3PackItem> items;
    for (int l = 0; l < d->num_layers; ++l) {
      char *is = ws + p.w3src + (size_t)l * p.w3src_stride, *il = ws + p.w3lin + (size_t)l * p.w3lin_stride;
      if (w3_src) {
        items.push_back(W3PackItem{lw[l].wpre[0][0] + h, is, 3 * (int64_t)h, h, h, 2 * h, 0});
        items.push_back(W3PackItem{lw[l].wpre[1][0] + h, is, 3 * (int64_t)h, h, h, 2 * h, h});
      }
      if (w3_lin) items.push_back(W3PackItem{lw[l].wlin, il, h, h, h, h, 0});
    }
    GS_TRY(launch_w3_pack((int)items.size(), items.data(), st));
  }
  hipStream_t sa = st;
  if (aux != nullptr) {
    sa = aux->stream;
    GS_HIP(hipEventRecord(aux->fork, st));
    GS_HIP(hipStreamWaitEvent(sa, aux->fork, 0));
  }
  auto fold_weights = [&](int phases) -> int {
    for (int l0 = 0; l0 < d->num_layers; l0 += GNNSAFT_MAX_FOLD_LAYERS) {
      const int nl = d->num_layers - l0 < GNNSAFT_MAX_FOLD_LAYERS ? d->num_layers - l0 : GNNSAFT_MAX_FOLD_LAYERS;
      const float *w0[GNNSAFT_MAX_FOLD_LAYERS], *w1[GNNSAFT_MAX_FOLD_LAYERS], *av[GNNSAFT_MAX_FOLD_LAYERS];
      const float *p0[GNNSAFT_MAX_FOLD_LAYERS], *p1[GNNSAFT_MAX_FOLD_LAYERS];
      for (int i = 0; i < nl; ++i) {
        w0[i] = lw[l0 + i].wpost[0][0];
        w1[i] = lw[l0 + i].wpost[1][0];
        av[i] = lw[l0 + i].avg;
        p0[i] = lw[l0 + i].wpre[0][0];
        p1[i] = lw[l0 + i].wpre[1][0];
      }
      GS_TRY(launch_fold_post_weights(nl, w0, w1, av, fold_dst ? p0 : nullptr, fold_dst ? p1 : nullptr,
                                      fold_dst ? D(p.gfold) + (int64_t)l0 * 6 * (h / 2) * h : nullptr, I(p.hist3), h,
                                      F(p.weff) + l0 * wstride, wstride, phases, sa,
                                      w3_upd ? ws + p.w3eff + (size_t)l0 * p.w3eff_stride : nullptr));
    }
    return GNNSAFT_OK;
  };
  if (d->fold_degree_scalers && fold_dst && !dst_in_prologue) GS_TRY(fold_weights(1));
  if (structure_in != nullptr) {
    // cached batch structure (gnnsaft_structure_build): one device copy instead of the K0 chain; the update-weight
    // fold (needs the cached degree histogram and the CURRENT weights) runs on its own
    GS_HIP(hipMemcpyAsync(ws + p.struct_begin, structure_in, p.struct_bytes, hipMemcpyDeviceToDevice, sa));
    if (d->fold_degree_scalers) GS_TRY(fold_weights(2));
  } else {
  // CSR + graph offsets (batch == NULL: un-batched Data, models.py:116, one graph spanning all nodes) + the first
  // pass of the degree bucketing, in 5 launches (the histogram was zeroed by the prologue)
  GS_REQUIRE(batch != nullptr || g == 1, GNNSAFT_ERR_SHAPE);
  if (!k0_fused)
    GS_TRY(launch_csr_build(edge_index, edge_attr, n, num_edges, d->num_bond_cols, d->bond_dims, d->self_loops,
                            I(p.rowptr), I(p.src), I(p.dst), I(p.combo), F(p.log_amp), F(p.log_att), err_flag,
                            ws + p.csr_ws, gnnsaft_csr_workspace_bytes(n, num_edges), batch, g, I(p.graph_ptr),
                            d->fold_degree_scalers ? I(p.hist3) + 2 * kDegreeBuckets : nullptr, true, sa,
                            d->fold_degree_scalers != 0));
  if (d->fold_degree_scalers) {
    if (d->num_layers <= GNNSAFT_MAX_FOLD_LAYERS) {  // weight fold rides along with the permutation fill
      const float *w0[GNNSAFT_MAX_FOLD_LAYERS], *w1[GNNSAFT_MAX_FOLD_LAYERS], *av[GNNSAFT_MAX_FOLD_LAYERS];
      for (int i = 0; i < d->num_layers; ++i) {
        w0[i] = lw[i].wpost[0][0];
        w1[i] = lw[i].wpost[1][0];
        av[i] = lw[i].avg;
      }
      const DegreeFoldRequest req{d->num_layers, w0, w1, av, fold_dst ? D(p.gfold) : nullptr, F(p.weff), wstride,
                                  w3_upd ? ws + p.w3eff : nullptr};
      const K0Installed inst{I(p.rowptr), I(p.src), I(p.dst), I(p.combo), p.ep, F(p.log_amp), F(p.log_att),
                             k0_fused ? err_flag + 1 : nullptr, k0_fused ? err_flag + 1 + GNNSAFT_K0_SYNC_WORDS : nullptr,
                             I(p.k0_lost)};
      GS_TRY(launch_degree_tiles(I(p.rowptr), n, h, I(p.perm), I(p.tiles), I(p.num_tiles), I(p.hist3), err_flag,
                                 true, sa, &req, k0_fused ? &inst : nullptr));
    } else {
      GS_TRY(launch_degree_tiles(I(p.rowptr), n, h, I(p.perm), I(p.tiles), I(p.num_tiles), I(p.hist3), err_flag,
                                 true, sa));
      GS_TRY(fold_weights(2));
    }
  }
  }  // structure built in place
  // ---- the edge-class tables of all layers, on the caller's stream (unless the prologue made them)
  for (int l0 = 0; !tables_in_prologue && l0 < d->num_layers; l0 += kMaxGemmBatch) {
    const int nl = d->num_layers - l0 < kMaxGemmBatch ? d->num_layers - l0 : kMaxGemmBatch;
    GemmBatchEntry e[kMaxGemmBatch];
    for (int i = 0; i < nl; ++i) e[i] = GemmBatchEntry{lw[l0 + i].we, lw[l0 + i].be, F(p.cenc) + (l0 + i) * cstride, 0};
    LinearEpilogue epi;
    GS_TRY(launch_linear(F(p.cemb), h, 0, nl, e, h, h, p.combos, h, h, epi, st));
  }
  for (int i0 = 0; !tables_in_prologue && i0 < 2 * d->num_layers; i0 += kMaxGemmBatch) {
    const int ne = 2 * d->num_layers - i0 < kMaxGemmBatch ? 2 * d->num_layers - i0 : kMaxGemmBatch;
    GemmBatchEntry e[kMaxGemmBatch];
    for (int i = 0; i < ne; ++i) {
      const int l = (i0 + i) / 2, t = (i0 + i) % 2;
      e[i] = GemmBatchEntry{lw[l].wpre[t][0] + 2 * h, lw[l].bpre[t][0], F(p.rtab) + l * rstride + t * h, l * cstride};
    }
    LinearEpilogue epi;
    GS_TRY(launch_linear(F(p.cenc), h, 0, ne, e, 3 * (int64_t)h, 2 * (int64_t)h, p.combos, h, h, epi, st));
  }
  bool joined = aux == nullptr;
  auto join_structure = [&]() -> int {  // everything below this call may read the CSR / tiles / folded weights
    if (!joined) {
      GS_HIP(hipEventRecord(aux->join, sa));
      GS_HIP(hipStreamWaitEvent(st, aux->join, 0));
      joined = true;
    }
    return GNNSAFT_OK;
  };

  const bool tape = d->save_tape != 0;
  float *xc = F(p.x0), *xn = tape ? F(p.x0) + p.sx : F(p.x1);
  // train-mode node BatchNorm applied by whoever reads the layer's output (linear_bn_relu / NodeState).
  // desc->unfused_bn_apply: 0 = the LAST layer only (applied by the pooling kernel on load: one pass over y instead of
  // an apply pass writing x_L and a pooling pass reading it -- measured faster at every size); 2 = every layer (the
  // next layer's message GEMM applies it on load: measured SLOWER than the apply launch it saves once the GEMMs run
  // on the bf16 pipe, +8 us per layer at C2, +50 us at C3: the GEMM's operand path is its bottleneck); 1 = never.
  const bool bn_all = d->training && d->unfused_bn_apply == 2 && h <= 256 && (h % 32) == 0;
  const bool bn_last = d->training && d->unfused_bn_apply != 1;
  NodeState state;
  for (int l = 0; l < d->num_layers; ++l) {
    if (state.y == nullptr) state.x = xc;   // (otherwise x_l is still pending: formed by this layer's message GEMM)
    const LayerW &w = lw[l];
    float *pq_l = F(p.pq) + l * p.spq, *agg_l = F(p.agg) + l * p.sagg, *u_l = F(p.u0) + l * p.su;
    float *y_l = F(p.y) + l * p.sy;
    const float *avg = w.avg;
    const float *const(*wpre)[8] = w.wpre;
    const float *const(*bpre)[8] = w.bpre;
    const float *const(*wpost)[8] = w.wpost;
    const float *const(*bpost)[8] = w.bpost;
    const float *wlin = w.wlin, *blin = w.blin;
    const BnPtrs bn = w.bn;
    const float *rtab = F(p.rtab) + l * rstride;
    const float *weff = F(p.weff) + l * wstride;

    // message: node terms (+ extra pre-layers on edge rows)
    {
      ProfScope ps(prof, GNNSAFT_PROF_NODE_TERMS, st);
      if (fold_dst)
        GS_TRY(src_terms(state, n, h, wpre[0][0], wpre[1][0], pq_l, st,
                         w3_src ? ws + p.w3src + (size_t)l * p.w3src_stride : nullptr));
      else
        GS_TRY(node_terms(state, n, h, wpre[0][0], wpre[1][0], pq_l, st));
      state = NodeState{};   // x_l (= xc) is in memory now
      state.x = xc;
    }
    GS_TRY(join_structure());
    const float *msgs = nullptr;
    if (d->pre_layers > 1 && tape) {
      // keep every edge-level tensor: h1pre, then the output of each extra pre layer
      float *m_l = F(p.msg0) + l * p.smsg;
      const int64_t ms = p.ep * (int64_t)(2 * h);
      GS_TRY(gnnsaft_pna_edge_preact(I(p.src), I(p.dst), I(p.combo), p.ep, h, pq_l, rtab, m_l, st));
      for (int j = 1; j < d->pre_layers; ++j) {
        float *in = m_l + (j - 1) * ms, *outp = m_l + j * ms;
        GemmBatchEntry e2[2] = {{wpre[0][j], bpre[0][j], outp, 0}, {wpre[1][j], bpre[1][j], outp + h, h}};
        LinearEpilogue epi;
        GS_TRY(launch_linear(in, 2 * (int64_t)h, 1, 2, e2, h, 2 * (int64_t)h, p.ep, h, h, epi, st));
      }
      msgs = m_l + (d->pre_layers - 1) * ms;
    } else if (d->pre_layers > 1) {
      float *ma = F(p.msg0), *mb = F(p.msg1);
      GemmBatchEntry e[2] = {{wpre[0][1], bpre[0][1], ma, 0}, {wpre[1][1], bpre[1][1], ma + h, 0}};
      GS_TRY(launch_pna_edge_mlp(I(p.src), I(p.dst), I(p.combo), p.ep, h, pq_l, rtab, e, 2 * (int64_t)h, st));
      for (int j = 2; j < d->pre_layers; ++j) {
        GemmBatchEntry e2[2] = {{wpre[0][j], bpre[0][j], mb, 0}, {wpre[1][j], bpre[1][j], mb + h, h}};
        LinearEpilogue epi;
        GS_TRY(launch_linear(ma, 2 * (int64_t)h, 1, 2, e2, h, 2 * (int64_t)h, p.ep, h, h, epi, st));
        float *t = ma;
        ma = mb;
        mb = t;
      }
      msgs = ma;
    }
    // K4 aggregation (unless the update does it itself)
    if (!fuse_agg) {
      ProfScope ps(prof, GNNSAFT_PROF_AGGREGATE, st);
      if (fold_dst)
        GS_TRY(gnnsaft_pna_aggregate_src(I(p.rowptr), I(p.src), I(p.combo), n, h, pq_l, rtab, agg_l, st));
      else
        GS_TRY(gnnsaft_pna_aggregate(I(p.rowptr), I(p.src), I(p.combo), n, h, pq_l, rtab, msgs, agg_l, st));
    }
    // update: first post-layer with scalers on load, then extra post-layers
    float *ua = u_l, *ub = F(p.u1);
    if (fuse_agg) {
      ProfScope ps(prof, GNNSAFT_PROF_UPDATE_AGG, st);
      GS_TRY(launch_pna_update_agg(xc, pq_l, rtab, (int)p.combos, I(p.rowptr), I(p.src), I(p.combo), I(p.perm), I(p.tiles),
                                   I(p.num_tiles), p.tile_cap, n, h, ws + p.w3eff + (size_t)l * p.w3eff_stride,
                                   bpost[0][0], bpost[1][0], ua, st));
    } else if (d->fold_degree_scalers && w3_upd_gemm) {
      ProfScope ps(prof, GNNSAFT_PROF_UPDATE, st);
      GS_TRY(launch_pna_update_folded_w3(xc, agg_l, I(p.perm), I(p.tiles), I(p.num_tiles), p.tile_cap, n, h,
                                         ws + p.w3eff + (size_t)l * p.w3eff_stride, bpost[0][0], bpost[1][0], ua, st));
    } else if (d->fold_degree_scalers && ar_upd) {
      ProfScope ps(prof, GNNSAFT_PROF_UPDATE, st);
      GS_TRY(launch_pna_update_folded_ar(xc, agg_l, I(p.perm), I(p.tiles), I(p.num_tiles), p.tile_cap, n, h,
                                         ws + p.w3eff + (size_t)l * p.w3eff_stride, bpost[0][0], bpost[1][0], ua, st));
    } else if (d->fold_degree_scalers) {
      ProfScope ps(prof, GNNSAFT_PROF_UPDATE, st);
      GS_TRY(launch_pna_update_folded(xc, agg_l, I(p.perm), I(p.tiles), I(p.num_tiles), p.tile_cap, n, h, weff,
                                      bpost[0][0], bpost[1][0], ua, st));
    } else {
      GemmBatchEntry e[2] = {{wpost[0][0], bpost[0][0], ua, 0}, {wpost[1][0], bpost[1][0], ua + h / 2, 0}};
      ProfScope ps(prof, GNNSAFT_PROF_UPDATE, st);
      GS_TRY(launch_pna_update(xc, agg_l, F(p.log_amp), F(p.log_att), avg, n, h, e, h, st));
    }
    for (int j = 1; j < d->post_layers; ++j) {
      if (tape) ub = u_l + (int64_t)j * n * h;  // keep every intermediate for the backward
      GemmBatchEntry e2[2] = {{wpost[0][j], bpost[0][j], ub, 0}, {wpost[1][j], bpost[1][j], ub + h / 2, h / 2}};
      LinearEpilogue epi;
      GS_TRY(launch_linear(ua, h, 1, 2, e2, h / 2, h, n, h / 2, h / 2, epi, st));
      float *t = ua;
      ua = ub;
      ub = t;
    }
    // lin -> BatchNorm -> ReLU -> (+ x)
    const bool bn_deferred = bn_all || (bn_last && l == d->num_layers - 1);
    GS_TRY(linear_bn_relu(ua, h, wlin, blin, n, h, h, bn, d, ws, p, y_l, d->skip_connections ? xc : nullptr, xn, st,
                          prof, F(p.bnstat) + (int64_t)l * 2 * h, bn_deferred,
                          w3_lin ? ws + p.w3lin + (size_t)l * p.w3lin_stride : nullptr));
    if (bn_deferred) {   // x_{l+1} = relu(y_l scale + shift) (+ x_l): formed by whoever reads it next
      state = NodeState{};
      state.y = y_l;
      state.xres = d->skip_connections ? xc : nullptr;
      state.scale = F(p.scale);
      state.shift = F(p.shift);
      state.xdst = xn;
    }
    if (tape) {
      xc = xn;
      xn = xn + p.sx;
    } else {
      float *t = xc;
      xc = xn;
      xn = t;
    }
  }

  // global_add_pool of the final node state; a pending BatchNorm of the last layer is applied on load (the tape
  // keeps x_L: the tests read the ReLU gates off it)
  // (the pooling launch also leaves the structure chain's "lost" word zero for the next call: every launch that reads
  // it -- the permutation fill -- is in front of this one, and I(p.k0_lost) carries its value to the end of the call)
  int32_t *k0_clear = k0_fused ? err_flag + 1 + kK0LostWord : nullptr;
  const int32_t *k0_lost = k0_fused ? I(p.k0_lost) : nullptr;
  auto pool = [&](hipStream_t s_) -> int {
    if (state.y != nullptr)
      return launch_add_pool_bn(state.y, state.xres, state.scale, state.shift, tape ? state.xdst : nullptr,
                                I(p.graph_ptr), g, n, h, F(p.pooled), s_, k0_clear);
    return launch_add_pool(xc, I(p.graph_ptr), g, n, h, F(p.pooled), s_, k0_clear);
  };
  // ---- readout: one launch (readout.hip) while its workgroups are co-resident, the per-op path beyond
  // (an eval-mode tape keeps the readout's pre-activations through the per-op path)
  if (!d->unfused_readout && !(d->save_tape && !d->training) && readout_fused_launchable(g, h, d->num_para, p.nb, false)) {
    GS_TRY(pool(st));
    ReadoutFusedParams rp{};
    rp.x = xc;
    rp.graph_ptr = I(p.graph_ptr);
    rp.g = g;
    rp.n = n;
    rp.h = h;
    rp.num_para = d->num_para;
    rp.nblocks = p.nb;
    rp.training = d->training;
    rp.momentum = d->bn_momentum;
    rp.eps = d->bn_eps;
    for (int i = 0; i <= p.nb; ++i) {
      rp.w[i] = wc.f();
      rp.b[i] = wc.f();
      if (i < p.nb) rp.bn[i] = wc.bn();
    }
    GS_REQUIRE(wc.ok && wc.i == num_weights, GNNSAFT_ERR_SHAPE);
    rp.target = loss3 != nullptr ? target : nullptr;
    rp.out = out;
    rp.loss3 = target != nullptr ? loss3 : nullptr;
    rp.pooled = F(p.pooled);
    rp.ry = F(p.ry);
    rp.ro = F(p.ro);
    rp.rstat = F(p.rstat);
    rp.scratch = ws + p.rd_scratch;
    rp.sync = I(p.rd_sync);
    rp.err = err_flag;
    rp.k0_lost = k0_lost;
    rp.barrier_extra = d->debug_barrier_extra;
    rp.dropout_p = d->readout_dropout;
    rp.dropout_seed = d->dropout_seed;
    rp.dropout_step = d->dropout_step;
    return launch_readout_fused(rp, st);
  }
  // ---- readout (per-op: no dropout kernel here -- the one-launch readout carries it)
  GS_REQUIRE(!(d->training && d->readout_dropout > 0.f), GNNSAFT_ERR_UNSUPPORTED);
  GS_TRY(pool(st));
  const float *cur = F(p.pooled);
  int width = h, bi = 0;
  const int64_t rs = g * (int64_t)h;  // floats per readout block buffer
  auto block = [&](int n_out) -> int {
    const float *w = wc.f(), *b = wc.f();
    BnPtrs bn = wc.bn();
    GS_REQUIRE(wc.ok, GNNSAFT_ERR_NULL);
    float *y_tmp = F(p.ry) + bi * rs, *o = F(p.ro) + bi * rs;  // kept for backward
    GS_TRY(linear_bn_relu(cur, width, w, b, g, n_out, width, bn, d, ws, p, y_tmp, nullptr, o, st, nullptr,
                          F(p.rstat) + (int64_t)bi * 2 * h));
    cur = o;
    ++bi;
    width = n_out;
    return GNNSAFT_OK;
  };
  for (int i = 0; i < d->num_mlp_layers; ++i) GS_TRY(block(h));
  GS_TRY(block(h / 2));
  GS_TRY(block(h / 4));
  {
    const float *w = wc.f(), *b = wc.f();
    GS_REQUIRE(wc.ok && wc.i == num_weights, GNNSAFT_ERR_SHAPE);
    GemmBatchEntry ent{w, b, out, 0};
    LinearEpilogue epi;
    GS_TRY(launch_linear(cur, width, 0, 1, &ent, width, d->num_para, g, d->num_para, width, epi, st));
  }
  if (target != nullptr && loss3 != nullptr) GS_TRY(gnnsaft_mape(out, target, g * d->num_para, loss3, st));
  // a structure chain that lost its barrier: NaN instead of the garbage computed on the empty structure
  if (k0_lost != nullptr)
    GS_TRY(launch_poison_if(k0_lost, out, g * d->num_para, target != nullptr ? loss3 : nullptr, st));
  return GNNSAFT_OK;
}
