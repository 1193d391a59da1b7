// Workspace layout and weight-table parsing shared by gnnsaft_forward and gnnsaft_backward.
#pragma once
#include <vector>

#include "common.hpp"

// Side stream + events of one device (include/gnnsaft.h: gnnsaft_aux_create).  The forward runs its structure chain
// on it, the backward everything that is off its critical path (weight / bias gradients, edge-class sums, the
// edge-table chain).  One call at a time per handle.
struct gnnsaft_aux {
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  std::vector<hipEvent_t> pool;  // dependency events between the two streams, created on first use
};

namespace gs {

// i-th dependency event of the handle (no timing), or nullptr when the runtime refuses to create one
static inline hipEvent_t aux_event(gnnsaft_aux *a, size_t i) {
  while (a->pool.size() <= i) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    a->pool.push_back(e);
  }
  return a->pool[i];
}

struct Plan {
  // sizes
  int64_t n, e, ep, g, combos;
  int h;
  // byte offsets into the workspace
  size_t csr_ws, rowptr, src, dst, combo, log_amp, log_att, graph_ptr;
  size_t x0, x1, pq, agg, u0, u1, y, msg0, msg1, cemb, cenc, rtab, stats, bnseg, bnseg_bytes, scale, shift, pooled, m0, m1, m2;
  size_t perm, tiles, num_tiles, hist3, weff, gfold;
  size_t w3src, w3lin, w3eff;   // W3 weight images (w3.hpp), per layer: source block of pre_nns [2H,H], lin [H,H], w_eff
  size_t w3src_stride, w3lin_stride, w3eff_stride;   // bytes per layer (0 where the images are not built)
  size_t rd_scratch, rd_sync;   // fused readout: per-workgroup partials, barrier counters (+ BatchNorm tail tickets)
  size_t bn_tail_seg;
  size_t k0_lost;               // one int: 1 = the structure chain of THIS call lost a barrier (written by every call)
  size_t struct_begin, struct_bytes;  // [rowptr .. hist3]: everything that depends on edge_index / batch only
  int64_t tile_cap;
  // per-layer strides in floats (0 unless desc->save_tape: then every layer keeps its own tensors for backward)
  int64_t sx, spq, sagg, su, sy, smsg;
  size_t bnstat;          // [L][2][H] batch mean, rstd of the node BatchNorms
  size_t ry, ro, rstat;   // readout blocks: pre-BN [nb][G,H], output [nb][G,H], (mean, rstd) [nb][2][H]
  int nb;                 // BatchNorm blocks in the readout = num_mlp_layers + 2
  size_t total;
};

static inline int make_plan(const gnnsaft_model_desc *d, int64_t n, int64_t e, int64_t g, Plan &p) {
  GS_REQUIRE(d != nullptr, GNNSAFT_ERR_NULL);
  GS_REQUIRE(d->hidden >= 32 && (d->hidden % 32) == 0 && d->hidden <= 1024, GNNSAFT_ERR_UNSUPPORTED);
  GS_REQUIRE(d->num_layers >= 0 && d->pre_layers >= 1 && d->post_layers >= 1 && d->num_mlp_layers >= 0 &&
                 d->num_para >= 1,
             GNNSAFT_ERR_UNSUPPORTED);
  GS_REQUIRE(d->num_atom_cols >= 1 && d->num_atom_cols <= GNNSAFT_MAX_TABLES && d->num_bond_cols >= 1 &&
                 d->num_bond_cols <= GNNSAFT_MAX_TABLES,
             GNNSAFT_ERR_UNSUPPORTED);
  GS_REQUIRE(n >= 0 && e >= 0 && g >= 0, GNNSAFT_ERR_SHAPE);
  p.n = n;
  p.e = e;
  p.g = g;
  p.h = d->hidden;
  p.ep = e + (d->self_loops ? n : 0);
  GS_REQUIRE(p.ep + n < ((int64_t)1 << 31) - 1, GNNSAFT_ERR_SHAPE);
  p.combos = 1;
  for (int k = 0; k < d->num_bond_cols; ++k) {
    GS_REQUIRE(d->bond_dims[k] >= 1, GNNSAFT_ERR_SHAPE);
    p.combos *= d->bond_dims[k];
    GS_REQUIRE(p.combos <= (1 << 20), GNNSAFT_ERR_UNSUPPORTED);
  }
  const size_t h = (size_t)d->hidden;
  const size_t nn = (size_t)(n > 0 ? n : 1), ee = (size_t)(p.ep > 0 ? p.ep : 1), gg = (size_t)(g > 0 ? g : 1);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off += gs_align_up(bytes, 256);
    return o;
  };
  p.csr_ws = take(gnnsaft_csr_workspace_bytes(n, e));
  p.rowptr = take((nn + 1) * 4);
  p.src = take(ee * 4);
  p.dst = take(ee * 4);
  p.combo = take(ee * 4);
  p.log_amp = take(nn * 4);
  p.log_att = take(nn * 4);
  p.graph_ptr = take((gg + 1) * 4);
  // degree tiles directly behind the CSR: together they are the batch STRUCTURE, one contiguous segment that can be
  // cached per batch (gnnsaft_structure_build) and handed back to gnnsaft_forward
  p.tile_cap = gnnsaft_degree_tiles_capacity(n, d->hidden);
  p.perm = take(nn * 4);
  p.tiles = take((size_t)p.tile_cap * 16);
  p.num_tiles = take(4);
  p.hist3 = take(gnnsaft_degree_scratch_ints(n) * 4);
  p.struct_begin = p.rowptr;
  p.struct_bytes = off - p.rowptr;
  const size_t nlay = (size_t)(d->num_layers > 0 ? d->num_layers : 1);
  const bool tape = d->save_tape != 0;
  const size_t rep = tape ? nlay : 1;
  p.sx = tape ? (int64_t)(nn * h) : 0;
  p.spq = tape ? (int64_t)(nn * 4 * h) : 0;
  p.sagg = tape ? (int64_t)(nn * 8 * h) : 0;
  // tape: every post layer keeps its (pre-ReLU) output: layer l, post layer j at u0 + (l*q + j) * nn*h
  p.su = tape ? (int64_t)(nn * h) * d->post_layers : 0;
  p.sy = tape ? (int64_t)(nn * h) : 0;
  p.x0 = take((tape ? nlay + 1 : 1) * nn * h * 4);  // tape: x_0 .. x_L contiguous
  p.x1 = tape ? p.x0 : take(nn * h * 4);
  p.pq = take(rep * nn * 4 * h * 4);
  p.agg = take(rep * nn * 8 * h * 4);
  p.u0 = take(rep * (tape ? (size_t)d->post_layers : 1) * nn * h * 4);
  p.u1 = d->post_layers > 1 ? take(nn * h * 4) : p.u0;
  p.y = take(rep * nn * h * 4);
  p.bnstat = take(nlay * 2 * h * 4);
  p.nb = d->num_mlp_layers + 2;
  p.ry = take((size_t)p.nb * gg * h * 4);
  p.ro = take((size_t)p.nb * gg * h * 4);
  p.rstat = take((size_t)p.nb * 2 * h * 4);
  // edge-level tensors of the extra pre layers.  Tape: per conv layer, buffer 0 = pre-activation of the first
  // pre layer (P[dst] + Q[src] + R[class]), buffer j = (pre-ReLU) output of pre layer j: msg0 + (l*p + j) * ee*2h
  p.smsg = tape ? (int64_t)(ee * 2 * h) * d->pre_layers : 0;
  p.msg0 = d->pre_layers > 1 ? take((tape ? rep * (size_t)d->pre_layers : 1) * ee * 2 * h * 4) : 0;
  p.msg1 = d->pre_layers > 2 ? take(ee * 2 * h * 4) : p.msg0;
  p.cemb = take((size_t)p.combos * h * 4);
  const size_t nl = (size_t)(d->num_layers > 0 ? d->num_layers : 1);
  p.cenc = take(nl * (size_t)p.combos * h * 4);
  p.rtab = take(nl * (size_t)p.combos * 2 * h * 4);
  const size_t max_rows = nn > gg ? nn : gg;
  p.stats = take(((max_rows + kBnRowsPerGroup - 1) / kBnRowsPerGroup) * 2 * h * 4);
  p.bnseg_bytes = gnnsaft_bn_train_scratch_bytes((int64_t)max_rows, (int32_t)h);
  p.bnseg = take(p.bnseg_bytes + 8);
  p.scale = take(h * 4);
  p.shift = take(h * 4);
  p.pooled = take(gg * h * 4);
  p.m0 = take(gg * h * 4);
  p.m1 = take(gg * h * 4);
  p.m2 = take(gg * h * 4);
  {
    const size_t wgs = (gg + 63) / 64;   // readout.hip: 64 graphs per workgroup
    p.rd_scratch = take(gs_align_up((size_t)p.nb * wgs * 2 * h * 4, 256) + gs_align_up(wgs * 4, 256));
    // barrier / ticket counters zeroed by the prologue launch of every forward: the fused readout's (kRdSyncInts = 16),
    // then k_bn_stats_close's slab tickets (kBnTailCounterInts)
    p.rd_sync = take((16 + (size_t)kBnTailCounterInts) * 4);
    p.bn_tail_seg = take((size_t)64 * 2 * h * 8);   // [kBnMaxSegments][2][H] f64 segment sums of k_bn_stats_close
    p.k0_lost = take(4);
  }
  p.weff = d->fold_degree_scalers ? take(nl * (size_t)kDegreeBuckets * 5 * h * h * 4) : 0;
  p.gfold = d->fold_degree_scalers ? take(nl * 2 * 3 * (h / 2) * h * 8) : 0;   // float64 (fold.hpp)
  // weight images of the split-bf16 GEMMs (6 bytes per weight); K of every image (H, 5H) must be a multiple of 32
  const bool w3 = (h % 64) == 0;
  p.w3src_stride = w3 ? gs_align_up(2 * h * h * 6, 256) : 0;
  p.w3lin_stride = w3 ? gs_align_up(h * h * 6, 256) : 0;
  p.w3eff_stride = w3 && d->fold_degree_scalers ? (size_t)kDegreeBuckets * 5 * h * h * 6 : 0;
  p.w3src = take(nl * p.w3src_stride);
  p.w3lin = take(nl * p.w3lin_stride);
  p.w3eff = take(nl * p.w3eff_stride);
  p.total = off;
  return GNNSAFT_OK;
}

struct BnPtrs {
  const float *gamma, *beta;
  float *rmean, *rvar;
  int64_t *nbt;
};

struct WeightCursor {
  const void *const *w;
  int n;
  int i = 0;
  bool ok = true;
  const float *f() {
    if (i >= n || w[i] == nullptr) {
      ok = false;
      ++i;
      return nullptr;
    }
    return static_cast<const float *>(w[i++]);
  }
  BnPtrs bn() {
    BnPtrs b;
    b.gamma = f();
    b.beta = f();
    b.rmean = const_cast<float *>(f());
    b.rvar = const_cast<float *>(f());
    b.nbt = reinterpret_cast<int64_t *>(const_cast<float *>(f()));
    return b;
  }
};

#define GS_TRY(expr)                  \
  do {                                \
    const int rc__ = (expr);          \
    if (rc__ != GNNSAFT_OK) return rc__; \
  } while (0)


struct LayerW {
  const float *avg, *we, *be, *wlin, *blin;
  const float *wpre[2][8], *bpre[2][8], *wpost[2][8], *bpost[2][8];
  BnPtrs bn;
};

struct ReadoutW {
  const float *w, *b;
  BnPtrs bn;   // unused for the final Linear
  int n_in, n_out;
};

// walks the canonical weight table (header of forward.hip); `idx` (optional) receives the table
// index of every tensor in the same walk so that the backward can address the matching gradient slots
struct ParsedWeights {
  const float *atom_tab[GNNSAFT_MAX_TABLES], *bond_tab[GNNSAFT_MAX_TABLES];
  int atom0, bond0;                 // table indices of the first atom / bond table
  std::vector<LayerW> layers;
  std::vector<int> layer_base;      // table index of layer l's first entry (avg_deg_log)
  std::vector<ReadoutW> readout;    // num_mlp_layers + 2 BN blocks, then the final Linear
  std::vector<int> readout_base;    // table index of each readout entry's weight
};

static inline int parse_weights(const gnnsaft_model_desc *d, const void *const *weights_host, int num_weights,
                                ParsedWeights &pw) {
  GS_REQUIRE(d->pre_layers <= 8 && d->post_layers <= 8, GNNSAFT_ERR_UNSUPPORTED);
  WeightCursor wc{weights_host, num_weights};
  pw.atom0 = wc.i;
  for (int k = 0; k < d->num_atom_cols; ++k) pw.atom_tab[k] = wc.f();
  pw.bond0 = wc.i;
  for (int k = 0; k < d->num_bond_cols; ++k) pw.bond_tab[k] = wc.f();
  pw.layers.resize(d->num_layers);
  pw.layer_base.resize(d->num_layers);
  for (int l = 0; l < d->num_layers; ++l) {
    LayerW &w = pw.layers[l];
    pw.layer_base[l] = wc.i;
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
  }
  const int h = d->hidden;
  int width = h;
  auto block = [&](int n_out, bool with_bn) {
    ReadoutW r;
    pw.readout_base.push_back(wc.i);
    r.w = wc.f();
    r.b = wc.f();
    if (with_bn) r.bn = wc.bn();
    r.n_in = width;
    r.n_out = n_out;
    width = n_out;
    pw.readout.push_back(r);
  };
  for (int i = 0; i < d->num_mlp_layers; ++i) block(h, true);
  block(h / 2, true);
  block(h / 4, true);
  block(d->num_para, false);
  GS_REQUIRE(wc.ok && wc.i == num_weights, GNNSAFT_ERR_SHAPE);
  return GNNSAFT_OK;
}

}  // namespace gs
