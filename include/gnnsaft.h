/*
 * gnnsaft.h -- C ABI of the MI355X (gfx950) implementation of the GNN-ePC-SAFT
 * message-passing forward (PNAPCSAFT.forward + MAPE training loss).
 *
 * Reference interface this library replaces (all paths relative to
 * /root/reference): the reference has NO native code; the seam is the Python
 * method gnnepcsaft/train/models.py:105-135 (PNAPCSAFT.forward) and the loss
 * at models.py:191-194, whose arithmetic is executed by third-party ATen
 * kernels issued by PyG / ogb / torchmetrics ("implicit
 * device ops" table).  Every entry point below names the reference line(s)
 * whose arithmetic it performs.  A maintainer binds this file with ctypes
 * (see INTEGRATION.md); no torch type appears in any signature.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter says "host";
 *   - float tensors are row-major f32, index tensors are int64 exactly as PyG
 *     delivers them (x [N,C], edge_index [2,E], edge_attr [E,B], batch [N]);
 *   - all work is enqueued on `stream` (a hipStream_t); no entry point
 *     synchronises, allocates or frees device memory (graph-capture safe);
 *   - return value: 0 = ok, >0 = hipError_t, <0 = GNNSAFT_ERR_*;
 *   - invalid indices never fault: they are clamped and bit(s) are OR-ed into
 *     the int32 word `err_flag` (may be NULL), see GNNSAFT_FLAG_*.
 */
#ifndef GNNSAFT_H
#define GNNSAFT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNNSAFT_ABI_VERSION 5

/* The library is built with -fvisibility=hidden: only the functions declared  */
/* in this header (GNNSAFT_API) are exported; kernels, launchers and helpers   */
/* stay internal, so a second HIP library in the process cannot interpose.     */
#ifndef GNNSAFT_API
#define GNNSAFT_API __attribute__((visibility("default")))
#endif

#define GNNSAFT_OK 0
#define GNNSAFT_ERR_SHAPE (-1)      /* unsupported / inconsistent sizes          */
#define GNNSAFT_ERR_WORKSPACE (-2)  /* workspace too small                       */
#define GNNSAFT_ERR_NULL (-3)       /* required pointer is NULL                  */
#define GNNSAFT_ERR_UNSUPPORTED (-4)/* configuration outside the shape envelope  */

#define GNNSAFT_FLAG_BAD_EDGE 1     /* edge_index value outside [0,N)            */
#define GNNSAFT_FLAG_BAD_ATTR 2     /* categorical index outside its vocabulary  */
#define GNNSAFT_FLAG_BAD_BATCH 4    /* batch not sorted / outside [0,G)          */
#define GNNSAFT_FLAG_BAD_DEGREE 8   /* in-degree >= gnnsaft_degree_buckets() with */
                                    /* degree folding on: disable folding         */
#define GNNSAFT_K0_SYNC_WORDS 8      /* see gnnsaft_model_desc.persistent_sync_words */
#define GNNSAFT_FLAG_BARRIER_TIMEOUT 16 /* a grid barrier of the fused readout gave  */
                                    /* up waiting (workgroups not co-resident):    */
                                    /* the outputs of that call are invalid         */

#define GNNSAFT_MAX_TABLES 16
#define GNNSAFT_TOWERS 2            /* models.py:76 towers=2                      */

typedef void *gnnsaft_stream_t;     /* hipStream_t */

GNNSAFT_API int gnnsaft_abi_version(void);
GNNSAFT_API const char *gnnsaft_error_string(int code);

/* ------------------------------------------------------------------------ */
/* K0: graph structure.  Replaces add_self_loops (models.py:118-121) and the  */
/* per-layer index_select / scatter index plumbing PyG re-derives from the    */
/* unsorted edge list: builds ONE destination-sorted CSR per batch, reused by */
/* all layers.  Edges of a node keep their edge_index order; the self-loop    */
/* (if enabled) is the LAST in-edge of every node, as the reference appends   */
/* loops after all real edges.  combo[r] is the mixed-radix id of the edge's   */
/* categorical attributes (self-loop = all-zero attributes = id 0).           */
/* log_amp[i] = log(deg_i + 1), log_att[i] = log(max(deg_i,1) + 1) with deg_i  */
/* the run-time in-degree INCLUDING the self-loop (PyG DegreeScalerAggregation).*/
/* ------------------------------------------------------------------------ */
GNNSAFT_API size_t gnnsaft_csr_workspace_bytes(int64_t num_nodes, int64_t num_edges);

GNNSAFT_API int gnnsaft_csr_build(const int64_t *edge_index, const int64_t *edge_attr,
                      int64_t num_nodes, int64_t num_edges,
                      int32_t num_bond_cols, const int32_t *bond_dims_host,
                      int32_t self_loops,
                      int32_t *rowptr,   /* [N+1]            */
                      int32_t *src,      /* [E'] E'=E+N*loops */
                      int32_t *dst,      /* [E'] destination of CSR row r */
                      int32_t *combo,    /* [E']             */
                      float *log_amp,    /* [N]              */
                      float *log_att,    /* [N]              */
                      int32_t *err_flag, void *workspace, size_t workspace_bytes,
                      gnnsaft_stream_t stream);

/* graph pointer from PyG's sorted `batch` vector (global_add_pool, models.py:133) */
GNNSAFT_API int gnnsaft_batch_to_ptr(const int64_t *batch, int64_t num_nodes, int64_t num_graphs,
                         int32_t *graph_ptr /* [G+1] */, int32_t *err_flag,
                         gnnsaft_stream_t stream);

/* ------------------------------------------------------------------------ */
/* K1: ogb AtomEncoder / BondEncoder (models.py:65-66,122-123): out[i,:] =     */
/* sum_k table_k[idx[i,k], :], left-to-right.  `tables_host` is a HOST array   */
/* of `num_cols` device pointers, `dims_host` the vocabulary sizes.            */
/* ------------------------------------------------------------------------ */
GNNSAFT_API int gnnsaft_embed_sum(const int64_t *idx, int64_t num_rows, int32_t num_cols,
                      const float *const *tables_host, const int32_t *dims_host,
                      int32_t hidden, float *out, int32_t *err_flag,
                      gnnsaft_stream_t stream);

/* BondEncoder evaluated once per attribute combination instead of once per   */
/* edge: out[c,:] = sum_k table_k[digit_k(c), :], c in [0, prod(dims)).        */
GNNSAFT_API int gnnsaft_bond_combo_embed(int32_t num_cols, const float *const *tables_host,
                             const int32_t *dims_host, int32_t hidden, float *out,
                             gnnsaft_stream_t stream);

/* ------------------------------------------------------------------------ */
/* Dense layer on the f32 matrix cores: out = act_out(act_in(A) W^T + b).     */
/* torch.nn.Linear / PyG Linear as used at models.py:84-103 and inside        */
/* PNAConv (edge_encoder, lin, extra pre/post layers).  W is [n_out, ldw]     */
/* row-major (torch layout), columns [0,k) are used.  `stats` (may be NULL)   */
/* receives per-row-group (mean, M2) column partials for train-mode BatchNorm */
/* ([ceil(M/rows_per_group), 2, n_out], rows_per_group from                   */
/* gnnsaft_bn_rows_per_group()).  If scale/shift are non-NULL the epilogue    */
/* applies y*scale[c]+shift[c] (eval-mode BatchNorm folded), then ReLU if     */
/* relu_out, then adds residual[row,c] if non-NULL.                           */
/* ------------------------------------------------------------------------ */
GNNSAFT_API int gnnsaft_linear(const float *a, int64_t lda, int32_t relu_in,
                   const float *w, int64_t ldw, const float *bias,
                   float *out, int64_t ldo,
                   int64_t m, int32_t n_out, int32_t k,
                   const float *scale, const float *shift, int32_t relu_out,
                   const float *residual, int64_t ldr,
                   float *stats, gnnsaft_stream_t stream);

GNNSAFT_API int32_t gnnsaft_bn_rows_per_group(void);

/* ------------------------------------------------------------------------ */
/* PNAConv message, first pre-layer (PyG PNAConv.message; models.py:69-80,128):*/
/* pre_nns[t][0](cat[x_dst, x_src, edge_encoder(e)]) split by linearity into   */
/*   pq[i, 0:2F]  = [W_t[:,0:F]   x_i]_t   (destination term, per node)        */
/*   pq[i, 2F:4F] = [W_t[:,F:2F]  x_i]_t   (source term, per node)             */
/*   rtab[c, 0:2F] = [W_t[:,2F:3F] (W_e emb_c + b_e) + b_t]_t  (per edge class) */
/* so that msg[e,t,:] = pq[dst,tF:(t+1)F] + pq[src,2F+tF:...] + rtab[c(e),...]. */
/* ------------------------------------------------------------------------ */
GNNSAFT_API int gnnsaft_pna_node_terms(const float *x, int64_t num_nodes, int32_t hidden,
                           const float *w_pre0, const float *w_pre1, /* [F,3F] each */
                           float *pq /* [N,4F] */, gnnsaft_stream_t stream);

GNNSAFT_API int gnnsaft_pna_edge_table(const float *combo_emb, int32_t num_combos, int32_t hidden,
                           const float *w_edge, const float *b_edge,   /* edge_encoder */
                           const float *w_pre0, const float *b_pre0,
                           const float *w_pre1, const float *b_pre1,
                           float *enc_tmp /* [C,F] scratch */, float *rtab /* [C,2F] */,
                           gnnsaft_stream_t stream);

/* remaining pre-layers (pre_layers >= 2): msgs[r,t,:] =                       */
/*   W2_t relu(pq[dst_r] + pq[src_r] + rtab[combo_r]) + b2_t  in CSR row order. */
GNNSAFT_API int gnnsaft_pna_edge_mlp(const int32_t *src, const int32_t *dst, const int32_t *combo,
                         int64_t num_rows, int32_t hidden,
                         const float *pq, const float *rtab,
                         const float *w2_t0, const float *b2_t0,
                         const float *w2_t1, const float *b2_t1,
                         float *msgs /* [E',2F] */, gnnsaft_stream_t stream);

/* pre-activation of the first pre layer per CSR row, out[r,:] = pq[dst_r] + pq[src_r] + rtab[combo_r]  */
/* ([E',2F]); materialised only when the backward of pre_layers >= 2 needs it.                          */
GNNSAFT_API int gnnsaft_pna_edge_preact(const int32_t *src, const int32_t *dst, const int32_t *combo,
                            int64_t num_rows, int32_t hidden, const float *pq, const float *rtab,
                            float *out, gnnsaft_stream_t stream);

/* ------------------------------------------------------------------------ */
/* K4 (the measured "scatter-add" kernel): PyG MultiAggregation               */
/* [mean,min,max,std] over the in-edges of every node (models.py:59,128):      */
/* one pass over the destination-sorted rows, sum / sum-of-squares / min / max */
/* kept in registers, no atomics, bitwise reproducible.                        */
/*   agg[i,t,:] = [mean | min | max | std]  ([N, 2, 4F])                       */
/* std = sqrt(clamp(mean(m^2) - mean(m)^2, 1e-5)), zeroed where <= sqrt(1e-5). */
/* Exactly one of (pq,rtab) / msgs is used: msgs == NULL selects the fused     */
/* gather form for pre_layers == 1.                                            */
/* ------------------------------------------------------------------------ */
GNNSAFT_API int gnnsaft_pna_aggregate(const int32_t *rowptr, const int32_t *src, const int32_t *combo,
                          int64_t num_nodes, int32_t hidden,
                          const float *pq, const float *rtab, const float *msgs,
                          float *agg, gnnsaft_stream_t stream);

/* ------------------------------------------------------------------------ */
/* PNAConv update, first post-layer (PyG PNAConv.forward; models.py:128):      */
/* u[i, tF/2:(t+1)F/2] = post_nns[t][0](cat[x_i, A, A*amp_i, A*att_i]) with    */
/* A = agg[i,t,:], amp_i = log_amp[i]/avg_deg_log, att_i = avg_deg_log/log_att[i];*/
/* the [N,T,13F] input is never materialised (scalers applied on operand load).*/
/* ------------------------------------------------------------------------ */
GNNSAFT_API int gnnsaft_pna_update(const float *x, const float *agg, const float *log_amp,
                       const float *log_att, const float *avg_deg_log /* device, [1] */,
                       int64_t num_nodes, int32_t hidden,
                       const float *w_post0, const float *b_post0, /* [F/2,13F] */
                       const float *w_post1, const float *b_post1,
                       float *u /* [N,F] */, gnnsaft_stream_t stream);

/* ------------------------------------------------------------------------ */
/* Degree-folded form of the same update (default in gnnsaft_forward): nodes  */
/* are grouped by in-degree into tiles of one degree each, the three scalers  */
/* are folded into per-degree weights W_eff(d) = [W_x | W_id + amp(d) W_amp + */
/* att(d) W_att], and the GEMM's K shrinks from 13F to 5F.                    */
/*   gnnsaft_degree_tiles: perm[N] (node ids grouped by degree), tiles[cap,4] */
/*     = (degree, first slot, rows, 0), num_tiles[1]; `scratch` holds          */
/*     gnnsaft_degree_scratch_ints(N) int32, the first `buckets` of which are   */
/*     the degree histogram afterwards.  Nodes of equal degree keep ascending   */
/*     node order (no atomics: deterministic);                                  */
/*   gnnsaft_pna_fold_post_weights: w_eff[buckets,2,F/2,5F] (only degrees      */
/*     present in the histogram are written);                                  */
/*   gnnsaft_pna_update_folded: u[N,F] as gnnsaft_pna_update.                  */
/* ------------------------------------------------------------------------ */
GNNSAFT_API int32_t gnnsaft_degree_buckets(void);
GNNSAFT_API int64_t gnnsaft_degree_tiles_capacity(int64_t num_nodes, int32_t hidden);
GNNSAFT_API size_t gnnsaft_degree_scratch_ints(int64_t num_nodes);
GNNSAFT_API int gnnsaft_degree_tiles(const int32_t *rowptr, int64_t num_nodes, int32_t hidden, int32_t *perm,
                         int32_t *tiles, int32_t *num_tiles, int32_t *scratch, int32_t *err_flag,
                         gnnsaft_stream_t stream);
GNNSAFT_API int gnnsaft_pna_fold_post_weights(const float *w_post0, const float *w_post1,
                                  const float *avg_deg_log, const int32_t *hist, int32_t hidden,
                                  float *w_eff, gnnsaft_stream_t stream);
#define GNNSAFT_MAX_FOLD_LAYERS 8
/* the same for several layers in one launch (weights of layer i at w_eff + i*layer_stride).  */
/* If w_pre0_host / w_pre1_host (pre_nns[t][0].weight per layer) are given, the DESTINATION    */
/* term of the message is folded too: msg = W_dst x_dst + m~ shifts mean/min/max of a node by   */
/* the constant P_i = W_dst x_i and leaves std unchanged, so W_eff's x-block gains              */
/* sum_s scale_s(d) (W_s,mean+W_s,min+W_s,max) W_dst (for d > 0) and the aggregates are taken   */
/* over m~ = W_src x_src + edge term (gnnsaft_pna_aggregate_src).  g_scratch: L*2*3*(F/2)*F f64  */
/* (8-byte aligned).  Every folded weight is accumulated in float64 and rounded to f32 once.      */
GNNSAFT_API int gnnsaft_pna_fold_post_weights_multi(int32_t num_layers, const float *const *w_post0_host,
                                        const float *const *w_post1_host,
                                        const float *const *avg_deg_log_host,
                                        const float *const *w_pre0_host /* or NULL */,
                                        const float *const *w_pre1_host /* or NULL */,
                                        void *g_scratch /* or NULL */, const int32_t *hist,
                                        int32_t hidden, float *w_eff, int64_t layer_stride,
                                        gnnsaft_stream_t stream);
/* source term only: q[i, tF:(t+1)F] = W_t[:,F:2F] x_i   ([N,2F]) */
GNNSAFT_API int gnnsaft_pna_src_terms(const float *x, int64_t num_nodes, int32_t hidden, const float *w_pre0,
                          const float *w_pre1, float *q /* [N,2F] */, gnnsaft_stream_t stream);
/* K4 over m~ = q[src] + rtab[class]: same outputs as gnnsaft_pna_aggregate minus the per-node shift P_i */
GNNSAFT_API int gnnsaft_pna_aggregate_src(const int32_t *rowptr, const int32_t *src, const int32_t *combo,
                              int64_t num_nodes, int32_t hidden, const float *q, const float *rtab,
                              float *agg, gnnsaft_stream_t stream);
GNNSAFT_API int gnnsaft_pna_update_folded(const float *x, const float *agg, const int32_t *perm,
                              const int32_t *tiles, const int32_t *num_tiles, int64_t num_nodes,
                              int32_t hidden, const float *w_eff, const float *b_post0,
                              const float *b_post1, float *u, gnnsaft_stream_t stream);
/* Envelope of every GEMM entry point (gnnsaft_linear, gnnsaft_pna_*, gnnsaft_forward): outputs and residuals are    */
/* addressed with 32-bit ELEMENT offsets, (rows + 1) * leading dimension < 2^31 (e.g. 4 M edge rows at 2H = 512);       */
/* beyond it the call returns GNNSAFT_ERR_SHAPE -- split the rows.  Non-finite operands in split-bf16 mode (the        */
/* default): +-inf or NaN anywhere in a row of A (or of W) makes the outputs it contributes to NaN (an f32 fma chain   */
/* would keep some of them +-inf).                                                                                     */
/* tuning / test hook: gnnsaft_linear (no epilogue options) with an explicit tile configuration    */
/* 0..5 = 256x32, 128x64, 128x128, 64x64, 64x128, 128x32; per call, the library keeps no global state */
GNNSAFT_API int gnnsaft_debug_linear_tile(const float *a, int64_t lda, const float *w, int64_t ldw, const float *bias,
                              float *out, int64_t ldo, int64_t m, int32_t n_out, int32_t k,
                              float *stats /* or NULL */, int32_t tile_config, gnnsaft_stream_t stream);

/* ------------------------------------------------------------------------ */
/* "W3" weight images (csrc/w3.hpp): a weight matrix [rows, k] (k % 32 == 0) split ONCE into the three bf16     */
/* planes of its exact f32 split, laid out as the split-bf16 GEMM's LDS stage wants them, so the GEMM copies    */
/* its B tile with direct-to-LDS loads.  gnnsaft_forward builds the images of pre_nns / lin / the folded update  */
/* weights (models.py:69-80,128) in its workspace on every call; these entry points serve the stage tests and    */
/* tools/gemm_tune.py.  tile_config: 0..5 = 128x128, 128x256, 64x128, 64x64, 128x64, 64x256 (one LDS stage, several */
/* workgroups per CU), 6 / 7 = 128x128 / 128x256 double-buffered (one workgroup per CU).                        */
GNNSAFT_API size_t gnnsaft_w3_image_bytes(int64_t rows, int64_t k);
GNNSAFT_API int gnnsaft_w3_pack(const float *w, int64_t ldw, int32_t rows, int32_t k, void *image /* 16-B aligned */,
                    gnnsaft_stream_t stream);
GNNSAFT_API int gnnsaft_debug_linear_w3(const float *a, int64_t lda, const void *w_image, const float *bias, float *out,
                            int64_t ldo, int64_t m, int32_t n_out, int32_t k, float *stats /* or NULL */,
                            int32_t tile_config, gnnsaft_stream_t stream);
/* K4 + degree-folded update in ONE launch (csrc/update_agg.hip): the aggregation of m~ = q[src] + rtab[class] over   */
/* the in-edges (what gnnsaft_pna_aggregate_src computes) happens inside the update GEMM's operand path, the aggregates */
/* [N,2,4F] never reach HBM.  w_eff_images: the W3 images of the folded weights [D][2][F/2,5F] in the block order of  */
/* gnnsaft_pna_fold_post_weights' w_eff (gnnsaft_w3_pack of every [F/2,5F] block).  hidden % 128 == 0.               */
GNNSAFT_API int gnnsaft_pna_update_agg(const float *x, const float *q, const float *rtab, int32_t num_classes /* <= 64 */,
                           const int32_t *rowptr, const int32_t *src, const int32_t *combo, const int32_t *perm,
                           const int32_t *tiles, const int32_t *num_tiles, int64_t num_nodes, int32_t hidden,
                           const void *w_eff_images, const float *b_post0, const float *b_post1, float *u,
                           gnnsaft_stream_t stream);
/* development probe: s_memtime stamps of one tile of the fused kernel into a device buffer of 512 uint64 (NULL: off) */
GNNSAFT_API int gnnsaft_debug_update_agg_stamps(void *device_buffer);
/* the wave-specialised kernel (csrc/gemm_w3s.hip: consumer waves = fragment reads + MFMAs, producer waves =   */
/* operand path); tile_config 0 = 128x128, 1 = 64x128                                                          */
GNNSAFT_API int gnnsaft_debug_linear_w3s(const float *a, int64_t lda, const void *w_image, const float *bias, float *out,
                             int64_t ldo, int64_t m, int32_t n_out, int32_t k, float *stats /* or NULL */,
                             int32_t tile_config, gnnsaft_stream_t stream);

/* the A-operand-in-registers kernel (csrc/gemm_ar.hip: a wave owns 32 rows x all columns of the tile, loads and   */
/* splits its own A rows, B through an LDS ring of image stages); tile_config 0..3 = 128x128, 128x64, 64x128, 64x64 */
GNNSAFT_API int gnnsaft_debug_linear_ar(const float *a, int64_t lda, const void *w_image, const float *bias, float *out,
                            int64_t ldo, int64_t m, int32_t n_out, int32_t k, int32_t tile_config,
                            gnnsaft_stream_t stream);

/* the degree-folded update (gnnsaft_pna_update_folded) on k_gemm_ar with BOTH towers in one workgroup per degree     */
/* tile -- what gnnsaft_forward launches below 64 k nodes for hidden 128 / 256 (csrc/gemm_ar.hip); w_eff_images as for  */
/* gnnsaft_pna_update_agg.  Other hidden sizes: GNNSAFT_ERR_UNSUPPORTED.                                               */
GNNSAFT_API int gnnsaft_pna_update_folded_ar(const float *x, const float *agg, const int32_t *perm, const int32_t *tiles,
                                 const int32_t *num_tiles, int64_t num_nodes, int32_t hidden,
                                 const void *w_eff_images, const float *b_post0, const float *b_post1, float *u,
                                 gnnsaft_stream_t stream);
/* development probe: s_memtime stamps of one tile of k_gemm_ar into a device buffer of 256 uint64 (NULL: off) */
GNNSAFT_API int gnnsaft_debug_ar_stamps(void *device_buffer);

/* ------------------------------------------------------------------------ */
/* BatchNorm (PyG BatchNorm -> torch BatchNorm1d, models.py:82,87,94,98,128).  */
/* training != 0: combine the (mean, M2) partials written by gnnsaft_linear    */
/* into batch statistics, update running_mean / running_var (unbiased) /       */
/* num_batches_tracked, and emit scale = gamma*rstd, shift = beta - mean*scale.*/
/* training == 0: scale / shift from the running statistics.                   */
/* ------------------------------------------------------------------------ */
GNNSAFT_API int gnnsaft_bn_finalize(const float *stats, int64_t num_rows, int32_t channels,
                        const float *gamma, const float *beta,
                        float *running_mean, float *running_var, int64_t *num_batches_tracked,
                        float momentum, float eps, int32_t training,
                        float *scale, float *shift, gnnsaft_stream_t stream);

/* Train-mode BatchNorm + ReLU (+ residual): folds the (mean, M2) partials of            */
/* gnnsaft_linear, normalises y, updates the running statistics and the counter.  One    */
/* launch up to 4096 rows; above that a small pre-combine launch folds the partials into  */
/* <= 64 f64 segment sums in `scratch` (8-byte aligned, gnnsaft_bn_train_scratch_bytes).  */
GNNSAFT_API size_t gnnsaft_bn_train_scratch_bytes(int64_t num_rows, int32_t channels);
GNNSAFT_API int gnnsaft_bn_train_apply(const float *stats, const float *y, int64_t num_rows, int32_t channels,
                           const float *gamma, const float *beta, float *running_mean,
                           float *running_var, int64_t *num_batches_tracked, float momentum,
                           float eps, const float *residual, float *out,
                           float *save_mean_rstd /* [2*channels] or NULL: kept for backward */,
                           void *scratch /* may be NULL when the size above is 0 */,
                           size_t scratch_bytes, gnnsaft_stream_t stream);

/* out = relu(y*scale + shift) (+ residual)   (models.py:128-131) */
GNNSAFT_API int gnnsaft_bn_relu_residual(const float *y, const float *scale, const float *shift,
                             const float *residual, float *out,
                             int64_t num_rows, int32_t channels, gnnsaft_stream_t stream);

/* global_add_pool (models.py:133): out[g,:] = sum of rows graph_ptr[g]..graph_ptr[g+1]-1 */
GNNSAFT_API int gnnsaft_add_pool(const float *x, const int32_t *graph_ptr, int64_t num_graphs,
                     int64_t num_nodes, int32_t hidden, float *out, gnnsaft_stream_t stream);

/* torchmetrics MAPE (models.py:194): out[0] = sum(|p-t| / max(|t|,1.17e-6)) / n, */
/* out[1] = the sum, out[2] = n (for the cross-rank all-reduce).                 */
GNNSAFT_API int gnnsaft_mape(const float *pred, const float *target, int64_t numel, float *out3,
                 gnnsaft_stream_t stream);

/* ------------------------------------------------------------------------ */
/* Whole network: PNAPCSAFT.forward (models.py:105-135) + optional MAPE loss   */
/* (models.py:191-194) enqueued as one call.                                   */
/* `weights_host`: HOST array of device pointers in the canonical order        */
/* documented in DESIGN.md ("weight table"); `num_weights` guards it.          */
/* ------------------------------------------------------------------------ */
typedef struct gnnsaft_model_desc {
  int32_t hidden;            /* H in {64,128,256}; must be a multiple of 32    */
  int32_t num_layers;        /* propagation_depth                              */
  int32_t pre_layers;        /* >= 1                                           */
  int32_t post_layers;       /* >= 1                                           */
  int32_t num_mlp_layers;    /* >= 0                                           */
  int32_t num_para;          /* P                                              */
  int32_t skip_connections;
  int32_t self_loops;
  int32_t training;          /* BatchNorm mode                                 */
  int32_t num_atom_cols;
  int32_t num_bond_cols;
  int32_t atom_dims[GNNSAFT_MAX_TABLES];
  int32_t bond_dims[GNNSAFT_MAX_TABLES];
  float bn_eps;
  float bn_momentum;
  int32_t fold_degree_scalers; /* 1: degree-folded update (in-degrees < gnnsaft_degree_buckets()) */
  int32_t fold_dst_term;       /* 1: also fold the message's destination term (needs the above, pre_layers == 1) */
  int32_t save_tape;           /* 1: every layer keeps its tensors in the workspace for gnnsaft_backward */
  int32_t unfused_readout;     /* 1: per-op readout (pool, GEMMs, BatchNorm, MAPE launches) instead of readout.hip */
  double bn_eps_f64;           /* BatchNorm eps in full precision for the float64 kernels (0: use bn_eps) */
  int32_t debug_barrier_extra; /* 0.  Test hook: the grid barriers of the fused readout wait for this many arrivals */
                               /* more than there are workgroups, i.e. time out (flag + NaN outputs)                */
  float readout_dropout;       /* p of the readout MLP's Dropout layers (models.py:88,95,99); applied in training  */
  uint64_t dropout_seed;       /* Philox key of this call's dropout masks (the backward regenerates them from it)   */
  int32_t unfused_bn_apply;    /* train-mode node BatchNorm: 0 = combine + apply launches, except the LAST layer, whose */
                               /* normalisation the pooling kernel applies on load; 1 = combine + apply everywhere; */
                               /* 2 = statistics closed in one launch and applied on load by the next message GEMM  */
  int32_t persistent_sync_words; /* int32 words the caller keeps BEHIND the err_flag word (err_flag[1 .. words]) for state  */
                               /* that must survive between calls: zero before the FIRST call; the library keeps them   */
                               /* zero between calls by itself -- also after a call that lost a grid barrier            */
                               /* (GNNSAFT_FLAG_BARRIER_TIMEOUT): that call's outputs are NaN in every mode, the launch */
                               /* that installs the empty structure restores the words, and the next call is a correct  */
                               /* forward whether or not the host ever read the flag word.                              */
                               /* >= GNNSAFT_K0_SYNC_WORDS + num_nodes: gnnsaft_forward builds the batch structure by   */
                               /* cooperating workgroups of its FIRST launch (one grid barrier, fill cursors in these    */
                               /* words) beside the embedding work instead of four dependent launches.  0 (err_flag is  */
                               /* one word): the launches.  Calls that share the words must be ordered (one stream, or  */
                               /* events between streams); not for several PROCESSES sharing one GPU (their waits lose  */
                               /* scheduling quanta: correct, but several times slower than the launches).             */
  const uint64_t *dropout_step; /* NULL, or a DEVICE word added (times an odd constant) to dropout_seed by the kernels when */
                               /* they run: a forward captured in a hipGraph draws fresh dropout masks on every replay   */
                               /* if the caller bumps the word in front of it (stream-ordered); gnnsaft_backward, given  */
                               /* the same descriptor, regenerates the masks of the replay it belongs to.               */
} gnnsaft_model_desc;

GNNSAFT_API int32_t gnnsaft_num_weights(const gnnsaft_model_desc *desc);

/* Workgroups (of 64 graphs each) of the one-launch readout (forward: backward = 0; its backward: 1) that the  */
/* CURRENT device keeps co-resident for hidden size `hidden`: hipOccupancyMaxActiveBlocksPerMultiprocessor x    */
/* multiProcessorCount, 0 when the query fails.  gnnsaft_forward / gnnsaft_backward take the one-launch readout  */
/* (train-mode BatchNorm statistics meet at a grid barrier) up to this many workgroups and the per-op launches   */
/* beyond, unless desc->unfused_readout.  Needs a device.                                                        */
GNNSAFT_API int32_t gnnsaft_readout_resident_workgroups(int32_t hidden, int32_t backward);

/* Optional per-kernel timing: HIP event pairs recorded on the launch stream  */
/* around selected launches of gnnsaft_forward (bench.py's roofline figure).  */
/* `mask` selects the instrumented kernels (GNNSAFT_PROF_*); each instrumented */
/* launch consumes one of `capacity` event pairs.  summary() synchronises on   */
/* the recorded events and returns launch count and summed milliseconds.       */
#define GNNSAFT_PROF_AGGREGATE 1   /* K4 segmented multi-reduce               */
#define GNNSAFT_PROF_UPDATE 2      /* PNAConv update GEMM (scalers on load)   */
#define GNNSAFT_PROF_NODE_TERMS 4  /* message node-term GEMM                  */
#define GNNSAFT_PROF_LIN 8         /* lin GEMM (+ BN partials / epilogue)     */
#define GNNSAFT_PROF_UPDATE_AGG 16 /* fused aggregation + update launch (update_agg.hip; never while bit 1 is asked for) */
#define GNNSAFT_PROF_NUM_KERNELS 5
typedef struct gnnsaft_profile gnnsaft_profile;
GNNSAFT_API int gnnsaft_profile_create(int32_t capacity, uint32_t mask, gnnsaft_profile **out);
GNNSAFT_API void gnnsaft_profile_destroy(gnnsaft_profile *prof);
GNNSAFT_API int gnnsaft_profile_reset(gnnsaft_profile *prof);
GNNSAFT_API int gnnsaft_profile_summary(gnnsaft_profile *prof, uint32_t kernel_bit, int32_t *count, float *total_ms);

/* Optional side stream (+ events) lent to gnnsaft_forward / gnnsaft_backward.  Forward: the structure chain */
/* (CSR, graph ptr, degree tiles, folded weights) runs there, concurrently with the embedding */
/* / edge-table / first message GEMM chain on `stream`, and is joined back before the first   */
/* aggregation; capturable (the two chains become parallel branches of the hipGraph).  One    */
/* handle per concurrently running call; create it on the device it will be used on.          */
/* Backward: see gnnsaft_backward.                                                             */
typedef struct gnnsaft_aux gnnsaft_aux;
GNNSAFT_API int gnnsaft_aux_create(gnnsaft_aux **out);
GNNSAFT_API void gnnsaft_aux_destroy(gnnsaft_aux *aux);

GNNSAFT_API size_t gnnsaft_forward_workspace_bytes(const gnnsaft_model_desc *desc, int64_t num_nodes,
                                       int64_t num_edges, int64_t num_graphs);

GNNSAFT_API int gnnsaft_forward(const gnnsaft_model_desc *desc,
                    const void *const *weights_host, int32_t num_weights,
                    const int64_t *x, const int64_t *edge_index, const int64_t *edge_attr,
                    const int64_t *batch /* NULL => single graph */,
                    int64_t num_nodes, int64_t num_edges, int64_t num_graphs,
                    const float *target /* [G,P] or NULL */,
                    float *out /* [G,P] */, float *loss3 /* [3] or NULL */,
                    int32_t *err_flag /* device int32, zeroed by the caller, or NULL */,
                    void *workspace, size_t workspace_bytes,
                    const void *structure /* from gnnsaft_structure_build, or NULL: build it here */,
                    gnnsaft_profile *profile /* or NULL */,
                    gnnsaft_aux *aux /* or NULL: single stream */, gnnsaft_stream_t stream);

/* The batch STRUCTURE (CSR by destination, graph offsets, degree tiles: everything that depends   */
/* on edge_index / edge_attr / batch only) as one opaque device blob, for batches that are seen     */
/* again (epochs over a fixed batch list, repeated inference): build once, pass to every            */
/* gnnsaft_forward over the same (desc sizes, n, e, g) -- it replaces the K0 chain by one device    */
/* copy.  `workspace` as for gnnsaft_forward.  The blob must come from the same desc->hidden /      */
/* self_loops / bond_dims and the same (num_nodes, num_edges, num_graphs).                          */
GNNSAFT_API size_t gnnsaft_structure_bytes(const gnnsaft_model_desc *desc, int64_t num_nodes, int64_t num_edges,
                               int64_t num_graphs);
GNNSAFT_API int gnnsaft_structure_build(const gnnsaft_model_desc *desc, const int64_t *edge_index,
                            const int64_t *edge_attr, const int64_t *batch, int64_t num_nodes,
                            int64_t num_edges, int64_t num_graphs, void *structure_out,
                            int32_t *err_flag, void *workspace, size_t workspace_bytes,
                            gnnsaft_stream_t stream);

/* ------------------------------------------------------------------------ */
/* Backward of the path (what autograd does when Lightning calls               */
/* loss.backward() after training_step, models.py:191-202).  Run               */
/* gnnsaft_forward with desc->save_tape = 1, fold_degree_scalers = 1,           */
/* fold_dst_term = 0 (training = 1: batch statistics; training = 0: running     */
/* statistics, i.e. fine-tuning behind frozen BatchNorm -- the same value in     */
/* both calls) and keep its workspace (`tape`) untouched; then                  */
/* gnnsaft_backward writes dL/dparam for EVERY parameter into `grads_host`      */
/* (HOST array of device pointers, same order and shapes as `weights_host`;     */
/* entries of buffers -- avg_deg_log, running statistics, counters -- are       */
/* ignored and may be NULL) given grad_out = dL/d(forward output) [G,P].        */
/* Supported: 1 <= pre_layers, post_layers <= 8, hidden % 64 == 0.              */
/* `segment_events` (HOST array of num_layers + 2 hipEvent_t, or NULL): the      */
/* gradients complete in the order readout, layer L-1 .. layer 0, embeddings     */
/* (contiguous segments of the canonical weight order); event i is recorded      */
/* (on `stream`, or on the side stream for the layer segments) when segment i is  */
/* complete, so that a data-parallel caller can all-reduce it on another stream    */
/* under the remaining backward kernels.                                          */
/* `aux` (or NULL: single stream): with pre_layers == post_layers == 1 everything  */
/* off the critical path of the input gradients -- weight / bias gradients, edge-  */
/* class sums, the edge-table chain, the transposed CSR -- runs on the handle's    */
/* side stream, forked from and joined back into `stream` inside the call.         */
/* ------------------------------------------------------------------------ */
GNNSAFT_API size_t gnnsaft_backward_scratch_bytes(const gnnsaft_model_desc *desc, int64_t num_nodes,
                                      int64_t num_edges, int64_t num_graphs);
GNNSAFT_API int gnnsaft_backward(const gnnsaft_model_desc *desc, const void *const *weights_host,
                     void *const *grads_host, int32_t num_weights, const int64_t *x,
                     const int64_t *batch, int64_t num_nodes, int64_t num_edges, int64_t num_graphs,
                     const float *grad_out, void *tape, size_t tape_bytes, void *scratch,
                     size_t scratch_bytes, void *const *segment_events,
                     int32_t *err_flag /* GNNSAFT_FLAG_* word of the forward, or NULL: a lost grid barrier of */
                                       /* the fused readout backward raises bit 16 and poisons the gradients  */,
                     gnnsaft_aux *aux /* or NULL */, gnnsaft_stream_t stream);
/* d(MAPE)/d(pred) * dloss[0] (dloss NULL => 1): sign(p-t) / max(|t|,1.17e-6) / (G*P) */
GNNSAFT_API int gnnsaft_mape_backward(const float *pred, const float *target, int64_t num_graphs,
                          int32_t num_para, const float *dloss, float *dpred,
                          gnnsaft_stream_t stream);
/* tuning aid: gnnsaft_linear_wgrad with an explicit wave grid (wn x wk waves of 64 x 64 outputs, 1 x 1 = the    */
/* 64 x 64 four-wave kernel) and slab count (0 = the library's choice)                                          */
GNNSAFT_API int gnnsaft_debug_linear_wgrad(const float *dy, int64_t ldy, const float *a, int64_t lda, int64_t m,
                               int32_t n_out, int32_t k, float *dw, int64_t ld_dw, void *scratch,
                               size_t scratch_bytes, int32_t wn, int32_t wk, int64_t chunks,
                               gnnsaft_stream_t stream);
/* out[c, :] = sum of the rows of a [m, k] whose class id cls[row] is c (0 <= c < num_classes <= 256; other ids     */
/* contribute nothing): the backward's per-edge-class sums dR = OneHot(class)^T dm (models.py:59,128) as a one-hot  */
/* product on the matrix cores, slab-ordered (bitwise reproducible, no atomics).  mode 0: the library's choice (from  */
/* 16 k rows, <= 64 classes, k % 256 == 0: the streaming three-product bf16 kernel), 1: the f32 one-hot GEMM, 2: the   */
/* streaming kernel wherever its shape rules allow.  scratch: gnnsaft_wgrad_scratch_bytes(m, num_classes, k).          */
GNNSAFT_API int gnnsaft_sum_rows_by_class(const int32_t *cls, int32_t num_classes, const float *a, int64_t lda, int64_t m,
                              int32_t k, float *out, int64_t ld_out, void *scratch, size_t scratch_bytes,
                              int32_t mode, gnnsaft_stream_t stream);
/* dW[n_out,k] (+)= dY^T A (deterministic slab reduction), dbias (+)= column sums of dY  */
GNNSAFT_API size_t gnnsaft_wgrad_scratch_bytes(int64_t m, int32_t n_out, int32_t k);
GNNSAFT_API int gnnsaft_linear_wgrad(const float *dy, int64_t ldy, const float *a, int64_t lda, int32_t relu_a,
                         int64_t m, int32_t n_out, int32_t k, float *dw, int64_t ld_dw,
                         int32_t accumulate, float *dbias, void *scratch, size_t scratch_bytes,
                         gnnsaft_stream_t stream);

/* ------------------------------------------------------------------------ */
/* Per-graph fused EVAL-mode forward (one workgroup per molecular graph, the   */
/* whole network in one launch), in float32 or float64.  Serves the reference's */
/* inference callers, which run the module in float64 / eval mode on one        */
/* un-batched Data at a time (evaluations/evaluate_ensemble.py:67-77,145,185;   */
/* demo/utils.py:23-27,141-152; validation_step models.py:204-211), and the     */
/* single-molecule latency path in float32.                                     */
/*   gnnsaft_eval_pack: the module's parameters (weights_host as for            */
/*     gnnsaft_forward, every float tensor of type `dtype`) -> one PACK: Linears  */
/*     transposed, eval-mode BatchNorm folded, edge branch collapsed to a table   */
/*     per bond-attribute class.  Rebuild whenever a parameter / buffer changes.  */
/*   gnnsaft_graph_forward: out[G,P] (dtype) = PNAPCSAFT.forward(data).eval().    */
/* Supported: hidden a multiple of 32 up to 256; desc->training must be 0.      */
/* ------------------------------------------------------------------------ */
#define GNNSAFT_DTYPE_F32 0
#define GNNSAFT_DTYPE_F64 1
GNNSAFT_API size_t gnnsaft_eval_pack_bytes(const gnnsaft_model_desc *desc, int32_t dtype);
GNNSAFT_API int gnnsaft_eval_pack(const gnnsaft_model_desc *desc, const void *const *weights_host,
                      int32_t num_weights, int32_t dtype, void *pack, size_t pack_bytes,
                      gnnsaft_stream_t stream);
GNNSAFT_API size_t gnnsaft_graph_forward_workspace_bytes(const gnnsaft_model_desc *desc, int32_t dtype,
                                             int64_t num_nodes, int64_t num_edges, int64_t num_graphs);
GNNSAFT_API int gnnsaft_graph_forward(const gnnsaft_model_desc *desc, int32_t dtype, const void *pack,
                          const int64_t *x, const int64_t *edge_index, const int64_t *edge_attr,
                          const int64_t *batch /* NULL => single graph */, int64_t num_nodes,
                          int64_t num_edges, int64_t num_graphs, void *out /* [G,P] of dtype */,
                          int32_t *err_flag, void *workspace, size_t workspace_bytes,
                          gnnsaft_stream_t stream);

/* Debug / test taps: after gnnsaft_forward, byte offsets of intermediate      */
/* tensors inside the workspace (node state after each layer etc.).            */
typedef struct gnnsaft_workspace_map {
  size_t rowptr, src, dst, combo, log_amp, log_att, graph_ptr;
  size_t x_embed;   /* [N,H] node embedding                                    */
  size_t x_final;   /* [N,H] node state after the last layer                   */
  size_t pq, agg, u, y, rtab, pooled;
  size_t total;
  size_t ro;        /* [num_mlp_layers + 2][G, n_out] outputs of the readout's BatchNorm + ReLU blocks (block stride G*H) */
  size_t x_stride;  /* tape (save_tape): x_0 .. x_L contiguous from x_embed, x_stride BYTES apart; 0 without a tape */
  size_t bnstat;    /* [L][2][H] (batch mean | rstd) of the node BatchNorms, as the backward reads them             */
  size_t y_stride;  /* tape: BYTES between the pre-BatchNorm tensors y_l of consecutive layers (from `y`); 0 without */
  size_t ry;        /* [num_mlp_layers + 2][G, n_out] pre-BatchNorm outputs of the readout blocks (block stride G*H)  */
  size_t rstat;     /* [num_mlp_layers + 2][2][H]: per block (batch mean at [c], rstd at [n_out + c])                   */
} gnnsaft_workspace_map;

GNNSAFT_API int gnnsaft_forward_workspace_map(const gnnsaft_model_desc *desc, int64_t num_nodes,
                                  int64_t num_edges, int64_t num_graphs,
                                  gnnsaft_workspace_map *map);

/* ------------------------------------------------------------------------ */
/* Fused optimizer steps over one flat f32 parameter buffer (training-step    */
/* host loop).  Replace torch.optim.AdamW(amsgrad=True, */
/* eps=1e-5) / torch.optim.SGD(nesterov=True) of models.py:162-178; same        */
/* update rule, one launch.  `grad_scale` multiplies the gradient first (1 /    */
/* world size after a SUM all-reduce).  max_exp_avg_sq == NULL: amsgrad off.   */
/* All pointers 16-byte aligned, `count` floats each.  `step` counts from 1.   */
/* ------------------------------------------------------------------------ */
GNNSAFT_API int gnnsaft_adamw_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                       float *max_exp_avg_sq, int64_t count, float lr, float beta1, float beta2,
                       float eps, float weight_decay, int64_t step, float grad_scale,
                       gnnsaft_stream_t stream);
/* The same step with its per-step scalars (learning rate, bias corrections) read from DEVICE memory, for a     */
/* launch that lives in a captured hipGraph: gnnsaft_adamw_args enqueues, on `stream`, the write of the         */
/* gnnsaft_adamw_args_floats() floats of step `step` into DEVICE memory `args_dev`.  The values travel as kernel */
/* arguments (copied when the call returns), so a replay queued behind it reads exactly this step's scalars      */
/* however far the host runs ahead; call it outside the captured region, on the stream the graph is replayed on. */
GNNSAFT_API int32_t gnnsaft_adamw_args_floats(void);
GNNSAFT_API int gnnsaft_adamw_args(float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                       float grad_scale, float *args_dev, gnnsaft_stream_t stream);
GNNSAFT_API int gnnsaft_adamw_step_dev(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                           float *max_exp_avg_sq, int64_t count, const float *args_dev,
                           gnnsaft_stream_t stream);
GNNSAFT_API int gnnsaft_sgd_step(float *param, const float *grad, float *momentum_buf, int64_t count, float lr,
                     float momentum, float weight_decay, int32_t first_step, float grad_scale,
                     gnnsaft_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GNNSAFT_H */
