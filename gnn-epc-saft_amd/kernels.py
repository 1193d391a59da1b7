"""Per-kernel Python bindings over the C ABI (``include/gnnsaft.h``): torch tensors in,
torch tensors out, every launch on the current HIP stream.  These are the granular entry
points the stage-level parity tests drive; ``PNAPCSAFT.forward`` uses the single
``gnnsaft_forward`` call instead.  CUDA(HIP) tensors only -- nothing here computes on the CPU.
"""

from __future__ import annotations

import ctypes
from typing import Optional, Sequence, Tuple

import torch

from ._native import check, lib


def _stream(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("gnnsaft kernels need HIP device tensors")
    return torch.cuda.current_stream(t.device).cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _i32arr(vals: Sequence[int]):
    return (ctypes.c_int32 * len(vals))(*[int(v) for v in vals])


def _ptrarr(tensors: Sequence[torch.Tensor]):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def csr_build(edge_index: torch.Tensor, edge_attr: torch.Tensor, num_nodes: int, bond_dims: Sequence[int],
              self_loops: bool):
    """-> rowptr[N+1], src[E'], dst[E'], combo[E'] (int32), log_amp[N], log_att[N] (f32), err flag (int32[1])."""
    dev = edge_index.device
    e = int(edge_index.shape[1])
    ep = e + (num_nodes if self_loops else 0)
    i32 = dict(dtype=torch.int32, device=dev)
    rowptr = torch.empty(num_nodes + 1, **i32)
    src, dst, combo = (torch.empty(max(ep, 1), **i32) for _ in range(3))
    log_amp = torch.empty(max(num_nodes, 1), dtype=torch.float32, device=dev)
    log_att = torch.empty_like(log_amp)
    err = torch.zeros(1, **i32)
    nbytes = lib.gnnsaft_csr_workspace_bytes(num_nodes, e)
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
    ws_ptr = (ws.data_ptr() + 255) // 256 * 256
    edge_index = edge_index.contiguous()
    edge_attr = edge_attr.contiguous()
    check(lib.gnnsaft_csr_build(_p(edge_index) if e else None, _p(edge_attr) if e else None, num_nodes, e,
                                len(bond_dims), _i32arr(bond_dims), int(self_loops), _p(rowptr), _p(src), _p(dst),
                                _p(combo), _p(log_amp), _p(log_att), _p(err), ws_ptr, nbytes, _stream(edge_index)),
          "gnnsaft_csr_build")
    return rowptr, src[:ep], dst[:ep], combo[:ep], log_amp[:num_nodes], log_att[:num_nodes], err


def batch_to_ptr(batch: Optional[torch.Tensor], num_nodes: int, num_graphs: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
    ptr = torch.empty(num_graphs + 1, dtype=torch.int32, device=device)
    err = torch.zeros(1, dtype=torch.int32, device=device)
    check(lib.gnnsaft_batch_to_ptr(_p(batch), num_nodes, num_graphs, _p(ptr), _p(err), _stream(ptr)),
          "gnnsaft_batch_to_ptr")
    return ptr, err


def embed_sum(idx: torch.Tensor, tables: Sequence[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    idx = idx.contiguous()
    h = int(tables[0].shape[1])
    out = torch.empty((idx.shape[0], h), dtype=torch.float32, device=idx.device)
    err = torch.zeros(1, dtype=torch.int32, device=idx.device)
    check(lib.gnnsaft_embed_sum(_p(idx), idx.shape[0], len(tables), _ptrarr(tables),
                                _i32arr([t.shape[0] for t in tables]), h, _p(out), _p(err), _stream(idx)),
          "gnnsaft_embed_sum")
    return out, err


def bond_combo_embed(tables: Sequence[torch.Tensor]) -> torch.Tensor:
    h = int(tables[0].shape[1])
    combos = 1
    for t in tables:
        combos *= int(t.shape[0])
    out = torch.empty((combos, h), dtype=torch.float32, device=tables[0].device)
    check(lib.gnnsaft_bond_combo_embed(len(tables), _ptrarr(tables), _i32arr([t.shape[0] for t in tables]), h,
                                       _p(out), _stream(out)), "gnnsaft_bond_combo_embed")
    return out


def bn_rows_per_group() -> int:
    return int(lib.gnnsaft_bn_rows_per_group())


def linear(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], relu_in: bool = False,
           scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None, relu_out: bool = False,
           residual: Optional[torch.Tensor] = None, want_stats: bool = False, tile_config: Optional[int] = None):
    m, k = a.shape
    n_out = w.shape[0]
    out = torch.empty((m, n_out), dtype=torch.float32, device=a.device)
    stats = None
    if want_stats:
        groups = (m + bn_rows_per_group() - 1) // bn_rows_per_group()
        stats = torch.full((groups, 2, n_out), float("nan"), dtype=torch.float32, device=a.device)
    if tile_config is not None:     # test / tuning hook: explicit tile configuration, plain epilogue only
        assert not relu_in and scale is None and shift is None and not relu_out and residual is None
        check(lib.gnnsaft_debug_linear_tile(_p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(out), n_out, m, n_out,
                                            k, _p(stats), int(tile_config), _stream(a)), "gnnsaft_debug_linear_tile")
        return (out, stats) if want_stats else out
    check(lib.gnnsaft_linear(_p(a), a.stride(0), int(relu_in), _p(w), w.stride(0), _p(bias), _p(out), n_out, m, n_out,
                             k, _p(scale), _p(shift), int(relu_out), _p(residual),
                             0 if residual is None else residual.stride(0), _p(stats), _stream(a)), "gnnsaft_linear")
    return (out, stats) if want_stats else out


def w3_pack(w: torch.Tensor) -> torch.Tensor:
    """The "W3" image of a weight matrix [rows, k] (csrc/w3.hpp): bf16 planes of the exact f32 split in the GEMM's
    LDS-stage order.  k % 32 == 0."""
    rows, k = w.shape
    nbytes = lib.gnnsaft_w3_image_bytes(rows, k)
    if nbytes == 0:
        raise ValueError(f"no W3 image for a [{rows}, {k}] matrix (k must be a positive multiple of 32)")
    img = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    w = w if w.stride(1) == 1 else w.contiguous()
    check(lib.gnnsaft_w3_pack(_p(w), w.stride(0), rows, k, _p(img), _stream(w)), "gnnsaft_w3_pack")
    return img


def linear_w3(a: torch.Tensor, w_image: torch.Tensor, n_out: int, bias: Optional[torch.Tensor], tile_config: int,
              want_stats: bool = False, specialised: bool = False):
    """out = a W^T + bias on the split-bf16 kernels that read pre-split weights (csrc/gemm_w3.hip; ``specialised``:
    the consumer / producer wave form of csrc/gemm_w3s.hip); test / tuning hook."""
    m, k = a.shape
    out = torch.empty((m, n_out), dtype=torch.float32, device=a.device)
    stats = None
    if want_stats:
        groups = (m + bn_rows_per_group() - 1) // bn_rows_per_group()
        stats = torch.full((groups, 2, n_out), float("nan"), dtype=torch.float32, device=a.device)
    fn = lib.gnnsaft_debug_linear_w3s if specialised else lib.gnnsaft_debug_linear_w3
    check(fn(_p(a), a.stride(0), _p(w_image), _p(bias), _p(out), n_out, m, n_out, k, _p(stats), int(tile_config),
             _stream(a)), "gnnsaft_debug_linear_w3s" if specialised else "gnnsaft_debug_linear_w3")
    return (out, stats) if want_stats else out


def sum_rows_by_class(cls: torch.Tensor, num_classes: int, a: torch.Tensor, mode: int = 0) -> torch.Tensor:
    """out[c] = sum of the rows of ``a`` whose int32 class id is c (csrc/gemm_tn.hip; mode 0 = the library's choice,
    1 = the f32 one-hot GEMM, 2 = the streaming three-product bf16 kernel); test / tuning hook."""
    m, k = a.shape
    out = torch.empty((num_classes, k), dtype=torch.float32, device=a.device)
    need = max(lib.gnnsaft_wgrad_scratch_bytes(m, num_classes, k), 1024)
    scratch = torch.empty(need // 4 + 64, dtype=torch.float32, device=a.device)
    check(lib.gnnsaft_sum_rows_by_class(_p(cls), num_classes, _p(a), a.stride(0), m, k, _p(out), k, _p(scratch),
                                        scratch.numel() * 4, int(mode), _stream(a)), "gnnsaft_sum_rows_by_class")
    return out


def linear_ar(a: torch.Tensor, w_image: torch.Tensor, n_out: int, bias: Optional[torch.Tensor], tile_config: int):
    """out = a W^T + bias on the split-bf16 kernel that keeps the A operand in registers (csrc/gemm_ar.hip); test /
    tuning hook."""
    m, k = a.shape
    out = torch.empty((m, n_out), dtype=torch.float32, device=a.device)
    check(lib.gnnsaft_debug_linear_ar(_p(a), a.stride(0), _p(w_image), _p(bias), _p(out), n_out, m, n_out, k,
                                      int(tile_config), _stream(a)), "gnnsaft_debug_linear_ar")
    return out


def pna_node_terms(x: torch.Tensor, w_pre0: torch.Tensor, w_pre1: torch.Tensor) -> torch.Tensor:
    n, h = x.shape
    pq = torch.empty((n, 4 * h), dtype=torch.float32, device=x.device)
    check(lib.gnnsaft_pna_node_terms(_p(x), n, h, _p(w_pre0), _p(w_pre1), _p(pq), _stream(x)),
          "gnnsaft_pna_node_terms")
    return pq


def pna_edge_table(combo_emb, w_edge, b_edge, w_pre0, b_pre0, w_pre1, b_pre1) -> torch.Tensor:
    c, h = combo_emb.shape
    tmp = torch.empty((c, h), dtype=torch.float32, device=combo_emb.device)
    rtab = torch.empty((c, 2 * h), dtype=torch.float32, device=combo_emb.device)
    check(lib.gnnsaft_pna_edge_table(_p(combo_emb), c, h, _p(w_edge), _p(b_edge), _p(w_pre0), _p(b_pre0), _p(w_pre1),
                                     _p(b_pre1), _p(tmp), _p(rtab), _stream(tmp)), "gnnsaft_pna_edge_table")
    return rtab


def pna_edge_mlp(src, dst, combo, pq, rtab, w2_t0, b2_t0, w2_t1, b2_t1) -> torch.Tensor:
    rows = src.shape[0]
    h = rtab.shape[1] // 2
    msgs = torch.empty((rows, 2 * h), dtype=torch.float32, device=pq.device)
    check(lib.gnnsaft_pna_edge_mlp(_p(src), _p(dst), _p(combo), rows, h, _p(pq), _p(rtab), _p(w2_t0), _p(b2_t0),
                                   _p(w2_t1), _p(b2_t1), _p(msgs), _stream(pq)), "gnnsaft_pna_edge_mlp")
    return msgs


def pna_aggregate(rowptr, src, combo, hidden: int, pq=None, rtab=None, msgs=None) -> torch.Tensor:
    n = rowptr.shape[0] - 1
    agg = torch.empty((n, 2, 4 * hidden), dtype=torch.float32, device=rowptr.device)
    check(lib.gnnsaft_pna_aggregate(_p(rowptr), _p(src), _p(combo), n, hidden, _p(pq), _p(rtab), _p(msgs), _p(agg),
                                    _stream(agg)), "gnnsaft_pna_aggregate")
    return agg


def pna_update(x, agg, log_amp, log_att, avg_deg_log, w_post0, b_post0, w_post1, b_post1) -> torch.Tensor:
    n, h = x.shape
    u = torch.empty((n, h), dtype=torch.float32, device=x.device)
    check(lib.gnnsaft_pna_update(_p(x), _p(agg), _p(log_amp), _p(log_att), _p(avg_deg_log), n, h, _p(w_post0),
                                 _p(b_post0), _p(w_post1), _p(b_post1), _p(u), _stream(x)), "gnnsaft_pna_update")
    return u


def bn_finalize(stats, num_rows: int, gamma, beta, running_mean, running_var, num_batches_tracked, momentum: float,
                eps: float, training: bool):
    ch = gamma.shape[0]
    scale = torch.empty(ch, dtype=torch.float32, device=gamma.device)
    shift = torch.empty_like(scale)
    check(lib.gnnsaft_bn_finalize(_p(stats), num_rows, ch, _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                                  _p(num_batches_tracked), momentum, eps, int(training), _p(scale), _p(shift),
                                  _stream(gamma)), "gnnsaft_bn_finalize")
    return scale, shift


def bn_relu_residual(y, scale, shift, residual=None) -> torch.Tensor:
    out = torch.empty_like(y)
    check(lib.gnnsaft_bn_relu_residual(_p(y), _p(scale), _p(shift), _p(residual), _p(out), y.shape[0], y.shape[1],
                                       _stream(y)), "gnnsaft_bn_relu_residual")
    return out


def add_pool(x, graph_ptr) -> torch.Tensor:
    g = graph_ptr.shape[0] - 1
    out = torch.empty((g, x.shape[1]), dtype=torch.float32, device=x.device)
    check(lib.gnnsaft_add_pool(_p(x), _p(graph_ptr), g, x.shape[0], x.shape[1], _p(out), _stream(x)),
          "gnnsaft_add_pool")
    return out


def mape(pred, target) -> torch.Tensor:
    out = torch.empty(3, dtype=torch.float32, device=pred.device)
    check(lib.gnnsaft_mape(_p(pred.contiguous()), _p(target.contiguous()), pred.numel(), _p(out), _stream(pred)),
          "gnnsaft_mape")
    return out


def degree_tiles(rowptr: torch.Tensor, hidden: int):
    """-> perm[N], tiles[cap,4], num_tiles[1], hist3[3*buckets] (first `buckets` ints: degree histogram), err."""
    n = rowptr.shape[0] - 1
    dev = rowptr.device
    cap = int(lib.gnnsaft_degree_tiles_capacity(n, hidden))
    buckets = int(lib.gnnsaft_degree_buckets())
    perm = torch.empty(n, dtype=torch.int32, device=dev)
    tiles = torch.zeros((cap, 4), dtype=torch.int32, device=dev)
    num_tiles = torch.zeros(1, dtype=torch.int32, device=dev)
    hist3 = torch.zeros(int(lib.gnnsaft_degree_scratch_ints(n)), dtype=torch.int32, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    check(lib.gnnsaft_degree_tiles(_p(rowptr), n, hidden, _p(perm), _p(tiles), _p(num_tiles), _p(hist3), _p(err),
                                   _stream(rowptr)), "gnnsaft_degree_tiles")
    return perm, tiles, num_tiles, hist3, err


def pna_update_folded(x, agg, perm, tiles, num_tiles, hist3, avg_deg_log, w_post0, b_post0, w_post1, b_post1):
    n, h = x.shape
    buckets = int(lib.gnnsaft_degree_buckets())
    w_eff = torch.full((buckets, 2, h // 2, 5 * h), float("nan"), dtype=torch.float32, device=x.device)
    check(lib.gnnsaft_pna_fold_post_weights(_p(w_post0), _p(w_post1), _p(avg_deg_log), _p(hist3), h, _p(w_eff),
                                            _stream(x)), "gnnsaft_pna_fold_post_weights")
    u = torch.empty((n, h), dtype=torch.float32, device=x.device)
    check(lib.gnnsaft_pna_update_folded(_p(x), _p(agg), _p(perm), _p(tiles), _p(num_tiles), n, h, _p(w_eff),
                                        _p(b_post0), _p(b_post1), _p(u), _stream(x)), "gnnsaft_pna_update_folded")
    return u


def pna_update_agg(x, q, rtab, rowptr, src, combo, perm, tiles, num_tiles, hist3, avg_deg_log, w_post0, b_post0, w_post1,
                   b_post1):
    """Aggregation + degree-folded update in ONE launch (csrc/update_agg.hip): what ``pna_aggregate_src`` followed by
    ``pna_update_folded`` computes, without the aggregates ever reaching HBM.  (The folded weights are packed into W3
    images block by block here; ``gnnsaft_forward`` gets them from the fold kernel directly.)"""
    n, h = x.shape
    buckets = int(lib.gnnsaft_degree_buckets())
    w_eff = torch.zeros((buckets, 2, h // 2, 5 * h), dtype=torch.float32, device=x.device)
    check(lib.gnnsaft_pna_fold_post_weights(_p(w_post0), _p(w_post1), _p(avg_deg_log), _p(hist3), h, _p(w_eff),
                                            _stream(x)), "gnnsaft_pna_fold_post_weights")
    images = torch.cat([w3_pack(w_eff[d, t]) for d in range(buckets) for t in range(2)])
    u = torch.empty((n, h), dtype=torch.float32, device=x.device)
    check(lib.gnnsaft_pna_update_agg(_p(x), _p(q), _p(rtab), rtab.shape[0], _p(rowptr), _p(src), _p(combo), _p(perm), _p(tiles),
                                     _p(num_tiles), n, h, _p(images), _p(b_post0), _p(b_post1), _p(u), _stream(x)),
          "gnnsaft_pna_update_agg")
    return u


def pna_update_folded_ar(x, agg, perm, tiles, num_tiles, hist3, avg_deg_log, w_post0, b_post0, w_post1, b_post1):
    """The degree-folded update on the A-in-registers GEMM with both towers per workgroup (csrc/gemm_ar.hip): what
    ``pna_update_folded`` computes; hidden 128 / 256."""
    n, h = x.shape
    buckets = int(lib.gnnsaft_degree_buckets())
    w_eff = torch.zeros((buckets, 2, h // 2, 5 * h), dtype=torch.float32, device=x.device)
    check(lib.gnnsaft_pna_fold_post_weights(_p(w_post0), _p(w_post1), _p(avg_deg_log), _p(hist3), h, _p(w_eff),
                                            _stream(x)), "gnnsaft_pna_fold_post_weights")
    images = torch.cat([w3_pack(w_eff[d, t]) for d in range(buckets) for t in range(2)])
    u = torch.empty((n, h), dtype=torch.float32, device=x.device)
    check(lib.gnnsaft_pna_update_folded_ar(_p(x), _p(agg), _p(perm), _p(tiles), _p(num_tiles), n, h, _p(images),
                                           _p(b_post0), _p(b_post1), _p(u), _stream(x)), "gnnsaft_pna_update_folded_ar")
    return u


def bn_train_apply(stats, y, gamma, beta, running_mean, running_var, num_batches_tracked, momentum: float, eps: float,
                   residual=None) -> torch.Tensor:
    out = torch.empty_like(y)
    need = lib.gnnsaft_bn_train_scratch_bytes(y.shape[0], y.shape[1])
    scratch = torch.empty(max(need, 8) // 8, dtype=torch.float64, device=y.device)
    check(lib.gnnsaft_bn_train_apply(_p(stats), _p(y), y.shape[0], y.shape[1], _p(gamma), _p(beta), _p(running_mean),
                                     _p(running_var), _p(num_batches_tracked), momentum, eps, _p(residual), _p(out),
                                     None, _p(scratch), need, _stream(y)), "gnnsaft_bn_train_apply")
    return out


def pna_src_terms(x: torch.Tensor, w_pre0: torch.Tensor, w_pre1: torch.Tensor) -> torch.Tensor:
    n, h = x.shape
    q = torch.empty((n, 2 * h), dtype=torch.float32, device=x.device)
    check(lib.gnnsaft_pna_src_terms(_p(x), n, h, _p(w_pre0), _p(w_pre1), _p(q), _stream(x)), "gnnsaft_pna_src_terms")
    return q


def pna_aggregate_src(rowptr, src, combo, hidden: int, q, rtab) -> torch.Tensor:
    n = rowptr.shape[0] - 1
    agg = torch.empty((n, 2, 4 * hidden), dtype=torch.float32, device=rowptr.device)
    check(lib.gnnsaft_pna_aggregate_src(_p(rowptr), _p(src), _p(combo), n, hidden, _p(q), _p(rtab), _p(agg),
                                        _stream(agg)), "gnnsaft_pna_aggregate_src")
    return agg


def pna_update_folded_dst(x, agg_src, perm, tiles, num_tiles, hist3, avg_deg_log, w_post0, b_post0, w_post1, b_post1,
                          w_pre0, w_pre1):
    """Degree-folded update with the destination term folded in: `agg_src` comes from pna_aggregate_src."""
    n, h = x.shape
    buckets = int(lib.gnnsaft_degree_buckets())
    w_eff = torch.full((buckets, 2, h // 2, 5 * h), float("nan"), dtype=torch.float32, device=x.device)
    g = torch.empty((2, 3, h // 2, h), dtype=torch.float64, device=x.device)
    arr = lambda t: (ctypes.c_void_p * 1)(t.data_ptr())
    check(lib.gnnsaft_pna_fold_post_weights_multi(1, arr(w_post0), arr(w_post1), arr(avg_deg_log), arr(w_pre0),
                                                  arr(w_pre1), _p(g), _p(hist3), h, _p(w_eff), 0, _stream(x)),
          "gnnsaft_pna_fold_post_weights_multi")
    u = torch.empty((n, h), dtype=torch.float32, device=x.device)
    check(lib.gnnsaft_pna_update_folded(_p(x), _p(agg_src), _p(perm), _p(tiles), _p(num_tiles), n, h, _p(w_eff),
                                        _p(b_post0), _p(b_post1), _p(u), _stream(x)), "gnnsaft_pna_update_folded")
    return u
