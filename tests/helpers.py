"""Shared test plumbing: model pairs (HIP module + CPU oracle with one state_dict),
the parity metric, and small hand-built graphs."""

from __future__ import annotations

import copy
from typing import Dict, Optional

import numpy as np
import torch

from gnn_epc_saft_amd.data.synthetic import (GraphData, collate, degree_histogram, ethanol_all_atom, ethanol_heavy,
                                             make_synthetic_batch)
from oracle.pna_torch import OracleMlpParams, OraclePNAPCSAFT, OraclePnaParams


def oracle_model(hidden, depth, pre, post, mlp, num_para, skip, loops, deg, seed=0, dtype=torch.float32):
    torch.manual_seed(seed)
    m = OraclePNAPCSAFT(hidden, OraclePnaParams(depth, pre, post, deg, skip_connections=skip, self_loops=loops),
                        OracleMlpParams(mlp, num_para))
    randomize_norm_state(m, seed)
    return m.to(dtype)


def randomize_norm_state(model: torch.nn.Module, seed: int) -> None:
    """Default BatchNorm state (gamma=1, beta=0, mean=0, var=1) hides bugs in the affine /
    running-statistics plumbing: perturb it deterministically."""
    g = torch.Generator().manual_seed(1000 + seed)
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            with torch.no_grad():
                mod.weight.copy_(1.0 + 0.2 * torch.randn(mod.weight.shape, generator=g))
                mod.bias.copy_(0.1 * torch.randn(mod.bias.shape, generator=g))
                mod.running_mean.copy_(0.3 * torch.randn(mod.running_mean.shape, generator=g))
                mod.running_var.copy_(0.5 + torch.rand(mod.running_var.shape, generator=g))


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a-b| / max|b|: error relative to the scale of the reference tensor."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    scale = float(b.abs().max())
    if scale == 0.0:
        return float((a - b).abs().max())
    return float((a - b).abs().max()) / scale


def gate_err(a: torch.Tensor, b: torch.Tensor, per_row: bool = False, floor_rel: float = 1e-6):
    """The parity gate of SURVEY.md section 8(d), per ELEMENT: |a - b| / max(|b|, 1e-6 * max|b|).  Returns the
    maximum over all elements (or, with ``per_row``, the per-row maxima as a float64 tensor).  Unlike ``rel_err``
    it does not let small-magnitude outputs hide behind the largest one.  ``floor_rel``: the denominator's floor as a
    fraction of max|b| (1e-6 = the gate as stated)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    floor = floor_rel * float(b.abs().max()) if b.numel() else 0.0
    e = (a - b).abs() / b.abs().clamp(min=max(floor, 1e-300))
    if per_row:
        return e.reshape(e.shape[0], -1).amax(dim=1)
    return float(e.max()) if e.numel() else 0.0


TOL = 1e-5   # north_star / SURVEY 8(d)
FLIP = 5e-3  # what one std-threshold flip may move an element of a graph's output (relative to that element)


def check_population(out, want32, want64) -> str:
    """The SURVEY 8(d) gate |a - b| / max(|b|, 1e-6 max|b|) per element, maximised per graph, of the HIP output against
    the exact (f64) oracle -- judged against the same figure of the reference arithmetic evaluated in f32 (the f32
    oracle), because no f32 evaluation meets 1e-5 PER ELEMENT on an arbitrary batch:

      * outputs pass through zero: an element of 1/20 of the output scale carrying the usual 1e-6-of-scale f32 error
        is already 2e-5 off relative to itself (measured: the f32 oracle keeps 8 % .. 99 % of the graphs of a random
        96-graph batch within 1e-5, depending on depth and BatchNorm mode);
      * PyG's StdAggregation zeroes std where var <= 1e-5: a segment whose variance lies within rounding of that
        threshold is masked differently by ANY two evaluations; one such flip moves its graph by 1e-4 .. 1e-3.

    Bars: the 50 % and 90 % quantiles within 3x the f32 oracle's (or 1e-5, whichever is larger); the fraction of graphs
    within 1e-5 not below the f32 oracle's by more than sampling noise; the worst graph within 3x the f32 oracle's worst
    graph (or one flip).  Returns the printed summary."""
    err_h = gate_err(out, want64, per_row=True)
    err_o = gate_err(want32, want64, per_row=True)
    qs = torch.tensor([0.5, 0.9, 0.99, 1.0], dtype=torch.float64)
    qh, qo = torch.quantile(err_h, qs), torch.quantile(err_o, qs)
    frac_h, frac_o = float((err_h <= TOL).float().mean()), float((err_o <= TOL).float().mean())
    margin = max(0.05, 2.5 * (0.5 / max(int(err_h.numel()), 1)) ** 0.5)   # two binomial samples of n graphs
    msg = (f"per-element gate |a-b|/max(|b|,1e-6 max|b|), per-graph quantiles (50/90/99/100%) hip "
           f"{['%.1e' % v for v in qh.tolist()]} f32-oracle {['%.1e' % v for v in qo.tolist()]}; within {TOL}: hip "
           f"{frac_h:.4f} f32-oracle {frac_o:.4f}; scale-relative max {rel_err(out, want64):.1e} (f32 oracle "
           f"{rel_err(want32, want64):.1e})")
    print(msg)
    assert bool((qh[:2] <= torch.clamp(3 * qo[:2], min=TOL)).all()), msg
    assert frac_h >= min(0.99, frac_o - margin), msg
    # FROZEN as of round 3: no further escape hatch without a failing case committed first.
    if float(qh[3]) > max(FLIP, 3 * float(qo[3])):
        # The single worst element of a batch is usually an output passing through zero: |b| ~ 1e-5 of the output scale
        # turns the ordinary 2e-6-of-scale f32 error into 0.2 "relative", on WHICHEVER evaluation happens to hold the
        # larger absolute error there (the f32 oracle draws from the same lottery: its own worst graphs are 1e-2).
        # Such an excess is accepted only if it disappears once outputs below 1e-3 of the scale are judged against
        # 1e-3 of the scale (everything larger keeps its own magnitude) AND the largest absolute error stays within
        # the f32 oracle's -- a flipped decision or a wrong term would survive both.
        # The case that needs it is pinned: tests/test_gpu_forward.py::test_zero_crossing_case_behind_the_coarse_gate
        # (round 3 tried to drop this gate after forming all folded weights in float64: that case still draws a worst
        # element of 0.11 relative with an ABSOLUTE error below the f32 oracle's).
        coarse_h = float(gate_err(out, want64, per_row=True, floor_rel=1e-3).max())
        coarse_o = float(gate_err(want32, want64, per_row=True, floor_rel=1e-3).max())
        assert coarse_h <= max(FLIP, 3 * coarse_o) and rel_err(out, want64) <= max(3 * rel_err(want32, want64), 1e-6), \
            msg + f"; coarse gate (floor 1e-3 of scale) hip {coarse_h:.1e} f32-oracle {coarse_o:.1e}"
    return msg


def tape_std_masks(pred: torch.Tensor):
    """Per layer, which std entries the TAPED HIP forward behind ``pred`` left unmasked ([L] bool tensors [N,2,F]),
    read from the tape the autograd node keeps (the workspace of gnnsaft_forward with save_tape = 1)."""
    import ctypes

    from gnn_epc_saft_amd._native import WorkspaceMap, lib
    tape = pred.grad_fn.tape
    desc, n, e, g = tape["desc"], tape["n"], tape["e"], tape["g"]
    wmap = WorkspaceMap()
    assert lib.gnnsaft_forward_workspace_map(ctypes.byref(desc), n, e, g, ctypes.byref(wmap)) == 0
    base = tape["ws_ptr"] - tape["ws"].data_ptr()
    h, layers = desc.hidden, desc.num_layers
    agg = tape["ws"][base + wmap.agg: base + wmap.agg + 4 * layers * n * 8 * h].view(torch.float32)
    return [(a[..., 3 * h:] > 0).cpu() for a in agg.view(layers, n, 2, 4 * h)]


def oracle_decisions(model: torch.nn.Module, data, skip: bool):
    """The discrete decisions of one oracle evaluation (no grad): per layer the std mask and the ReLU gate of the node
    update (the sign of the BatchNorm output the ReLU sees), then the ReLU gates of the readout blocks.  Returns
    (stages, decisions)."""
    stages: Dict[str, torch.Tensor] = {}
    gates, node_gates = [], []
    hooks = [m.register_forward_hook(lambda mod, inp, out: gates.append(out.detach() > 0))
             for m in model.mlp.modules() if isinstance(m, torch.nn.ReLU)]
    hooks += [m.register_forward_hook(lambda mod, inp, out: node_gates.append(out.detach() > 0))
              for m in model.batch_norms]
    try:
        with torch.no_grad():
            model(data, stages)
    finally:
        for hk in hooks:
            hk.remove()
    dec = []
    layer = 0
    while f"l{layer}.agg" in stages:
        a = stages[f"l{layer}.agg"]
        f = a.shape[-1] // 4
        dec.append(a[..., 3 * f:] > 0)
        dec.append(node_gates[layer])
        layer += 1
    return stages, dec + gates


def tape_decisions(pred: torch.Tensor, skip: bool):
    """The same list of decisions as ``oracle_decisions``, read from the tape of the HIP forward behind ``pred``: std
    masks from the aggregates; node ReLU gates as the kernels take them, relu(y * scale + shift) > 0 with scale = rstd
    gamma, shift = beta - mean scale in float32 (csrc/bn_fold.hpp; forward and backward use this one expression --
    x_{l+1} - x_l > 0 would lose a gate whose activation the residual add absorbs); readout gates from the block
    outputs."""
    import ctypes

    from gnn_epc_saft_amd._native import WorkspaceMap, lib
    tape = pred.grad_fn.tape
    desc, n, e, g = tape["desc"], tape["n"], tape["e"], tape["g"]
    wmap = WorkspaceMap()
    assert lib.gnnsaft_forward_workspace_map(ctypes.byref(desc), n, e, g, ctypes.byref(wmap)) == 0
    base = tape["ws_ptr"] - tape["ws"].data_ptr()
    h, layers = desc.hidden, desc.num_layers

    def tap(off, count):
        return tape["ws"][base + off: base + off + 4 * count].view(torch.float32)

    agg = tap(wmap.agg, layers * n * 8 * h).view(layers, n, 2, 4 * h)
    ys = tap(wmap.y, layers * n * h).view(layers, n, h).cpu()
    stat = tap(wmap.bnstat, layers * 2 * h).view(layers, 2, h).cpu()
    module = pred.grad_fn.module
    dec = []
    for layer in range(layers):
        dec.append((agg[layer][..., 3 * h:] > 0).cpu())
        bn = module.batch_norms[layer].module
        scale = stat[layer, 1] * bn.weight.detach().float().cpu()
        shift = bn.bias.detach().float().cpu() - stat[layer, 0] * scale
        dec.append((ys[layer] * scale + shift) > 0)
    widths = [h] * desc.num_mlp_layers + [h // 2, h // 4]
    bns = [m for m in module.mlp.modules() if isinstance(m, torch.nn.BatchNorm1d)]
    rstat = tap(wmap.rstat, len(widths) * 2 * h).view(len(widths), 2 * h).cpu()
    for b, w in enumerate(widths):     # readout gates: the same expression on the blocks' pre-BatchNorm tensors
        y = tap(wmap.ry + 4 * b * g * h, g * w).view(g, w).cpu()
        scale = rstat[b, w:2 * w] * bns[b].weight.detach().float().cpu()
        shift = bns[b].bias.detach().float().cpu() - rstat[b, :w] * scale
        dec.append((y * scale + shift) > 0)
    return dec


def tape_dropout_masks(pred: torch.Tensor):
    """Keep-masks of the readout's Dropout layers as the taped HIP forward behind ``pred`` drew them: block outputs that
    are exactly zero although their ReLU gate is open were dropped (a closed gate's mask is unobservable and
    irrelevant: value and gradient are zero either way -- reported as kept)."""
    import ctypes

    from gnn_epc_saft_amd._native import WorkspaceMap, lib
    tape = pred.grad_fn.tape
    desc, n, e, g = tape["desc"], tape["n"], tape["e"], tape["g"]
    wmap = WorkspaceMap()
    assert lib.gnnsaft_forward_workspace_map(ctypes.byref(desc), n, e, g, ctypes.byref(wmap)) == 0
    base = tape["ws_ptr"] - tape["ws"].data_ptr()
    h = desc.hidden
    widths = [h] * desc.num_mlp_layers + [h // 2, h // 4]
    gates = tape_decisions(pred, True)[-len(widths):]
    masks = []
    for b, w in enumerate(widths):
        off = base + wmap.ro + 4 * b * g * h
        ro = tape["ws"][off: off + 4 * g * w].view(torch.float32).view(g, w).cpu()
        masks.append((ro != 0) | ~gates[b])
    return masks


def _csr_order(data, loops: bool):
    """Permutation of the oracle's edge list (real edges, then the appended self-loops) into the HIP path's CSR row
    order: by destination, edge-list order inside a destination, self-loop last."""
    n = data.x.shape[0]
    dst = data.edge_index[1]
    if loops:
        dst = torch.cat([dst, torch.arange(n)])
    return torch.argsort(dst, stable=True), dst


def oracle_routing(stages, data, loops: bool):
    """Per layer, which message rows attain the min / the max of their segment ([E', T*F] bool, CSR row order): the
    routing of the min / max aggregators' gradient."""
    order, dst = _csr_order(data, loops)
    out, layer = [], 0
    n = data.x.shape[0]
    while f"l{layer}.msgs" in stages:
        m = stages[f"l{layer}.msgs"]
        m = m.reshape(m.shape[0], -1)[order]
        seg = dst[order].view(-1, 1).expand_as(m)
        for red in ("amin", "amax"):
            ext = torch.zeros((n, m.shape[1]), dtype=m.dtype).scatter_reduce_(0, seg, m, reduce=red, include_self=False)
            out.append(m == ext.gather(0, seg))
        layer += 1
    return out


def tape_routing(pred: torch.Tensor):
    """The same masks for the taped HIP forward (pre_layers == 1): the float32 messages are recomputed on the CPU from
    the tape exactly as k_pna_aggregate / k_agg_bwd form them, (P[dst] + Q[src]) + R[class], in CSR row order."""
    import ctypes

    from gnn_epc_saft_amd._native import WorkspaceMap, lib
    tape = pred.grad_fn.tape
    desc, n, e, g = tape["desc"], tape["n"], tape["e"], tape["g"]
    assert desc.pre_layers == 1
    wmap = WorkspaceMap()
    assert lib.gnnsaft_forward_workspace_map(ctypes.byref(desc), n, e, g, ctypes.byref(wmap)) == 0
    base = tape["ws_ptr"] - tape["ws"].data_ptr()
    h, layers = desc.hidden, desc.num_layers
    ep = e + (n if desc.self_loops else 0)
    combos = 1
    for k in range(desc.num_bond_cols):
        combos *= desc.bond_dims[k]

    def tap(off, count, dtype):
        return tape["ws"][base + off: base + off + 4 * count].view(dtype).cpu()

    rowptr = tap(wmap.rowptr, n + 1, torch.int32).long()
    src = tap(wmap.src, ep, torch.int32).long()
    combo = tap(wmap.combo, ep, torch.int32).long()
    dst = torch.repeat_interleave(torch.arange(n), rowptr[1:] - rowptr[:-1])
    pq = tap(wmap.pq, layers * n * 4 * h, torch.float32).view(layers, n, 4 * h)
    rtab = tap(wmap.rtab, layers * combos * 2 * h, torch.float32).view(layers, combos, 2 * h)
    out = []
    for layer in range(layers):
        m = (pq[layer][dst, :2 * h] + pq[layer][src, 2 * h:]) + rtab[layer][combo]
        seg = dst.view(-1, 1).expand_as(m)
        for red in ("amin", "amax"):
            ext = torch.zeros((n, 2 * h)).scatter_reduce_(0, seg, m, reduce=red, include_self=False)
            out.append(m == ext.gather(0, seg))
    return out


def decisions_agree(a, b) -> bool:
    """Every discrete decision (std mask, ReLU gate, min / max routing) taken alike.  A gradient comparison is well
    defined only between evaluations that agree here: one std entry masked differently moves a row of the layer's
    message weights by ~1e-3 (1/std = 316 at the threshold), one ReLU gate flipped or one min / max gradient routed
    to the other of two nearly tied messages moves a row of a weight gradient by ~1/N of its scale (1e-2 on a 24..64
    graph batch; at f32 resolution a 64-graph H=128 batch holds a few such near-ties per layer) --
    tests/analysis_gradient_flips_gpu.py."""
    return len(a) == len(b) and all(torch.equal(x.cpu(), y.cpu()) for x, y in zip(a, b))


def std_masks_agree(stages_or_masks, stages64: Dict[str, torch.Tensor]) -> bool:
    """PyG's StdAggregation zeroes std where var <= 1e-5: a discrete decision per (node, tower, feature).  True when
    an evaluation (oracle ``stages`` dict, or the list ``tape_std_masks`` returns) took every one of them like the
    float64 oracle did.  A gradient comparison is only well defined between evaluations that agree here: one
    differently masked entry moves a whole row of that layer's message weights by ~1e-3 through the 1/std factor of
    d std / d m (profiles/r02_std_variance_gradient_analysis.txt)."""
    layer = 0
    while f"l{layer}.agg" in stages64:
        a64 = stages64[f"l{layer}.agg"]
        f = a64.shape[-1] // 4
        m64 = a64[..., 3 * f:] > 0
        if isinstance(stages_or_masks, dict):
            m = stages_or_masks[f"l{layer}.agg"][..., 3 * f:] > 0
        else:
            m = stages_or_masks[layer]
        if not torch.equal(m.cpu(), m64.cpu()):
            return False
        layer += 1
    return True


def mini4() -> GraphData:
    """4 graphs: a ring of 5, a single isolated atom (0 edges), a 2-atom molecule, a branched 6-atom tree."""
    g = torch.Generator().manual_seed(7)

    def feats(n):
        return torch.stack([torch.randint(0, d, (n,), generator=g) for d in (119, 5, 12, 12, 10, 6, 6, 2, 2)], 1)

    def graph(n, bonds):
        src, dst, attr = [], [], []
        for a, b in bonds:
            t = [int(torch.randint(0, d, (1,), generator=g)) for d in (5, 6, 2)]
            src += [a, b]
            dst += [b, a]
            attr += [t, t]
        ei = torch.tensor([src, dst], dtype=torch.int64).reshape(2, -1)
        ea = torch.tensor(attr, dtype=torch.int64).reshape(-1, 3)
        return GraphData(feats(n), ei, ea, para=torch.rand(5, generator=g) * 4.5 + 0.5)

    ring = graph(5, [(0, 1), (1, 2), (2, 3), (3, 4), (4, 0)])
    lone = graph(1, [])
    pair = graph(2, [(0, 1)])
    tree = graph(6, [(0, 1), (0, 2), (0, 3), (3, 4), (3, 5)])
    return collate([ring, lone, pair, tree])


def to_numpy_state(model: torch.nn.Module) -> Dict[str, np.ndarray]:
    return {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}


# ---------------------------------------------------------------------------------------------------------------------
# Branch-forced evaluation of the oracle.  The network is piecewise smooth: std masks (var <= 1e-5), ReLU gates and the
# min / max routing are discrete decisions.  On a small batch one can search for a seed on which every evaluation takes
# them alike; at BASELINE size (1024 graphs: ~1e7 decisions) some always differ.  `forced_forward` evaluates the
# oracle's own modules ON A GIVEN BRANCH -- the decisions are inputs, read off the HIP tape -- so that the HIP
# gradients can be compared with the exact (f64) gradients of the very function the HIP forward evaluated, at any size.
# Checked against the free oracle in tests/test_oracle_cpu.py (decisions taken from the oracle itself: same output,
# same gradients to 1e-12).  pre_layers = post_layers = 1 (no ReLU inside the towers).
# ---------------------------------------------------------------------------------------------------------------------
def branch_of_oracle(model: torch.nn.Module, data, skip: bool, loops: bool):
    """The branch the (free) oracle takes on ``data``: dict(std=[L], gate=[L], amin=[L], amax=[L], ro=[blocks]) with the
    routing masks in the ORACLE's edge order (real edges, then self-loops)."""
    stages, dec = oracle_decisions(model, data, skip)
    layers = len(model.convs)
    std, gate = [dec[2 * l] for l in range(layers)], [dec[2 * l + 1] for l in range(layers)]
    ro = dec[2 * layers:]
    order, _ = _csr_order(data, loops)
    inv = torch.empty_like(order)
    inv[order] = torch.arange(order.numel())
    routing = oracle_routing(stages, data, loops)                  # CSR order
    amin = [routing[2 * l][inv] for l in range(layers)]
    amax = [routing[2 * l + 1][inv] for l in range(layers)]
    return dict(std=std, gate=gate, amin=amin, amax=amax, ro=list(ro))


def branch_of_tape(pred: torch.Tensor, data, skip: bool, loops: bool):
    """The branch the taped HIP forward behind ``pred`` took, in the same form."""
    dec = tape_decisions(pred, skip)
    layers = pred.grad_fn.tape["desc"].num_layers
    std, gate = [dec[2 * l] for l in range(layers)], [dec[2 * l + 1] for l in range(layers)]
    ro = dec[2 * layers:]
    order, _ = _csr_order(data, loops)
    inv = torch.empty_like(order)
    inv[order] = torch.arange(order.numel())
    routing = tape_routing(pred)
    amin = [routing[2 * l][inv] for l in range(layers)]
    amax = [routing[2 * l + 1][inv] for l in range(layers)]
    return dict(std=std, gate=gate, amin=amin, amax=amax, ro=list(ro))


def branch_differences(a, b) -> Dict[str, int]:
    """Number of decisions two branches take differently, per kind (+ the total number of decisions)."""
    out = {}
    total = 0
    for key in ("std", "gate", "amin", "amax", "ro"):
        out[key] = int(sum(int((x.cpu() != y.cpu()).sum()) for x, y in zip(a[key], b[key])))
        total += sum(x.numel() for x in a[key])
    out["decisions"] = total
    return out


def forced_forward(model: torch.nn.Module, data, branch, dropout_masks=None) -> torch.Tensor:
    """The oracle's forward (oracle/pna_torch.py ``OraclePNAPCSAFT.forward``, same modules, same op order) with every
    discrete decision taken from ``branch`` instead of from the data: differentiable, float32 or float64."""
    from oracle.pna_torch import add_self_loops, global_add_pool, pna_scale, scatter_mean
    pp = model.pna_params
    assert pp.pre_layers == 1 and pp.post_layers == 1 and pp.dropout == 0.0
    x, edge_index, edge_attr = data.x, data.edge_index, data.edge_attr
    batch = getattr(data, "batch", None)
    if pp.self_loops:
        edge_index, edge_attr = add_self_loops(edge_index, edge_attr, x.size(0))
    x = model.node_embed(x)
    edge_emb = model.edge_embed(edge_attr)
    n = x.size(0)
    dst = edge_index[1]
    for l, (conv, bn) in enumerate(zip(model.convs, model.batch_norms)):
        x_prev = x
        xr = x.view(n, 1, conv.f_in).repeat(1, 2, 1)
        msgs = conv.messages(x, edge_index, edge_emb)                     # [E', T, F]
        mean = scatter_mean(msgs, dst, n)
        parts = [mean]
        for key in ("amin", "amax"):                                      # forced routing, ties share evenly (as torch)
            sel = branch[key][l].view_as(msgs).to(msgs.dtype)
            bidx = dst.view(-1, 1, 1).expand_as(msgs)
            cnt = torch.zeros_like(mean).scatter_add_(0, bidx, sel)
            parts.append(torch.zeros_like(mean).scatter_add_(0, bidx, msgs * sel) / cnt.clamp(min=1))
        var = scatter_mean(msgs * msgs, dst, n) - mean * mean
        keep = branch["std"][l].view_as(var)
        parts.append(torch.where(keep, var.clamp(min=1e-12).sqrt(), torch.zeros_like(var)))
        agg = torch.cat(parts, dim=-1)
        z = torch.cat([xr, pna_scale(agg, dst, n, conv.aggr_module.avg_deg_log)], dim=-1)
        u = torch.cat([net(z[:, t]) for t, net in enumerate(conv.post_nns)], dim=1)
        x = bn(conv.lin(u)) * branch["gate"][l].to(x.dtype)               # forced ReLU gate
        if pp.skip_connections:
            x = x + x_prev
    g = global_add_pool(x, batch)
    gates = iter(branch["ro"])
    drops = iter(dropout_masks) if dropout_masks is not None else None

    def run(seq, v):
        for mod in seq:
            if isinstance(mod, torch.nn.Sequential):
                v = run(mod, v)
            elif isinstance(mod, torch.nn.ReLU):
                v = v * next(gates).to(v.dtype)
            elif isinstance(mod, torch.nn.Dropout):
                if drops is None:
                    assert mod.p == 0.0
                else:     # the masks the HIP forward drew, scaled as torch's Dropout scales: x * mask / (1 - p)
                    keep = torch.tensor(1.0, dtype=torch.float32) / (torch.tensor(1.0, dtype=torch.float32) - mod.p)
                    v = v * next(drops).to(v.dtype) * keep.to(v.dtype)
            else:
                v = mod(v)
        return v

    return run(model.mlp, g)
