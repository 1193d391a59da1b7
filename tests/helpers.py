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


def gate_err(a: torch.Tensor, b: torch.Tensor, per_row: bool = False):
    """The parity gate of SURVEY.md section 8(d), per ELEMENT: |a - b| / max(|b|, 1e-6 * max|b|).  Returns the
    maximum over all elements (or, with ``per_row``, the per-row maxima as a float64 tensor).  Unlike ``rel_err``
    it does not let small-magnitude outputs hide behind the largest one."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    floor = 1e-6 * float(b.abs().max()) if b.numel() else 0.0
    e = (a - b).abs() / b.abs().clamp(min=max(floor, 1e-300))
    if per_row:
        return e.reshape(e.shape[0], -1).amax(dim=1)
    return float(e.max()) if e.numel() else 0.0


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
