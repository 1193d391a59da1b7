#!/usr/bin/env python3
"""GPU probe (test infrastructure, imports oracle/): where does a 1e-3 gradient error on a message-weight tensor come
from when the forward output agrees with the f64 oracle to 1e-5?  For the smoke configuration (H=128 L=3, 64 graphs)
it lists, per layer, the std entries that the HIP forward masks differently from the f64 oracle (PyG zeroes std where
var <= 1e-5) and the rows of convs.l.pre_nns.t.0.weight that carry the gradient error.

    python tests/analysis_gradient_flips_gpu.py [first_seed]"""
import copy
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gnn_epc_saft_amd._native import WorkspaceMap, lib  # noqa: E402
from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch  # noqa: E402
from gnn_epc_saft_amd.train.models import mape_loss  # noqa: E402
from helpers import oracle_model  # noqa: E402
from oracle.pna_torch import mape  # noqa: E402
from test_gpu_forward import hip_twin  # noqa: E402

DEV = "cuda:0"
H, L, G = 128, 3, 64


def main():
    for attempt in range(32):
        data = make_synthetic_batch(G, 900 + H + L + 1000 * attempt, num_para=3)
        oracle = oracle_model(H, L, 1, 1, 1, 3, True, True, degree_histogram(data), seed=L).train()
        with torch.no_grad():
            want64, want32 = copy.deepcopy(oracle).double()(data), copy.deepcopy(oracle)(data)
            probe = hip_twin(copy.deepcopy(oracle))
            probe.fold_dst_term = False
            got = probe(data.to(DEV)).cpu()
        scale = float(want64.abs().max())
        if max(float((want32.double() - want64).abs().max()), float((got.double() - want64).abs().max())) <= 1e-5 * scale:
            break
    print(f"batch seed {900 + H + L + 1000 * attempt} (attempt {attempt})")
    st64 = {}
    o64 = copy.deepcopy(oracle).double().train()
    mape(o64(data, st64), data.para.view(-1, 3).double()).backward()
    g64 = {k: p.grad.detach() for k, p in o64.named_parameters()}
    o32 = copy.deepcopy(oracle).train()
    mape(o32(data), data.para.view(-1, 3)).backward()
    g32 = {k: p.grad.detach().double() for k, p in o32.named_parameters()}
    hip = hip_twin(copy.deepcopy(oracle))
    dd = data.to(DEV)
    pred = hip(dd)
    tape = pred.grad_fn.tape
    n, e, g = tape["n"], tape["e"], tape["g"]
    wmap = WorkspaceMap()
    assert lib.gnnsaft_forward_workspace_map(ctypes.byref(tape["desc"]), n, e, g, ctypes.byref(wmap)) == 0
    base = tape["ws_ptr"] - tape["ws"].data_ptr()
    agg_all = tape["ws"][base + wmap.agg: base + wmap.agg + 4 * L * n * 8 * H].view(torch.float32).view(L, n, 2, 4 * H).cpu()
    mape_loss(pred, dd.para.view(-1, 3)).backward()
    gs = max(float(v.abs().max()) for v in g64.values())
    for layer in range(L):
        s64 = st64[f"l{layer}.agg"][..., 3 * H:]
        sh = agg_all[layer][..., 3 * H:].double()
        flips = ((s64 > 0) != (sh > 0)).nonzero()
        both = (s64 > 0) & (sh > 0)
        rel = ((sh - s64).abs() / s64.clamp(min=1e-30)).masked_fill(~both, 0)
        print(f"layer {layer}: HIP masks {flips.shape[0]} std entries differently from f64; max rel std error elsewhere {float(rel.max()):.1e}")
        for nd, t, f in flips.tolist()[:8]:
            m64 = st64[f"l{layer}.msgs"]
            print(f"    node {nd} tower {t} feature {f}: std f64 {float(s64[nd, t, f]):.6e} (var {float(s64[nd, t, f]) ** 2:.6e}), hip {float(sh[nd, t, f]):.6e}")
        for t in range(2):
            k = f"convs.{layer}.pre_nns.{t}.0.weight"
            gh = dict(hip.named_parameters())[k].grad.double().cpu()
            d, d32 = (gh - g64[k]).abs(), (g32[k] - g64[k]).abs()
            sc = max(float(g64[k].abs().max()), 1e-4 * gs)
            rows = (d.max(1).values > 0.2 * d.max()).nonzero().flatten().tolist()
            print(f"    {k}: hip {float(d.max()) / sc:.1e} (f32 oracle {float(d32.max()) / sc:.1e}); rows holding > 20% of the max error: {rows[:10]}")


if __name__ == "__main__":
    main()
