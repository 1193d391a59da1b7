#!/usr/bin/env python3
"""GPU probe (test infrastructure, imports oracle/): WHICH STAGE owns the distance of the golden fixture `mini4_b`
(H=64, L=2, pre=2, post=3, mlp=0, P=5; four graphs) to the f64 oracle in TRAIN mode, where the HIP path measured 2.8e-4
per element against the f32 oracle's 1.5e-4?  Walks the taped forward's
workspace -- node state after every layer, pooled rows, the readout's pre-BatchNorm tensors and block outputs, the
prediction -- and prints, per stage, the scale-relative and per-element errors of the HIP tensor and of the f32 oracle's
against the f64 oracle's, plus the per-column batch variance the 4-row BatchNorm divides by.

    python tests/analysis_mini4b_stages_gpu.py [case]       (default mini4_b)"""
import copy
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gnn_epc_saft_amd._native import WorkspaceMap, lib  # noqa: E402
from golden_util import fill_deterministic, load_case  # noqa: E402
from helpers import gate_err, rel_err  # noqa: E402
from oracle.pna_torch import OracleMlpParams, OraclePNAPCSAFT, OraclePnaParams  # noqa: E402
from test_gpu_forward import graph_data, hip_twin  # noqa: E402

DEV = "cuda:0"


def oracle_stages(model, data):
    """node state after every layer, pooled rows, every readout Linear's output (pre-BatchNorm), every block output"""
    stages, tensors = {}, {}
    hooks = []
    lin_i, blk_i = [0], [0]

    def on_linear(_m, _i, o):
        tensors[f"readout.y{lin_i[0]}"] = o.detach()
        lin_i[0] += 1

    def on_relu(_m, _i, o):
        tensors[f"readout.o{blk_i[0]}"] = o.detach()
        blk_i[0] += 1

    for m in model.mlp.modules():
        if isinstance(m, torch.nn.Linear):
            hooks.append(m.register_forward_hook(on_linear))
        elif isinstance(m, torch.nn.ReLU):
            hooks.append(m.register_forward_hook(on_relu))
    with torch.no_grad():
        out = model(data, stages)
    for h in hooks:
        h.remove()
    for k, v in stages.items():
        if k.endswith(".out") or k in ("embed", "pooled"):
            tensors[k] = v.detach()
    tensors["prediction"] = out.detach()
    return tensors


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "mini4_b"
    case = load_case(name)
    hidden, depth, pre, post, mlp, num_para, skip, loops = (int(v) for v in case["config"])
    oracle = OraclePNAPCSAFT(hidden, OraclePnaParams(depth, pre, post, torch.from_numpy(case["deg"]),
                                                     skip_connections=bool(skip), self_loops=bool(loops)),
                             OracleMlpParams(mlp, num_para))
    fill_deterministic(oracle, int(case["seed"][0]))
    oracle.train()
    data = graph_data(case)
    t64 = oracle_stages(copy.deepcopy(oracle).double(), data)
    t32 = oracle_stages(copy.deepcopy(oracle), data)
    hip = hip_twin(copy.deepcopy(oracle))
    pred = hip(data.to(DEV))                     # grad mode: the workspace is the tape
    torch.cuda.synchronize()
    tape = pred.grad_fn.tape
    desc, n, e, g = tape["desc"], tape["n"], tape["e"], tape["g"]
    wmap = WorkspaceMap()
    assert lib.gnnsaft_forward_workspace_map(ctypes.byref(desc), n, e, g, ctypes.byref(wmap)) == 0
    base = tape["ws_ptr"] - tape["ws"].data_ptr()
    tap = lambda off, cnt: tape["ws"][base + off: base + off + 4 * cnt].view(torch.float32).cpu()
    h = hidden
    got = {"embed": tap(wmap.x_embed, n * h).view(n, h)}
    for layer in range(depth):
        got[f"l{layer}.out"] = tap(wmap.x_embed + (layer + 1) * wmap.x_stride, n * h).view(n, h)
    got["pooled"] = tap(wmap.pooled, g * h).view(g, h)
    widths = [h] * mlp + [h // 2, h // 4]
    for b, w in enumerate(widths):
        got[f"readout.y{b}"] = tap(wmap.ry + 4 * b * g * h, g * w).view(g, w)
        got[f"readout.o{b}"] = tap(wmap.ro + 4 * b * g * h, g * w).view(g, w)
    got["prediction"] = pred.detach().cpu()
    print(f"{name} train mode ({g} graphs, {n} nodes; H={hidden} L={depth} pre={pre} post={post} mlp={mlp}): per stage, "
          "error against the f64 oracle -- scale-relative max | per-element gate max")
    print(f"{'stage':14s} {'HIP scale-rel':>14s} {'f32 scale-rel':>14s} {'HIP gate':>10s} {'f32 gate':>10s}   min column variance over the 4 rows (f64)")
    for key in ["embed"] + [f"l{i}.out" for i in range(depth)] + ["pooled"] + \
            [f"readout.{c}{b}" for b in range(len(widths)) for c in "yo"] + ["prediction"]:
        if key not in t64 or key not in got:
            continue
        w64, w32, hv = t64[key], t32[key].double(), got[key].double()
        extra = ""
        if key.startswith("readout.y"):
            var = w64.var(dim=0, unbiased=False)
            extra = f"   {float(var.min()):.2e} (max {float(var.max()):.2e})"
        print(f"{key:14s} {rel_err(hv, w64):14.2e} {rel_err(w32, w64):14.2e} {gate_err(hv, w64):10.2e} {gate_err(w32, w64):10.2e}{extra}")


if __name__ == "__main__":
    main()
