"""Generates the golden vectors under ``tests/golden`` from the CPU oracle.

PARITY UNPINNED: the reference itself cannot be run (see ``oracle/__init__.py``),
so these vectors are outputs of the oracle restatement, cross-checked against the
independent float64 loop restatement (``oracle/pna_loops.py``) at generation time.
Whoever has PyG + ogb can replay them through the true reference: the fixtures hold
inputs, the constructor arguments, and the weights are a closed-form hash
(``tests/golden_util.fill_deterministic``) keyed by the reference's own state_dict
names.

Run from the repository root:  python tests/make_golden.py
"""

from __future__ import annotations

import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from golden_util import fill_deterministic, save_case  # noqa: E402
from helpers import mini4  # noqa: E402
from oracle import pna_loops  # noqa: E402
from oracle.pna_torch import (OracleMlpParams, OraclePNAPCSAFT, OraclePnaParams, mape, scatter_mean)  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "gnn-epc-saft_amd", "data"))
import synthetic  # noqa: E402  (imported as a plain module: no HIP library needed to make fixtures)

CASES = {
    # name: (graph builder, hidden, depth, pre, post, mlp, P, skip, loops, modes)
    "ethanol3_default": (synthetic.ethanol_heavy, 64, 6, 1, 1, 1, 5, True, True, ("eval",)),
    "ethanol9_default": (synthetic.ethanol_all_atom, 64, 6, 1, 1, 1, 5, True, True, ("eval",)),
    "mini4_a": (mini4, 64, 2, 1, 1, 1, 3, True, True, ("eval", "train")),
    "mini4_b": (mini4, 64, 2, 2, 3, 0, 5, False, False, ("eval", "train")),
    "mini4_c": (mini4, 32, 3, 3, 2, 2, 5, True, False, ("eval", "train")),
    "synth4_c2model": (lambda: synthetic.make_synthetic_batch(4, 99, num_para=3), 128, 3, 1, 1, 1, 3, True, True,
                       ("eval", "train")),
}


def flip_margin(stages, edge_dst, n):
    """Smallest distance of any segment variance to the 1e-5 std threshold, relative to the
    rounding noise an f32 evaluation of mean(m^2) - mean(m)^2 carries."""
    worst = np.inf
    for k, msgs in stages.items():
        if not k.endswith(".msgs"):
            continue
        mean = scatter_mean(msgs, edge_dst, n)
        msq = scatter_mean(msgs * msgs, edge_dst, n)
        var = msq - mean * mean
        noise = 6e-8 * (msq + 1e-30) * 8
        cnt = torch.bincount(edge_dst, minlength=n).view(-1, 1, 1)
        ratio = ((var - 1e-5).abs() / noise)[(cnt > 1).expand_as(var)]
        if ratio.numel():
            worst = min(worst, float(ratio.min()))
    return worst


def build(name):
    """Tries weight seeds until no segment variance of any layer / mode sits near the
    std threshold, so that the fixture asserts a 1e-5 parity that is well defined."""
    for attempt in range(64):
        try:
            return _build(name, zlib_seed(name) + attempt)
        except NearThreshold as exc:
            print(f"{name}: seed attempt {attempt} rejected ({exc})")
    raise RuntimeError(f"{name}: no threshold-free seed found")


class NearThreshold(Exception):
    pass


def _build(name, seed):
    builder, hidden, depth, pre, post, mlp, num_para, skip, loops, modes = CASES[name]
    data = builder()
    deg = synthetic.degree_histogram(data)
    if deg.sum() == 0 or deg.numel() < 2:
        deg = torch.tensor([0, 2, 1])
    model = OraclePNAPCSAFT(hidden, OraclePnaParams(depth, pre, post, deg, skip_connections=skip, self_loops=loops),
                            OracleMlpParams(mlp, num_para))
    checksum = fill_deterministic(model, seed=seed)
    out = {
        "x": data.x.numpy(), "edge_index": data.edge_index.numpy(), "edge_attr": data.edge_attr.numpy(),
        "deg": deg.numpy(),
        "config": np.array([hidden, depth, pre, post, mlp, num_para, int(skip), int(loops)], dtype=np.int64),
        "weights_checksum": np.array([checksum]), "seed": np.array([seed], dtype=np.int64),
    }
    if data.batch is not None:
        out["batch"] = data.batch.numpy()
    para = data.para.reshape(-1, 5 if data.para.numel() % 5 == 0 else num_para)[:, :num_para].contiguous() \
        if data.para is not None else None
    if para is not None:
        out["para"] = para.numpy()
    n = data.x.shape[0]
    for mode in modes:
        m32 = copy.deepcopy(model).train(mode == "train")
        m64 = copy.deepcopy(model).double().train(mode == "train")
        stages = {}
        with torch.no_grad():
            o32 = m32(data)
            o64 = m64(data, stages)
        sd64 = {k: v.numpy() for k, v in copy.deepcopy(model).double().state_dict().items()}
        loops_out = pna_loops.forward_loops(sd64, out["x"], out["edge_index"], out["edge_attr"], out.get("batch"),
                                            hidden=hidden, depth=depth, pre_layers=pre, post_layers=post,
                                            num_mlp_layers=mlp, skip=skip, self_loops=loops,
                                            training=(mode == "train"))
        agree = float(np.abs(loops_out - o64.numpy()).max() / np.abs(o64.numpy()).max())
        assert agree < 1e-12, (name, mode, agree)
        dst = data.edge_index[1]
        if loops:
            dst = torch.cat([dst, torch.arange(n)])
        margin = flip_margin(stages, dst, n)
        if margin <= (1.0 if name.startswith("synth") else 4.0):
            raise NearThreshold(f"{mode}: margin {margin:.2f}")
        out[f"out_{mode}_f32"] = o32.numpy()
        out[f"out_{mode}_f64"] = o64.numpy()
        for key in ("embed", "l0.agg", "l0.post", "l0.conv", "l0.out", "pooled"):
            if not name.startswith("synth") or key == "pooled":
                out[f"{mode}.{key}"] = stages[key].numpy()
        if para is not None and o64.shape[0] * num_para == para.numel():
            out[f"loss_{mode}_f64"] = np.array([float(mape(o64, para.double().view(-1, num_para)))])
        if mode == "train":
            bn0 = m64.batch_norms[0].module
            out["train.bn0_running_mean"] = bn0.running_mean.numpy()
            out["train.bn0_running_var"] = bn0.running_var.numpy()
            tail_bn = m64.mlp[4 * mlp][5]
            out["train.tail_bn_running_mean"] = tail_bn.running_mean.numpy()
            out["train.tail_bn_running_var"] = tail_bn.running_var.numpy()
        print(f"{name:18s} {mode:5s} loops-vs-torch {agree:.1e}  flip margin {margin:9.1f}  "
              f"f32-vs-f64 {float((o32.double() - o64).abs().max() / o64.abs().max()):.1e}")
    path = save_case(name, out)
    print("   wrote", path, os.path.getsize(path), "bytes")


def zlib_seed(name: str) -> int:
    import zlib
    return zlib.crc32(name.encode()) % 100000


if __name__ == "__main__":
    torch.set_num_threads(4)
    for case in CASES:
        build(case)
