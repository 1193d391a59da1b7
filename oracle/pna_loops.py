"""Independent float64 per-node / per-edge restatement.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see ``oracle/__init__.py``).  This second restatement is
written from the prose of SURVEY.md Appendix A with explicit Python loops over
nodes and edges and plain numpy dot products; it shares no code with
``oracle/pna_torch.py`` so that the two, written from the same specification of
``/root/reference/gnnepcsaft/train/models.py:105-135,191-194``, can catch each
other's mistakes (concat orders, tower slicing, std clamp / mask, scaler
formulas, empty segments, self-loop placement).  Small inputs only.
"""

from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np

TOWERS = 2


def _lin(sd: Dict[str, np.ndarray], prefix: str, v: np.ndarray) -> np.ndarray:
    return sd[prefix + ".weight"].astype(np.float64) @ v + sd[prefix + ".bias"].astype(np.float64)


def _tower_net(sd, prefix: str, depth: int, v: np.ndarray) -> np.ndarray:
    # Sequential(Linear, [ReLU, Linear] * (depth-1)): Linear layers sit at even indices
    out = _lin(sd, f"{prefix}.0", v)
    for j in range(1, depth):
        out = _lin(sd, f"{prefix}.{2 * j}", np.maximum(out, 0.0))
    return out


def _batchnorm(sd, prefix: str, rows: np.ndarray, training: bool, eps: float = 1e-5) -> np.ndarray:
    g = sd[prefix + ".weight"].astype(np.float64)
    b = sd[prefix + ".bias"].astype(np.float64)
    if training:
        if rows.shape[0] < 2:
            raise ValueError("Expected more than 1 value per channel when training")
        mu = rows.mean(axis=0)
        var = ((rows - mu) ** 2).mean(axis=0)  # biased
    else:
        mu = sd[prefix + ".running_mean"].astype(np.float64)
        var = sd[prefix + ".running_var"].astype(np.float64)
    return (rows - mu) / np.sqrt(var + eps) * g + b


def forward_loops(sd: Dict[str, np.ndarray], x_idx: np.ndarray, edge_index: np.ndarray, edge_attr: np.ndarray,
                  batch: Optional[np.ndarray], *, hidden: int, depth: int, pre_layers: int, post_layers: int,
                  num_mlp_layers: int, skip: bool, self_loops: bool, training: bool) -> np.ndarray:
    """Returns the [G, P] prediction in float64."""
    n = x_idx.shape[0]
    f_in = hidden
    # --- edge list, with loops appended after every real edge (A.5)
    edges = [(int(edge_index[0, e]), int(edge_index[1, e]), [int(a) for a in edge_attr[e]])
             for e in range(edge_index.shape[1])]
    if self_loops:
        edges += [(i, i, [0] * edge_attr.shape[1]) for i in range(n)]
    # --- categorical embeddings (A.1)
    h = np.zeros((n, hidden))
    for i in range(n):
        for k in range(x_idx.shape[1]):
            h[i] += sd[f"node_embed.atom_embedding_list.{k}.weight"][int(x_idx[i, k])].astype(np.float64)
    edge_emb = []
    for (_, _, attr) in edges:
        v = np.zeros(hidden)
        for k, a in enumerate(attr):
            v += sd[f"edge_embed.bond_embedding_list.{k}.weight"][a].astype(np.float64)
        edge_emb.append(v)
    in_edges = [[] for _ in range(n)]
    for eid, (_, d, _) in enumerate(edges):
        in_edges[d].append(eid)

    for l in range(depth):
        cp = f"convs.{l}"
        avg_log = float(sd[cp + ".aggr_module.avg_deg_log"][0])
        new_rows = np.zeros((n, hidden))
        enc = [_lin(sd, cp + ".edge_encoder", ev) for ev in edge_emb]
        for i in range(n):
            towers_out = []
            cnt = len(in_edges[i])
            for t in range(TOWERS):
                ms = []
                for eid in in_edges[i]:
                    s = edges[eid][0]
                    cat3 = np.concatenate([h[i], h[s], enc[eid]])  # destination, source, edge
                    ms.append(_tower_net(sd, f"{cp}.pre_nns.{t}", pre_layers, cat3))
                if cnt == 0:
                    mean = np.zeros(f_in); mn = np.zeros(f_in); mx = np.zeros(f_in); std = np.zeros(f_in)
                else:
                    m = np.stack(ms)
                    mean = m.sum(axis=0) / cnt
                    mn = m.min(axis=0)
                    mx = m.max(axis=0)
                    var = (m * m).sum(axis=0) / cnt - mean * mean
                    std = np.where(var <= 1e-5, 0.0, np.sqrt(np.maximum(var, 1e-5)))
                agg = np.concatenate([mean, mn, mx, std])
                amp = agg * (math.log(cnt + 1) / avg_log)
                att = agg * (avg_log / math.log(max(cnt, 1) + 1))
                z = np.concatenate([h[i], agg, amp, att])
                towers_out.append(_tower_net(sd, f"{cp}.post_nns.{t}", post_layers, z))
            new_rows[i] = _lin(sd, cp + ".lin", np.concatenate(towers_out))
        y = _batchnorm(sd, f"batch_norms.{l}.module", new_rows, training)
        y = np.maximum(y, 0.0)
        h = y + h if skip else y

    # --- add-pool (A.4)
    if batch is None:
        pooled = h.sum(axis=0, keepdims=True)
    else:
        g = int(batch.max()) + 1
        pooled = np.zeros((g, hidden))
        for i in range(n):
            pooled[int(batch[i])] += h[i]

    # --- readout MLP (models.py:84-103)
    z = pooled
    for i in range(num_mlp_layers):
        z = np.stack([_lin(sd, f"mlp.{4 * i}", r) for r in z])
        z = np.maximum(_batchnorm(sd, f"mlp.{4 * i + 1}", z, training), 0.0)
    fp = f"mlp.{4 * num_mlp_layers}"
    z = np.stack([_lin(sd, fp + ".0", r) for r in z])
    z = np.maximum(_batchnorm(sd, fp + ".1", z, training), 0.0)
    z = np.stack([_lin(sd, fp + ".4", r) for r in z])
    z = np.maximum(_batchnorm(sd, fp + ".5", z, training), 0.0)
    z = np.stack([_lin(sd, fp + ".8", r) for r in z])
    return z


def mape_loops(pred: np.ndarray, target: np.ndarray) -> float:
    tot = 0.0
    for p, t in zip(pred.reshape(-1), target.reshape(-1)):
        tot += abs(p - t) / max(abs(t), 1.17e-06)
    return tot / target.size
