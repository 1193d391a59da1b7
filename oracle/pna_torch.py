"""Vectorised CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see ``oracle/__init__.py``): the reference cannot be imported
here, and ships no golden vectors; this file restates, op for op, what
``/root/reference/gnnepcsaft/train/models.py:105-135`` (``PNAPCSAFT.forward``)
and ``models.py:191-194`` (MAPE training loss) compute, including the
third-party pieces (PyG ``PNAConv`` / ``DegreeScalerAggregation`` / ``BatchNorm``
/ ``global_add_pool`` / ``add_self_loops``, ogb ``AtomEncoder`` /
``BondEncoder``, torchmetrics MAPE) whose semantics are written down in
SURVEY.md Appendix A.  It issues the same sequence of dense ops PyG issues
(``index_select``, ``repeat``, ``cat``, ``addmm``, ``scatter_add_``,
``scatter_reduce_`` ... and materialises the ``[N,T,13F]`` update input), so it
doubles as the "CPU restatement (PyG-equivalent op sequence)" baseline that
``bench.py`` times next to the MI355X number.

Module / parameter / buffer names follow SURVEY.md Appendix C so that one
``state_dict`` loads into the reference, into this oracle and into the HIP
module alike.
"""

from __future__ import annotations

import dataclasses
import math
from typing import Dict, List, Optional, Sequence

import torch
from torch import nn

# ogb >= 1.3 vocabulary sizes
ATOM_FEATURE_DIMS = (119, 5, 12, 12, 10, 6, 6, 2, 2)
BOND_FEATURE_DIMS = (5, 6, 2)
TOWERS = 2  # models.py:76
MAPE_EPS = 1.17e-06  # torchmetrics, Appendix A.6


# --------------------------------------------------------------------------
# parameter records (models.py:28-45)
# --------------------------------------------------------------------------
@dataclasses.dataclass
class OraclePnaParams:
    propagation_depth: int
    pre_layers: int
    post_layers: int
    deg: torch.Tensor
    dropout: float = 0.0
    skip_connections: bool = False
    self_loops: bool = False


@dataclasses.dataclass
class OracleMlpParams:
    num_mlp_layers: int
    num_para: int
    dropout: float = 0.0


# --------------------------------------------------------------------------
# third-party building blocks, restated
# --------------------------------------------------------------------------
class _CategoricalSum(nn.Module):
    """ogb Atom/BondEncoder: sum over columns of one nn.Embedding per column
    (Appendix A.1), xavier-uniform tables, left-to-right accumulation."""

    def __init__(self, list_name: str, dims: Sequence[int], emb_dim: int):
        super().__init__()
        tables = nn.ModuleList()
        for d in dims:
            emb = nn.Embedding(int(d), emb_dim)
            nn.init.xavier_uniform_(emb.weight.data)
            tables.append(emb)
        self._list_name = list_name
        setattr(self, list_name, tables)

    def forward(self, idx: torch.Tensor) -> torch.Tensor:
        tables = getattr(self, self._list_name)
        out = 0
        for k in range(idx.shape[1]):
            out = out + tables[k](idx[:, k])
        return out


class _DegreeScalerBuffers(nn.Module):
    """Holds PyG DegreeScalerAggregation's two buffers (Appendix A.2)."""

    def __init__(self, deg: torch.Tensor):
        super().__init__()
        d = deg.to(torch.float)
        total = int(d.sum())
        bins = torch.arange(d.numel())
        lin = float((bins * d).sum()) / total
        log = float(((bins + 1).log() * d).sum()) / total
        self.register_buffer("avg_deg_lin", torch.full((1,), lin))
        self.register_buffer("avg_deg_log", torch.full((1,), log))


def _tower_mlp(n_in: int, n_hidden: int, depth: int) -> nn.Sequential:
    mods: List[nn.Module] = [nn.Linear(n_in, n_hidden)]
    for _ in range(depth - 1):
        mods += [nn.ReLU(), nn.Linear(n_hidden, n_hidden)]
    return nn.Sequential(*mods)


def scatter_mean(src: torch.Tensor, index: torch.Tensor, dim_size: int) -> torch.Tensor:
    """PyG ``scatter(..., reduce='mean')`` on dim 0 (Appendix A.2 step 3)."""
    count = src.new_zeros(dim_size)
    count.scatter_add_(0, index, src.new_ones(src.size(0)))
    count = count.clamp(min=1)
    bidx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    out = src.new_zeros((dim_size,) + tuple(src.shape[1:])).scatter_add_(0, bidx, src)
    return out / count.view(-1, *([1] * (src.dim() - 1)))


def scatter_minmax(src: torch.Tensor, index: torch.Tensor, dim_size: int, which: str) -> torch.Tensor:
    bidx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    out = src.new_zeros((dim_size,) + tuple(src.shape[1:]))
    if src.numel() == 0:
        return out
    return out.scatter_reduce_(0, bidx, src, reduce=which, include_self=False)


def pna_aggregate(msgs: torch.Tensor, dst: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """mean | min | max | std over in-edges -> [N,T,4F]  (Appendix A.2 step 3)."""
    mean = scatter_mean(msgs, dst, num_nodes)
    mn = scatter_minmax(msgs, dst, num_nodes, "amin")
    mx = scatter_minmax(msgs, dst, num_nodes, "amax")
    mean_sq = scatter_mean(msgs * msgs, dst, num_nodes)
    var = mean_sq - mean * mean
    std = var.clamp(min=1e-5).sqrt()
    std = std.masked_fill(std <= math.sqrt(1e-5), 0.0)
    return torch.cat([mean, mn, mx, std], dim=-1)


def pna_scale(agg: torch.Tensor, dst: torch.Tensor, num_nodes: int, avg_deg_log: torch.Tensor) -> torch.Tensor:
    """identity | amplification | attenuation -> [N,T,12F] (Appendix A.2 step 4)."""
    deg = agg.new_zeros(num_nodes).scatter_add_(0, dst, agg.new_ones(dst.numel()))
    deg = deg.view(-1, 1, 1)
    amp = agg * (torch.log(deg + 1) / avg_deg_log)
    att = agg * (avg_deg_log / torch.log(deg.clamp(min=1) + 1))
    return torch.cat([agg, amp, att], dim=-1)


class OraclePNAConv(nn.Module):
    """PyG PNAConv(H, H, [mean,min,max,std], [identity,amplification,attenuation],
    deg, edge_dim=H, towers=2, pre_layers=p, post_layers=q, divide_input=False)
    as constructed at models.py:69-80."""

    def __init__(self, hidden: int, deg: torch.Tensor, pre_layers: int, post_layers: int):
        super().__init__()
        self.f_in = hidden
        self.f_out = hidden // TOWERS
        self.aggr_module = _DegreeScalerBuffers(deg)
        self.edge_encoder = nn.Linear(hidden, self.f_in)
        self.pre_nns = nn.ModuleList(_tower_mlp(3 * self.f_in, self.f_in, pre_layers) for _ in range(TOWERS))
        self.post_nns = nn.ModuleList(_tower_mlp(13 * self.f_in, self.f_out, post_layers) for _ in range(TOWERS))
        self.lin = nn.Linear(hidden, hidden)

    def messages(self, x: torch.Tensor, edge_index: torch.Tensor, edge_emb: torch.Tensor) -> torch.Tensor:
        n = x.size(0)
        xr = x.view(n, 1, self.f_in).repeat(1, TOWERS, 1)
        x_j = xr.index_select(0, edge_index[0])  # source
        x_i = xr.index_select(0, edge_index[1])  # destination
        ee = self.edge_encoder(edge_emb).view(-1, 1, self.f_in).repeat(1, TOWERS, 1)
        h = torch.cat([x_i, x_j, ee], dim=-1)
        return torch.stack([net(h[:, t]) for t, net in enumerate(self.pre_nns)], dim=1)

    def forward(self, x, edge_index, edge_emb, stages: Optional[Dict[str, torch.Tensor]] = None):
        n = x.size(0)
        xr = x.view(n, 1, self.f_in).repeat(1, TOWERS, 1)
        msgs = self.messages(x, edge_index, edge_emb)
        agg = pna_aggregate(msgs, edge_index[1], n)
        scaled = pna_scale(agg, edge_index[1], n, self.aggr_module.avg_deg_log)
        z = torch.cat([xr, scaled], dim=-1)  # [N,T,13F]
        u = torch.cat([net(z[:, t]) for t, net in enumerate(self.post_nns)], dim=1)
        out = self.lin(u)
        if stages is not None:
            stages["msgs"] = msgs
            stages["agg"] = agg
            stages["post"] = u
            stages["conv"] = out
        return out


class _NodeBatchNorm(nn.Module):
    """PyG BatchNorm: a BatchNorm1d stored as attribute ``module`` (Appendix A.3)."""

    def __init__(self, hidden: int):
        super().__init__()
        self.module = nn.BatchNorm1d(hidden)

    def forward(self, x):
        return self.module(x)


def add_self_loops(edge_index: torch.Tensor, edge_attr: torch.Tensor, num_nodes: int):
    """PyG add_self_loops(edge_index, edge_attr, 0, num_nodes) (Appendix A.5)."""
    loop = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device).repeat(2, 1)
    ei = torch.cat([edge_index, loop], dim=1)
    ea = torch.cat([edge_attr, edge_attr.new_full((num_nodes,) + tuple(edge_attr.shape[1:]), 0)], dim=0)
    return ei, ea


def global_add_pool(x: torch.Tensor, batch: Optional[torch.Tensor]) -> torch.Tensor:
    """PyG global_add_pool (Appendix A.4)."""
    if batch is None:
        return x.sum(dim=0, keepdim=True)
    size = int(batch.max()) + 1
    out = x.new_zeros(size, x.size(1))
    return out.scatter_add_(0, batch.view(-1, 1).expand_as(x), x)


def mape(preds: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """torchmetrics mean_absolute_percentage_error (Appendix A.6)."""
    ape = torch.abs(preds - target) / torch.clamp(torch.abs(target), min=MAPE_EPS)
    return ape.sum() / target.numel()


# --------------------------------------------------------------------------
# the network (models.py:48-135)
# --------------------------------------------------------------------------
class OraclePNAPCSAFT(nn.Module):
    def __init__(self, hidden_dim: int, pna_params: OraclePnaParams, mlp_params: OracleMlpParams,
                 atom_dims: Sequence[int] = ATOM_FEATURE_DIMS, bond_dims: Sequence[int] = BOND_FEATURE_DIMS):
        super().__init__()
        self.pna_params = pna_params
        self.mlp_params = mlp_params
        self.convs = nn.ModuleList()          # registration order of models.py:62-66 (fixes parameters() order)
        self.batch_norms = nn.ModuleList()
        self.node_embed = _CategoricalSum("atom_embedding_list", atom_dims, hidden_dim)
        self.edge_embed = _CategoricalSum("bond_embedding_list", bond_dims, hidden_dim)
        for _ in range(pna_params.propagation_depth):
            self.convs.append(OraclePNAConv(hidden_dim, pna_params.deg, pna_params.pre_layers, pna_params.post_layers))
            self.batch_norms.append(_NodeBatchNorm(hidden_dim))
        h = hidden_dim
        self.mlp = nn.Sequential()
        for _ in range(mlp_params.num_mlp_layers):
            self.mlp.append(nn.Linear(h, h))
            self.mlp.append(nn.BatchNorm1d(h))
            self.mlp.append(nn.ReLU())
            self.mlp.append(nn.Dropout(p=mlp_params.dropout))
        self.mlp.append(nn.Sequential(
            nn.Linear(h, h // 2), nn.BatchNorm1d(h // 2), nn.ReLU(), nn.Dropout(p=mlp_params.dropout),
            nn.Linear(h // 2, h // 4), nn.BatchNorm1d(h // 4), nn.ReLU(), nn.Dropout(p=mlp_params.dropout),
            nn.Linear(h // 4, mlp_params.num_para),
        ))

    def forward(self, data, stages: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
        x = data.x
        edge_index = data.edge_index
        edge_attr = data.edge_attr
        batch = getattr(data, "batch", None)
        if self.pna_params.self_loops:
            edge_index, edge_attr = add_self_loops(edge_index, edge_attr, x.size(0))
        x = self.node_embed(x)
        edge_emb = self.edge_embed(edge_attr)
        if stages is not None:
            stages["embed"] = x
            stages["edge_embed"] = edge_emb
        for l, (conv, bn) in enumerate(zip(self.convs, self.batch_norms)):
            x_prev = x
            per_layer = {} if stages is not None else None
            x = torch.relu(bn(conv(x, edge_index, edge_emb, per_layer)))
            x = torch.nn.functional.dropout(x, p=self.pna_params.dropout, training=self.training)
            if self.pna_params.skip_connections:
                x = x + x_prev
            if stages is not None:
                for k, v in per_layer.items():
                    stages[f"l{l}.{k}"] = v
                stages[f"l{l}.out"] = x
        g = global_add_pool(x, batch)
        if stages is not None:
            stages["pooled"] = g
        out = self.mlp(g)
        return out


def training_loss(model: nn.Module, data, num_para: int) -> torch.Tensor:
    """models.py:191-194: target = para.view(-1, P); loss = mape(model(data), target)."""
    target = data.para.view(-1, num_para)
    return mape(model(data), target)
