"""Model factory and degree histogram -- the constructor contract of the hot path
(``/root/reference/gnnepcsaft/train/utils.py:26-85``)."""

from __future__ import annotations

from typing import Iterable

import torch

from . import models
from .models import _cfg


def calc_deg(dataset, workdir: str = "") -> torch.Tensor:
    """In-degree histogram for the PNA degree scalers (train/utils.py:26-49).

    The reference loads the Ramirez / Esper ``InMemoryDataset`` named by a string; those
    datasets are DVC pointers that cannot be fetched, and dataset I/O is outside the hot
    path, so this takes the graphs themselves: any iterable of objects with ``edge_index``
    and ``x`` (or ``num_nodes``).  Raw edges only -- no self-loops, even if the model adds them.
    """
    if isinstance(dataset, str):     # the reference call shape calc_deg("esper", workdir) (train.py:128)
        raise NotImplementedError(
            f"calc_deg({dataset!r}, {workdir!r}): loading the reference datasets by name is out of scope (DVC/GCS, "
            "rdkit, ogb absent); pass the graphs themselves -- calc_deg(graphs, workdir) -- instead")
    per_graph = []
    max_degree = -1
    for data in dataset:
        n = int(data.x.shape[0]) if getattr(data, "x", None) is not None else int(data.num_nodes)
        d = torch.bincount(data.edge_index[1].reshape(-1).cpu(), minlength=n)
        per_graph.append(d)
        if d.numel():
            max_degree = max(max_degree, int(d.max()))
    deg = torch.zeros(max_degree + 1, dtype=torch.long)
    for d in per_graph:
        deg += torch.bincount(d, minlength=deg.numel())
    return deg


def create_model(config, deg: torch.Tensor) -> torch.nn.Module:
    """train/utils.py:52-85: ``config.model`` in {"PNA", "PNAL"}."""
    pna_params = models.PnaconvsParams(
        propagation_depth=_cfg(config, "propagation_depth"),
        pre_layers=_cfg(config, "pre_layers"),
        post_layers=_cfg(config, "post_layers"),
        deg=deg,
        skip_connections=_cfg(config, "skip_connections"),
        self_loops=_cfg(config, "add_self_loops"),
    )
    mlp_params = models.ReadoutMLPParams(
        num_mlp_layers=_cfg(config, "num_mlp_layers"),
        num_para=_cfg(config, "num_para"),
        dropout=_cfg(config, "dropout_rate"),
    )
    kind = _cfg(config, "model")
    if kind == "PNA":
        return models.PNAPCSAFT(hidden_dim=_cfg(config, "hidden_dim"), pna_params=pna_params, mlp_params=mlp_params)
    if kind == "PNAL":
        return models.PNApcsaftL(pna_params=pna_params, mlp_params=mlp_params, config=config)
    raise ValueError(f"Unsupported model: {kind}.")
