"""Synthetic molecular graphs and the PyG-free ``Data`` / ``Batch`` stand-ins.

The reference's datasets are DVC pointers to a bucket that cannot be fetched
(``/root/reference/.dvc/config:4``), so every test and benchmark input is
synthetic, shaped like what ``ogb.utils.mol.smiles2graph`` hands to
``/root/reference/gnnepcsaft/data/graph.py:28-37``: int64 categorical node
features ``[N,9]``, int64 ``edge_index [2,E]`` with ``(i,j),(j,i)`` per bond and
int64 ``edge_attr [E,3]`` duplicated per direction.
"""

from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch

ATOM_FEATURE_DIMS = (119, 5, 12, 12, 10, 6, 6, 2, 2)
BOND_FEATURE_DIMS = (5, 6, 2)


class GraphData:
    """Attribute bag with the fields ``PNAPCSAFT.forward`` reads
    (``/root/reference/gnnepcsaft/train/models.py:111-116``).  A real PyG
    ``Data`` / ``Batch`` works in its place."""

    def __init__(self, x, edge_index, edge_attr, batch=None, ptr=None, para=None, num_graphs=None):
        self.x = x
        self.edge_index = edge_index
        self.edge_attr = edge_attr
        self.batch = batch
        self.ptr = ptr
        self.para = para
        if num_graphs is None:
            num_graphs = 1 if batch is None else (int(ptr.numel()) - 1 if ptr is not None else int(batch.max()) + 1)
        self.num_graphs = num_graphs
        self.gnnsaft_structure = None   # optional cached CSR / degree tiles (PNAPCSAFT.build_structure)

    @property
    def num_nodes(self) -> int:
        return int(self.x.shape[0])

    def to(self, device, non_blocking: bool = False) -> "GraphData":
        mv = lambda t: None if t is None else t.to(device, non_blocking=non_blocking)
        out = GraphData(mv(self.x), mv(self.edge_index), mv(self.edge_attr), mv(self.batch), mv(self.ptr),
                        mv(self.para), self.num_graphs)
        out.gnnsaft_structure = mv(self.gnnsaft_structure)
        return out


def collate(graphs: Sequence[GraphData]) -> GraphData:
    """What PyG's ``Batch.from_data_list`` does to the fields the path reads:
    concatenate, offset ``edge_index`` by the running node count, build
    ``batch`` and ``ptr``."""
    xs, eis, eas, bs, paras = [], [], [], [], []
    ptr = [0]
    for g, d in enumerate(graphs):
        n = d.x.shape[0]
        xs.append(d.x)
        eis.append(d.edge_index + ptr[-1])
        eas.append(d.edge_attr)
        bs.append(torch.full((n,), g, dtype=torch.int64))
        if d.para is not None:
            paras.append(d.para.reshape(-1))
        ptr.append(ptr[-1] + n)
    return GraphData(torch.cat(xs), torch.cat(eis, dim=1), torch.cat(eas), torch.cat(bs),
                     torch.tensor(ptr, dtype=torch.int64), torch.cat(paras) if paras else None, len(graphs))


def ethanol_heavy() -> GraphData:
    """Heavy-atom ethanol ``CCO`` as the reference pipeline builds it
    (``data/graph.py:9-13`` default ``with_hydrogen=False``): 3 nodes, 4 directed
    edges."""
    x = torch.tensor([[5, 0, 4, 5, 3, 0, 2, 0, 0], [5, 0, 4, 5, 2, 0, 2, 0, 0], [7, 0, 2, 5, 1, 0, 2, 0, 0]])
    ei = torch.tensor([[0, 1, 1, 2], [1, 0, 2, 1]])
    ea = torch.zeros((4, 3), dtype=torch.int64)
    return GraphData(x, ei, ea, para=torch.tensor([2.3827, 3.1771, 198.24, 0.032384, 2653.4]))


def ethanol_all_atom() -> GraphData:
    """Hand-built 9-atom ethanol (BASELINE.json configs[0]): C C O + 6 H, 8 bonds."""
    c1, c2, o = 0, 1, 2
    bonds = [(c1, c2), (c2, o), (c1, 3), (c1, 4), (c1, 5), (c2, 6), (c2, 7), (o, 8)]
    x = torch.tensor([[5, 0, 4, 5, 3, 0, 2, 0, 0], [5, 0, 4, 5, 2, 0, 2, 0, 0], [7, 0, 2, 5, 1, 0, 2, 0, 0]]
                     + [[0, 0, 1, 5, 0, 0, 0, 0, 0]] * 6)
    src, dst = [], []
    for a, b in bonds:
        src += [a, b]
        dst += [b, a]
    ei = torch.tensor([src, dst])
    ea = torch.zeros((len(src), 3), dtype=torch.int64)
    return GraphData(x, ei, ea, para=torch.tensor([2.3827, 3.1771, 198.24, 0.032384, 2653.4]))


def _random_molecule(rng: np.random.Generator, n: int):
    """Random spanning tree with valence cap 4 plus Binomial(n, 0.05) ring
    closures (rejected when a partner already has 4 bonds or the bond exists)."""
    degree = np.zeros(n, dtype=np.int64)
    bonds = []
    have = set()
    for i in range(1, n):
        open_nodes = np.flatnonzero(degree[:i] < 4)
        p = int(open_nodes[rng.integers(open_nodes.size)])
        bonds.append((p, i))
        have.add((p, i))
        degree[p] += 1
        degree[i] += 1
    for _ in range(int(rng.binomial(n, 0.05))):
        a, b = (int(v) for v in rng.integers(0, n, size=2))
        if a == b:
            continue
        lo, hi = min(a, b), max(a, b)
        if (lo, hi) in have or degree[a] >= 4 or degree[b] >= 4:
            continue
        bonds.append((lo, hi))
        have.add((lo, hi))
        degree[a] += 1
        degree[b] += 1
    return bonds


def make_synthetic_batch(num_graphs: int, seed: int, num_para: int = 3, n_min: int = 12, n_max: int = 28,
                         atom_dims: Sequence[int] = ATOM_FEATURE_DIMS,
                         bond_dims: Sequence[int] = BOND_FEATURE_DIMS) -> GraphData:
    """SURVEY.md §8(d) / Appendix D workload: |V| ~ U[12,28] (mean 20), |E| ~ 2|V|."""
    rng = np.random.default_rng(seed)
    sizes = rng.integers(n_min, n_max + 1, size=num_graphs)
    ptr = np.zeros(num_graphs + 1, dtype=np.int64)
    np.cumsum(sizes, out=ptr[1:])
    src_parts: List[np.ndarray] = []
    dst_parts: List[np.ndarray] = []
    nb_parts: List[int] = []
    for g in range(num_graphs):
        bonds = np.asarray(_random_molecule(rng, int(sizes[g])), dtype=np.int64).reshape(-1, 2) + ptr[g]
        a, b = bonds[:, 0], bonds[:, 1]
        # (i,j),(j,i) per bond, in bond order
        src_parts.append(np.stack([a, b], axis=1).reshape(-1))
        dst_parts.append(np.stack([b, a], axis=1).reshape(-1))
        nb_parts.append(bonds.shape[0])
    n_total = int(ptr[-1])
    nb_total = int(sum(nb_parts))
    x = np.stack([rng.integers(0, d, size=n_total) for d in atom_dims], axis=1).astype(np.int64)
    bond_attr = np.stack([rng.integers(0, d, size=nb_total) for d in bond_dims], axis=1).astype(np.int64)
    edge_attr = np.repeat(bond_attr, 2, axis=0)
    edge_index = np.stack([np.concatenate(src_parts), np.concatenate(dst_parts)]).astype(np.int64)
    batch = np.repeat(np.arange(num_graphs, dtype=np.int64), sizes)
    para = rng.uniform(0.5, 5.0, size=num_graphs * num_para).astype(np.float32)
    return GraphData(torch.from_numpy(x), torch.from_numpy(edge_index), torch.from_numpy(edge_attr),
                     torch.from_numpy(batch), torch.from_numpy(ptr), torch.from_numpy(para), num_graphs)


def synthetic_dataset(num_graphs: int, seed: int, num_para: int = 3) -> List[GraphData]:
    """``num_graphs`` single-molecule ``GraphData`` objects (graph-local node ids, ``para`` [num_para]): the
    synthetic stand-in for an ``InMemoryDataset`` of the reference."""
    b = make_synthetic_batch(num_graphs, seed, num_para=num_para)
    ptr = b.ptr.tolist()
    counts = torch.bincount(b.batch[b.edge_index[1]], minlength=num_graphs)   # edges per graph
    eptr = [0] + counts.cumsum(0).tolist()                    # edges are graph-contiguous by construction
    para = b.para.view(num_graphs, num_para)
    out = []
    for g in range(num_graphs):
        n0, n1, e0, e1 = ptr[g], ptr[g + 1], eptr[g], eptr[g + 1]
        out.append(GraphData(b.x[n0:n1], b.edge_index[:, e0:e1] - n0, b.edge_attr[e0:e1], para=para[g]))
    return out


def split_graphs(data: GraphData, world_size: int, rank: int) -> GraphData:
    """Contiguous ``ptr`` range of graphs for one rank:
    nodes and edges follow their graph, edge ids are re-based locally."""
    g = data.num_graphs
    lo = (g * rank) // world_size
    hi = (g * (rank + 1)) // world_size
    ptr = data.ptr
    n0, n1 = int(ptr[lo]), int(ptr[hi])
    dst = data.edge_index[1]
    emask = (dst >= n0) & (dst < n1)
    p = data.para.numel() // g if data.para is not None else 0
    return GraphData(data.x[n0:n1], data.edge_index[:, emask] - n0, data.edge_attr[emask],
                     data.batch[n0:n1] - lo, ptr[lo:hi + 1] - n0,
                     None if data.para is None else data.para.view(g, p)[lo:hi].reshape(-1), hi - lo)


def degree_histogram(graphs_or_batch) -> torch.Tensor:
    """In-degree histogram of ``edge_index[1]`` over raw graphs (no self-loops),
    length ``max_degree + 1`` -- the tensor ``calc_deg`` returns
    (``/root/reference/gnnepcsaft/train/utils.py:38-49``)."""
    items = graphs_or_batch if isinstance(graphs_or_batch, (list, tuple)) else [graphs_or_batch]
    counts = []
    for d in items:
        counts.append(torch.bincount(d.edge_index[1].reshape(-1), minlength=d.x.shape[0]))
    indeg = torch.cat(counts)
    return torch.bincount(indeg, minlength=int(indeg.max()) + 1 if indeg.numel() else 1)
