#!/usr/bin/env python3
"""Latency of one eval-mode forward for a single molecule (un-batched Data, batch=None) -- the call pattern of
/root/reference/gnnepcsaft/demo/utils.py:141-152 and models.py:204-211 -- eager and from a captured hipGraph, for
the per-graph fused kernel (float32 and float64) and for the batched pipeline (graph_kernel_max_graphs = 0)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_epc_saft_amd as G  # noqa: E402
from gnn_epc_saft_amd.data.synthetic import (degree_histogram, ethanol_all_atom, ethanol_heavy,  # noqa: E402
                                             make_synthetic_batch, synthetic_dataset)

deg = degree_histogram(make_synthetic_batch(256, 1))


def measure(m, mol, reps=500):
    with torch.no_grad():
        for _ in range(20):
            out = m(mol)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = m(mol)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / reps * 1e6
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            out = m(mol)
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / reps * 1e6
    return eager, graph, out


for hidden, depth, para in ((64, 6, 5), (128, 3, 3), (256, 5, 3)):
    torch.manual_seed(0)
    m = G.PNAPCSAFT(hidden, G.PnaconvsParams(depth, 1, 1, deg, skip_connections=True, self_loops=True),
                    G.ReadoutMLPParams(1, para)).to("cuda:0").eval()
    mols = {"ethanol 3 heavy atoms": ethanol_heavy(), "ethanol 9 atoms": ethanol_all_atom(),
            "synthetic 20-atom molecule": synthetic_dataset(8, 3, para)[2]}
    for name, mol in mols.items():
        mol = mol.to("cuda:0")
        m.graph_kernel_max_graphs, m.graph_kernel_max_nodes = 1, 1 << 30
        e1, g1, o1 = measure(m, mol)
        m.graph_kernel_max_graphs = 0
        e0, g0, o0 = measure(m, mol)
        m64 = G.PNAPCSAFT(hidden, G.PnaconvsParams(depth, 1, 1, deg, skip_connections=True, self_loops=True),
                          G.ReadoutMLPParams(1, para)).to("cuda:0", torch.float64).eval()
        m64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in m.state_dict().items()})
        e2, g2, o2 = measure(m64, mol)
        print(f"H={hidden} L={depth} {name}: per-graph kernel f32 {g1:.0f} us / replay ({e1:.0f} eager), f64 {g2:.0f} us "
              f"({e2:.0f} eager); batched pipeline f32 {g0:.0f} us ({e0:.0f} eager); |f32 - f64| / scale "
              f"{float((o1.double() - o2).abs().max() / o2.abs().max()):.1e}", flush=True)
