#!/usr/bin/env python3
"""Latency of one eval-mode forward for a single molecule (un-batched Data, batch=None) -- the call pattern of
/root/reference/gnnepcsaft/demo/utils.py:141-152 -- eager and from a captured hipGraph."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_epc_saft_amd as G  # noqa: E402
from gnn_epc_saft_amd.data.synthetic import degree_histogram, ethanol_all_atom, make_synthetic_batch  # noqa: E402

deg = degree_histogram(make_synthetic_batch(256, 1))
torch.manual_seed(0)
m = G.PNAPCSAFT(64, G.PnaconvsParams(6, 1, 1, deg, skip_connections=True, self_loops=True),
                G.ReadoutMLPParams(1, 5)).to("cuda:0").eval()      # configs/default.py shape
mol = ethanol_all_atom().to("cuda:0")
with torch.no_grad():
    for _ in range(20):
        out = m(mol)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        out = m(mol)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 500 * 1e6
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        out = m(mol)
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 500 * 1e6
print(f"single molecule (9 atoms), H=64 L=6 P=5, eval: {eager:.0f} us per call eager, {graph:.0f} us per hipGraph replay; "
      f"out = {[round(float(v), 4) for v in out[0]]}")
