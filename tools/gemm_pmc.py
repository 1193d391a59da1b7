#!/usr/bin/env python3
"""A few launches of the three per-layer GEMMs at C2 / C3 shapes, for rocprofv3 --pmc runs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_epc_saft_amd.kernels as K
from gnn_epc_saft_amd.data.synthetic import make_synthetic_batch
DEV = "cuda:0"
SEL = {"C2": (1024, 128), "C3": (8192, 256)}
for graphs, h in ([SEL[a] for a in sys.argv[1:]] or list(SEL.values())):
    d = make_synthetic_batch(graphs, 1)
    n = d.x.shape[0]
    x = torch.randn(n, h, device=DEV)
    w_pre = [torch.randn(h, 3 * h, device=DEV) / 20 for _ in range(2)]
    w_lin, b_lin = torch.randn(h, h, device=DEV) / 10, torch.randn(h, device=DEV)
    w_post = [torch.randn(h // 2, 13 * h, device=DEV) / 40 for _ in range(2)]
    b_post = [torch.randn(h // 2, device=DEV) for _ in range(2)]
    avg = torch.tensor([1.1], device=DEV)
    rowptr, src, dst, combo, la, lt, _ = K.csr_build(d.edge_index.to(DEV), d.edge_attr.to(DEV), n, (5, 6, 2), True)
    agg = torch.randn(n, 2, 4 * h, device=DEV)
    perm, tiles, nt, hist3, _ = K.degree_tiles(rowptr, h)
    for _ in range(3):
        K.pna_src_terms(x, w_pre[0], w_pre[1])
        K.pna_update_folded(x, agg, perm, tiles, nt, hist3, avg, w_post[0], b_post[0], w_post[1], b_post[1])
        K.linear(x, w_lin, b_lin, want_stats=True)
    torch.cuda.synchronize()
