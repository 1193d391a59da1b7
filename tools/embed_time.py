#!/usr/bin/env python3
"""Times gnnsaft_embed_sum alone (AtomEncoder: the sum of nine embedding rows per node, models.py:122) at the node counts
of BASELINE configs 2 and 3: what share of the forward's first launch (k_forward_prologue: 34 / 250 us) the embedding sum is.
Measured (round 4): C2 11.3 us, C3 98.9 us = 15 TB/s of table rows out of L1 / L2."""
import os, sys, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/tools") else os.getcwd())
import gnn_epc_saft_amd.kernels as K
from tools.gemm_tune import timeit
DEV = "cuda:0"
dims = [119, 5, 12, 12, 10, 6, 6, 2, 2]
for name, n, h in (("C2", 20409, 128), ("C3", 163967, 256)):
    tabs = [torch.randn(d, h, device=DEV) for d in dims]
    idx = torch.stack([torch.randint(0, d, (n,)) for d in dims], 1).to(DEV)
    idx[:, 0] = torch.randint(5, 9, (n,), device=DEV)   # a few hot atom types
    t = min(timeit(lambda: K.embed_sum(idx, tabs), iters=10) for _ in range(3))
    print(f"{name}: embed_sum of {n} nodes x {h}: {t:7.1f} us  (write {n*h*4/t/1e6:.2f} TB/s, table reads {9*n*h*4/t/1e6:.2f} TB/s)", flush=True)
