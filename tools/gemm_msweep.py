#!/usr/bin/env python3
"""Row-count sweep of the short-K GEMMs (source terms [M,H]x[2H,H], lin [M,H]x[H,H] with / without BatchNorm
partials) for every tile configuration: exposes workgroup quantisation (tail) effects.  usage: gemm_msweep.py [H]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_epc_saft_amd.kernels as K  # noqa: E402
from gemm_tune import timeit  # noqa: E402

DEV = "cuda:0"
h = int(sys.argv[1]) if len(sys.argv) > 1 else 128
w_src = torch.randn(2 * h, h, device=DEV) / 20
w_lin, b_lin = torch.randn(h, h, device=DEV) / 10, torch.randn(h, device=DEV)
cfgs = [(3, "64x64"), (1, "128x64"), (2, "128x128"), (4, "64x128"), (5, "128x32")]
for m in (1024, 8192, 16384, 20480, 24576, 65536, 163840):
    x = torch.randn(m, h, device=DEV)
    for name, fn, flop in (("src_terms", lambda c: K.linear(x, w_src, None, tile_config=c), 2 * m * h * 2 * h),
                           ("lin+stats", lambda c: K.linear(x, w_lin, b_lin, want_stats=True, tile_config=c), 2 * m * h * h),
                           ("lin", lambda c: K.linear(x, w_lin, b_lin, tile_config=c), 2 * m * h * h)):
        row = []
        for c, cn in cfgs:
            try:
                t = min(timeit(lambda: fn(c), 30) for _ in range(3))
            except Exception:
                t = float("nan")
            row.append(f"{cn}:{t:6.1f}us/{flop / t / 1e6:5.1f}TF")
        print(f"M={m:6d} {name:10s} " + "  ".join(row), flush=True)
