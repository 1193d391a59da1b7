#!/usr/bin/env python3
"""How much of a GEMM launch's time is round quantisation: k_gemm_w3 128x256 (two workgroups per CU = 512 resident) on
[M, 256] x [256, 256] for row counts around whole multiples of 512 row tiles.  usage: round_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_epc_saft_amd.kernels as K  # noqa: E402
from tools.gemm_tune import timeit  # noqa: E402

DEV = "cuda:0"
k, n_out = 256, 256
w = torch.randn(n_out, k, device=DEV) / 16
b = torch.randn(n_out, device=DEV)
img = K.w3_pack(w)
for tiles in (512, 513, 520, 576, 640, 768, 1023, 1024, 1025, 1030, 1088, 1152, 1280, 1281, 1536, 1537):
    m = tiles * 128
    a = torch.randn(m, k, device=DEV)
    t = min(timeit(lambda: K.linear_w3(a, img, n_out, b, 1)) for _ in range(3))
    t0 = min(timeit(lambda: K.linear_w3(a, img, n_out, b, 0)) for _ in range(3))
    ta = min(timeit(lambda: K.linear_ar(a, img, n_out, b, 0)) for _ in range(3))
    print(f"{tiles:5d} row tiles ({tiles / 512:5.3f} rounds of 512): w3 128x256 {t:7.1f} us  ({t / tiles * 512:6.1f} per 512 tiles) | "
          f"w3 128x128 {t0:7.1f} | ar 128x128 {ta:7.1f}", flush=True)
