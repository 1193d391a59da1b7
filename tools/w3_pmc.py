#!/usr/bin/env python3
"""A few launches of the ablation builds of k_gemm_w3 / k_gemm_w3s on the C3 update shape, for rocprofv3 --pmc passes
(tools/w3_pmc_round.sh; GNNSAFT_LIB = the variants library)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_epc_saft_amd.kernels as K  # noqa: E402

DEV = "cuda:0"
n, k, n_out = 163277, 1280, 128
a = torch.randn(n, k, device=DEV)
w = torch.randn(n_out, k, device=DEV) / k ** 0.5
img = K.w3_pack(w)
for _ in range(3):
    for var in (0, 7, 55, 24, 8):
        K.linear_w3(a, img, n_out, None, 64 * var)
    for var in (0, 1, 2):
        K.linear_w3(a, img, n_out, None, 64 * var, specialised=True)
    K.linear(a, w, None)
torch.cuda.synchronize()
