#!/usr/bin/env python3
"""Per-class row sums dR = OneHot(class)^T dm of the backward (csrc/gemm_tn.hip) at BASELINE configs 2 and 3: the f32
one-hot GEMM (k_gemm_tn<TnPlain, Y_CLASS>) against the streaming three-product bf16 kernel (k_class_sum_x3), both with
their slab reduction.  hipGraph replays, best of 3."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_epc_saft_amd.kernels as K  # noqa: E402
from tools.gemm_tune import timeit  # noqa: E402

DEV = "cuda:0"
for name, rows, k in (("C2", 60603, 256), ("C3", 486461, 512)):
    a = torch.randn(rows, k, device=DEV)
    cls = torch.randint(0, 60, (rows,), dtype=torch.int32, device=DEV)
    cls[::3] = 59
    for mode, label in ((1, "f32 one-hot GEMM"), (2, "streaming x3")):
        out = K.sum_rows_by_class(cls, 60, a, mode)
        scratch_free = None
        t = min(timeit(lambda: K.sum_rows_by_class(cls, 60, a, mode), iters=10) for _ in range(3))
        print(f"{name} [{rows}, {k}] 60 classes, {label:18s}: {t:8.1f} us  ({rows * k * 4 / t / 1e6:6.2f} TB/s of A)", flush=True)
