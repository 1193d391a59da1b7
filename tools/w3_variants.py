#!/usr/bin/env python3
"""Where the time of k_gemm_w3 goes: ablation builds (make -C gnn-epc-saft_amd/csrc variants; run with
GNNSAFT_LIB=gnn-epc-saft_amd/lib/libgnnsaft_variants.so) that drop parts of the kernel -- A loads, A split + LDS
writes, B copies, MFMAs, fragment reads, barriers -- on the C3 GEMM shapes.  Results of the ablated kernels are garbage;
only their time is read.  hipGraph replays, best of 3."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_epc_saft_amd.kernels as K  # noqa: E402
from tools.gemm_tune import timeit  # noqa: E402

DEV = "cuda:0"
VARIANTS = [(0, "full kernel"), (64, "loads NOT pinned at the top of the stage"), (3, "- A loads, A split"), (4, "- B copies"),
            (7, "- all staging (MFMA + fragment reads + barriers)"), (39, "MFMA + fragment reads, no staging, no barriers"),
            (55, "MFMA only"), (8, "- MFMAs"), (24, "staging + barriers only"), (16, "- fragment reads")]


def main():
    n = 163277
    for sname, k, n_out, cfg in (("update [N,1280]x[128]", 1280, 128, 0), ("src [N,256]x[512]", 256, 512, 1),
                                 ("src [N,256]x[512]", 256, 512, 0), ("lin [N,256]x[256]", 256, 256, 1)):
        a = torch.randn(n, k, device=DEV)
        w = torch.randn(n_out, k, device=DEV) / k ** 0.5
        img = K.w3_pack(w)
        flop = 12.0 * n * n_out * k
        print(f"== {sname} tile {'128x128' if cfg == 0 else '128x256'}: {flop / 1e9:.0f} GFLOP bf16 issued = "
              f"{flop / 2.5e15 * 1e6:.0f} us at 2.5 PF")
        for var, what in VARIANTS:
            t = min(timeit(lambda: K.linear_w3(a, img, n_out, None, cfg + 64 * var)) for _ in range(3))
            print(f"   var {var:3d} {t:8.1f} us   {what}")
        if cfg == 0:   # the wave-specialised kernel (gemm_w3s.hip), 128 x 128
            for var, what in ((0, "full kernel"), (1, "producers idle"), (2, "consumers idle")):
                t = min(timeit(lambda: K.linear_w3(a, img, n_out, None, 64 * var, specialised=True)) for _ in range(3))
                print(f"   specialised var {var} {t:8.1f} us   {what}")


if __name__ == "__main__":
    main()
