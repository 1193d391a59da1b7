#!/usr/bin/env python3
"""Times the split-bf16 GEMM on pre-split weight images (csrc/gemm_w3.hip) in every tile configuration on the plain
GEMM shapes of the PNAPCSAFT forward at BASELINE.json configs 2 and 3, next to the in-kernel-split kernel
(k_gemm_f32<X6>, its measured tile choice).  hipGraph replays, best of 3 (tools/gemm_tune.py: timeit)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_epc_saft_amd.kernels as K  # noqa: E402
from tools.gemm_tune import timeit  # noqa: E402

DEV = "cuda:0"
TILES = ["128x128", "128x256", "64x128", "64x64", "128x64", "64x256", "128x128d", "128x256d", "S128x128", "S64x128"]


def main():
    sel = sys.argv[1:] or ["C2", "C3"]
    for name, n, h in (("C2", 20409, 128), ("C3", 163277, 256)):
        if name not in sel:
            continue
        shapes = [("src terms [N,H]x[2H,H]", h, 2 * h, False), ("lin+stats [N,H]x[H,H]", h, h, True),
                  ("update    [N,5H]x[H/2,5H]", 5 * h, h // 2, False)]
        print(f"== {name}: N={n} H={h}")
        for sname, k, n_out, stats in shapes:
            a = torch.randn(n, k, device=DEV)
            w = torch.randn(n_out, k, device=DEV) / k ** 0.5
            b = torch.randn(n_out, device=DEV)
            img = K.w3_pack(w)
            ref = K.linear(a, w, b, want_stats=stats)
            base = min(timeit(lambda: K.linear(a, w, b, want_stats=stats)) for _ in range(3))
            row = []
            for cfg in range(len(TILES)):
                try:
                    kw = dict(specialised=True) if cfg >= 8 else {}
                    cc = cfg - 8 if cfg >= 8 else cfg
                    out = K.linear_w3(a, img, n_out, b, cc, want_stats=stats, **kw)
                    o, r = (out[0], ref[0]) if stats else (out, ref)
                    d = float((o - r).abs().max() / r.abs().max())
                    t = min(timeit(lambda: K.linear_w3(a, img, n_out, b, cc, want_stats=stats, **kw)) for _ in range(3))
                    row.append(f"{TILES[cfg]}:{t:7.1f}" + ("" if d < 1e-5 else f"(!{d:.0e})"))
                except Exception as e:  # noqa: BLE001
                    row.append(f"{TILES[cfg]}:   n/a")
            flop = 12.0 * n * n_out * k   # bf16 MFMA FLOP issued (6 products)
            best = min(float(r.split(":")[1].split("(")[0]) for r in row if "n/a" not in r)
            print(f"  {sname:28s} x6 auto {base:7.1f} us | " + " ".join(row) +
                  f" | best {flop / best / 1e6:6.0f} TF bf16 = {flop / best / 1e6 / 2500:.2f} of peak")
        tpack = min(timeit(lambda: K.w3_pack(w)) for _ in range(3))
        print(f"  pack of the last weight matrix: {tpack:.1f} us")


if __name__ == "__main__":
    main()
