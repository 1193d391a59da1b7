#!/usr/bin/env python3
"""Times the split-bf16 GEMM that keeps its A operand in registers (csrc/gemm_ar.hip) on the plain GEMM shapes of the
PNAPCSAFT forward at BASELINE.json configs 2 and 3, next to k_gemm_w3's measured tile choice and the in-kernel-split
kernel (k_gemm_f32<X6>).  hipGraph replays, best of 3 (tools/gemm_tune.py: timeit)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_epc_saft_amd.kernels as K  # noqa: E402
from tools.gemm_tune import timeit  # noqa: E402

DEV = "cuda:0"
AR = ["128x128", "128x64", "64x128", "64x64", "256x128"]
W3 = {"C3": {"src": 1, "lin": 1, "update": 0}, "C2": {"src": 4, "lin": 6, "update": 3}}


def main():
    sel = sys.argv[1:] or ["C2", "C3"]
    for name, n, h in (("C2", 20409, 128), ("C3", 163277, 256)):
        if name not in sel:
            continue
        shapes = [("src", "src terms [N,H]x[2H,H]", h, 2 * h), ("lin", "lin [N,H]x[H,H]", h, h),
                  ("update", "update [N,5H]x[H/2,5H]", 5 * h, h // 2)]
        print(f"== {name}: N={n} H={h}", flush=True)
        for key, sname, k, n_out in shapes:
            a = torch.randn(n, k, device=DEV)
            w = torch.randn(n_out, k, device=DEV) / k ** 0.5
            b = torch.randn(n_out, device=DEV)
            img = K.w3_pack(w)
            ref = K.linear(a, w, b)
            want = (a.double() @ w.double().t() + b.double())
            scale = float((a.double().abs() @ w.double().abs().t()).max())
            base = min(timeit(lambda: K.linear(a, w, b)) for _ in range(3))
            cfg3 = W3[name][key]
            t3 = min(timeit(lambda: K.linear_w3(a, img, n_out, b, cfg3)) for _ in range(3))
            row = []
            for cfg, tname in enumerate(AR):
                try:
                    out = K.linear_ar(a, img, n_out, b, cfg)
                    torch.cuda.synchronize()
                    d = float((out - ref).abs().max() / ref.abs().max())
                    e64 = float((out.double() - want).abs().max()) / scale
                    t = min(timeit(lambda: K.linear_ar(a, img, n_out, b, cfg)) for _ in range(3))
                    row.append(f"{tname}:{t:7.1f}" + (f" (vs x6 {d:.0e}, vs f64 {e64:.1e} of sum|a||w|)"))
                except Exception as e:  # noqa: BLE001
                    row.append(f"{tname}: n/a ({e})")
            flop = 12.0 * n * n_out * k
            print(f"  {sname:26s} x6 {base:7.1f} us | w3 cfg {cfg3} {t3:7.1f} us | AR " + " | ".join(row), flush=True)
            best = min(float(r.split(":")[1].split("(")[0]) for r in row if "n/a" not in r)
            print(f"      best AR {flop / best / 1e6:6.0f} TF bf16 issued = {flop / best / 1e6 / 2500:.2f} of peak", flush=True)


if __name__ == "__main__":
    main()
