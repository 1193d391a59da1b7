#!/usr/bin/env python3
"""Where one tile of k_gemm_ar (csrc/gemm_ar.hip) spends its time: s_memtime stamps of a mid-grid tile's first wave
(gnnsaft_debug_ar_stamps), two per 32-k stage: in front of the stage's counted wait + barrier and behind it.
Needs the probe build (the product kernel has no stamps):
  make -C gnn-epc-saft_amd/csrc stamps; GNNSAFT_LIB=gnn-epc-saft_amd/lib/libgnnsaft_stamps.so python tools/ar_stamps.py [cfg]
Timing only -- the stamped kernel's extra branches make hipcc copy registers of loads still in flight (tools/check_ar_isa.py
on a -DGS_AR_STAMPS build), its results are not checked.  Shape: the update of BASELINE config 3, [163277, 1280] x [128, 1280]."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_epc_saft_amd.kernels as K  # noqa: E402
from gnn_epc_saft_amd._native import check, lib  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n, k, n_out = 163277, 1280, 128
a = torch.randn(n, k, device="cuda:0")
w = torch.randn(n_out, k, device="cuda:0") / k ** 0.5
b = torch.randn(n_out, device="cuda:0")
img = K.w3_pack(w)
for _ in range(3):
    K.linear_ar(a, img, n_out, b, cfg)
torch.cuda.synchronize()
buf = torch.zeros(256, dtype=torch.int64, device="cuda:0")
check(lib.gnnsaft_debug_ar_stamps(buf.data_ptr()), "stamps")
K.linear_ar(a, img, n_out, b, cfg)
torch.cuda.synchronize()
check(lib.gnnsaft_debug_ar_stamps(None), "stamps off")
st = buf.cpu().view(-1, 2)
nk = int((st[:, 0] > 0).sum())
print(f"cfg {cfg}: {nk} stages; cycles (s_memtime, 100 MHz or shader clock as the counter runs)")
print("stage | compute (previous barrier -> this wait)   wait + barrier")
prev = None
tot_c = tot_w = 0
for s in range(nk):
    t0, t1 = int(st[s, 0]), int(st[s, 1])
    c = t0 - prev if prev is not None else 0
    print(f"{s:5d} | {c:10d} {t1 - t0:10d}")
    if prev is not None:
        tot_c += c
    tot_w += t1 - t0
    prev = t1
print(f"total compute {tot_c}, total wait {tot_w}, first -> last {int(st[nk - 1, 1]) - int(st[0, 0])}")
