#!/usr/bin/env python3
"""Times the weight-gradient (TN) GEMM dW = dY^T A for the shapes the backward issues at BASELINE.json configs 2, 3
and the C5 stand-in, over wave grids (wn x wk waves of 64 x 64 outputs; 1x1 = the 64 x 64 four-wave kernel) and slab
counts; every variant is checked against torch.  Replayed from a captured hipGraph (kernel time, not launch cost)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_epc_saft_amd._native import check, lib  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (5 * iters) * 1e3


def main():
    torch.manual_seed(0)
    shapes = [("C2 lin", 20505, 128, 128), ("C2 dPQ^T x", 20505, 512, 128), ("C3 lin", 163907, 256, 256),
              ("C3 dPQ^T x", 163907, 1024, 256), ("C5 lin", 10319, 64, 64), ("C5 dPQ^T x", 10319, 256, 64)]
    for name, m, n_out, k in shapes:
        dy = torch.randn(m, n_out, device=DEV)
        a = torch.randn(m, k, device=DEV)
        ref = dy.double().t() @ a.double()
        need = max(lib.gnnsaft_wgrad_scratch_bytes(m, n_out, k), 1024 * n_out * k * 4)
        scratch = torch.empty(need + 256, dtype=torch.uint8, device=DEV)
        dw = torch.empty(n_out, k, device=DEV)
        gflop = 2.0 * m * n_out * k / 1e9
        print(f"== {name}: m={m} n_out={n_out} k={k} ({gflop:.2f} GFLOP)")
        # 0x0: the library's choice; + 16 / + 32 on wn: split-bf16 / f32 kernel
        grids = [(0, 0), (1, 1), (2, 2), (4, 2), (4 + 32, 4), (4 + 16, 4)]
        for wn, wk in grids:
            line = f"  {wn % 16}x{wk}{' x6 ' if 16 <= wn < 32 else (' f32' if wn >= 32 else '    ')}:"
            for chunks in ((0,) if wn == 0 else (0, 64, 128, 256, 512)):
                if chunks and chunks * 64 > m:
                    continue

                def fn():
                    check(lib.gnnsaft_debug_linear_wgrad(dy.data_ptr(), n_out, a.data_ptr(), k, m, n_out, k, dw.data_ptr(), k,
                                                         scratch.data_ptr(), need, wn, wk, chunks,
                                                         torch.cuda.current_stream().cuda_stream), "wgrad")
                us = timeit(fn)
                err = float((dw.double() - ref).abs().max() / ref.abs().max())
                line += f"  z={chunks or 'auto'}: {us:7.1f} us {gflop / us * 1e3:5.1f} TF" + ("" if err < 3e-6 else f" ERR {err:.1e}")
            print(line, flush=True)


if __name__ == "__main__":
    main()
