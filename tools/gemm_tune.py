#!/usr/bin/env python3
"""Times every GEMM tile configuration on the shapes the PNAPCSAFT forward issues at
BASELINE.json configs 2 and 3 (interleaved rounds in one process, median of rounds)."""

import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_epc_saft_amd.kernels as K  # noqa: E402
from gnn_epc_saft_amd.data.synthetic import make_synthetic_batch  # noqa: E402

DEV = "cuda:0"
CFG_NAMES = ["256x32", "128x64", "128x128", "64x64", "64x128", "128x32"]


def timeit(fn, iters=20):
    """us per call, replayed from a captured hipGraph so that the Python / allocator / launch overhead of the
    per-kernel wrappers (a few us per call, more than the small kernels themselves) is not measured."""
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
    return s.elapsed_time(e) / (5 * iters) * 1e3  # us


def main():
    for name, graphs, h in (("C2", 1024, 128), ("C3", 8192, 256), ("C5", 512, 64)):
        d = make_synthetic_batch(graphs, 1)
        n = d.x.shape[0]
        x = torch.randn(n, h, device=DEV)
        w_pre = [torch.randn(h, 3 * h, device=DEV) / 20 for _ in range(2)]
        w_lin, b_lin = torch.randn(h, h, device=DEV) / 10, torch.randn(h, device=DEV)
        w_post = [torch.randn(h // 2, 13 * h, device=DEV) / 40 for _ in range(2)]
        b_post = [torch.randn(h // 2, device=DEV) for _ in range(2)]
        avg = torch.tensor([1.1], device=DEV)
        rowptr, src, dst, combo, la, lt, _ = K.csr_build(d.edge_index.to(DEV), d.edge_attr.to(DEV), n, (5, 6, 2), True)
        agg = torch.randn(n, 2, 4 * h, device=DEV)
        w_src = torch.cat([w_pre[0][:, h:2 * h], w_pre[1][:, h:2 * h]]).contiguous()
        shapes = {
            "src_terms  [N,H]x[2H,H]": lambda c: K.linear(x, w_src, None, tile_config=c),
            "lin+stats  [N,H]x[H,H]": lambda c: K.linear(x, w_lin, b_lin, want_stats=True, tile_config=c),
            "lin plain  [N,H]x[H,H]": lambda c: K.linear(x, w_lin, b_lin, tile_config=c),
        }
        print(f"== {name}: N={n} H={h}")
        for sname, fn in shapes.items():
            for mode, off in (("f32", 32), ("x6 ", 16)):   # + 32: f32 matrix cores, + 16: split-bf16 (gemm.hip)
                row = []
                for cfg in range(6):        # explicit tile configuration per call (gnnsaft_debug_linear_tile)
                    try:
                        row.append(min(timeit(lambda: fn(cfg + off)) for _ in range(3)))
                    except Exception:
                        row.append(float("nan"))
                row.append(min(timeit(lambda: fn(None)) for _ in range(3)))
                print(f"  {sname:26s} {mode} " + " ".join(f"{c}:{t:7.1f}" for c, t in zip(CFG_NAMES + ["auto"], row)))
        perm, tiles, nt, hist3, _ = K.degree_tiles(rowptr, h)
        fn = lambda: K.pna_update_folded(x, agg, perm, tiles, nt, hist3, avg, w_post[0], b_post[0], w_post[1], b_post[1])
        print(f"  {'update folded K=5H (+fold)':26s} auto:{min(timeit(fn) for _ in range(3)):7.1f}")
        fn = lambda: K.pna_update(x, agg, la, lt, avg, w_post[0], b_post[0], w_post[1], b_post[1])
        print(f"  {'update unfolded K=13H':26s} auto:{min(timeit(fn) for _ in range(3)):7.1f}")
        fn = lambda: K.pna_aggregate(rowptr, src, combo, h, pq=torch.empty(0, device=DEV) if False else pq, rtab=rtab)
        pq = torch.randn(n, 4 * h, device=DEV)
        rtab = torch.randn(60, 2 * h, device=DEV)
        print(f"  {'K4 aggregate':26s} {min(timeit(fn) for _ in range(3)):7.1f} us")


if __name__ == "__main__":
    main()
