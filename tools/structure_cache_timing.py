#!/usr/bin/env python3
"""Forward + loss at C2 with and without a cached batch structure (eager and hipGraph replay)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_epc_saft_amd as G  # noqa: E402
from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch  # noqa: E402

data = make_synthetic_batch(1024, 1236)
torch.manual_seed(0)
m = G.PNAPCSAFT(128, G.PnaconvsParams(3, 1, 1, degree_histogram(data), skip_connections=True, self_loops=True),
                G.ReadoutMLPParams(1, 3)).to("cuda:0").train()
dd = data.to("cuda:0")
tgt = dd.para.view(-1, 3)


def timed(label):
    with torch.no_grad():
        for _ in range(10):
            m.run(dd, target=tgt)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            m.run(dd, target=tgt)
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            g.replay()
        torch.cuda.synchronize()
        print(f"{label}: {(time.perf_counter() - t0) / 300 * 1e3:.4f} ms per step (hipGraph replay)")


timed("CSR rebuilt every step")
dd.gnnsaft_structure = m.build_structure(dd)
timed("cached structure     ")
