#!/usr/bin/env python3
"""Cost of the input pipeline alone: ms per 512-graph batch of the C5 stand-in through GraphLoader (PREFETCH=0: caller's
thread), with a host profile."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_epc_saft_amd as G
from gnn_epc_saft_amd.data.loader import GraphLoader
from gnn_epc_saft_amd.data.synthetic import synthetic_dataset
graphs = synthetic_dataset(2000, 1239, num_para=5)
loader = GraphLoader(graphs, 512, shuffle=True, device="cuda:0", seed=0, prefetch=int(os.environ.get("PREFETCH", "2")))
it = loader.forever()
for _ in range(5): next(it)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): b = next(it)
torch.cuda.synchronize()
print("loader only: %.3f ms per batch" % ((time.perf_counter() - t0) * 10))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(100): b = next(it)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
