#!/usr/bin/env python3
"""Cost of the input pipeline alone: ms per 512-graph batch of the C5 stand-in through GraphLoader (PREFETCH=0: collation
and staging in the caller's thread; default 2: background thread), followed by a host profile of 100 batches."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_epc_saft_amd.data.loader import GraphLoader  # noqa: E402
from gnn_epc_saft_amd.data.synthetic import synthetic_dataset  # noqa: E402

graphs = synthetic_dataset(2000, 1239, num_para=5)
loader = GraphLoader(graphs, 512, shuffle=True, device="cuda:0", seed=0, prefetch=int(os.environ.get("PREFETCH", "2")))
it = loader.forever()
for _ in range(5):
    next(it)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    batch = next(it)
torch.cuda.synchronize()
print("loader only: %.3f ms per batch" % ((time.perf_counter() - t0) * 10))
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    batch = next(it)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
