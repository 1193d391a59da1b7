#!/usr/bin/env python3
"""Where one tile of the fused aggregation + update kernel spends its time: s_memtime stamps of tile 5's first producer
and first consumer wave (gnnsaft_debug_update_agg_stamps).  Prints, per stage, the producer's stash / build / barrier
wait and the consumer's MFMA interval / barrier wait in cycles.  usage: update_agg_stamps.py [C2|C3]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_epc_saft_amd._native import check, lib  # noqa: E402

buf = torch.zeros(2 * 64 * 4, dtype=torch.int64, device="cuda:0")
check(lib.gnnsaft_debug_update_agg_stamps(buf.data_ptr()), "stamps")
sys.argv = [sys.argv[0]] + [a for a in sys.argv[1:] if a in ("C2", "C3")][:1]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "update_agg_pmc.py")).read())
torch.cuda.synchronize()
st = buf.cpu().view(2, 64, 4)
p, c = st[0], st[1]
nk = int((p[:, 0] > 0).sum())
t0 = int(p[0, 0])
print(f"{nk} stages; cycles (shader clock) relative to the producer's first stamp")
print("stage | producer: start  stash  build  barrier-wait | consumer: start  mfma-interval  (next start - end = barrier wait)")
for s in range(nk):
    ps, pa, pb, pc = (int(v) - t0 for v in p[s])
    cs, ce = int(c[s, 0]) - t0, int(c[s, 1]) - t0
    nxt = int(c[s + 1, 0]) - t0 if s + 1 < nk and int(c[s + 1, 0]) > 0 else ce
    print(f"{s:5d} | {ps:9d} {pa - ps:6d} {pb - pa:6d} {pc - pb:8d} | {cs:9d} {ce - cs:8d} {nxt - ce:8d}")
check(lib.gnnsaft_debug_update_agg_stamps(None), "stamps off")
