#!/usr/bin/env python3
"""A few launches of the fused aggregation + update kernel (csrc/update_agg.hip) and of the two launches it replaces at
C2 / C3 shapes, for rocprofv3 --pmc passes and timing.  usage: update_agg_pmc.py [C2|C3] [time]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_epc_saft_amd.kernels as K  # noqa: E402
from gnn_epc_saft_amd._native import check, lib  # noqa: E402
from gnn_epc_saft_amd.data.synthetic import make_synthetic_batch  # noqa: E402

DEV = "cuda:0"
SEL = {"C2": (1024, 128), "C3": (8192, 256)}
which = [a for a in sys.argv[1:] if a in SEL] or ["C3"]
for name in which:
    graphs, h = SEL[name]
    d = make_synthetic_batch(graphs, 1)
    n = d.x.shape[0]
    x = torch.randn(n, h, device=DEV)
    q = torch.randn(n, 2 * h, device=DEV)
    rtab = torch.randn(60, 2 * h, device=DEV)
    w_post = [torch.randn(h // 2, 13 * h, device=DEV) / 40 for _ in range(2)]
    b_post = [torch.randn(h // 2, device=DEV) for _ in range(2)]
    avg = torch.tensor([1.1], device=DEV)
    rowptr, src, dst, combo, la, lt, _ = K.csr_build(d.edge_index.to(DEV), d.edge_attr.to(DEV), n, (5, 6, 2), True)
    perm, tiles, nt, hist3, _ = K.degree_tiles(rowptr, h)
    buckets = int(lib.gnnsaft_degree_buckets())
    w_eff = torch.zeros((buckets, 2, h // 2, 5 * h), dtype=torch.float32, device=DEV)
    check(lib.gnnsaft_pna_fold_post_weights(w_post[0].data_ptr(), w_post[1].data_ptr(), avg.data_ptr(), hist3.data_ptr(), h,
                                            w_eff.data_ptr(), torch.cuda.current_stream().cuda_stream), "fold")
    images = torch.cat([K.w3_pack(w_eff[dd, t]) for dd in range(buckets) for t in range(2)])
    u = torch.empty((n, h), dtype=torch.float32, device=DEV)

    def fused():
        check(lib.gnnsaft_pna_update_agg(x.data_ptr(), q.data_ptr(), rtab.data_ptr(), 60, rowptr.data_ptr(), src.data_ptr(),
                                         combo.data_ptr(), perm.data_ptr(), tiles.data_ptr(), nt.data_ptr(), n, h,
                                         images.data_ptr(), b_post[0].data_ptr(), b_post[1].data_ptr(), u.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream),
              "gnnsaft_pna_update_agg")

    def two():
        agg = K.pna_aggregate_src(rowptr, src, combo, h, q, rtab)
        return K.pna_update_folded(x, agg, perm, tiles, nt, hist3, avg, w_post[0], b_post[0], w_post[1], b_post[1])

    for _ in range(3):
        fused()
        two()
    torch.cuda.synchronize()
    if "time" in sys.argv:
        from tools.gemm_tune import timeit
        print(f"{name}: fused {min(timeit(fused) for _ in range(3)):.1f} us, two launches {min(timeit(two) for _ in range(3)):.1f} us "
              "(aggregation + folded update incl. its weight fold launch)")
