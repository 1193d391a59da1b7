#!/usr/bin/env python3
"""Development probe: phase timing inside k_graph_forward (needs a library built with -DGS_GF_TIMING:
make -C gnn-epc-saft_amd/csrc timing; GNNSAFT_LIB=gnn-epc-saft_amd/lib/libgnnsaft_timing.so python tools/graph_kernel_phases.py)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_epc_saft_amd as G  # noqa: E402
from gnn_epc_saft_amd import _native  # noqa: E402
from gnn_epc_saft_amd.data.synthetic import degree_histogram, ethanol_all_atom, ethanol_heavy, make_synthetic_batch  # noqa: E402

lib = ctypes.CDLL(_native.LIB_PATH)
deg = degree_histogram(make_synthetic_batch(256, 1))
for hidden, depth, para, dt in ((64, 6, 5, torch.float32), (64, 6, 5, torch.float64), (256, 5, 3, torch.float32)):
    torch.manual_seed(0)
    m = G.PNAPCSAFT(hidden, G.PnaconvsParams(depth, 1, 1, deg, skip_connections=True, self_loops=True),
                    G.ReadoutMLPParams(1, para)).to("cuda:0", dt).eval()
    for name, mol in (("ethanol-3", ethanol_heavy()), ("ethanol-9", ethanol_all_atom())):
        mol = mol.to("cuda:0")
        with torch.no_grad():
            for _ in range(5):
                m(mol)
        torch.cuda.synchronize()
        buf = (ctypes.c_longlong * 256)()
        assert lib.gnnsaft_debug_graph_stamps(buf) == 0
        t = lambda i: (buf[i] - buf[250]) / 100.0   # us (100 MHz)
        print(f"H={hidden} L={depth} {dt} {name}: total {t(251):.1f} us; structure+embed {t(0):.1f}")
        for l in range(depth):
            prev = t(0) if l == 0 else t(5 * l)
            print(f"   layer {l}: PQ {t(1 + 5 * l) - prev:.1f} | aggregate {t(2 + 5 * l) - t(1 + 5 * l):.1f} | update "
                  f"{t(3 + 5 * l) - t(2 + 5 * l):.1f} | lin {t(4 + 5 * l) - t(3 + 5 * l):.1f} | rest of tiles "
                  f"{t(5 + 5 * l) - t(4 + 5 * l):.1f}")
        print(f"   readout {t(251) - t(5 * depth):.1f}")
