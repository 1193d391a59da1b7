"""Prints what the occupancy calculator says about the fused readout kernels on this device
(gnnsaft_readout_resident_workgroups) and checks that a 130-graph train-mode forward takes the one-launch path."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_epc_saft_amd._native import lib  # noqa: E402

torch.cuda.init()
torch.zeros(1, device="cuda")
for h in (32, 64, 128, 256):
    print(h, "forward", lib.gnnsaft_readout_resident_workgroups(h, 0), "backward", lib.gnnsaft_readout_resident_workgroups(h, 1))
