#!/usr/bin/env python3
"""Host-side (Python + launch) cost of a training step: cProfile of N steps at a bench configuration.
usage: train_step_hostprofile.py [config=2] [steps=100]"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_epc_saft_amd as G  # noqa: E402
from bench import CONFIGS  # noqa: E402
from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch  # noqa: E402

cfg = CONFIGS[int(sys.argv[1]) if len(sys.argv) > 1 else 2]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
data = make_synthetic_batch(cfg["graphs"], 1234, num_para=3)
torch.manual_seed(0)
lit = G.PNApcsaftL(G.PnaconvsParams(cfg["depth"], 1, 1, degree_histogram(data), skip_connections=True, self_loops=True),
                   G.ReadoutMLPParams(1, 3),
                   dict(hidden_dim=cfg["hidden"], num_para=3, optimizer="adam", learning_rate=1e-3, weight_decay=1e-2,
                        warmup_steps=100, momentum=0.9)).to("cuda:0").train()
dd = data.to("cuda:0")
conf = lit.configure_optimizers()
opt, sched = conf["optimizer"], conf["lr_scheduler"]["scheduler"]


def step():
    opt.zero_grad(set_to_none=True)
    loss = lit.training_step(dd)
    loss.backward()
    opt.step()
    sched.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
