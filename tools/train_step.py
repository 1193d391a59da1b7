#!/usr/bin/env python3
"""N full training steps (forward + MAPE + backward + fused AdamW) at a bench configuration, for rocprofv3 runs.
usage: train_step.py [config=2] [steps=10]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_epc_saft_amd as G  # noqa: E402
from bench import CONFIGS  # noqa: E402
from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch  # noqa: E402

cfg = CONFIGS[int(sys.argv[1]) if len(sys.argv) > 1 else 2]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
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


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print(f"{cfg['name']}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per training step")
