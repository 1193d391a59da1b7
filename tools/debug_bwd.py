import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from helpers import oracle_model
from test_gpu_forward import hip_twin
from test_gpu_backward import grads_of
from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch
from gnn_epc_saft_amd.train.models import mape_loss
for (hidden, depth, mlp, P, skip, loops) in [(64,1,1,3,False,True),(64,2,1,3,False,True),(64,2,1,3,True,True)]:
    data = make_synthetic_batch(48, 7, num_para=P)
    oracle = oracle_model(hidden, depth, 1, 1, mlp, P, skip, loops, degree_histogram(data), seed=2).train()
    l64, g64 = grads_of(oracle, data, P, torch.float64)
    hip = hip_twin(copy.deepcopy(oracle)); dd = data.to("cuda:0")
    loss = mape_loss(hip(dd), dd.para.view(-1, P)); loss.backward()
    gs = max(float(g.abs().max()) for g in g64.values())
    print("==", hidden, depth, mlp, P, skip, loops)
    for name, p in hip.named_parameters():
        sc = max(float(g64[name].abs().max()), 1e-4*gs)
        e = float((p.grad.double().cpu()-g64[name]).abs().max())/sc
        if e > 5e-5: print(f"   {name:45s} {e:.2e}")
