import copy, sys, os
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,"tests"))
import torch
from helpers import oracle_model
from test_gpu_forward import hip_twin
from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch
from gnn_epc_saft_amd.train.models import mape_loss
data = make_synthetic_batch(24, 77, num_para=3); dd = data.to("cuda:0")
for post in (1,2,3):
    oracle = oracle_model(128, 2, 1, post, 1, 3, True, True, degree_histogram(data), seed=2).train()
    hip = hip_twin(copy.deepcopy(oracle))
    runs=[]
    for r in range(3):
        hip.zero_grad()
        junk = torch.full((50_000_000,), float(r+1)*1e3, device="cuda:0"); del junk   # dirty the allocator's free blocks
        mape_loss(hip(dd), dd.para.view(-1,3)).backward()
        runs.append({n:p.grad.clone() for n,p in hip.named_parameters()})
    for n in runs[0]:
        d = max(float((runs[0][n]-runs[k][n]).abs().max()) for k in (1,2)); s=float(runs[0][n].abs().max())+1e-30
        if d/s > 1e-5: print(post, n, d/s)
    print("post", post, "done")
