import copy, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import oracle_model, branch_of_tape, forced_forward
from oracle.pna_torch import mape
from test_gpu_forward import hip_twin
from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch
from gnn_epc_saft_amd.train.models import mape_loss
graphs, hidden, depth = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
fused = sys.argv[4] == "1"
data = make_synthetic_batch(graphs, 1236, num_para=3)
oracle = oracle_model(hidden, depth, 1, 1, 1, 3, True, True, degree_histogram(data), seed=2).train()
hip = hip_twin(copy.deepcopy(oracle)); hip.fused_batchnorm = fused
hip.fused_readout = os.environ.get("FUSED_READOUT", "1") == "1"
dd = data.to("cuda:0")
pred = hip(dd)
branch = branch_of_tape(pred, data, True, True)
m = copy.deepcopy(oracle).double().train()
mape(forced_forward(m, data, branch), data.para.view(-1, 3).double()).backward()
g64 = {k: p.grad for k, p in m.named_parameters()}
mape_loss(pred, dd.para.view(-1, 3)).backward()
gs = max(float(g.abs().max()) for g in g64.values())
worst = sorted(((float((p.grad.double().cpu() - g64[k]).abs().max()) / max(float(g64[k].abs().max()), 1e-4 * gs), k) for k, p in hip.named_parameters()), reverse=True)[:4]
print(f"fused_readout={hip.fused_readout} X6={os.environ.get('GNNSAFT_GEMM_X6','1')} fused_bn={fused} G={graphs} H={hidden} L={depth}:", [(f"{e:.1e}", k) for e, k in worst])
