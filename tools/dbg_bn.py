import copy, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import oracle_model, rel_err
from test_gpu_forward import hip_twin
from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch
data = make_synthetic_batch(96, 107, num_para=3)
for depth in (1, 2):
    oracle = oracle_model(64, depth, 1, 1, 1, 3, True, True, degree_histogram(data), seed=4).train()
    with torch.no_grad():
        want = copy.deepcopy(oracle).double()(data)
    for fused in (True, False):
        hip = hip_twin(copy.deepcopy(oracle)); hip.fused_batchnorm = fused
        with torch.no_grad():
            out = hip(data.to("cuda:0"))
        print("depth", depth, "fused", fused, "err vs f64 oracle", rel_err(out, want), "flags", hip.input_error_flags())
