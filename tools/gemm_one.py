"""Times ONE GEMM shape with an explicit tile configuration: gemm_one.py M N_OUT K CFG"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_epc_saft_amd.kernels as K
from tools.gemm_tune import timeit
m, n_out, k, cfg = (int(v) for v in sys.argv[1:5])
a = torch.randn(m, k, device="cuda"); w = torch.randn(n_out, k, device="cuda") / k ** 0.5
t = min(timeit(lambda: K.linear(a, w, None, tile_config=cfg)) for _ in range(3))
print(f"{os.environ.get('GNNSAFT_LIB', 'default'):60s} [{m},{k}]x[{k},{n_out}] cfg {cfg}: {t:8.1f} us  {2.0*m*n_out*k/t/1e6:7.1f} TF-equivalent")
