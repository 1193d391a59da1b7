"""One rank of the two-process data-parallel rehearsal ON THE GPU (tests/test_gpu_two_rank.py): gloo backend, both
ranks on cuda:0 (a one-GPU box), the HIP path doing the per-rank compute.  What is under test is everything a rank
does between the kernels when world > 1: strided shards from GraphLoader, the rank-0 broadcast of parameters /
buffers / optimizer state (ranks start from DIFFERENT seeds), the exact global MAPE, the flat gradient all-reduce
feeding the fused optimizer through ``use_reduced_gradient`` -- with one frozen parameter, so that the gradients do
NOT sit in backward's flat buffer -- and per-rank BatchNorm statistics."""

import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import gnn_epc_saft_amd as G  # noqa: E402
from gnn_epc_saft_amd import parallel  # noqa: E402
from gnn_epc_saft_amd.data.loader import GraphLoader  # noqa: E402
from gnn_epc_saft_amd.data.synthetic import degree_histogram, synthetic_dataset  # noqa: E402
from gnn_epc_saft_amd.train.loop import training_loop  # noqa: E402

CFG = dict(propagation_depth=2, hidden_dim=64, pre_layers=1, post_layers=1, num_mlp_layers=1, num_para=3,
           skip_connections=True, add_self_loops=True, dropout_rate=0.0, model="PNAL", optimizer="adam",
           learning_rate=2e-3, weight_decay=1e-2, warmup_steps=8, momentum=0.9, num_train_steps=2,
           log_every_steps=1, checkpoint_every_steps=0)
FROZEN = "model.node_embed.atom_embedding_list.3.weight"


def build(seed, graphs, frozen=True):
    torch.manual_seed(seed)
    lit = G.create_model(CFG, degree_histogram(graphs)).to("cuda:0")
    if frozen:
        dict(lit.named_parameters())[FROZEN].requires_grad_(False)
    return lit


def main():
    out_dir = sys.argv[1]
    overlap = len(sys.argv) > 2 and sys.argv[2] == "overlap"   # segment-wise exchange under the backward: no frozen
    rank, _, world = parallel.init_from_env("gloo")            # parameter, so the gradients sit in backward's buffer
    assert world == 2 and dist.get_backend() == "gloo"
    torch.cuda.set_device(0)
    graphs = synthetic_dataset(96, 31, num_para=3)
    lit = build(100 + rank, graphs, frozen=not overlap)   # ranks start from different weights on purpose
    loader = GraphLoader(graphs, 24, shuffle=False, device="cuda:0", rank=rank, world_size=world)
    assert len(loader) == 2
    hist = training_loop(lit, loader, overlap_gradient_exchange=overlap)   # 2 steps: broadcast, fwd, bwd, all-reduce, step
    if overlap:
        segs = lit.model.gradient_segments()
        total = lit.model.flat_layout()[2]
        assert sorted(segs)[0][0] == 0 and max(b for _, b in segs) == total and len(segs) == CFG["propagation_depth"] + 2
        assert sum(b - a for a, b in segs) == total
    # eval-mode global MAPE over all 96 graphs from the per-rank [sum(ape), count] pairs
    lit.eval()
    parts = torch.zeros(3, device="cuda:0")
    with torch.no_grad():
        for b in GraphLoader(graphs, 24, shuffle=False, device="cuda:0", rank=rank, world_size=world):
            p3 = lit.model.run(b, target=b.para.view(-1, 3))[1]
            parts[1:] += p3[1:]
    glob = parallel.global_mape(parts)
    torch.save({"hist": hist, "global_mape": float(glob),
                "state": {k: v.detach().cpu() for k, v in lit.state_dict().items()}},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()
    print("GLOO_HIP_OK")


if __name__ == "__main__":
    main()
