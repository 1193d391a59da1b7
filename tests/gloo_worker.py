"""One rank of the two-process gloo rehearsal (tests/test_host_cpu.py).  The per-rank compute is
the CPU oracle (no GPU in this test); what is under test is the multi-rank logic of
``gnn_epc_saft_amd.parallel``: contiguous graph shards, the exact global MAPE from per-rank
[sum(ape), count], and the single flat gradient all-reduce."""

import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from gnn_epc_saft_amd import parallel  # noqa: E402
from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch  # noqa: E402
from helpers import oracle_model  # noqa: E402
from oracle.pna_torch import MAPE_EPS, mape  # noqa: E402


def main():
    rank, local_rank, world = parallel.init_from_env("gloo")
    assert world == 2 and dist.get_backend() == "gloo"
    full = make_synthetic_batch(24, 77)                      # identical on both ranks
    model = oracle_model(32, 2, 1, 1, 1, 3, True, True, degree_histogram(full)).double().eval()
    mine = parallel.shard(full, rank, world)
    assert mine.num_graphs == 12
    tgt = mine.para.view(-1, 3).double()
    pred = model(mine)
    ape = (pred - tgt).abs() / tgt.abs().clamp(min=MAPE_EPS)
    loss3 = torch.stack([ape.mean(), ape.sum(), torch.tensor(float(ape.numel()), dtype=torch.float64)])
    got = parallel.global_mape(loss3.detach())
    with torch.no_grad():
        want = mape(model(full), full.para.view(-1, 3).double())   # eval mode: shards are independent
    assert abs(float(got) - float(want)) < 1e-12 * float(want), (float(got), float(want))

    # flat gradient all-reduce == mean of the per-rank gradients
    ape.mean().backward()
    local = [p.grad.clone() for p in model.parameters() if p.requires_grad]
    parallel.FlatGradientAllReduce(model.parameters())()
    gathered = []
    for g in local:
        both = [torch.zeros_like(g) for _ in range(world)]
        dist.all_gather(both, g)
        gathered.append(sum(both) / world)
    for p, w in zip([p for p in model.parameters() if p.requires_grad], gathered):
        assert torch.allclose(p.grad, w, rtol=1e-12, atol=1e-15)
    # the layout gnnsaft_backward leaves behind: every .grad a view of ONE flat buffer -> reduced in place, and
    # allreduce_flat_sum hands back the 1/world factor the fused optimizer folds into its kernel
    params = [p for p in model.parameters() if p.requires_grad]
    flat = torch.zeros(sum((p.numel() + 63) // 64 * 64 for p in params), dtype=torch.float64)
    off = 0
    for p, g in zip(params, local):
        view = flat[off:off + p.numel()].view(p.shape)
        view.copy_(g)
        p.grad = view
        off += (p.numel() + 63) // 64 * 64
    assert parallel.common_gradient_buffer(params) is flat
    before = flat.clone()
    scale = parallel.allreduce_flat_sum(flat)
    assert scale == 1.0 / world
    for p, w in zip(params, gathered):
        assert torch.allclose(p.grad * scale, w, rtol=1e-12, atol=1e-15)      # views see the reduced buffer
    for p, g in zip(params, local):
        p.grad.copy_(g)
    parallel.FlatGradientAllReduce(params)()                                   # zero-copy path of the class
    for p, w in zip(params, gathered):
        assert torch.allclose(p.grad, w, rtol=1e-12, atol=1e-15)
    assert not torch.equal(before, flat) or world == 1
    dist.barrier()
    dist.destroy_process_group()
    print("GLOO_OK")


if __name__ == "__main__":
    main()
