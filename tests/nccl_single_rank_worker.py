"""ONE rank on the RCCL backend ("nccl" on ROCm) with every collective of the data-parallel path forced through it
(GNNSAFT_FORCE_COLLECTIVES=1; tests/test_gpu_nccl_single_rank.py starts this under torch.distributed.run).  A one-GPU box
cannot measure scaling, but it can run what every real N > 1 rank runs beside its kernels: communicator creation, the
asynchronous [sum(ape), count] all-reduce behind a hipGraph replay of the forward (bench.py's step), the rank-0
broadcast of the training state, ONE flat all-reduce of the 28 MB gradient buffer of the BASELINE configs[2] / [3] model,
and the segment-wise exchange on its side stream behind gnnsaft_backward's events -- with the cooperative structure
chain of the forward (the default) resident next to RCCL's kernels.  An all-reduce over one rank changes nothing, so
every result must equal the collective-free run bit for bit."""

import copy
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import gnn_epc_saft_amd as G  # noqa: E402
from gnn_epc_saft_amd import parallel  # noqa: E402
from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch  # noqa: E402
from gnn_epc_saft_amd.train.loop import training_loop  # noqa: E402

CFG = dict(propagation_depth=5, hidden_dim=256, pre_layers=1, post_layers=1, num_mlp_layers=1, num_para=3,
           skip_connections=True, add_self_loops=True, dropout_rate=0.0, model="PNAL", optimizer="adam",
           learning_rate=1e-3, weight_decay=1e-2, warmup_steps=8, momentum=0.9, num_train_steps=2,
           log_every_steps=1, checkpoint_every_steps=0)


def build(data):
    torch.manual_seed(7)
    return G.create_model(CFG, degree_histogram(data)).to("cuda:0")


def main():
    assert os.environ.get("GNNSAFT_FORCE_COLLECTIVES") == "1"
    rank, local_rank, world = parallel.init_from_env("nccl")
    assert world == 1 and dist.is_initialized() and dist.get_backend() == "nccl", (world, dist.is_initialized())
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    data = make_synthetic_batch(256, 11, num_para=3)
    ddev = data.to(dev)
    ref = build(data)          # collective-free twin
    lit = copy.deepcopy(ref)
    assert lit.model.fused_structure_chain, "the cooperative structure chain is the default every rank runs"

    # ---- (1) forward + loss replayed from a hipGraph, the loss exchange issued behind every replay (bench.py)
    os.environ["GNNSAFT_FORCE_COLLECTIVES"] = "0"
    with torch.no_grad():
        want = parallel.global_mape(ref.training_step_parts(ddev)).clone()
    os.environ["GNNSAFT_FORCE_COLLECTIVES"] = "1"
    assert parallel.exchange_active()
    stream = torch.cuda.Stream(dev)
    with torch.no_grad(), torch.cuda.stream(stream):
        for _ in range(2):
            first = parallel.global_mape_async(lit.training_step_parts(ddev))
        assert first.work is not None, "the loss all-reduce was not issued"
        eager = first.result().clone()
        torch.cuda.synchronize(dev)
        cap = torch.cuda.Stream(dev)
        cap.wait_stream(stream)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=cap, capture_error_mode="thread_local"):   # beside a live process group
            parts = lit.training_step_parts(ddev)
        stream.wait_stream(cap)
        pend = []
        for _ in range(5):
            graph.replay()
            pend.append(parallel.global_mape_async(parts))
        got = [h.result().clone() for h in pend]
        dist.barrier()
        torch.cuda.synchronize(dev)
    assert torch.equal(eager, want) and all(torch.equal(g, want) for g in got), (float(eager), float(want))
    assert lit.model.input_error_flags() == 0

    # ---- (2) training: broadcast of the state, single flat all-reduce, then the segment-wise overlapped exchange
    def train(model, overlap, forced):
        os.environ["GNNSAFT_FORCE_COLLECTIVES"] = "1" if forced else "0"
        hist = training_loop(model, [ddev], overlap_gradient_exchange=overlap)
        torch.cuda.synchronize(dev)
        return hist, {k: v.detach().clone() for k, v in model.state_dict().items()}

    for overlap in (False, True):
        a, b = copy.deepcopy(ref), copy.deepcopy(ref)
        hist_ref, state_ref = train(a, overlap, forced=False)
        hist, state = train(b, overlap, forced=True)
        assert hist == hist_ref, (hist, hist_ref)
        for k, v in state.items():
            assert torch.equal(v, state_ref[k]), (overlap, k)
        assert a.model.input_error_flags() == 0 and b.model.input_error_flags() == 0
    os.environ["GNNSAFT_FORCE_COLLECTIVES"] = "1"

    # ---- (3) the flat gradient buffer of this model through ONE RCCL all-reduce, timed by events on the stream
    conf = lit.configure_optimizers()
    opt = conf["optimizer"]
    opt.zero_grad(set_to_none=True)
    lit.train()
    lit.training_step(ddev).backward()
    flat = opt._flat_grad()
    before = flat.clone()
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    ev[0].record()
    scale = parallel.allreduce_flat_sum(flat)
    ev[1].record()
    torch.cuda.synchronize(dev)
    assert scale == 1.0 and torch.equal(flat, before) and flat.numel() * 4 > 28_000_000
    print(f"flat gradient all-reduce on RCCL, one rank: {flat.numel() * 4 / 1e6:.1f} MB in {ev[0].elapsed_time(ev[1]):.3f} ms")
    dist.barrier()
    dist.destroy_process_group()
    print("NCCL_SINGLE_OK")


if __name__ == "__main__":
    main()
