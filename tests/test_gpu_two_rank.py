"""world_size = 2 on ONE GPU: two fresh child processes (gloo backend, both on cuda:0) run the HIP path through
``training_loop``; the parent replays the same two steps in one process -- two replicas with their own BatchNorm
buffers, gradients summed and scaled by 1/2 exactly as the all-reduce + fused optimizer do -- and demands the same
bits.  (An 8-GPU RCCL run is the driver's; this keeps every line of the N > 1 host logic executed on hardware.)"""

import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


@pytest.mark.parametrize("overlap", [False, True], ids=["flat-allreduce", "segment-overlap"])
def test_two_processes_on_one_gpu_match_a_single_process_replay(tmp_path, overlap):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gloo_hip_worker as W
    from gnn_epc_saft_amd.data.loader import GraphLoader
    from gnn_epc_saft_amd.data.synthetic import collate, synthetic_dataset
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYTHONPATH=ROOT)
    worker = os.path.join(ROOT, "tests", "gloo_hip_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(tmp_path)] + (["overlap"] if overlap else []),
                              env=dict(env, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=420)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "GLOO_HIP_OK" in o, o[-3000:]
    got = [torch.load(os.path.join(str(tmp_path), f"rank{r}.pt"), weights_only=True) for r in range(2)]

    # ---- single-process replay: both replicas start from rank 0's weights (seed 100), own BatchNorm buffers
    graphs = synthetic_dataset(96, 31, num_para=3)
    reps = [W.build(100, graphs, frozen=not overlap) for _ in range(2)]
    confs = [m.configure_optimizers() for m in reps]
    shards = [list(GraphLoader(graphs, 24, shuffle=False, device=DEV, rank=r, world_size=2)) for r in range(2)]
    logged = []
    for step in range(2):
        losses = []
        for m, c, sh in zip(reps, confs, shards):
            m.train()
            c["optimizer"].zero_grad(set_to_none=True)
            loss = m.training_step(sh[step])
            loss.backward()
            losses.append(float(loss))
        trainable = [[p for p in m.parameters() if p.requires_grad] for m in reps]
        for p0, p1 in zip(*trainable):          # what the SUM all-reduce leaves on both ranks
            total = p0.grad + p1.grad
            p0.grad, p1.grad = total.clone(), total.clone()
        for c in confs:
            c["optimizer"].grad_scale = 0.5     # ... and the 1 / world folded into the optimizer kernel
            c["optimizer"].step()
            c["lr_scheduler"]["scheduler"].step()
        logged.append((step + 1, (losses[0] + losses[1]) / 2))     # sync_dist=True: mean over ranks
    for r in range(2):
        for (s_got, v_got), (s_want, v_want) in zip(got[r]["hist"], logged):
            assert s_got == s_want and abs(v_got - v_want) <= 1e-6 * abs(v_want)
        want = {k: v.detach().cpu() for k, v in reps[r].state_dict().items()}
        for k, v in got[r]["state"].items():
            assert torch.equal(v, want[k]), (r, k)
    # replicas agree on every parameter; BatchNorm running statistics stay per rank (no SyncBatchNorm, as the reference)
    names = dict(reps[0].named_parameters())
    assert all(torch.equal(got[0]["state"][k], got[1]["state"][k]) for k in names)
    assert any(not torch.equal(got[0]["state"][k], got[1]["state"][k]) for k in got[0]["state"] if "running_mean" in k)
    if not overlap:
        assert torch.equal(got[0]["state"][W.FROZEN], W.build(100, graphs).state_dict()[W.FROZEN].cpu())
    # global MAPE from the two ranks' [sum(ape), count] == one process over all 96 graphs (eval mode: rank 0's model)
    assert got[0]["global_mape"] == got[1]["global_mape"]
    reps[0].eval()
    with torch.no_grad():
        full = collate(graphs).to(DEV)
        want = float(reps[0].model.run(full, target=full.para.view(-1, 3))[1][0])
    # each rank evaluates its strided shard with ITS OWN BatchNorm statistics; rank 0's differ from rank 1's, so
    # compare with the same mixture computed here
    parts = torch.zeros(2, dtype=torch.float64)
    for r in range(2):
        reps[r].eval()
        with torch.no_grad():
            for b in GraphLoader(graphs, 24, shuffle=False, device=DEV, rank=r, world_size=2):
                parts += reps[r].model.run(b, target=b.para.view(-1, 3))[1][1:].double().cpu()
    assert abs(got[0]["global_mape"] - float(parts[0] / parts[1])) <= 1e-6 * want
