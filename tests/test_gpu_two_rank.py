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


@pytest.mark.parametrize("overlap,k0_fused", [(False, "0"), (True, "0"), (False, "1")],
                         ids=["flat-allreduce", "segment-overlap", "flat-allreduce-cooperative-structure-chain"])
def test_two_processes_on_one_gpu_match_a_single_process_replay(tmp_path, overlap, k0_fused):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gloo_hip_worker as W
    from gnn_epc_saft_amd.data.loader import GraphLoader
    from gnn_epc_saft_amd.data.synthetic import collate, synthetic_dataset
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    # (two processes share this GPU: the cooperative structure chain loses scheduling quanta at its waits, so the
    #  rehearsals run the launches, which build the same structure -- tests/test_gpu_forward.py compares them bit for
    #  bit; the third case runs the chain anyway, as every real rank does: slow is fine, a lost barrier is not -- the
    #  workers' training loop raises on GNNSAFT_FLAG_BARRIER_TIMEOUT -- and the bits must be the same)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYTHONPATH=ROOT, GNNSAFT_K0_FUSED=k0_fused)
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


@pytest.mark.parametrize("config", [2, 5], ids=["C2-forward", "C5-training-loop"])
def test_bench_two_rank_rehearsal_prints_one_whole_job_line(config):
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one rank per process), but with
    both ranks on cuda:0 over gloo (GNNSAFT_BENCH_REHEARSAL=1): argument / environment handling, shard-by-rank
    workloads, the loss / gradient exchange, barrier + max-over-ranks timing and the single rank-0 JSON line.  The
    numbers mean nothing (two ranks share one GPU); the RCCL backend itself is the driver's to exercise."""
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, GNNSAFT_BENCH_REHEARSAL="1", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4",
           "--warmup", "2", "--config", str(config), "--train-steps", "2", "--no-cpu-baseline", "--no-c3"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]          # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["higher_is_better"] is True and d["data"] == "synthetic"
    per_step = 512 if config == 5 else 1024          # graphs per rank and step (C5: the last batch of a rank's epoch
    ratio = d["value"] * d["ms_per_step"] * 1e-3 / (2 * per_step)     # is short: 1000 graphs = 512 + 488)
    assert (0.9 < ratio <= 1.001) if config == 5 else abs(ratio - 1.0) < 0.02   # whole-job graphs / max-rank time
    assert d["input_error_flags"] == 0                # every line carries the flag word; non-zero fails the run
    if config == 2:
        assert "error" not in (d.get("train_step") or {}), d.get("train_step")
        assert d["input_error_flags_by_block"] == {"c2": 0, "c4": 0}
        # BASELINE.json configs[3]: 8192 graphs per rank on the H=256 / L=5 model, forward + loss with the loss
        # all-reduce, and the training step with the flat gradient all-reduce timed on its own, in both exchange modes
        c4 = d["c4"]
        assert "8192 synthetic molecular graphs per GPU" in c4["workload"] and "H=256 L=5" in c4["workload"]
        fl = c4["forward_loss"]
        assert abs(fl["graphs_per_s"] * fl["ms_per_step"] * 1e-3 / (2 * 8192) - 1.0) < 0.02 and fl["global_mape"] > 0
        tr = c4["train_step"]
        assert tr["gradient_bytes"] > 28_000_000           # 7.05 M parameters (+ 64-float padding) in one flat buffer
        assert tr["single_collective"]["allreduce_ms"] > 0 and tr["single_collective"]["ms_per_step"] > 0
        assert tr["segment_overlapped"]["exposed_exchange_ms"] >= 0 and tr["segment_overlapped"]["segments"] == 7
