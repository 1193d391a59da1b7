"""Training-step host loop (train/loop.py): gnnsaft_forward -> MAPE -> gnnsaft_backward -> fused AdamW, scheduler
per step, logging, Lightning-dialect checkpoints and resume (reference: Trainer.fit over PNApcsaftL,
/root/reference/gnnepcsaft/train/train.py:142-185 with models.py:162-202)."""

import copy
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from test_gpu_stages import DEV  # noqa: E402


def _setup(seed=0):
    import gnn_epc_saft_amd as G
    from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch
    batches = [make_synthetic_batch(64, 500 + i).to(DEV) for i in range(3)]
    cfg = dict(propagation_depth=2, hidden_dim=64, pre_layers=1, post_layers=1, num_mlp_layers=1, num_para=3,
               skip_connections=True, add_self_loops=True, dropout_rate=0.0, model="PNAL", optimizer="adam",
               learning_rate=2e-3, weight_decay=1e-2, warmup_steps=8, momentum=0.9, num_train_steps=12,
               log_every_steps=3, checkpoint_every_steps=6)
    torch.manual_seed(seed)
    lit = G.create_model(cfg, degree_histogram(batches)).to(DEV)
    return lit, batches, cfg


def test_loop_trains_logs_checkpoints_and_resumes(tmp_path):
    from gnn_epc_saft_amd.train import checkpoint as C
    from gnn_epc_saft_amd.train.loop import training_loop
    lit, batches, cfg = _setup()
    start = copy.deepcopy(lit.state_dict())
    seen = []
    hist = training_loop(lit, batches, workdir=str(tmp_path), on_log=lambda s, v, lr: seen.append((s, v, lr)))
    assert [s for s, _ in hist] == [3, 6, 9, 12] and [s for s, _, _ in seen] == [3, 6, 9, 12]
    assert hist[-1][1] < hist[0][1]                                  # it learns
    lrs = [lr for _, _, lr in seen]
    assert lrs[0] < 2e-3 and lrs[2] > lrs[1]                         # cosine decay, restart after warmup_steps = 8
    ck6 = os.path.join(str(tmp_path), "train", "checkpoints", "step=6.ckpt")
    ck12 = os.path.join(str(tmp_path), "train", "checkpoints", "step=12.ckpt")
    assert os.path.exists(ck6) and os.path.exists(ck12)
    final = {k: v.detach().cpu() for k, v in lit.state_dict().items()}
    # forward, backward and the fused optimizer are free of atomics: a second uninterrupted run gives the same bits
    lit_b, _, _ = _setup()
    training_loop(lit_b, batches, checkpoint_every_steps=0)
    for k, v in lit_b.state_dict().items():
        assert torch.equal(v.cpu(), final[k]), k
    # resume from step 6 in a fresh module
    lit2, _, _ = _setup(seed=1)
    ck = C.load_checkpoint(lit2, ck6)
    assert ck["global_step"] == 6 and "optimizer_states" in ck and "lr_schedulers" in ck
    conf = lit2.configure_optimizers()                               # restored state is exact
    assert C.resume(ck, conf["optimizer"], conf["lr_scheduler"]["scheduler"]) == 6
    st = conf["optimizer"].state_dict()["state"]
    for i, want in ck["optimizer_states"][0]["state"].items():
        assert float(st[i]["step"]) == 6.0
        for key in ("exp_avg", "exp_avg_sq", "max_exp_avg_sq"):
            assert torch.equal(st[i][key].cpu(), want[key]), (i, key)
    assert conf["lr_scheduler"]["scheduler"].last_epoch == 6
    hist2 = training_loop(lit2, batches, resume_from=ck)
    assert [s for s, _ in hist2] == [9, 12]
    for (s1, v1), (s2, v2) in zip(hist[2:], hist2):
        assert s1 == s2 and v1 == v2                                 # the resumed trajectory is the same trajectory
    for k, v in lit2.state_dict().items():
        assert torch.equal(v.cpu(), final[k]), k                     # ... bit for bit (num_batches_tracked: 12 on both)
    # the checkpoint loads into the bare module through the other dialect's entry point too
    import gnn_epc_saft_amd as G
    bare = G.PNAPCSAFT(64, lit.model.pna_params, lit.model.mlp_params)
    C.load_checkpoint(bare, ck12)
    for k, v in bare.state_dict().items():
        assert torch.equal(v, final["model." + k])
    assert any(not torch.equal(start[k].cpu(), final[k]) for k in final)


def test_sgd_configuration_steps_too():
    from gnn_epc_saft_amd.train.loop import training_loop
    from gnn_epc_saft_amd.train.optim import FusedSGD
    lit, batches, cfg = _setup()
    lit.config = dict(cfg, optimizer="sgd", learning_rate=1e-3)
    assert isinstance(lit.configure_optimizers()["optimizer"], FusedSGD)
    hist = training_loop(lit, batches, max_steps=8, log_every_steps=4, checkpoint_every_steps=0)
    assert len(hist) == 2 and all(v == v for _, v in hist)


def test_loader_feeds_the_loop_with_overlapped_copies():
    """GraphLoader (pinned staging, copies on a side stream) delivers the same device batches as a direct
    ``.to(device)`` of the collated graphs, and the training loop consumes it over several epochs."""
    import gnn_epc_saft_amd as G
    from gnn_epc_saft_amd.data.loader import GraphLoader
    from gnn_epc_saft_amd.data.synthetic import collate, degree_histogram
    from gnn_epc_saft_amd.train.loop import training_loop
    from test_host_cpu import _graph_list
    graphs = _graph_list(150, 9)
    ld = GraphLoader(graphs, 64, shuffle=False, device=DEV)
    got = list(ld)
    assert [b.num_graphs for b in got] == [64, 64, 22]
    for k, b in enumerate(got):
        want = collate(graphs[64 * k:64 * (k + 1)])
        for f in ("x", "edge_index", "edge_attr", "batch", "ptr", "para"):
            assert getattr(b, f).device.type == "cuda" and torch.equal(getattr(b, f).cpu(), getattr(want, f)), f
    cfg = dict(propagation_depth=2, hidden_dim=64, pre_layers=1, post_layers=1, num_mlp_layers=1, num_para=3,
               skip_connections=True, add_self_loops=True, dropout_rate=0.0, model="PNAL", optimizer="adam",
               learning_rate=2e-3, weight_decay=1e-2, warmup_steps=50, momentum=0.9, num_train_steps=12,
               log_every_steps=3, checkpoint_every_steps=0)
    torch.manual_seed(0)
    lit = G.create_model(cfg, degree_histogram(graphs)).to(DEV)
    hist = training_loop(lit, GraphLoader(graphs, 64, shuffle=True, device=DEV, seed=1))      # 4 epochs of 3 batches
    assert [s for s, _ in hist] == [3, 6, 9, 12] and hist[-1][1] < hist[0][1]


def test_loader_with_cached_batches_and_structure():
    """shuffle=False + cache_on_device: batches (and their CSR / degree tiles) are built once, later epochs reuse
    them; the training trajectory equals the uncached one bit for bit."""
    import gnn_epc_saft_amd as G
    from gnn_epc_saft_amd.data.loader import GraphLoader
    from gnn_epc_saft_amd.data.synthetic import degree_histogram
    from gnn_epc_saft_amd.train.loop import training_loop
    from test_host_cpu import _graph_list
    graphs = _graph_list(150, 9)
    cfg = dict(propagation_depth=2, hidden_dim=64, pre_layers=1, post_layers=1, num_mlp_layers=1, num_para=3,
               skip_connections=True, add_self_loops=True, dropout_rate=0.0, model="PNAL", optimizer="adam",
               learning_rate=2e-3, weight_decay=1e-2, warmup_steps=50, momentum=0.9, num_train_steps=9,
               log_every_steps=3, checkpoint_every_steps=0)
    def run(cached):
        torch.manual_seed(0)
        lit = G.create_model(cfg, degree_histogram(graphs)).to(DEV)
        ld = GraphLoader(graphs, 64, device=DEV, cache_on_device=cached, structure_for=lit.model if cached else None)
        hist = training_loop(lit, ld)
        if cached:
            assert ld._cache is not None and all(b.gnnsaft_structure is not None for b in ld._cache)
        return hist, [p.detach().clone() for p in lit.parameters()]
    h0, p0 = run(False)
    h1, p1 = run(True)
    assert h0 == h1
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)
    with pytest.raises(ValueError):
        GraphLoader(graphs, 64, shuffle=True, device=DEV, cache_on_device=True)


def test_full_size_training_steps_are_finite_and_learn():
    """BASELINE.json config 2 (1024 graphs, H=128, L=3): a few full training steps at the benchmark size -- every
    gradient finite, the loss goes down, running statistics move (the oracle is too slow to check gradients here;
    tests/test_gpu_backward.py does that on small batches)."""
    import gnn_epc_saft_amd as G
    from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch
    data = make_synthetic_batch(1024, 1236, num_para=3)
    cfg = dict(hidden_dim=128, num_para=3, optimizer="adam", learning_rate=1e-3, weight_decay=1e-2, warmup_steps=100,
               momentum=0.9)
    torch.manual_seed(0)
    lit = G.PNApcsaftL(G.PnaconvsParams(3, 1, 1, degree_histogram(data), skip_connections=True, self_loops=True),
                       G.ReadoutMLPParams(1, 3), cfg).to(DEV).train()
    conf = lit.configure_optimizers()
    opt, sched = conf["optimizer"], conf["lr_scheduler"]["scheduler"]
    dd = data.to(DEV)
    losses = []
    for _ in range(8):
        opt.zero_grad(set_to_none=True)
        loss = lit.training_step(dd)
        loss.backward()
        for name, p in lit.named_parameters():
            assert p.grad is not None and torch.isfinite(p.grad).all(), name
        opt.step()
        sched.step()
        losses.append(float(loss))
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses
    assert int(lit.model.batch_norms[0].module.num_batches_tracked) == 8
    assert lit.model.input_error_flags() == 0


def test_c5_standin_default_model_training_loop():
    """BASELINE.json configs[4] as SURVEY.md 8(d) states its stand-in (the ThermoML-derived dataset is a DVC pointer
    that cannot be fetched): 2 000 synthetic graphs, batch 512 (configs/default.py:20), the default model H=64 L=6
    pre=post=1 mlp=1 P=5 skip + self-loops (configs/default.py:35-45), AdamW(amsgrad) + CosineAnnealingWarmRestarts,
    shuffled epochs, `num_train_steps` shortened.  Checks: the first step's loss equals the f64 oracle's on the same
    batch, the loop learns, no index was clamped, and the TRAINED weights give the oracle's predictions (eval mode,
    per-element gate) -- i.e. the loop trained the model the reference arithmetic describes."""
    import gnn_epc_saft_amd as G
    from gnn_epc_saft_amd.data.loader import GraphLoader
    from gnn_epc_saft_amd.data.synthetic import collate, degree_histogram, synthetic_dataset
    from gnn_epc_saft_amd.train.loop import training_loop
    from helpers import check_population
    from oracle.pna_torch import OracleMlpParams, OraclePNAPCSAFT, OraclePnaParams, mape
    graphs = synthetic_dataset(2000, 1239, num_para=5)
    deg = degree_histogram(graphs)
    cfg = dict(propagation_depth=6, hidden_dim=64, pre_layers=1, post_layers=1, num_mlp_layers=1, num_para=5,
               skip_connections=True, add_self_loops=True, dropout_rate=0.0, model="PNAL", optimizer="adam",
               learning_rate=1e-3, weight_decay=1e-2, warmup_steps=100, momentum=0.9, num_train_steps=24,
               log_every_steps=1, checkpoint_every_steps=0)
    torch.manual_seed(0)
    lit = G.create_model(cfg, deg).to(DEV)
    oracle = OraclePNAPCSAFT(64, OraclePnaParams(6, 1, 1, deg, skip_connections=True, self_loops=True),
                             OracleMlpParams(1, 5))
    oracle.load_state_dict({k: v.detach().cpu() for k, v in lit.model.state_dict().items()})
    loader = GraphLoader(graphs, 512, shuffle=True, device=DEV, seed=3)
    assert len(loader) == 4                                           # 512, 512, 512, 464 graphs per epoch
    first = collate([graphs[i] for i in GraphLoader(graphs, 512, shuffle=True, seed=3)._batches()[0].tolist()])
    with torch.no_grad():
        want_first = float(mape(copy.deepcopy(oracle).double().train()(first), first.para.view(-1, 5).double()))
    hist = training_loop(lit, loader)                                 # 24 steps = 6 epochs
    assert [s for s, _ in hist] == list(range(1, 25))
    assert abs(hist[0][1] - want_first) <= 1e-5 * want_first, (hist[0][1], want_first)
    assert all(v == v for _, v in hist) and hist[-1][1] < 0.9 * hist[0][1], hist
    assert lit.model.input_error_flags() == 0
    assert int(lit.model.batch_norms[0].module.num_batches_tracked) == 24
    # the trained weights, evaluated by the oracle in f64 on a held-in batch, against the HIP eval-mode forward
    oracle.load_state_dict({k: v.detach().cpu() for k, v in lit.model.state_dict().items()})
    probe = collate(graphs[:256])
    lit.eval()
    with torch.no_grad():
        got = lit(probe.to(DEV)).cpu()
        want32 = copy.deepcopy(oracle).eval()(probe)
        want = oracle.double().eval()(probe)
    print(f"C5 stand-in: loss {hist[0][1]:.4f} -> {hist[-1][1]:.4f}; eval predictions of the trained model:")
    check_population(got, want32, want)


def test_graphed_training_step_auto_choice_trains_like_eager_steps():
    """GraphedTrainingStep(choose="auto") times replays against eager steps at construction (the graphed
    step must never be the slower one) -- every trial step is a real training step, replay and eager step compute the
    same bits, so whatever mix of the two ran, the trajectory equals the plain eager loop's."""
    import gnn_epc_saft_amd as G
    lit_a, batches, _ = _setup()
    lit_b, _, _ = _setup()
    batch = batches[0]
    conf = lit_b.configure_optimizers()
    opt_b, sched_b = conf["optimizer"], conf["lr_scheduler"]["scheduler"]
    lit_b.train()
    graphed = G.GraphedTrainingStep(lit_b, opt_b, batch, scheduler=sched_b, warmup=2, choose="auto", trial_steps=3)
    assert graphed.mode in ("graph", "eager") and set(graphed.trial_ms) == {"graph", "eager"}
    print(f"auto choice: {graphed.mode}, trial ms per step {graphed.trial_ms}")
    for _ in range(3):
        graphed()
    total = 2 + (1 + 3) + (1 + 3) + 3      # warm-up, the two trials (one untimed step each), the calls above
    torch.cuda.synchronize()
    conf = lit_a.configure_optimizers()
    opt_a, sched_a = conf["optimizer"], conf["lr_scheduler"]["scheduler"]
    lit_a.train()
    for _ in range(total):
        opt_a.zero_grad(set_to_none=True)
        lit_a.training_step(batch).backward()
        opt_a.step()
        sched_a.step()
    assert sched_b.last_epoch == sched_a.last_epoch == total
    for (ka, va), (kb, vb) in zip(lit_a.state_dict().items(), lit_b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka


@pytest.mark.parametrize("two_streams", [False, True], ids=["one-stream", "two-stream-capture"])
def test_graphed_training_step_equals_eager_steps(two_streams):
    """GraphedTrainingStep (forward + backward + fused AdamW in one captured hipGraph, learning rate and bias
    corrections refreshed through the captured host-to-device copy) against the same number of eager steps on the
    same batch: the kernels and their order are the same, so parameters, optimizer state and BatchNorm statistics
    must agree bit for bit."""
    import gnn_epc_saft_amd as G
    lit_a, batches, _ = _setup()
    lit_b, _, _ = _setup()
    # the backward's side stream forks from and joins the capturing stream inside the capture (both modules alike:
    # the two schedules give the same bits, test_backward_schedules_agree)
    lit_a.model.backward_side_stream = lit_b.model.backward_side_stream = two_streams
    batch = batches[0]
    steps, warm = 7, 2

    conf = lit_a.configure_optimizers()
    opt_a, sched_a = conf["optimizer"], conf["lr_scheduler"]["scheduler"]
    lit_a.train()
    losses_a = []
    for _ in range(steps):
        opt_a.zero_grad(set_to_none=True)
        loss = lit_a.training_step(batch)
        loss.backward()
        opt_a.step()
        sched_a.step()
        losses_a.append(float(loss))

    conf = lit_b.configure_optimizers()
    opt_b, sched_b = conf["optimizer"], conf["lr_scheduler"]["scheduler"]
    lit_b.train()
    graphed = G.GraphedTrainingStep(lit_b, opt_b, batch, scheduler=sched_b, warmup=warm)
    losses_b = [float(graphed()) for _ in range(steps - warm)]
    torch.cuda.synchronize()
    assert losses_b == losses_a[warm:]
    assert losses_b[-1] < losses_a[0]
    assert sched_b.last_epoch == sched_a.last_epoch == steps
    for (ka, va), (kb, vb) in zip(lit_a.state_dict().items(), lit_b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka
    sa, sb = opt_a.state_dict()["state"], opt_b.state_dict()["state"]
    for i in sa:
        assert float(sa[i]["step"]) == float(sb[i]["step"]) == steps
        for key in ("exp_avg", "exp_avg_sq", "max_exp_avg_sq"):
            assert torch.equal(sa[i][key], sb[i][key]), (i, key)


def test_graphed_replays_issued_back_to_back_equal_eager_steps():
    """N replays enqueued WITHOUT a host synchronisation in between (the host runs several steps ahead of the GPU:
    bench.py's hipgraph loop, any fixed-batch training loop) against N eager steps: same bits.  Early steps are the
    sensitive ones (the learning rate / bias corrections of steps 3 and 4 differ by ~27 %): a replay that read the
    per-step scalars of a LATER step -- a host buffer rewritten while replays are still queued -- fails here."""
    import gnn_epc_saft_amd as G
    lit_a, batches, _ = _setup()
    lit_b, _, _ = _setup()
    batch = batches[0]
    steps, warm = 12, 1
    conf = lit_a.configure_optimizers()
    opt_a, sched_a = conf["optimizer"], conf["lr_scheduler"]["scheduler"]
    lit_a.train()
    for _ in range(steps):
        opt_a.zero_grad(set_to_none=True)
        lit_a.training_step(batch).backward()
        opt_a.step()
        sched_a.step()
    conf = lit_b.configure_optimizers()
    opt_b, sched_b = conf["optimizer"], conf["lr_scheduler"]["scheduler"]
    lit_b.train()
    graphed = G.GraphedTrainingStep(lit_b, opt_b, batch, scheduler=sched_b, warmup=warm)
    torch.cuda.synchronize()
    # keep the GPU busy so that the replays below queue up behind it while the host runs ahead
    ballast = torch.randn(8192, 8192, device=DEV)
    for _ in range(4):
        ballast = ballast @ ballast * 1e-4
    for _ in range(steps - warm):
        graphed()                 # no .item(), no synchronize: the host publishes step k+1 while step k is queued
    torch.cuda.synchronize()
    for (ka, va), (kb, vb) in zip(lit_a.state_dict().items(), lit_b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka
    sa, sb = opt_a.state_dict()["state"], opt_b.state_dict()["state"]
    for i in sa:
        for key in ("exp_avg", "exp_avg_sq", "max_exp_avg_sq"):
            assert torch.equal(sa[i][key], sb[i][key]), (i, key)


def test_loader_prefetch_thread_yields_the_same_batches_and_stops_cleanly():
    """GraphLoader(prefetch=2) -- batches collated, staged and copied by a background thread -- against prefetch=0
    (caller's thread): identical batches in identical order over two reshuffled epochs; forever() continues across
    the epoch boundary with the same sequence; abandoning an iteration leaves no producer thread behind."""
    import threading
    from gnn_epc_saft_amd.data.loader import GraphLoader
    from gnn_epc_saft_amd.data.synthetic import synthetic_dataset
    graphs = synthetic_dataset(300, 77, num_para=3)

    def same(x, y):
        assert x.num_graphs == y.num_graphs
        for name in ("x", "edge_index", "edge_attr", "batch", "ptr", "para"):
            assert torch.equal(getattr(x, name), getattr(y, name)), name

    a = GraphLoader(graphs, 64, shuffle=True, device=DEV, seed=3, prefetch=0)
    b = GraphLoader(graphs, 64, shuffle=True, device=DEV, seed=3, prefetch=2)
    expected = []
    for _ in range(2):
        ea, eb = list(a), list(b)
        assert len(ea) == len(eb) == 5
        for x, y in zip(ea, eb):
            same(x, y)
        expected.extend(ea)
    c = GraphLoader(graphs, 64, shuffle=True, device=DEV, seed=3, prefetch=2)
    it = c.forever()
    for want in expected:
        same(next(it), want)
    it.close()                                   # generator abandoned mid-stream
    for x in GraphLoader(graphs, 64, shuffle=False, device=DEV, prefetch=2):
        break                                    # ... and a plain epoch left after its first batch
    torch.cuda.synchronize()
    import time
    deadline = time.time() + 5.0
    while any(t.name == "gnnsaft-loader" and t.is_alive() for t in threading.enumerate()) and time.time() < deadline:
        time.sleep(0.05)
    assert not any(t.name == "gnnsaft-loader" and t.is_alive() for t in threading.enumerate())
