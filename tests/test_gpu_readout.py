"""The one-launch readout (csrc/readout.hip: add-pool -> Linear/BatchNorm/ReLU blocks -> Linear -> MAPE, train-mode
BatchNorm statistics across workgroups through a grid barrier) against the nine per-op launches it replaces
(``fused_readout = False``) and against the f64 oracle (reference: models.py:84-103,133-134,191-194)."""

import copy
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import check_population, oracle_model, rel_err  # noqa: E402
from oracle.pna_torch import mape  # noqa: E402
from test_gpu_forward import hip_twin  # noqa: E402

DEV = "cuda:0"


@pytest.mark.parametrize("hidden,mlp,num_para,graphs", [(128, 1, 3, 1024), (64, 1, 5, 511), (256, 2, 3, 200),
                                                        (64, 0, 5, 2), (32, 1, 3, 70), (128, 2, 5, 65)])
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_fused_readout_equals_per_op_readout_and_oracle(hidden, mlp, num_para, graphs, mode):
    from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch
    data = make_synthetic_batch(graphs, 10 + graphs, num_para=num_para)
    oracle = oracle_model(hidden, 1, 1, 1, mlp, num_para, True, True, degree_histogram(data), seed=graphs)
    oracle.train(mode == "train")
    dd, tgt = data.to(DEV), data.para.view(-1, num_para).to(DEV)
    runs = {}
    for fused in (True, False):
        m = hip_twin(copy.deepcopy(oracle))
        m.fused_readout = fused
        m.graph_kernel_max_graphs = 0
        with torch.no_grad():
            pred, loss3 = m.run(dd, target=tgt)
            pred2 = m(dd)                              # no target: the loss part is skipped
        assert m.input_error_flags() == 0
        assert torch.equal(pred, pred2) or mode == "train"   # train mode: same batch statistics, same outputs
        runs[fused] = (pred.cpu(), loss3.cpu(), {k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
    (p1, l1, s1), (p0, l0, s0) = runs[True], runs[False]
    assert rel_err(p1, p0) < 2e-6, rel_err(p1, p0)
    assert abs(float(l1[0]) - float(l0[0])) < 1e-6 * abs(float(l0[0])) and float(l1[2]) == float(l0[2]) == graphs * num_para
    for k in s0:
        if "running_" in k:
            assert rel_err(s1[k], s0[k]) < 2e-6, k
        if k.endswith("num_batches_tracked"):
            assert int(s1[k]) == int(s0[k])
    o64 = copy.deepcopy(oracle).double()
    with torch.no_grad():
        want = o64(data)
        want32 = copy.deepcopy(oracle)(data)
    if graphs >= 16:
        check_population(p1, want32, want)        # std-threshold flips: judged against the f32 oracle's own figures
    else:
        assert rel_err(p1, want) < max(1e-5, 3 * rel_err(want32, want))
    assert abs(float(l1[0]) - float(mape(want, data.para.view(-1, num_para).double()))) < 1e-4 * float(l1[0])
    if mode == "train":
        with torch.no_grad():
            o64(data)                                      # the HIP module above saw the batch twice (run + forward)
        for k, v in o64.state_dict().items():
            if k.startswith("mlp") and "running_" in k:
                assert rel_err(s1[k], v) < 1e-4, k


def test_fused_readout_feeds_the_backward_tape():
    """Gradients through the taped forward are the same whether the readout ran fused or per-op."""
    from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch
    from gnn_epc_saft_amd.train.models import mape_loss
    data = make_synthetic_batch(96, 3)
    oracle = oracle_model(64, 2, 1, 1, 1, 3, True, True, degree_histogram(data), seed=5).train()
    dd = data.to(DEV)
    grads = {}
    for fused in (True, False):
        m = hip_twin(copy.deepcopy(oracle))
        m.fused_readout = fused
        mape_loss(m(dd), dd.para.view(-1, 3)).backward()
        grads[fused] = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    gscale = max(float(v.abs().max()) for v in grads[False].values())
    for k in grads[True]:     # biases in front of a train-mode BatchNorm have an exactly-zero gradient: global floor
        scale = max(float(grads[False][k].abs().max()), 1e-4 * gscale)
        assert float((grads[True][k] - grads[False][k]).abs().max()) <= 2e-5 * scale + 1e-7 * gscale, k


def test_large_batches_take_the_per_op_readout_and_many_steps_reuse_the_counters():
    """Repeated steps and a hipGraph replay keep working because the prologue re-zeroes the barrier counters; a batch
    with more workgroups' worth of graphs than the device keeps co-resident (occupancy x CU count, asked of the
    runtime per device) takes the per-op path behind the same API."""
    from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch
    data = make_synthetic_batch(300, 8)
    oracle = oracle_model(64, 1, 1, 1, 1, 3, True, True, degree_histogram(data), seed=1).train()
    m = hip_twin(copy.deepcopy(oracle))
    dd, tgt = data.to(DEV), data.para.view(-1, 3).to(DEV)
    ref = hip_twin(copy.deepcopy(oracle))
    ref.fused_readout = False
    with torch.no_grad():
        for _ in range(5):
            p1, l1 = m.run(dd, target=tgt)
            p0, l0 = ref.run(dd, target=tgt)
            assert rel_err(p1, p0) < 2e-6 and abs(float(l1[0]) - float(l0[0])) < 1e-6 * float(l0[0])
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            pg, lg = m.run(dd, target=tgt)
        for _ in range(3):
            g.replay()
            p0, l0 = ref.run(dd, target=tgt)
        torch.cuda.synchronize()
        assert rel_err(pg, p0) < 2e-6
    assert m.input_error_flags() == 0
    big = make_synthetic_batch(16384 + 64, 9, n_min=2, n_max=3)
    ob = oracle_model(64, 1, 1, 1, 0, 3, False, True, degree_histogram(big), seed=2).eval()
    mb, rb = hip_twin(copy.deepcopy(ob)), hip_twin(copy.deepcopy(ob))
    rb.fused_readout = False
    with torch.no_grad():     # fused or per-op, whatever this device's residency bound says: same results
        assert rel_err(mb(big.to(DEV)), rb(big.to(DEV))) < 2e-6
        assert rel_err(mb.train()(big.to(DEV)), rb.train()(big.to(DEV))) < 2e-6
    assert mb.input_error_flags() == 0


def test_a_lost_grid_barrier_raises_the_flag_and_poisons_the_results():
    """The fused readout's grid barriers are only safe while every workgroup is resident.  The launcher asks the
    occupancy calculator; what that cannot see (another process on the GPU, a CU mask) ends in the bounded spin.  Force
    it (desc.debug_barrier_extra: the barriers wait for one arrival that never comes) and demand the loud failure:
    GNNSAFT_FLAG_BARRIER_TIMEOUT raised, predictions / loss NaN in the forward, every gradient below the readout NaN in
    the backward, the training loop's flag check fatal -- wrong BatchNorm statistics can never train silently."""
    from gnn_epc_saft_amd._native import lib
    from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch
    from gnn_epc_saft_amd.train.models import mape_loss
    # the residency query must WORK on this device (a failing query would silently route everything per-op)
    assert lib.gnnsaft_readout_resident_workgroups(64, 0) >= 64 and lib.gnnsaft_readout_resident_workgroups(64, 1) >= 64
    assert lib.gnnsaft_readout_resident_workgroups(256, 0) >= 64 and lib.gnnsaft_readout_resident_workgroups(256, 1) >= 64
    data = make_synthetic_batch(130, 21)
    oracle = oracle_model(64, 1, 1, 1, 1, 3, True, True, degree_histogram(data), seed=3).train()
    m = hip_twin(copy.deepcopy(oracle))
    dd, tgt = data.to(DEV), data.para.view(-1, 3).to(DEV)
    with torch.no_grad():
        good, gl = m.run(dd, target=tgt)
        assert bool(torch.isfinite(good).all()) and m.input_error_flags() == 0
        m._debug_barrier_extra = 1
        bad, bl = m.run(dd, target=tgt)
        torch.cuda.synchronize()
        assert bool(torch.isnan(bad).all()) and bool(torch.isnan(bl[0]))
        assert m.input_error_flags() & 16
        assert m.input_error_flags() == 0                 # cleared by the read
        # eval mode has no batch statistics, but the structure chain inside the forward's first launch meets at a grid
        # barrier too (elementwise.hip: k0_chain_body).  A lost one raises the flag, the next launch installs an EMPTY
        # structure instead of indexing with half-built tables AND restores the chain's persistent words, and the end
        # of the forward writes NaN in EVERY mode and on BOTH readout paths -- then, WITHOUT the host ever reading the
        # flag word, the next call is a correct forward again (the same bits as before the loss).
        m.eval()
        m.fused_structure_chain = True
        m._debug_barrier_extra = 0
        good_eval = m(dd).clone()
        m._debug_barrier_extra = 1
        bad_eval = m(dd)
        torch.cuda.synchronize()
        assert bool(torch.isnan(bad_eval).all()), "eval mode: finite garbage after a lost structure barrier"
        m.fused_readout = False
        bad_perop = m(dd)
        assert bool(torch.isnan(bad_perop).all()), "per-op readout: finite garbage after a lost structure barrier"
        m._debug_barrier_extra = 0
        again_perop = m(dd)                               # (no input_error_flags() in between)
        m.fused_readout = True
        again = m(dd)
        assert torch.equal(again, good_eval) and bool(torch.isfinite(again_perop).all())
        m.train()
        ok, okl = m.run(dd, target=tgt)                   # training mode right behind it: not poisoned by the sticky flag
        assert torch.equal(ok, good) and torch.equal(okl, gl)
        assert m.input_error_flags() & 16                 # the sticky word still tells the host what happened
        assert m.input_error_flags() == 0
        m._debug_barrier_extra = 1                        # training mode: both the chain's and the readout's barriers are lost
        bad, bl = m.run(dd, target=tgt)
        assert bool(torch.isnan(bad).all()) and bool(torch.isnan(bl[0]))
        m._debug_barrier_extra = 0
        ok, okl = m.run(dd, target=tgt)
        assert torch.equal(ok, good) and torch.equal(okl, gl)
        assert m.input_error_flags() & 16 and m.input_error_flags() == 0
    # backward: the forward's barriers pass, the backward's are lost
    m._debug_barrier_extra = 0
    pred = m(dd)
    pred.grad_fn.tape["desc"].debug_barrier_extra = 1
    mape_loss(pred, tgt).backward()
    torch.cuda.synchronize()
    assert m.input_error_flags() & 16
    grads = dict(m.named_parameters())
    assert bool(torch.isnan(grads["convs.0.lin.weight"].grad).all())
    assert bool(torch.isnan(grads["node_embed.atom_embedding_list.0.weight"].grad).any())


def _with_dropout(oracle, p):
    oracle.mlp_params.dropout = p
    for mod in oracle.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = p
    return oracle


def test_readout_dropout_masks_statistics_and_gradients():
    """Dropout of the readout MLP in training (models.py:88,95,99; config.dropout_rate through train/utils.py:66-70):
    Philox-keyed masks inside the one-launch readout, regenerated by its backward.  Checked by
      * statistics: of the open ReLU gates of every block a fraction 1 - p survives (binomial bound), survivors are
        scaled by 1 / (1 - p);
      * repeatability: the same torch seed gives the same bits, another seed another mask; p = 0 and eval mode are the
        dropout-free path bit for bit, and p = 1e-12 (mask all ones, scale exactly 1) equals it too;
      * exactness: the f64 oracle evaluated with the SAME masks (read off the tape) and on the same branch reproduces
        the prediction and, through autograd, every gradient -- the backward applies the forward's masks."""
    from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch
    from gnn_epc_saft_amd.train.models import mape_loss
    from helpers import branch_of_tape, forced_forward, tape_decisions, tape_dropout_masks
    p_drop = 0.3
    data = make_synthetic_batch(300, 41)
    oracle = _with_dropout(oracle_model(64, 2, 1, 1, 1, 3, True, True, degree_histogram(data), seed=6), p_drop).train()
    dd, tgt = data.to(DEV), data.para.view(-1, 3).to(DEV)
    hip = hip_twin(copy.deepcopy(oracle))
    assert hip.mlp_params.dropout == p_drop
    torch.manual_seed(123)
    pred = hip(dd)
    masks = tape_dropout_masks(pred)
    gates = tape_decisions(pred, True)[-3:]
    for b, (m, gte) in enumerate(zip(masks, gates)):
        open_gates = int(gte.sum())
        kept = int((m & gte).sum())
        sigma = (open_gates * p_drop * (1 - p_drop)) ** 0.5
        print(f"block {b}: {kept} of {open_gates} open gates kept ({kept / open_gates:.4f}, expected {1 - p_drop})")
        assert abs(kept - open_gates * (1 - p_drop)) <= 4.5 * sigma + 1
    branch = branch_of_tape(pred, data, True, True)          # (the tape is released by the backward)
    loss = mape_loss(pred, tgt)
    loss.backward()
    grads = {k: p.grad.detach().double().cpu() for k, p in hip.named_parameters()}
    # the exact function on the same branch with the same masks
    o64 = copy.deepcopy(oracle).double().train()
    out64 = forced_forward(o64, data, branch, dropout_masks=masks)
    mape(out64, data.para.view(-1, 3).double()).backward()
    assert rel_err(pred.detach(), out64.detach()) < 1e-5
    o32 = copy.deepcopy(oracle).train()
    mape(forced_forward(o32, data, branch, dropout_masks=masks), data.para.view(-1, 3)).backward()
    gscale = max(float(p.grad.abs().max()) for p in o64.parameters())
    g32 = dict(o32.named_parameters())
    for k, p64 in o64.named_parameters():
        scale = max(float(p64.grad.abs().max()), 1e-4 * gscale)
        if float(p64.grad.abs().max()) < 1e-6 * gscale:
            continue                                        # exactly-zero gradients (biases in front of a BatchNorm)
        e_hip = float((grads[k] - p64.grad).abs().max()) / scale
        e_f32 = float((g32[k].grad.double() - p64.grad).abs().max()) / scale
        assert e_hip <= max(3 * e_f32, 2e-5), (k, e_hip, e_f32)
    # repeatability / other seed / the dropout-free path
    twin = hip_twin(copy.deepcopy(oracle))       # (constructing a module draws from torch's generator: seed after it)
    torch.manual_seed(123)
    again = twin(dd)
    assert torch.equal(again, pred)
    twin.load_state_dict(hip.state_dict())       # (same weights; the running statistics moved, the output does not care)
    torch.manual_seed(124)
    other = twin(dd)
    assert not torch.equal(other, pred)
    plain = hip_twin(_with_dropout(copy.deepcopy(oracle), 0.0))
    tiny = hip_twin(_with_dropout(copy.deepcopy(oracle), 1e-12))
    with torch.no_grad():
        a, b = plain(dd), tiny(dd)
        assert torch.equal(a, b)
        ev = hip_twin(copy.deepcopy(oracle)).eval()
        ev0 = hip_twin(_with_dropout(copy.deepcopy(oracle), 0.0)).eval()
        assert torch.equal(ev(dd), ev0(dd))                 # eval mode: Dropout is the identity
    hip.fused_readout = False
    with pytest.raises(NotImplementedError):
        hip(dd)
    assert hip.input_error_flags() == 0


def test_readout_dropout_draws_fresh_masks_on_every_hipgraph_replay():
    """a captured training forward used to be refused with readout dropout (the Philox key is a kernel
    argument: every replay would draw ONE mask for ever).  The kernels now add a device word to the key when they run;
    ``bump_dropout_step()`` in front of a replay gives it fresh masks, and a replay WITHOUT the bump reproduces the
    previous one bit for bit (what the backward of a replayed step relies on to regenerate its forward's masks).
    GraphedTrainingStep (forward + backward + optimizer in one graph) trains with dropout the same way."""
    import gnn_epc_saft_amd as G
    from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch
    data = make_synthetic_batch(200, 77)
    oracle = _with_dropout(oracle_model(64, 2, 1, 1, 1, 3, True, True, degree_histogram(data), seed=8), 0.3).train()
    hip = hip_twin(copy.deepcopy(oracle))
    dd = data.to(DEV)
    with torch.no_grad():
        hip(dd)                                   # eager warm-up: allocates workspace and the device word
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            hip(dd)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        stats_before = {k: v.clone() for k, v in hip.state_dict().items() if "running" in k}
        with torch.cuda.graph(g):
            out = hip(dd)
        outs = []
        for bump in (True, True, False, True):
            if bump:
                hip.bump_dropout_step()
            g.replay()
            outs.append(out.clone())
    torch.cuda.synchronize()
    assert not torch.equal(outs[0], outs[1]) and not torch.equal(outs[1], outs[3])     # fresh masks
    assert torch.equal(outs[1], outs[2])                                               # same word, same masks
    assert all(torch.isfinite(o).all() for o in outs) and hip.input_error_flags() == 0
    assert stats_before                                                                # (train mode: statistics exist)
    # the whole training step from one graph, with dropout
    torch.manual_seed(5)
    lit = G.create_model(dict(propagation_depth=2, hidden_dim=64, pre_layers=1, post_layers=1, num_mlp_layers=1, num_para=3,
                              skip_connections=True, add_self_loops=True, dropout_rate=0.3, model="PNAL", optimizer="adam",
                              learning_rate=1e-3, weight_decay=1e-2, warmup_steps=8, momentum=0.9),
                         degree_histogram(data)).to(DEV)
    conf = lit.configure_optimizers()
    step = G.GraphedTrainingStep(lit, conf["optimizer"], dd, scheduler=conf["lr_scheduler"]["scheduler"], warmup=2)
    losses = [float(step()) for _ in range(6)]
    assert all(math.isfinite(v) for v in losses) and len(set(losses)) == len(losses)
    assert lit.model.input_error_flags() == 0
