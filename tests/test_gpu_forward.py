"""End-to-end parity of the HIP ``PNAPCSAFT`` against the CPU oracle and the committed golden
vectors, plus size-independent properties at the BASELINE.json sizes.

Tolerance.  north_star asks for 1e-5 relative f32; SURVEY 8(d) states the gate per element:
|a - b| / max(|b|, 1e-6 max|b|) <= 1e-5.  Measured facts that shape how it is asserted (helpers.check_population):
the reference arithmetic has a discontinuity (PyG StdAggregation zeroes std where var <= 1e-5) that makes ANY two
f32 evaluations of it -- including the oracle run in f32 vs f64 -- disagree by 1e-4..1e-3 on the graphs whose segment
variance lands on the threshold, and outputs of a random batch pass through zero, where no f32 evaluation is
1e-5-accurate relative to the element itself.  Therefore:
  * golden fixtures are generated threshold-free; there the per-element gate is asserted outright in eval mode
    (<= 1e-5) and against 3x the f32 oracle's own value in train mode (BatchNorm over FOUR rows);
  * on random batches (shape envelope, BASELINE config 2 at full size) the per-graph distribution of the gate must be
    as good as the f32 oracle's own (50 % / 90 % quantiles within 3x, fraction within 1e-5 not lower beyond sampling
    noise, worst graph within 3x the f32 oracle's worst or one flip).  Both distributions are printed.
"""

import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import fill_deterministic, list_cases, load_case  # noqa: E402
from helpers import check_population, gate_err, mini4, oracle_model, rel_err  # noqa: E402
from oracle.pna_torch import OracleMlpParams, OraclePNAPCSAFT, OraclePnaParams, mape  # noqa: E402

DEV = "cuda:0"
TOL = 1e-5


def hip_twin(oracle: OraclePNAPCSAFT):
    """HIP module with the oracle's constructor arguments and state_dict."""
    import gnn_epc_saft_amd as G
    p, q = oracle.pna_params, oracle.mlp_params
    hidden = oracle.node_embed.atom_embedding_list[0].weight.shape[1]
    m = G.PNAPCSAFT(hidden, G.PnaconvsParams(p.propagation_depth, p.pre_layers, p.post_layers, p.deg,
                                             skip_connections=p.skip_connections, self_loops=p.self_loops),
                    G.ReadoutMLPParams(q.num_mlp_layers, q.num_para, dropout=q.dropout))
    missing = m.load_state_dict({k: v.float() for k, v in oracle.state_dict().items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m.to(DEV).train(oracle.training)


def graph_data(case):
    from gnn_epc_saft_amd.data.synthetic import GraphData
    t = lambda k: torch.from_numpy(case[k]) if k in case else None
    return GraphData(t("x"), t("edge_index"), t("edge_attr"), t("batch"), None, t("para"))


@pytest.mark.parametrize("name", list_cases())
def test_golden_vectors(name):
    case = load_case(name)
    hidden, depth, pre, post, mlp, num_para, skip, loops = (int(v) for v in case["config"])
    deg = torch.from_numpy(case["deg"])
    oracle = OraclePNAPCSAFT(hidden, OraclePnaParams(depth, pre, post, deg, skip_connections=bool(skip),
                                                     self_loops=bool(loops)), OracleMlpParams(mlp, num_para))
    checksum = fill_deterministic(oracle, int(case["seed"][0]))
    assert checksum == float(case["weights_checksum"][0])
    data = graph_data(case)
    for mode in ("eval", "train"):
        if f"out_{mode}_f64" not in case:
            continue
        oracle.train(mode == "train")
        hip = hip_twin(copy.deepcopy(oracle))
        with torch.no_grad():
            out = hip(data.to(DEV))
        assert hip.input_error_flags() == 0
        want64 = torch.from_numpy(case[f"out_{mode}_f64"])
        want32 = torch.from_numpy(case[f"out_{mode}_f32"])
        bar = max(TOL, 3 * rel_err(want32, want64))
        assert rel_err(out, want64) <= bar, (name, mode, rel_err(out, want64), bar)
        # the SURVEY 8(d) gate, per element.  Eval mode: 1e-5 outright.  Train mode on these 4-graph fixtures runs
        # BatchNorm1d over FOUR rows in the readout -- the reference arithmetic in f32 (the f32 oracle) is itself
        # 1.3e-5 .. 1.5e-4 away from the exact result per element there -- so: 3x the f32 oracle's own gate value.
        gate, gate32 = gate_err(out, want64), gate_err(want32, want64)
        print(f"{name} {mode}: per-element gate |a-b|/max(|b|,1e-6 max|b|) = {gate:.2e} (f32 oracle {gate32:.2e}), "
              f"scale-relative {rel_err(out, want64):.2e}")
        assert gate <= (TOL if mode == "eval" else max(TOL, 3 * gate32)), (name, mode, gate, gate32)
        if mode == "train":
            bn0 = hip.batch_norms[0].module
            assert rel_err(bn0.running_mean, torch.from_numpy(case["train.bn0_running_mean"])) < TOL
            assert rel_err(bn0.running_var, torch.from_numpy(case["train.bn0_running_var"])) < TOL
            tail = hip.mlp[4 * mlp][5]
            assert rel_err(tail.running_mean, torch.from_numpy(case["train.tail_bn_running_mean"])) < 10 * TOL
            assert rel_err(tail.running_var, torch.from_numpy(case["train.tail_bn_running_var"])) < 10 * TOL
            assert int(bn0.num_batches_tracked) == 4 and int(tail.num_batches_tracked) == 4
        if f"loss_{mode}_f64" in case:
            tgt = torch.from_numpy(case["para"]).to(DEV)
            with torch.no_grad():
                _, loss3 = hip_twin(copy.deepcopy(oracle)).run(data.to(DEV), target=tgt)
            want = float(case[f"loss_{mode}_f64"][0])
            assert abs(float(loss3[0]) - want) <= 10 * bar * abs(want)


def test_workspace_taps_match_oracle_stages():
    """Intermediate tensors inside the workspace against the fixture's per-stage values."""
    import ctypes
    from gnn_epc_saft_amd._native import WorkspaceMap, lib
    case = load_case("mini4_a")
    hidden, depth, pre, post, mlp, num_para, skip, loops = (int(v) for v in case["config"])
    oracle = OraclePNAPCSAFT(hidden, OraclePnaParams(depth, pre, post, torch.from_numpy(case["deg"]),
                                                     skip_connections=bool(skip), self_loops=bool(loops)),
                             OracleMlpParams(mlp, num_para))
    fill_deterministic(oracle, int(case["seed"][0]))
    hip = hip_twin(oracle.eval())
    data = graph_data(case).to(DEV)
    with torch.no_grad():
        hip(data)
    torch.cuda.synchronize()
    n, e, g = data.x.shape[0], data.edge_index.shape[1], 4
    wmap = WorkspaceMap()
    desc = hip._model_desc()
    assert lib.gnnsaft_forward_workspace_map(ctypes.byref(desc), n, e, g, ctypes.byref(wmap)) == 0
    base = (hip._workspace.data_ptr() + 255) // 256 * 256 - hip._workspace.data_ptr()

    def tap(off, shape):
        cnt = int(np.prod(shape))
        return hip._workspace[base + off: base + off + 4 * cnt].view(torch.float32).view(shape).cpu()

    assert rel_err(tap(wmap.pooled, (g, hidden)), torch.from_numpy(case["eval.pooled"])) < TOL
    ptr = torch.tensor([0, 5, 6, 8, 14])
    pooled = tap(wmap.pooled, (g, hidden))
    xf = tap(wmap.x_final, (n, hidden))
    for gi in range(g):
        assert rel_err(xf[ptr[gi]:ptr[gi + 1]].sum(0), pooled[gi]) < 1e-6


ENVELOPE = [
    # hidden, depth, pre, post, mlp, P, skip, loops
    (64, 6, 1, 1, 1, 5, True, True),     # configs/default.py
    (128, 3, 1, 1, 1, 3, True, True),    # BASELINE config 2 model
    (256, 5, 1, 1, 1, 3, True, True),    # BASELINE config 3 model
    (128, 2, 1, 3, 1, 3, True, True),    # compare.ipynb "model6"
    (64, 2, 2, 2, 0, 5, False, False),
    (128, 2, 2, 1, 2, 3, False, True),
    (64, 3, 1, 2, 2, 5, True, False),
]


@pytest.mark.parametrize("cfg", ENVELOPE, ids=[str(c) for c in ENVELOPE])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_shape_envelope_vs_oracle(cfg, mode):
    from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch
    hidden, depth, pre, post, mlp, num_para, skip, loops = cfg
    data = make_synthetic_batch(96, 4321 + hidden + depth, num_para=num_para)
    oracle = oracle_model(hidden, depth, pre, post, mlp, num_para, skip, loops, degree_histogram(data), seed=depth)
    oracle.train(mode == "train")
    hip = hip_twin(copy.deepcopy(oracle))
    o64 = copy.deepcopy(oracle).double()
    stages = {}
    with torch.no_grad():
        out = hip(data.to(DEV)).cpu()
        want32 = copy.deepcopy(oracle)(data)
        want64 = o64(data, stages)
    check_population(out, want32, want64)
    # the literal north_star number where it holds: EVAL mode, every element within 1e-5 of the output scale of the f64
    # oracle (no population statistics); train mode (BatchNorm batch statistics amplify rounding) prints its value
    scale = float(want64.abs().max())
    e_hip = float((out.double() - want64).abs().max()) / scale
    e_f32 = float((want32.double() - want64).abs().max()) / scale
    print(f"envelope {cfg} {mode}: max |err| / max |oracle| = {e_hip:.2e} (f32 oracle {e_f32:.2e})")
    if mode == "eval":
        assert e_hip <= TOL, "eval mode: the literal 1e-5 (of the output scale) against the f64 oracle"
    if mode == "train":  # running statistics and counters of every BatchNorm moved like the oracle's
        sd_h, sd_o = hip.state_dict(), o64.state_dict()
        for k in sd_o:
            if "running_" in k:
                assert rel_err(sd_h[k], sd_o[k]) < 1e-4, k
            if k.endswith("num_batches_tracked"):
                assert int(sd_h[k]) == int(sd_o[k]) == 1


@pytest.mark.parametrize("cfg", [(64, 2, True, 96), (128, 3, True, 1024), (256, 2, False, 300), (64, 6, True, 3000)],
                         ids=["H64", "C2", "H256-noskip", "default-model-3000"])
def test_deferred_batchnorm_equals_the_separate_launches_bit_for_bit(cfg):
    """Train-mode node BatchNorm without an apply pass -- statistics closed by ONE launch (k_bn_stats_close: segment
    folds, then the last-arriving workgroup of a column slab finalises; csrc/bn_fold.hpp), normalisation + ReLU +
    residual applied on load by the next layer's message GEMM (BnResA) resp. by the pooling kernel -- against
    k_bn_combine + k_bn_train_apply: the same sums in the same order, so outputs, loss, running statistics and (taped
    forward) every gradient are EQUAL, whatever workgroup happens to arrive last."""
    from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch
    from gnn_epc_saft_amd.train.models import mape_loss
    hidden, depth, skip, graphs = cfg
    data = make_synthetic_batch(graphs, 11 + graphs, num_para=3)
    oracle = oracle_model(hidden, depth, 1, 1, 1, 3, skip, True, degree_histogram(data), seed=4).train()
    dd, tgt = data.to(DEV), data.para.view(-1, 3).to(DEV)
    res = {}
    for fused in (True, False, "pool"):
        hip = hip_twin(copy.deepcopy(oracle))
        hip.fused_batchnorm = fused
        with torch.no_grad():
            for _ in range(3):                      # repeated launches reuse (and must re-zero) the ticket counters
                pred, loss3 = hip.run(dd, target=tgt)
        out = hip(dd)                               # taped forward (no destination fold: the 4H-wide message GEMM)
        mape_loss(out, tgt).backward()
        torch.cuda.synchronize()
        assert hip.input_error_flags() == 0
        res[fused] = (pred.clone(), loss3.clone(), out.detach().clone(),
                      {k: v.detach().clone() for k, v in hip.state_dict().items()},
                      {k: p.grad.detach().clone() for k, p in hip.named_parameters()})
    for other in (False, "pool"):
        a, b = res[True], res[other]
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
        for k in a[3]:
            assert torch.equal(a[3][k], b[3][k]), (other, k)
        for k in a[4]:
            assert torch.equal(a[4][k], b[4][k]), (other, k)


def test_zero_crossing_case_behind_the_coarse_gate():
    """The one case of the suite that needs check_population's coarse gate, pinned with its numbers: envelope case
    (64, 2, 2, 2, 0, 5, no skip, no loops), train mode.  Its worst per-element gate value is ~0.1 for the HIP path
    against ~0.02 for the f32 oracle -- on an output element whose exact value is ~1e-5 of the output scale, where the
    HIP path's ABSOLUTE error is no larger than the f32 oracle's worst absolute error.  If this test starts failing,
    the excess is no longer a zero crossing and the coarse gate must not absorb it."""
    from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch
    hidden, depth, pre, post, mlp, num_para, skip, loops = 64, 2, 2, 2, 0, 5, False, False
    data = make_synthetic_batch(96, 4321 + hidden + depth, num_para=num_para)
    oracle = oracle_model(hidden, depth, pre, post, mlp, num_para, skip, loops, degree_histogram(data), seed=depth).train()
    hip = hip_twin(copy.deepcopy(oracle))
    with torch.no_grad():
        out = hip(data.to(DEV)).cpu().double()
        want32 = copy.deepcopy(oracle)(data).double()
        want64 = copy.deepcopy(oracle).double()(data)
    scale = float(want64.abs().max())
    gate = (out - want64).abs() / want64.abs().clamp(min=1e-6 * scale)
    worst = int(gate.argmax())
    b = float(want64.view(-1)[worst])
    abs_h = float((out - want64).abs().view(-1)[worst]) / scale
    abs_o_max = float((want32 - want64).abs().max()) / scale
    print(f"worst element: exact value {b / scale:.2e} of the output scale, HIP gate value {float(gate.view(-1)[worst]):.2e}, "
          f"HIP absolute error there {abs_h:.2e} of scale; f32 oracle's largest absolute error {abs_o_max:.2e} of scale")
    # the condition under which check_population takes its coarse gate -- this case must still NEED it: if the excess is
    # gone (e.g. after a kernel change), this test fails on purpose and the gate is to be removed from helpers.py
    worst_h = float(gate_err(out, want64, per_row=True).max())
    worst_o = float(gate_err(want32, want64, per_row=True).max())
    assert worst_h > max(FLIP_BAR, 3 * worst_o), \
        (f"worst per-graph gate value hip {worst_h:.2e} vs f32 oracle {worst_o:.2e}: no case of the suite needs "
         "check_population's coarse gate any more -- delete that branch of tests/helpers.py::check_population")
    assert abs(b) <= 1e-3 * scale          # the excess sits on an output passing through zero ...
    assert abs_h <= 1.5 * abs_o_max        # ... and is an ordinary f32 absolute error


FLIP_BAR = 5e-3


ABLATION = [(128, 3, 1, 1, 1, 3, True, True), (64, 6, 1, 1, 1, 5, True, True), (256, 5, 1, 1, 1, 3, True, True)]


@pytest.mark.parametrize("cfg", ABLATION, ids=[str(c) for c in ABLATION])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_fold_ablation_against_f64_oracle(cfg, mode):
    """Which algebraic fold owns how much of the distance to the exact (f64) result?  The same 96-graph batch through
    (a) the unfolded update (scalers on load, K = 13F: the reference's own formulation of the update), (b) the
    degree-folded update, (c) degree fold + destination-term fold (what a no-grad forward runs), each judged by the
    per-element gate per graph against the f64 oracle, next to the f32 oracle.  The folded weights are accumulated in
    float64 and rounded once (csrc/fold.hpp), so a fold must not cost accuracy: the 50 % / 90 % quantiles of (b) and (c)
    within 1.5x of the larger of (a)'s and the f32 oracle's (floor: the 1e-5 gate itself); the 99 % quantile -- the
    second-worst of 96 graphs, an output passing through zero: a lottery all evaluations draw from -- within 3x."""
    from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch
    hidden, depth, pre, post, mlp, num_para, skip, loops = cfg
    data = make_synthetic_batch(96, 4321 + hidden + depth, num_para=num_para)
    oracle = oracle_model(hidden, depth, pre, post, mlp, num_para, skip, loops, degree_histogram(data), seed=depth)
    oracle.train(mode == "train")
    with torch.no_grad():
        want32 = copy.deepcopy(oracle)(data)
        want64 = copy.deepcopy(oracle).double()(data)
    qs = torch.tensor([0.5, 0.9, 0.99, 1.0], dtype=torch.float64)
    q32 = torch.quantile(gate_err(want32, want64, per_row=True), qs)
    rows = {}
    for name, (fold_deg, fold_dst) in (("unfolded", (False, False)), ("degree fold", (True, False)),
                                       ("degree + destination fold", (True, True))):
        hip = hip_twin(copy.deepcopy(oracle))
        hip.fold_degree_scalers, hip.fold_dst_term = fold_deg, fold_dst
        with torch.no_grad():
            out = hip(data.to(DEV)).cpu()
        assert hip.input_error_flags() == 0
        rows[name] = torch.quantile(gate_err(out, want64, per_row=True), qs)
    fmt = lambda q: "[" + ", ".join("%.1e" % v for v in q.tolist()) + "]"
    print(f"fold ablation H={hidden} L={depth} {mode}: per-graph gate quantiles 50/90/99/100% -- f32 oracle {fmt(q32)}; "
          + "; ".join(f"{k} {fmt(v)}" for k, v in rows.items()))
    bar = torch.maximum(torch.maximum(rows["unfolded"], q32), torch.full_like(q32, TOL))
    for name in ("degree fold", "degree + destination fold"):
        assert bool((rows[name][:2] <= 1.5 * bar[:2]).all()) and float(rows[name][2]) <= 3 * float(bar[2]), \
            (name, fmt(rows[name]), fmt(bar))


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_nine_layers_take_the_chunked_weight_paths(mode):
    """More layers than one fused launch carries (GNNSAFT_MAX_FOLD_LAYERS = 8): the destination-term fold, the
    update-weight fold and the edge-class tables then run as their own chunked launches / GEMMs instead of riding in
    the prologue -- same bar as the rest of the envelope."""
    from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch
    data = make_synthetic_batch(64, 97, num_para=3)
    oracle = oracle_model(64, 9, 1, 1, 1, 3, True, True, degree_histogram(data), seed=9)
    oracle.train(mode == "train")
    hip = hip_twin(copy.deepcopy(oracle))
    with torch.no_grad():
        out = hip(data.to(DEV)).cpu()
        want32 = copy.deepcopy(oracle)(data)
        want64 = copy.deepcopy(oracle).double()(data)
    assert hip.input_error_flags() == 0
    check_population(out, want32, want64)


def test_single_graph_unbatched_and_one_node_graphs():
    from gnn_epc_saft_amd.data.synthetic import GraphData, collate, ethanol_heavy
    d = ethanol_heavy()
    deg = torch.tensor([0, 2, 1])
    oracle = oracle_model(64, 3, 1, 1, 1, 5, True, True, deg).eval()
    hip = hip_twin(copy.deepcopy(oracle))
    with torch.no_grad():
        out = hip(d.to(DEV)).cpu()          # batch is None -> [1, P]
        want = oracle.double()(d)
    assert out.shape == (1, 5) and gate_err(out, want) < TOL
    # batch of one graph == un-batched; one-node / zero-edge graphs are legal, with and without loops
    lone = GraphData(d.x[:1], d.edge_index[:, :0], d.edge_attr[:0])
    both = collate([d, lone, lone])
    for loops in (True, False):
        oracle = oracle_model(64, 2, 1, 1, 0, 3, False, loops, deg).eval()
        hip = hip_twin(copy.deepcopy(oracle))
        with torch.no_grad():
            out = hip(both.to(DEV)).cpu()
            want = oracle.double()(both)
        assert gate_err(out, want) < TOL
        assert torch.equal(out[1], out[2])


def test_train_mode_single_row_raises_like_batchnorm():
    from gnn_epc_saft_amd.data.synthetic import ethanol_heavy
    oracle = oracle_model(64, 1, 1, 1, 0, 3, False, True, torch.tensor([0, 2, 1])).train()
    hip = hip_twin(oracle)
    with pytest.raises(ValueError):
        with torch.no_grad():
            hip(ethanol_heavy().to(DEV))


def test_cpu_tensors_fail_loudly_and_eval_mode_builds_a_graph():
    from gnn_epc_saft_amd.data.synthetic import ethanol_heavy
    oracle = oracle_model(64, 1, 1, 1, 0, 3, False, True, torch.tensor([0, 2, 1])).eval()
    hip = hip_twin(oracle)
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            hip(ethanol_heavy())                 # CPU batch: no fallback
    # grad mode in eval(): a graph over frozen BatchNorm statistics (one un-batched molecule: a single pooled row,
    # which train-mode BatchNorm would refuse)
    pred = hip(ethanol_heavy().to(DEV))
    assert pred.requires_grad and pred.shape == (1, 3)
    pred.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in hip.parameters())


@pytest.mark.parametrize("config_id", [2, 3])
def test_full_size_properties(config_id):
    """BASELINE.json configs 2 and 3: properties that do not need the (slow) oracle at full
    size, then the population bar against the oracle for config 2."""
    from gnn_epc_saft_amd.data.synthetic import GraphData, degree_histogram, make_synthetic_batch
    g, hidden, depth = (1024, 128, 3) if config_id == 2 else (8192, 256, 5)
    data = make_synthetic_batch(g, 1234 + config_id)
    oracle = oracle_model(hidden, depth, 1, 1, 1, 3, True, True, degree_histogram(data), seed=config_id)
    hip = hip_twin(copy.deepcopy(oracle).eval())
    dd = data.to(DEV)
    with torch.no_grad():
        a = hip(dd)
        b = hip(dd)
        assert torch.equal(a, b)                                   # bitwise reproducible
        # edge order is irrelevant up to f32 summation order
        perm = torch.randperm(data.edge_index.shape[1], generator=torch.Generator().manual_seed(1))
        shuffled = GraphData(data.x, data.edge_index[:, perm], data.edge_attr[perm], data.batch, data.ptr, data.para,
                             data.num_graphs).to(DEV)
        c = hip(shuffled)
        assert rel_err(c, a) < TOL
        # eval mode: graphs are independent -> any sub-batch reproduces its rows
        from gnn_epc_saft_amd.data.synthetic import split_graphs
        part = split_graphs(data, 4, 2).to(DEV)
        lo, hi = g // 2, 3 * g // 4
        assert rel_err(hip(part), a[lo:hi]) < TOL
        # train mode: bitwise reproducible too, finite, running stats move
        hip.train()
        t1 = hip(dd)
        assert torch.isfinite(t1).all()
    if config_id == 2:
        o = copy.deepcopy(oracle)
        for mode in (False, True):
            o.train(mode)
            hip = hip_twin(copy.deepcopy(o))
            stages = {}
            with torch.no_grad():
                out = hip(dd).cpu()
                want32 = copy.deepcopy(o)(data)
                want64 = copy.deepcopy(o).double()(data, stages)
            check_population(out, want32, want64)
            tgt = data.para.view(-1, 3)
            with torch.no_grad():
                _, loss3 = hip_twin(copy.deepcopy(o)).run(dd, target=tgt.to(DEV))
            want = float(mape(want64, tgt.double()))
            assert abs(float(loss3[0]) - want) < 1e-4 * want


@pytest.mark.parametrize("config_id", [2, 3])
def test_full_size_against_the_oracle_with_the_literal_1e5_bar(config_id):
    """BASELINE.json configs[1] and configs[2] AT FULL SIZE against the oracle run once per mode (f32 and f64; at C3 --
    8192 graphs, 164 k nodes, H=256, L=5: the size at which k_pna_aggregate<2, true> (streaming stores) and the big GEMM
    tiles are the kernels that run -- about a minute of host time per mode):
      * the literal north_star number where it holds: EVAL mode, every element within 1e-5 of the output scale of the
        f64 oracle (measured ~2e-6) -- an oracle-independent bar, no population statistics; the train-mode value is
        printed beside it and must stay within 3x the f32 oracle's own;
      * the frozen population bar (helpers.check_population) in both modes, and the MAPE loss to 1e-4;
      * train mode, grad enabled: the discrete decisions (std masks, ReLU gates, min / max routing) read off the HIP
        tape against the free f64 oracle's -- printed, and bounded at 2e-5 of all decisions as in the gradient test."""
    from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch
    from helpers import branch_differences, branch_of_oracle, branch_of_tape
    g, hidden, depth = (1024, 128, 3) if config_id == 2 else (8192, 256, 5)
    data = make_synthetic_batch(g, 1234 + config_id)
    oracle = oracle_model(hidden, depth, 1, 1, 1, 3, True, True, degree_histogram(data), seed=config_id)
    dd = data.to(DEV)
    tgt = data.para.view(-1, 3)
    for mode in (False, True):
        o = copy.deepcopy(oracle)
        o.train(mode)
        hip = hip_twin(copy.deepcopy(o))
        with torch.no_grad():
            out, loss3 = hip.run(dd, target=tgt.to(DEV))
            out = out.cpu()
            want32 = copy.deepcopy(o)(data)
            want64 = copy.deepcopy(o).double()(data)
        assert hip.input_error_flags() == 0
        e_hip, e_f32 = rel_err(out, want64), rel_err(want32, want64)
        print(f"C{config_id} full size, {'train' if mode else 'eval'}: max |hip - f64 oracle| / max |f64 oracle| = {e_hip:.2e} "
              f"(f32 oracle {e_f32:.2e})")
        if not mode:
            assert e_hip <= TOL, "eval mode: the literal 1e-5 (of the output scale) against the f64 oracle"
        else:
            assert e_hip <= max(3 * e_f32, TOL)
        check_population(out, want32, want64)
        want = float(mape(want64, tgt.double()))
        assert abs(float(loss3[0]) - want) < 1e-4 * want
        del want32, want64
    # decisions of the taped train-mode forward at this size
    o = copy.deepcopy(oracle).train()
    hip = hip_twin(copy.deepcopy(o))
    pred = hip(dd)
    branch = branch_of_tape(pred, data, True, True)
    free = branch_of_oracle(copy.deepcopy(o).double(), data, True, True)
    diff = branch_differences(branch, free)
    print(f"C{config_id} full size: decisions taken differently from the free f64 oracle: {diff}")
    assert sum(v for k, v in diff.items() if k != "decisions") <= 2e-5 * diff["decisions"], diff


def test_side_stream_and_graph_replay_give_the_same_bits():
    """The optional gnnsaft_aux side stream (structure chain beside the embedding / edge-table chain) and a hipGraph
    replay of the step only change the schedule: outputs and the loss are bit-identical to the eager single-stream
    run, also when the same module runs many steps back to back."""
    from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch
    data = make_synthetic_batch(96, 321)
    oracle = oracle_model(128, 3, 1, 1, 1, 3, True, True, degree_histogram(data), seed=2).train()
    dd = data.to(DEV)
    tgt = dd.para.view(-1, 3)

    def run(side, steps):
        m = hip_twin(copy.deepcopy(oracle))
        m.use_side_stream = side
        outs = []
        with torch.no_grad():
            for _ in range(steps):
                pred, loss3 = m.run(dd, target=tgt)
                outs.append((pred.clone(), loss3.clone()))
        torch.cuda.synchronize()
        return m, outs

    _, base = run(False, 4)
    _, side = run(True, 4)
    for (p0, l0), (p1, l1) in zip(base, side):
        assert torch.equal(p0, p1) and torch.equal(l0, l1)
    # graph capture with the side stream: fork / join become parallel branches of the graph
    m, _ = run(True, 1)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g, stream=s):
        pred, loss3 = m.run(dd, target=tgt)
    for k in range(1, 4):            # the module has seen one step already: replays are steps 2..4
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(pred, base[k][0]) and torch.equal(loss3, base[k][1])


def test_cached_structure_gives_the_same_bits_and_is_validated():
    """gnnsaft_structure_build + gnnsaft_forward(structure=...): the cached CSR / degree tiles replace the K0
    chain by one device copy; outputs, loss and gradients are bit-identical; a blob of another batch is rejected."""
    from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch
    from gnn_epc_saft_amd.train.models import mape_loss
    data = make_synthetic_batch(80, 99)
    other = make_synthetic_batch(81, 98)
    oracle = oracle_model(64, 2, 1, 1, 1, 3, True, True, degree_histogram(data), seed=4).train()
    m = hip_twin(copy.deepcopy(oracle))
    dd, tgt = data.to(DEV), data.para.view(-1, 3).to(DEV)
    with torch.no_grad():
        p0, l0 = m.run(dd, target=tgt)
        p0, l0 = p0.clone(), l0.clone()
        dd.gnnsaft_structure = m.build_structure(dd)
        p1, l1 = m.run(dd, target=tgt)
        assert torch.equal(p0, p1) and torch.equal(l0, l1)
        m.eval()
        e1 = m(dd).clone()
        dd.gnnsaft_structure = None
        assert torch.equal(m(dd), e1)
        m.train()
    # training step through the tape with the cached structure
    def grads(with_structure):
        mm = hip_twin(copy.deepcopy(oracle))
        d2 = data.to(DEV)
        if with_structure:
            d2.gnnsaft_structure = mm.build_structure(d2)
        mape_loss(mm(d2), tgt).backward()
        return [p.grad.clone() for p in mm.parameters()]
    for a, b in zip(grads(False), grads(True)):
        assert torch.equal(a, b)
    od = other.to(DEV)
    od.gnnsaft_structure = dd.gnnsaft_structure if dd.gnnsaft_structure is not None else m.build_structure(dd)
    with pytest.raises(ValueError), torch.no_grad():
        m(od)
    assert m.input_error_flags() == 0


def test_hip_path_properties_node_relabelling_explicit_loops_isolated_nodes():
    """The invariances the oracle is held to (tests/test_oracle_cpu.py), on the HIP path itself: relabelling the
    nodes of a graph, appending the self-loops explicitly instead of setting the flag, and nodes without in-edges."""
    from gnn_epc_saft_amd.data.synthetic import GraphData, degree_histogram, make_synthetic_batch
    data = make_synthetic_batch(40, 77)
    n = data.x.shape[0]
    oracle = oracle_model(64, 3, 1, 1, 1, 3, True, True, degree_histogram(data), seed=6).eval()
    hip = hip_twin(copy.deepcopy(oracle))
    with torch.no_grad():
        ref = hip(data.to(DEV))
        # relabel the nodes inside every graph (reverse their order): rows of x / batch permuted, endpoints mapped
        p = torch.cat([torch.arange(int(data.ptr[g + 1]) - 1, int(data.ptr[g]) - 1, -1) for g in range(data.num_graphs)])
        inv = torch.empty_like(p)
        inv[p] = torch.arange(n)
        relabelled = GraphData(data.x[p], inv[data.edge_index], data.edge_attr, data.batch[p], data.ptr, data.para,
                               data.num_graphs)
        assert rel_err(hip(relabelled.to(DEV)), ref) < TOL
        # self_loops=True  ==  explicit (i, i) edges with attribute [0, 0, 0] appended last, flag off
        plain = hip_twin(copy.deepcopy(oracle))
        plain.pna_params = copy.copy(plain.pna_params)
        plain.pna_params.self_loops = False
        loops = torch.arange(n).repeat(2, 1)
        explicit = GraphData(data.x, torch.cat([data.edge_index, loops], dim=1),
                             torch.cat([data.edge_attr, torch.zeros((n, 3), dtype=torch.int64)]), data.batch,
                             data.ptr, data.para, data.num_graphs)
        assert torch.equal(plain(explicit.to(DEV)), ref)          # same rows in the same order: same bits
        # nodes without in-edges (no self-loops): aggregates are exactly zero, degree 0 is a legal bucket
        keep = data.edge_index[1] % 3 != 0                          # drop every edge into nodes 0, 3, 6, ...
        sparse = GraphData(data.x, data.edge_index[:, keep], data.edge_attr[keep], data.batch, data.ptr, data.para,
                           data.num_graphs)
        o2 = copy.deepcopy(oracle)
        o2.pna_params = copy.copy(o2.pna_params)
        o2.pna_params.self_loops = False
        h2 = hip_twin(copy.deepcopy(o2))
        h2.pna_params = copy.copy(h2.pna_params)
        h2.pna_params.self_loops = False
        want = copy.deepcopy(o2).double()(sparse)
        got = h2(sparse.to(DEV)).cpu()
        assert h2.input_error_flags() == 0
        assert rel_err(got, want) < max(TOL, 3 * rel_err(copy.deepcopy(o2)(sparse), want))


@pytest.mark.parametrize("graphs", [150, 1800])   # 1800: 36 k nodes = 141 groups of 256 on 128 workgroups (two rounds
@pytest.mark.parametrize("chain", ["prologue workgroups", "launches"])   # of the look-back scan)
@pytest.mark.parametrize("loops", [True, False])
def test_the_forward_builds_the_csr_of_the_general_chain(loops, chain, graphs):
    """gnnsaft_forward builds its CSR with the slotted chain (csr.hip: in-degree < 32 promised by the folded update,
    one pass over the edge list, per-node sort in registers) -- by cooperating workgroups of its first launch with
    grid barriers among them (elementwise.hip: k0_chain_body; needs the module's persistent barrier words), or as
    launches; gnnsaft_csr_build keeps the general histogram / scan / fill chain.  Same rows bit for bit, on an edge list in RANDOM order (the per-node sort has
    work to do) with a hub of 13 in-edges (more than the 8 the register sort takes: the in-memory path) and the edge
    list's order kept inside every row (stable: the float sums downstream depend on it)."""
    import ctypes

    import gnn_epc_saft_amd.kernels as K
    from gnn_epc_saft_amd._native import WorkspaceMap, lib
    from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch
    data = make_synthetic_batch(graphs, 31)
    n = data.x.shape[0]
    g = torch.Generator().manual_seed(5)
    hub = torch.stack([torch.arange(1, 14), torch.zeros(13, dtype=torch.int64)])
    ei = torch.cat([data.edge_index, hub], dim=1)
    ea = torch.cat([data.edge_attr, torch.randint(0, 2, (13, 3), generator=g)])
    order = torch.randperm(ei.shape[1], generator=g)
    data.edge_index, data.edge_attr = ei[:, order].contiguous(), ea[order].contiguous()
    oracle = oracle_model(64, 2, 1, 1, 1, 3, True, loops, degree_histogram(data), seed=2).train()
    m = hip_twin(copy.deepcopy(oracle))
    assert m.fold_degree_scalers
    m.fused_structure_chain = chain == "prologue workgroups"
    pred = m(data.to(DEV))
    again = m(data.to(DEV))     # the barrier words are left zero: the next call meets the same barriers
    assert torch.equal(pred, again)
    assert m.input_error_flags() == 0 and int(m._err_flag.abs().sum()) == 0
    tape = pred.grad_fn.tape
    desc, e, gg = tape["desc"], tape["e"], tape["g"]
    wmap = WorkspaceMap()
    assert lib.gnnsaft_forward_workspace_map(ctypes.byref(desc), n, e, gg, ctypes.byref(wmap)) == 0
    base = tape["ws_ptr"] - tape["ws"].data_ptr()
    ep = e + (n if loops else 0)
    view = lambda off, count: tape["ws"][base + off: base + off + 4 * count].view(torch.int32)
    want = K.csr_build(data.edge_index.to(DEV), data.edge_attr.to(DEV), n, (5, 6, 2), loops)
    assert int(want[-1].item()) == 0
    assert torch.equal(view(wmap.rowptr, n + 1), want[0])
    assert int((want[0][1:] - want[0][:-1]).max()) >= 13
    for name, got, ref in (("src", view(wmap.src, ep), want[1]), ("dst", view(wmap.dst, ep), want[2]),
                           ("combo", view(wmap.combo, ep), want[3])):
        assert torch.equal(got, ref), name
    la = tape["ws"][base + wmap.log_amp: base + wmap.log_amp + 4 * n].view(torch.float32)
    assert torch.equal(la, want[4])


def test_an_in_degree_beyond_the_buckets_is_flagged_by_the_forward():
    """A hub with 40 in-edges: the slotted chain keeps 32 of them, drops the rest and raises
    GNNSAFT_FLAG_BAD_DEGREE (8) -- no fault, no silent wrong answer."""
    from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch
    data = make_synthetic_batch(20, 3)
    deg = degree_histogram(data)
    hub = torch.stack([torch.arange(1, 41), torch.zeros(40, dtype=torch.int64)])
    data.edge_index = torch.cat([data.edge_index, hub], dim=1).contiguous()
    data.edge_attr = torch.cat([data.edge_attr, torch.zeros((40, 3), dtype=torch.int64)]).contiguous()
    m = hip_twin(oracle_model(64, 1, 1, 1, 1, 3, True, True, deg, seed=2).eval())
    with torch.no_grad():
        out = m(data.to(DEV))
    torch.cuda.synchronize()
    assert m.input_error_flags() & 8 and out.shape[0] == 20


@pytest.mark.parametrize("graphs,calls", [(260, 300), (1800, 60)])
def test_the_cooperative_structure_chain_is_stable_over_many_calls(graphs, calls):
    """The K0 chain inside the prologue launch synchronises through one grid barrier, relaxed look-back words and a
    ticket (elementwise.hip: k0_chain_body); a race there would show as an occasional different row order or offset.
    Back-to-back calls on one stream (the persistent words of call k are the entry state of call k + 1): every call
    must return the bits of the first one, raise no flag and leave the persistent words zero."""
    from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch
    data = make_synthetic_batch(graphs, 77)
    oracle = oracle_model(64, 2, 1, 1, 1, 3, True, True, degree_histogram(data), seed=5).train()
    m = hip_twin(copy.deepcopy(oracle))
    assert m.fused_structure_chain
    dd, tgt = data.to(DEV), data.para.view(-1, 3).to(DEV)
    with torch.no_grad():
        first, loss0 = m.run(dd, target=tgt)
        first, loss0 = first.clone(), loss0.clone()
        outs = [m.run(dd, target=tgt) for _ in range(calls)]   # enqueued back to back, checked afterwards
    torch.cuda.synchronize()
    for k, (out, loss) in enumerate(outs):
        assert torch.equal(out, first) and torch.equal(loss, loss0), f"call {k} differs"
    assert m.input_error_flags() == 0 and int(m._err_flag.abs().sum()) == 0
    m.fused_structure_chain = False
    with torch.no_grad():
        ref, _ = m.run(dd, target=tgt)
    assert torch.equal(ref, first)     # and the launches build the same structure
