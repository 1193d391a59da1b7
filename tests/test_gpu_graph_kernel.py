"""The per-graph fused eval-mode kernel (csrc/graph_eval.hip, gnnsaft_graph_forward): float64 modules
-- ``create_model(config, deg).to(device, torch.float64).eval()`` as the reference's inference callers build them
(/root/reference/gnnepcsaft/evaluations/evaluate_ensemble.py:67-77, demo/utils.py:23-27,141-152) -- and the float32
single-molecule / small-batch path.  Bars: SURVEY 8(d): float64 <= 1e-12 against the f64 oracle (asserted per
element: |a-b| / max(|b|, 1e-6 max|b|)), float32 as in tests/test_gpu_forward.py."""

import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import fill_deterministic, list_cases, load_case  # noqa: E402
from helpers import check_population, gate_err, oracle_model, rel_err  # noqa: E402
from oracle.pna_torch import OracleMlpParams, OraclePNAPCSAFT, OraclePnaParams  # noqa: E402
from test_gpu_forward import graph_data, hip_twin  # noqa: E402

DEV = "cuda:0"
F64_TOL = 1e-12


def assert_f64(got, want):
    """float64 bar on random batches: 1e-13 of the output scale per element (measured: <= 1e-15).  The per-element
    relative gate is printed; it is <= 1e-12 wherever no output lies within 1e-3 of zero (all fixtures: <= 1.1e-14),
    and an output of 1e-4 of the scale necessarily shows rounding of 1e-15 of the scale as 1e-11 of itself."""
    rel, gate = rel_err(got, want), gate_err(got, want)
    print(f"float64: max|a-b| / max|b| = {rel:.1e}, per-element gate {gate:.1e}")
    assert rel <= 1e-13 and gate <= 1e-9, (rel, gate)


def force_graph_kernel(m):
    """float32 modules take the per-graph kernel for single small molecules only (measured crossover); the tests
    push every input through it."""
    m.graph_kernel_max_graphs, m.graph_kernel_max_nodes = 1 << 30, 1 << 30
    return m


def twin64(oracle):
    """HIP module in float64 with the oracle's weights (what .to(device, torch.float64) gives the reference)."""
    m = hip_twin(copy.deepcopy(oracle)).to(DEV, torch.float64)
    m.load_state_dict(copy.deepcopy(oracle).double().state_dict())
    return m.eval()


@pytest.mark.parametrize("name", list_cases())
def test_float64_module_on_golden_vectors(name):
    case = load_case(name)
    hidden, depth, pre, post, mlp, num_para, skip, loops = (int(v) for v in case["config"])
    oracle = OraclePNAPCSAFT(hidden, OraclePnaParams(depth, pre, post, torch.from_numpy(case["deg"]),
                                                     skip_connections=bool(skip), self_loops=bool(loops)),
                             OracleMlpParams(mlp, num_para))
    fill_deterministic(oracle, int(case["seed"][0]))     # float32 weights, as the fixtures were generated
    oracle = oracle.double()
    hip = twin64(oracle)
    assert all(p.dtype == torch.float64 for p in hip.parameters())
    data = graph_data(case)
    with torch.no_grad():
        out = hip(data.to(DEV))
    assert out.dtype == torch.float64 and out.is_cuda and hip.input_error_flags() == 0
    gate = gate_err(out, torch.from_numpy(case["out_eval_f64"]))
    print(f"{name}: float64 module, per-element gate vs the stored f64 oracle output {gate:.1e}, scale-relative "
          f"{rel_err(out, torch.from_numpy(case['out_eval_f64'])):.1e}")
    assert gate <= F64_TOL
    # float32 module through the same kernel (small batch, eval): the plain 1e-5 bar per element
    hip32 = force_graph_kernel(hip_twin(copy.deepcopy(oracle).float()).eval())
    with torch.no_grad():
        out32 = hip32(data.to(DEV))
        hip32.graph_kernel_max_graphs = 0          # the batched pipeline (gnnsaft_forward) on the same input
        ref32 = hip32(data.to(DEV))
    g32 = gate_err(out32, torch.from_numpy(case["out_eval_f64"]))
    print(f"{name}: float32 per-graph kernel gate {g32:.1e}; vs batched pipeline {rel_err(out32, ref32):.1e}")
    assert out32.dtype == torch.float32 and g32 <= 1e-5 and rel_err(out32, ref32) < 2e-6


ENVELOPE = [
    # hidden, depth, pre, post, mlp, P, skip, loops
    (64, 6, 1, 1, 1, 5, True, True),     # configs/default.py
    (128, 3, 1, 1, 1, 3, True, True),    # BASELINE config 2 model
    (256, 5, 1, 1, 1, 3, True, True),    # BASELINE config 3 model
    (128, 2, 1, 3, 1, 3, True, True),    # compare.ipynb "model6"
    (64, 2, 2, 2, 0, 5, False, False),
    (128, 2, 2, 1, 2, 3, False, True),
    (64, 3, 3, 2, 2, 5, True, False),
]


@pytest.mark.parametrize("cfg", ENVELOPE, ids=[str(c) for c in ENVELOPE])
def test_float64_shape_envelope(cfg):
    from gnn_epc_saft_amd.data.synthetic import GraphData, collate, degree_histogram, synthetic_dataset
    hidden, depth, pre, post, mlp, num_para, skip, loops = cfg
    graphs = synthetic_dataset(23, 77 + hidden + depth, num_para=num_para)
    graphs.insert(5, GraphData(graphs[0].x[:1], torch.zeros((2, 0), dtype=torch.int64),
                               torch.zeros((0, 3), dtype=torch.int64), para=graphs[0].para))    # 1 node, 0 edges
    data = collate(graphs)
    oracle = oracle_model(hidden, depth, pre, post, mlp, num_para, skip, loops, degree_histogram(graphs), seed=depth,
                          dtype=torch.float64).eval()
    hip = twin64(oracle)
    with torch.no_grad():
        want = oracle(data)
        got = hip(data.to(DEV))
        assert_f64(got, want)
        # un-batched Data (batch=None -> [1, P]), as validation_step / predparams call the model
        for gi in (0, 5, 11):
            one = graphs[gi]
            assert_f64(hip(one.to(DEV)), oracle(one))
        # float32 twin of the same weights through the per-graph kernel: population bar against the f32 oracle
        hip32 = force_graph_kernel(hip_twin(copy.deepcopy(oracle).float()).eval())
        out32 = hip32(data.to(DEV)).cpu()
        check_population(out32, copy.deepcopy(oracle).float()(data), want)
    assert hip.input_error_flags() == 0 and hip32.input_error_flags() == 0


def test_large_single_graph_and_many_graphs_take_the_global_structure_path():
    """One graph beyond the in-kernel CSR limits (64 nodes / 256 edges) and beyond the LDS node-state budget; and a
    batch with more graphs than CUs."""
    from gnn_epc_saft_amd.data.synthetic import GraphData, degree_histogram, make_synthetic_batch
    n = 300
    g = torch.Generator().manual_seed(5)
    x = torch.stack([torch.randint(0, d, (n,), generator=g) for d in (119, 5, 12, 12, 10, 6, 6, 2, 2)], 1)
    a = torch.arange(n - 1)
    extra = torch.randint(0, n, (2, 60), generator=g)
    src = torch.cat([a, a + 1, extra[0], extra[1]])
    dst = torch.cat([a + 1, a, extra[1], extra[0]])
    attr = torch.stack([torch.randint(0, d, (src.numel(),), generator=g) for d in (5, 6, 2)], 1)
    big = GraphData(x, torch.stack([src, dst]), attr)
    oracle = oracle_model(128, 3, 1, 2, 1, 3, True, True, degree_histogram(big), seed=2, dtype=torch.float64).eval()
    hip = twin64(oracle)
    with torch.no_grad():
        assert_f64(hip(big.to(DEV)), oracle(big))
        many = make_synthetic_batch(700, 4)
        o2 = oracle_model(64, 2, 1, 1, 1, 3, True, True, degree_histogram(many), seed=3, dtype=torch.float64).eval()
        assert_f64(twin64(o2)(many.to(DEV)), o2(many))


def test_float64_is_eval_only_and_float32_weights_are_tracked():
    from gnn_epc_saft_amd.data.synthetic import degree_histogram, ethanol_heavy, make_synthetic_batch
    from gnn_epc_saft_amd.train.models import mape_loss
    from gnn_epc_saft_amd.train.optim import FusedAdamW
    data = make_synthetic_batch(16, 9)
    oracle = oracle_model(64, 2, 1, 1, 1, 3, True, True, degree_histogram(data), seed=1, dtype=torch.float64)
    hip = twin64(oracle)
    dd = data.to(DEV)
    with pytest.raises(NotImplementedError):
        hip(dd)                                   # grad mode
    hip.train()
    with pytest.raises(NotImplementedError), torch.no_grad():
        hip(dd)                                   # train-mode BatchNorm in float64
    hip.eval()
    with pytest.raises(NotImplementedError), torch.no_grad():
        hip.run(dd, target=dd.para.view(-1, 3))   # no float64 loss kernel
    # weights changed through torch (load_state_dict) and through the fused optimizer: the packed copy follows
    o32 = copy.deepcopy(oracle).float()
    m = force_graph_kernel(hip_twin(copy.deepcopy(o32)).eval())
    eth = ethanol_heavy().to(DEV)
    with torch.no_grad():
        before = m(eth).clone()
        sd = {k: (v * 1.01 if v.is_floating_point() and v.dim() == 2 else v) for k, v in m.state_dict().items()}
        m.load_state_dict(sd)
        o32.load_state_dict({k: v.cpu() for k, v in sd.items()})
        after = m(eth)
        assert not torch.equal(before, after)
        assert gate_err(after, copy.deepcopy(o32).double().eval()(ethanol_heavy())) <= 1e-5
    m.train()
    params, offsets, total = m.flat_layout()
    opt = FusedAdamW(params, lr=1e-2, amsgrad=True, eps=1e-5, layout=(offsets, total))
    opt.on_parameters_rewritten = m.invalidate_eval_pack
    mape_loss(m(dd), dd.para.view(-1, 3)).backward()
    m.eval()
    with torch.no_grad():
        pre_step = m(eth).clone()
        opt.step()                                # raw kernel: no torch version counter moves
        post_step = m(eth)
    assert not torch.equal(pre_step, post_step)
    o32.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    with torch.no_grad():
        assert gate_err(post_step, o32.double().eval()(ethanol_heavy())) <= 1e-5


def test_single_molecule_call_replays_from_a_hipgraph():
    from gnn_epc_saft_amd.data.synthetic import ethanol_all_atom
    oracle = oracle_model(64, 6, 1, 1, 1, 5, True, True, torch.tensor([0, 6, 2, 0, 1]), seed=4).eval()
    hip = force_graph_kernel(hip_twin(copy.deepcopy(oracle)))
    d = ethanol_all_atom().to(DEV)
    with torch.no_grad():
        eager = hip(d).clone()                    # builds the pack outside the capture
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            out = hip(d)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager)
        assert gate_err(eager, copy.deepcopy(oracle).double()(ethanol_all_atom())) <= 1e-5
