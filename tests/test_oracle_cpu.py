"""CPU tests of the oracle itself: golden vectors, the independent loop restatement, and the
properties SURVEY.md section 8(c) lists to keep an unpinned oracle honest."""

import copy

import numpy as np
import pytest
import torch

from golden_util import fill_deterministic, list_cases, load_case
from helpers import mini4, oracle_model, rel_err, to_numpy_state
from oracle import pna_loops
from oracle.pna_torch import (OracleMlpParams, OraclePNAPCSAFT, OraclePnaParams, add_self_loops, global_add_pool, mape,
                              pna_aggregate)


class Bag:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def case_model_and_data(case):
    hidden, depth, pre, post, mlp, num_para, skip, loops = (int(v) for v in case["config"])
    model = OraclePNAPCSAFT(hidden, OraclePnaParams(depth, pre, post, torch.from_numpy(case["deg"]),
                                                    skip_connections=bool(skip), self_loops=bool(loops)),
                            OracleMlpParams(mlp, num_para))
    checksum = fill_deterministic(model, int(case["seed"][0]))
    assert checksum == float(case["weights_checksum"][0]), "deterministic weights do not reproduce"
    t = lambda k: torch.from_numpy(case[k]) if k in case else None
    data = Bag(x=t("x"), edge_index=t("edge_index"), edge_attr=t("edge_attr"), batch=t("batch"))
    return model, data


@pytest.mark.parametrize("name", list_cases())
def test_oracle_reproduces_golden_vectors(name):
    case = load_case(name)
    model, data = case_model_and_data(case)
    for mode in ("eval", "train"):
        if f"out_{mode}_f64" not in case:
            continue
        m64 = copy.deepcopy(model).double().train(mode == "train")
        stages = {}
        with torch.no_grad():
            out = m64(data, stages)
        assert rel_err(out, torch.from_numpy(case[f"out_{mode}_f64"])) < 1e-12
        for key in ("embed", "l0.agg", "l0.post", "l0.conv", "l0.out", "pooled"):
            if f"{mode}.{key}" in case:
                assert rel_err(stages[key], torch.from_numpy(case[f"{mode}.{key}"])) < 1e-12, key
        m32 = copy.deepcopy(model).train(mode == "train")
        with torch.no_grad():
            out32 = m32(data)
        # the f32 evaluation is threshold-free on these fixtures by construction (make_golden.py)
        assert rel_err(out32, torch.from_numpy(case[f"out_{mode}_f64"])) < 5e-5


@pytest.mark.parametrize("name", [n for n in list_cases() if not n.startswith("synth")])
def test_loop_restatement_agrees_with_golden(name):
    case = load_case(name)
    model, data = case_model_and_data(case)
    hidden, depth, pre, post, mlp, num_para, skip, loops = (int(v) for v in case["config"])
    sd = to_numpy_state(copy.deepcopy(model).double())
    for mode in ("eval", "train"):
        if f"out_{mode}_f64" not in case:
            continue
        out = pna_loops.forward_loops(sd, case["x"], case["edge_index"], case["edge_attr"], case.get("batch"),
                                      hidden=hidden, depth=depth, pre_layers=pre, post_layers=post,
                                      num_mlp_layers=mlp, skip=bool(skip), self_loops=bool(loops),
                                      training=(mode == "train"))
        assert rel_err(torch.from_numpy(out), torch.from_numpy(case[f"out_{mode}_f64"])) < 1e-12
        if f"loss_{mode}_f64" in case:
            assert abs(pna_loops.mape_loops(out, case["para"]) - float(case[f"loss_{mode}_f64"][0])) < 1e-12


def small_model(loops=True, skip=True, seed=0):
    data = mini4()
    from gnn_epc_saft_amd.data.synthetic import degree_histogram
    return oracle_model(32, 2, 1, 2, 1, 3, skip, loops, degree_histogram(data), seed=seed,
                        dtype=torch.float64).eval(), data


def test_edge_order_and_node_relabelling_invariance():
    model, d = small_model()
    with torch.no_grad():
        ref = model(d)
        perm = torch.randperm(d.edge_index.shape[1], generator=torch.Generator().manual_seed(3))
        shuffled = Bag(x=d.x, edge_index=d.edge_index[:, perm], edge_attr=d.edge_attr[perm], batch=d.batch)
        assert rel_err(model(shuffled), ref) < 1e-12
        # relabel nodes inside graph 0 (nodes 0..4): a permutation of rows + edge endpoints
        p = torch.arange(d.x.shape[0])
        p[:5] = torch.tensor([3, 0, 4, 1, 2])
        inv = torch.empty_like(p)
        inv[p] = torch.arange(p.numel())
        relabelled = Bag(x=d.x[p], edge_index=inv[d.edge_index], edge_attr=d.edge_attr, batch=d.batch[p])
        assert rel_err(model(relabelled), ref) < 1e-12


def test_batch_of_one_equals_unbatched_and_batching_invariance_in_eval():
    from gnn_epc_saft_amd.data.synthetic import collate, ethanol_all_atom, ethanol_heavy
    model, d = small_model()
    a, b = ethanol_heavy(), ethanol_all_atom()
    with torch.no_grad():
        ua, ub = model(a), model(b)
        assert ua.shape == (1, 3)
        both = model(collate([a, b]))
        one = model(collate([a]))
    assert rel_err(one, ua) < 1e-12
    assert rel_err(both, torch.cat([ua, ub])) < 1e-12


def test_isolated_node_aggregates_to_zero_and_loops_equal_explicit_loops():
    torch.manual_seed(0)
    msgs = torch.randn(5, 2, 8, dtype=torch.float64)
    dst = torch.tensor([0, 0, 2, 2, 2])
    agg = pna_aggregate(msgs, dst, 4)
    assert torch.count_nonzero(agg[1]) == 0 and torch.count_nonzero(agg[3]) == 0   # nodes 1 and 3: no in-edges
    # single-edge segments have var = 0 -> std masked to exactly 0
    one = pna_aggregate(msgs[:1], dst[:1], 1)
    assert torch.count_nonzero(one[..., 24:]) == 0
    # self_loops=True == the same model without the flag on a graph with explicit loop edges appended last
    m_loops, d = small_model(loops=True)
    m_plain = copy.deepcopy(m_loops)
    m_plain.pna_params = copy.copy(m_loops.pna_params)
    m_plain.pna_params.self_loops = False
    ei, ea = add_self_loops(d.edge_index, d.edge_attr, d.x.shape[0])
    with torch.no_grad():
        assert rel_err(m_plain(Bag(x=d.x, edge_index=ei, edge_attr=ea, batch=d.batch)), m_loops(d)) < 1e-12


def test_pool_and_mape_small_cases():
    x = torch.arange(12, dtype=torch.float64).view(6, 2)
    assert torch.equal(global_add_pool(x, None), x.sum(0, keepdim=True))
    pooled = global_add_pool(x, torch.tensor([0, 0, 1, 3, 3, 3]))
    assert pooled.shape == (4, 2) and torch.equal(pooled[2], torch.zeros(2, dtype=torch.float64))
    pred, tgt = torch.tensor([[1.0, 2.0]], dtype=torch.float64), torch.tensor([[2.0, 0.0]], dtype=torch.float64)
    assert abs(float(mape(pred, tgt)) - (0.5 + 2.0 / 1.17e-06) / 2) < 1e-6  # |target| clamped at 1.17e-6


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_branch_forced_oracle_equals_the_free_oracle_on_its_own_branch(mode):
    """helpers.forced_forward (the oracle's modules with the discrete decisions as inputs: what the full-size gradient
    test compares the HIP backward with) on the branch the free oracle itself takes: same output and same parameter
    gradients to float64 rounding -- i.e. forcing a branch changes nothing but who takes the decisions.  Forcing ONE
    other std decision must change the gradient (the test is not vacuous)."""
    import copy
    from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch
    from helpers import branch_differences, branch_of_oracle, forced_forward, oracle_model
    data = make_synthetic_batch(20, 5, num_para=3)
    model = oracle_model(64, 2, 1, 1, 1, 3, True, True, degree_histogram(data), seed=2, dtype=torch.float64)
    model.train(mode == "train")
    branch = branch_of_oracle(copy.deepcopy(model), data, True, True)
    assert branch_differences(branch, branch)["std"] == 0
    target = data.para.view(-1, 3).double()

    def grads(fn):
        m = copy.deepcopy(model)
        out = fn(m)
        mape(out, target).backward()
        return out.detach(), {k: p.grad.clone() for k, p in m.named_parameters()}

    out_free, g_free = grads(lambda m: m(data))
    out_forced, g_forced = grads(lambda m: forced_forward(m, data, branch))
    assert float((out_free - out_forced).abs().max()) <= 1e-12 * float(out_free.abs().max())
    scale = max(float(g.abs().max()) for g in g_free.values())
    for k in g_free:
        assert float((g_free[k] - g_forced[k]).abs().max()) <= 1e-11 * scale, k
    other = {k: [t.clone() for t in v] for k, v in branch.items()}
    idx = int(torch.nonzero(other["std"][0].reshape(-1))[0])
    other["std"][0].view(-1)[idx] = False                       # mask one std entry the oracle keeps
    _, g_other = grads(lambda m: forced_forward(m, data, other))
    assert max(float((g_other[k] - g_free[k]).abs().max()) for k in g_free) > 1e-7 * scale
