"""Fused optimizers (gnnsaft_adamw_step / gnnsaft_sgd_step) against torch.optim on the CPU in float64 and float32,
on the same parameters and gradient sequences (reference: torch.optim.AdamW(amsgrad=True, eps=1e-5) /
torch.optim.SGD(nesterov=True), /root/reference/gnnepcsaft/train/models.py:162-178).

Tolerance: after 25 steps every parameter must agree with the float64 torch trajectory to 2e-6 of the tensor's
scale, and be no further from it than 4x the float32 torch trajectory is (floor 2e-7)."""

import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import oracle_model, rel_err  # noqa: E402
from test_gpu_forward import hip_twin  # noqa: E402
from test_gpu_stages import DEV  # noqa: E402

SHAPES = [(37, 5), (64,), (128, 64), (3,), (1,), (9, 7, 3)]


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(s, generator=g) for s in SHAPES]


def _grads(step, seed):
    g = torch.Generator().manual_seed(1000 * seed + step)
    return [torch.randn(s, generator=g) * (0.1 + 0.05 * step) for s in SHAPES]


def _run_torch(make, dtype, steps, sched=False):
    ps = [torch.nn.Parameter(p.to(dtype)) for p in _params(0)]
    opt = make(ps)
    sch = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, 7) if sched else None
    for t in range(steps):
        for p, g in zip(ps, _grads(t, 1)):
            p.grad = g.to(dtype)
        opt.step()
        if sch:
            sch.step()
    return [p.detach() for p in ps], opt


def _run_fused(make, steps, sched=False, flat_grads=False):
    ps = [torch.nn.Parameter(p.to(DEV)) for p in _params(0)]
    opt = make(ps)
    sch = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, 7) if sched else None
    for t in range(steps):
        gs = _grads(t, 1)
        if flat_grads:      # gradients delivered as views of one flat buffer in the optimizer's own layout
            flat = torch.zeros(opt._total, device=DEV)
            for p, g, off in zip(ps, gs, opt._offsets):
                v = flat[off:off + p.numel()].view(p.shape)
                v.copy_(g)
                p.grad = v
        else:
            for p, g in zip(ps, gs):
                p.grad = g.to(DEV)
        opt.step()
        if sch:
            sch.step()
    return [p.detach().cpu() for p in ps], opt


@pytest.mark.parametrize("flat", [False, True])
@pytest.mark.parametrize("kind", ["adamw_amsgrad", "adamw", "sgd"])
def test_fused_step_matches_torch_optim(kind, flat):
    from gnn_epc_saft_amd.train.optim import FusedAdamW, FusedSGD
    if kind == "sgd":
        kw = dict(lr=3e-2, momentum=0.9, weight_decay=1e-2, nesterov=True)
        t_make = lambda ps: torch.optim.SGD(ps, **kw)
        f_make = lambda ps: FusedSGD(ps, **kw)
    else:
        kw = dict(lr=2e-3, weight_decay=1e-2, amsgrad=kind == "adamw_amsgrad", eps=1e-5)
        t_make = lambda ps: torch.optim.AdamW(ps, **kw)
        f_make = lambda ps: FusedAdamW(ps, **kw)
    steps = 25
    want64, _ = _run_torch(t_make, torch.float64, steps, sched=True)
    want32, _ = _run_torch(t_make, torch.float32, steps, sched=True)
    got, opt = _run_fused(f_make, steps, sched=True, flat_grads=flat)
    for w64, w32, g in zip(want64, want32, got):
        e_hip, e_f32 = rel_err(g, w64), rel_err(w32, w64)
        assert e_hip < 2e-6 and e_hip <= max(4 * e_f32, 2e-7), (kind, tuple(g.shape), e_hip, e_f32)
    # state_dict in torch's per-parameter format, and a round trip through load_state_dict
    sd = copy.deepcopy(opt.state_dict())
    key = "momentum_buffer" if kind == "sgd" else "exp_avg"
    assert sd["state"][0][key].shape == torch.Size(SHAPES[0]) and float(sd["state"][0]["step"]) == steps
    ps2 = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in got]
    opt2 = f_make(ps2)
    opt2.load_state_dict(sd)
    opt2.param_groups[0]["lr"] = opt.param_groups[0]["lr"]
    ps1 = opt._params
    for t in range(steps, steps + 3):
        for p1, p2, g in zip(ps1, ps2, _grads(t, 1)):
            p1.grad = g.to(DEV)
            p2.grad = g.to(DEV)
        opt.step()
        opt2.step()
    for p1, p2 in zip(ps1, ps2):
        assert torch.equal(p1.detach(), p2.detach())


def test_grad_scale_and_zero_copy_path_with_the_model():
    """configure_optimizers() adopts the model's flat layout: backward leaves .grad as views of one buffer and the
    step consumes it as it is; a training run with the fused AdamW follows torch.optim.AdamW on the f64 oracle."""
    import gnn_epc_saft_amd as G
    from gnn_epc_saft_amd.data.synthetic import degree_histogram, make_synthetic_batch
    from gnn_epc_saft_amd.train.optim import FusedAdamW
    data = make_synthetic_batch(48, 77)
    oracle = oracle_model(64, 2, 1, 1, 1, 3, True, True, degree_histogram(data), seed=5).train()
    hip = hip_twin(copy.deepcopy(oracle))
    cfg = dict(hidden_dim=64, num_para=3, optimizer="adam", learning_rate=1e-3, weight_decay=1e-2, warmup_steps=50,
               momentum=0.9)
    lit = G.PNApcsaftL(hip.pna_params, hip.mlp_params, cfg).to(DEV).train()
    lit.model.load_state_dict(hip.state_dict())
    conf = lit.configure_optimizers()
    opt, sched = conf["optimizer"], conf["lr_scheduler"]["scheduler"]
    assert isinstance(opt, FusedAdamW) and isinstance(opt, torch.optim.Optimizer)
    dd = data.to(DEV)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        loss = lit.training_step(dd)
        loss.backward()
        g0 = next(lit.parameters()).grad
        assert g0._base is not None and opt._flat_grad().data_ptr() == g0._base.data_ptr()   # zero-copy
        opt.step()
        sched.step()
        losses.append(float(loss))
    # the same six steps with torch.optim.AdamW through the oracle in float64
    from oracle.pna_torch import mape
    ref = copy.deepcopy(oracle).double().train()
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-3, weight_decay=1e-2, amsgrad=True, eps=1e-5)
    rs = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(ropt, 50)
    want = []
    for _ in range(6):
        ropt.zero_grad()
        l = mape(ref(data), data.para.view(-1, 3).double())
        l.backward()
        ropt.step()
        rs.step()
        want.append(float(l))
    assert losses[-1] < losses[0]
    for a, b in zip(losses, want):
        assert abs(a - b) < 2e-3 * abs(b), (losses, want)     # trajectories diverge slowly (f32 vs f64, Adam's 1/sqrt(v))
    # grad_scale: a gradient k times too large with grad_scale = 1/k gives the same step
    p1 = torch.nn.Parameter(torch.linspace(-1, 1, 200, device=DEV))
    p2 = torch.nn.Parameter(p1.detach().clone())
    o1, o2 = FusedAdamW([p1], lr=1e-2, amsgrad=True), FusedAdamW([p2], lr=1e-2, amsgrad=True)
    o2.grad_scale = 0.125
    g = torch.randn(200, device=DEV)
    p1.grad, p2.grad = g.clone(), g * 8
    o1.step()
    o2.step()
    assert torch.equal(p1.detach(), p2.detach())
