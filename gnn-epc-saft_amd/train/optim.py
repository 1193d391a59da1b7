"""Fused optimizers over one flat parameter buffer -- the optimizer half of the training-step host loop
(SURVEY.md section 8(f) rank 2).  Drop-ins for what ``configure_optimizers`` builds at
``/root/reference/gnnepcsaft/train/models.py:162-178``: ``torch.optim.AdamW(amsgrad=True, eps=1e-5)`` and
``torch.optim.SGD(nesterov=True)``.  Both are ``torch.optim.Optimizer`` subclasses, so ``param_groups[0]["lr"]``
schedulers (``CosineAnnealingWarmRestarts``), ``zero_grad`` and ``state_dict`` / ``load_state_dict`` in torch's
per-parameter format keep working; one step is ONE HIP launch (``gnnsaft_adamw_step`` / ``gnnsaft_sgd_step``).

Layout: every parameter's ``.data`` is re-pointed at a slice of one flat f32 buffer (offsets padded to 64
floats).  With ``layout=model.flat_layout()`` the slices coincide with the flat gradient buffer
``gnnsaft_backward`` writes, so the step consumes that buffer as it is -- no gather, no per-tensor launches;
any other gradient source is gathered into a flat buffer first (slower, still correct).
"""

from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple

import torch

from .._native import check, lib

_PAD = 64  # floats; matches PNAPCSAFT._backward


def default_layout(params: Sequence[torch.nn.Parameter]) -> Tuple[List[int], int]:
    offs, off = [], 0
    for p in params:
        offs.append(off)
        off += (p.numel() + _PAD - 1) // _PAD * _PAD
    return offs, off


class _FlatOptimizer(torch.optim.Optimizer):
    _STATE_KEYS: Tuple[str, ...] = ()

    def __init__(self, params: Iterable[torch.nn.Parameter], defaults: dict,
                 layout: Optional[Tuple[Sequence[int], int]] = None):
        params = list(params)
        if any(isinstance(p, dict) for p in params):
            raise NotImplementedError("one parameter group only (the reference builds one: models.py:163-178)")
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("optimizer got an empty parameter list")
        super().__init__(params, defaults)
        self._params: List[torch.nn.Parameter] = params
        dev = params[0].device
        for p in params:
            if p.device != dev or p.dtype != torch.float32 or not p.is_cuda:
                raise RuntimeError("fused optimizers need float32 parameters on one HIP device: move the module "
                                   "to the GPU before configure_optimizers()")
        self._offsets, self._total = (list(layout[0]), int(layout[1])) if layout is not None else default_layout(params)
        if len(self._offsets) != len(params):
            raise ValueError("layout does not match the parameter list")
        self._flat: Optional[torch.Tensor] = None
        self._flat_state = {}
        self._gather: Optional[torch.Tensor] = None
        self._reduced: Optional[torch.Tensor] = None   # flat gradient handed over by the data-parallel exchange
        self._steps = 0
        self.grad_scale = 1.0          # set to 1 / world_size by the data-parallel loop (SUM all-reduce)
        self.on_parameters_rewritten = None   # callable: the step kernel writes parameters behind torch's version counters
        self._step_tensor = None
        self.capturable = False              # FusedAdamW: per-step scalars from device memory (hipGraph capture)
        self._args_dev = None
        self._adopt()

    # ---- flat parameter buffer
    def _adopt(self) -> None:
        """(Re-)point every parameter at its slice of the flat buffer; needed again after module.to() / load."""
        dev = self._params[0].device
        flat = torch.zeros(self._total, dtype=torch.float32, device=dev)
        for p, off in zip(self._params, self._offsets):
            view = flat[off:off + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
        self._flat = flat

    def _adopted(self) -> bool:
        """Are the parameters still views of the flat buffer?  What breaks the adoption (``module.to()``, a loader that
        assigns ``p.data``) replaces EVERY parameter's storage, so three probes per step stand for the full walk
        (~110 tensors at L = 6: 0.08 ms per step); the full walk runs when a probe fails and before checkpoints."""
        base = self._flat.data_ptr()
        n = len(self._params)
        for i in (0, n // 2, n - 1):
            p = self._params[i]
            if p.data_ptr() != base + 4 * self._offsets[i] or not p.is_contiguous():
                return False
        return True

    def _adopted_all(self) -> bool:
        base = self._flat.data_ptr()
        return all(p.data_ptr() == base + 4 * off and p.is_contiguous() for p, off in zip(self._params, self._offsets))

    def _state_buffer(self, key: str) -> torch.Tensor:
        buf = self._flat_state.get(key)
        if buf is None:
            buf = self._flat_state[key] = torch.zeros_like(self._flat)
            for p, off in zip(self._params, self._offsets):   # torch-format views, so state_dict() just works
                self.state[p][key] = buf[off:off + p.numel()].view(p.shape)
        return buf

    def _flat_grad(self) -> torch.Tensor:
        """The gradients as one flat tensor in this optimizer's layout: zero-copy when they already are one."""
        first = self._params[0].grad
        if first is None:
            raise RuntimeError("step() without gradients")
        base = first.data_ptr() - 4 * self._offsets[0]
        owner = getattr(first, "_base", None)
        if owner is not None and owner.dim() == 1 and owner.data_ptr() == base and owner.numel() >= self._total \
                and owner.dtype == torch.float32:
            # views of ONE flat buffer in this optimizer's layout (what PNAPCSAFT's backward hands over)?
            zero_copy = True
            for p in self._params:      # (a view of `owner` with the parameter's shape: offsets were fixed at creation)
                g = p.grad
                if g is None or g._base is not owner:
                    zero_copy = False
                    break
            if zero_copy:      # ... laid out as this optimizer expects (first / middle / last, as _adopted() probes)
                n = len(self._params)
                for i in (0, n // 2, n - 1):
                    g = self._params[i].grad
                    if g.storage_offset() != self._offsets[i] or not g.is_contiguous():
                        zero_copy = False
                        break
            if zero_copy:
                return owner
        if self._gather is None:
            self._gather = torch.zeros_like(self._flat)
        for p, off in zip(self._params, self._offsets):
            dst = self._gather[off:off + p.numel()].view(p.shape)
            if p.grad is None:
                dst.zero_()
            else:
                dst.copy_(p.grad)
        return self._gather

    def use_reduced_gradient(self, flat: torch.Tensor, grad_scale: float) -> None:
        """The data-parallel exchange hands its all-reduced flat buffer (this optimizer's layout) to the NEXT
        ``step()``, which consumes it once instead of gathering the -- still local -- ``p.grad`` tensors again."""
        if flat.numel() < self._total or flat.dtype != torch.float32 or flat.device != self._flat.device:
            raise ValueError("reduced gradient buffer does not match this optimizer's flat layout")
        self._reduced = flat
        self.grad_scale = float(grad_scale)

    def _step_gradient(self) -> torch.Tensor:
        if self._reduced is not None:
            grad, self._reduced = self._reduced, None
            return grad
        return self._flat_grad()

    def zero_grad(self, set_to_none: bool = True) -> None:
        self._reduced = None
        if set_to_none:                       # torch's loop carries profiler / foreach bookkeeping (0.09 ms at L = 6)
            for p in self._params:
                p.grad = None
            return
        super().zero_grad(set_to_none=False)

    def flat_parameters(self) -> torch.Tensor:
        """The one flat f32 buffer every parameter is a view of (broadcast / checksum it as a whole)."""
        if not self._adopted_all():
            self._adopt()
        return self._flat

    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)          # torch's loader leaves freshly cloned per-parameter tensors
        loaded = {id(p): dict(self.state[p]) for p in self._params if p in self.state}
        self._flat_state = {}
        steps = 0
        for key in self._STATE_KEYS:
            if any(key in st for st in loaded.values()):
                buf = self._state_buffer(key)
                for p, off in zip(self._params, self._offsets):
                    src = loaded.get(id(p), {}).get(key)
                    if src is not None:
                        buf[off:off + p.numel()].view(p.shape).copy_(src)
        for p in self._params:
            st = loaded.get(id(p), {})
            if "step" in st:
                steps = max(steps, int(float(st["step"])))
                self.state[p]["step"] = torch.tensor(float(steps))
        self._steps = steps

    def _tick(self) -> int:
        self._steps += 1
        if self.on_parameters_rewritten is not None:
            self.on_parameters_rewritten()
        # torch keeps a host scalar tensor per parameter; here every parameter's entry is ONE shared tensor, updated
        # in place (a fresh torch.tensor per parameter per step was 0.1 ms of the step's host time)
        t = self._step_tensor
        if t is None or self.state[self._params[0]].get("step") is not t:
            t = self._step_tensor = torch.tensor(float(self._steps))
            for p in self._params:
                self.state[p]["step"] = t
        else:
            t.fill_(float(self._steps))
        return self._steps


class FusedAdamW(_FlatOptimizer):
    """``torch.optim.AdamW`` semantics (decoupled weight decay, optional AMSGrad), one launch per step."""

    _STATE_KEYS = ("exp_avg", "exp_avg_sq", "max_exp_avg_sq")

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 amsgrad: bool = False, layout=None):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad),
                         layout=layout)

    # ---- hipGraph capture (train/loop.py::GraphedTrainingStep): the step's scalars live in device memory.  They are
    # published by a one-thread launch OUTSIDE the graph whose values travel as kernel arguments (copied at enqueue
    # time, stream-ordered in front of the replay): no host buffer that a queued replay could read in a later step's
    # state (a pinned buffer re-read by a captured copy raced with the host rewriting it for the next step).
    def _arg_buffer(self):
        if self._args_dev is None or self._args_dev.device != self._flat.device:
            n = int(lib.gnnsaft_adamw_args_floats())
            self._args_dev = torch.zeros(n, dtype=torch.float32, device=self._flat.device)
        return self._args_dev

    def _write_args(self, step: int) -> None:
        g = self.param_groups[0]
        dev = self._arg_buffer()
        stream = torch.cuda.current_stream(dev.device).cuda_stream
        with torch.cuda.device(dev.device):
            check(lib.gnnsaft_adamw_args(float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                         float(g["weight_decay"]), step, float(self.grad_scale), dev.data_ptr(),
                                         stream), "gnnsaft_adamw_args")

    def prepare_replay(self) -> None:
        """Before every replay of a graph that holds a captured ``step()``, on the stream of the replay: advance the
        step count and enqueue this step's learning rate / bias corrections in front of it."""
        if not self.capturable:
            raise RuntimeError("prepare_replay() belongs to a capturable optimizer (opt.capturable = True)")
        self._write_args(self._tick())

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not self._adopted():
            self._adopt()
        g = self.param_groups[0]
        grad = self._step_gradient()
        m, v = self._state_buffer("exp_avg"), self._state_buffer("exp_avg_sq")
        vmax = self._state_buffer("max_exp_avg_sq") if g["amsgrad"] else None
        stream = torch.cuda.current_stream(self._flat.device).cuda_stream
        if self.capturable:
            dev = self._arg_buffer()
            if not torch.cuda.is_current_stream_capturing():
                self._write_args(self._tick())          # eager use of a capturable optimizer
            # (while capturing, only the step kernel is recorded and nothing is counted: prepare_replay() publishes
            # the scalars and counts, once per replay, outside the graph)
            with torch.cuda.device(self._flat.device):
                check(lib.gnnsaft_adamw_step_dev(self._flat.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(),
                                                 None if vmax is None else vmax.data_ptr(), self._total,
                                                 dev.data_ptr(), stream), "gnnsaft_adamw_step_dev")
            return loss
        step = self._tick()
        with torch.cuda.device(self._flat.device):
            check(lib.gnnsaft_adamw_step(self._flat.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(),
                                         None if vmax is None else vmax.data_ptr(), self._total, float(g["lr"]),
                                         float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                         float(g["weight_decay"]), step, float(self.grad_scale), stream),
                  "gnnsaft_adamw_step")
        return loss


class FusedSGD(_FlatOptimizer):
    """``torch.optim.SGD(momentum, nesterov=True, dampening=0)`` semantics, one launch per step."""

    _STATE_KEYS = ("momentum_buffer",)

    def __init__(self, params, lr: float = 1e-3, momentum: float = 0.0, weight_decay: float = 0.0,
                 nesterov: bool = True, layout=None):
        if lr < 0 or momentum < 0 or weight_decay < 0:
            raise ValueError("invalid SGD hyper-parameter")
        if not nesterov:
            raise NotImplementedError("only the Nesterov form the reference uses (models.py:171-177)")
        if momentum <= 0:
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")   # torch's own check
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, nesterov=True,
                                      dampening=0), layout=layout)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not self._adopted():
            self._adopt()
        g = self.param_groups[0]
        grad = self._step_gradient()
        first = "momentum_buffer" not in self._flat_state
        buf = self._state_buffer("momentum_buffer")
        self._tick()
        stream = torch.cuda.current_stream(self._flat.device).cuda_stream
        with torch.cuda.device(self._flat.device):
            check(lib.gnnsaft_sgd_step(self._flat.data_ptr(), grad.data_ptr(), buf.data_ptr(), self._total,
                                       float(g["lr"]), float(g["momentum"]), float(g["weight_decay"]), int(first),
                                       float(self.grad_scale), stream), "gnnsaft_sgd_step")
        return loss
